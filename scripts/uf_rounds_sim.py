"""How many rounds would a deterministic-reservation Felzenszwalb-Huttenlocher take on the bench scene's edge list?  (numpy
simulation of the parallel scheme: per round every undecided edge of a window reserves both its components with its index
(minimum wins); an edge holding both reservations decides exactly as the sequential loop would.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 2000, 12)))
window = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
sc = make_scene(V, S, N, seed=20260)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc); l.prepare(); l.match_views(); l.finish(False)
A, n = l.affinity()
order = np.argsort(A["w"], kind="stable")
E = A[order]
# drop the reversed twin right behind an edge (as the product does)
keep = np.ones(len(E), bool)
tw = (E["i"][1:] == E["j"][:-1]) & (E["j"][1:] == E["i"][:-1]) & (E["w"][1:] == E["w"][:-1])
k = 0
idx = []
i = 0
while i < len(E):
    idx.append(i)
    i += 2 if i + 1 < len(E) and tw[i] else 1
E = E[np.array(idx)]
m = len(E)
print("nodes", n, "edges", m)
parent = np.arange(n); size = np.ones(n, np.int64); thr = np.full(n, 1.0, np.float32); rank = np.zeros(n, np.int32)
def roots(x):
    r = parent[x]
    while True:
        rr = parent[r]
        if np.array_equal(rr, r): return r
        r = rr
active = np.arange(m)
rounds = 0; decided_total = 0
t0 = time.time()
hist = []
while len(active):
    w = active[:window]
    ra = roots(E["i"][w]); rb = roots(E["j"][w])
    same = ra == rb
    res = np.full(n, m + 1, np.int64)
    cand = w[~same]; ca = ra[~same]; cb = rb[~same]
    np.minimum.at(res, ca, cand); np.minimum.at(res, cb, cand)
    win = (res[ca] == cand) & (res[cb] == cand)
    # winners decide (their components are pairwise distinct by construction)
    for e, a, b in zip(cand[win], ca[win], cb[win]):
        we = E["w"][e]
        if we <= thr[a] and we <= thr[b]:
            if rank[a] > rank[b]: parent[b] = a; size[a] += size[b]; r = a
            else:
                parent[a] = b; size[b] += size[a]
                if rank[a] == rank[b]: rank[b] += 1
                r = b
            thr[r] = np.float32(we + np.float32(1.0) / np.float32(size[r]))
    done = np.zeros(len(w), bool); done[same] = True
    tmp = np.zeros((~same).sum(), bool); tmp[win] = True
    done[~same] = tmp
    hist.append(int(done.sum()))
    active = np.concatenate([w[~done], active[window:]])
    rounds += 1
print("window", window, "rounds", rounds, "decided per round: first 10", hist[:10], "median", int(np.median(hist)), "min", min(hist), "sim %.1f s" % (time.time() - t0))
