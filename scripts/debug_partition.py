import os, sys, threading
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
from helpers import thread_exchange
V, S, N, W, warm, seed = [int(x) for x in sys.argv[1:7]]
scene = make_scene(V, S, N, seed=seed)
ref = Line3D("", matchingNeighbors=N); ref.keep_view_matches(True); load_scene(ref, scene); ref.prepare(); ref.match_views()
rl = {v["id"]: ref.view_matches(v["id"]) for v in scene.views}
rp = ref.resident_products()
ref.finish(False)
rA, rn = ref.affinity(); rres = ref.getResult(); rhyp = ref.resident_products()["hyp"]
make, calls = thread_exchange(W)
ls = []
for r in range(W):
    l = Line3D("", matchingNeighbors=N); l.keep_view_matches(True); load_scene(l, scene); l.prepare(); ls.append(l)
shares = [None] * W; errors = []
def run(r):
    try:
        v = ls[r].partition_run(r, W, make(r), None, warm)
        shares[r] = (ls[r].partition_info(), ls[r].resident_products())
        ls[r].finish_sharded(False)
    except Exception as e:
        errors.append((r, repr(e)))
th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
[x.start() for x in th]; [x.join() for x in th]
print("errors", errors)
ids = [v["id"] for v in scene.views]
sb = rp["seg_base"]
for r, l in enumerate(ls):
    info, prod = shares[r]
    print("rank", r, info)
    for vi in range(info["held"][0], info["held"][1]):
        m, med = l.view_matches(ids[vi])
        if m.tobytes() != rl[ids[vi]][0].tobytes(): print("  view", vi, "LIST differs", len(m), len(rl[ids[vi]][0]))
        if np.float32(med) != np.float32(rl[ids[vi]][1]): print("  view", vi, "MEDIAN differs", med, rl[ids[vi]][1])
    ps, pt = prod["pot_start"], prod["pot_tgt"]
    nbad = 0
    for d in range(int(sb[info["rows"][0]]), int(sb[info["rows"][1]])):
        if not np.array_equal(pt[ps[d]:ps[d+1]], rp["pot_tgt"][rp["pot_start"][d]:rp["pot_start"][d+1]]):
            nbad += 1
            if nbad < 4: print("  row", d, "view", np.searchsorted(sb, d, side="right") - 1, "differs", len(pt[ps[d]:ps[d+1]]), len(rp["pot_tgt"][rp["pot_start"][d]:rp["pot_start"][d+1]]))
    print("  bad rows", nbad)
    lo, hi = int(sb[info["held"][0]]), int(sb[info["held"][1]])
    bb = prod["best"][lo:hi]; rb = rp["best"][lo:hi]
    nb = int((bb.view(np.uint8).reshape(len(bb), -1) != rb.view(np.uint8).reshape(len(rb), -1)).any(axis=1).sum())
    print("  best differs at", nb, "segments")
    A, n = l.affinity()
    print("  affinity equal", n == rn and A.tobytes() == rA.tobytes(), len(A), len(rA), n, rn, "hyp equal", l.resident_products()["hyp"].tobytes() == rhyp.tobytes())
    if len(A) == len(rA) and A.tobytes() != rA.tobytes():
        k = int(np.nonzero((A["i"] != rA["i"]) | (A["j"] != rA["j"]) | (A["w"] != rA["w"]))[0][0]); print("   first diff at entry", k, A[k], rA[k])
    print("  lines", len(l.getResult()), len(rres))
print([c[0] for c in calls])
