#!/usr/bin/env python3
"""Soak of the block modes over many scenes (virtual ranks as threads on one GPU, all-gather through the host):
    python scripts/soak_blocks.py [n_scenes]
Every scene runs (a) l3d_match_chain_blocks (replicas) and (b) l3d_match_chain_partition + the collective finish, each with the default warm-up and
with FORCED FAILURES -- warm-ups far shorter than the chain's memory (one window, one view, none).  Round 5: a block whose speculation fails is
re-run warm from its predecessor's true lists, so EVERY run must end exact -- the ONE chain's kept lists, products, affinity list and lines byte
for byte, on every rank -- with no fall-through to another mode; the number of repaired blocks is reported.  (c) "segments": the partitioned
segment-sharded run (l3d_shard_chain_partition: no speculation) + the same collective finish; every third scene a scattered one (cameras in no
order, neighbours from shared world points: non-mutual, far apart in the chain) in all three modes."""
import hashlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from line3d_amd.pipeline import Line3D, load_scene, load_scene_worldpoints   # noqa: E402
from line3d_amd.synth import make_scene, make_scene_scattered               # noqa: E402
from helpers import thread_exchange as _thread_exchange  # noqa: E402


def digest(l, scene, views=None):
    h = hashlib.sha256()
    for v in scene.views if views is None else views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes())
        h.update(np.float32(med).tobytes())
    return h.hexdigest()


def digest_products(l):
    h = hashlib.sha256()
    p = l.resident_products()
    for k in ("pot_start", "pot_tgt", "best"):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    return h.hexdigest()


def digest_result(l):
    h = hashlib.sha256()
    A, n = l.affinity()
    h.update(A.tobytes())
    for seg2, seg3 in l.getResult():
        h.update(np.array(seg2, np.int64).tobytes())
        for P, Q in seg3:
            h.update(np.asarray(P, np.float64).tobytes()); h.update(np.asarray(Q, np.float64).tobytes())
    return h.hexdigest()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    # (views, segments, neighbours, ranks, warm-up views; -1 = the default of eight windows)
    shapes = [(48, 150, 6, 3, -1), (64, 300, 8, 4, 4), (40, 200, 6, 2, 1), (96, 120, 6, 4, 0), (72, 400, 12, 3, -1), (80, 250, 10, 5, 5)]
    bad = repaired = 0
    for s in range(n):
        V, S, N, W, warm = shapes[s % len(shapes)]
        scattered = s % 3 == 2
        scene = make_scene_scattered(V, S, seed=4000 + s) if scattered else make_scene(V, S, N, seed=4000 + s)
        load = load_scene_worldpoints if scattered else load_scene
        by_id = sorted(scene.views, key=lambda v: v["id"])        # (the dense map: views in id order)
        ref = Line3D("", matchingNeighbors=N)
        ref.keep_view_matches(True)
        load(ref, scene)
        ref.prepare()
        ref.match_views()
        want, want_prod = digest(ref, scene), digest_products(ref)
        ref.finish(False)
        want_res = digest_result(ref)
        ref.close()
        for mode in ("replicas", "partition", "segments"):
            make, calls = _thread_exchange(W)
            ls, verdicts, errors, held, fell = [], [None] * W, [], [None] * W, [False] * W
            for r in range(W):
                l = Line3D("", matchingNeighbors=N)
                l.keep_view_matches(True)
                load(l, scene)
                l.prepare()
                ls.append(l)

            def run(r):
                try:
                    if mode == "replicas":
                        verdicts[r] = ls[r].block_run(r, W, make(r), None, warm)
                        if not verdicts[r]:
                            # the documented fall-through (a block shorter than the neighbour window -- scattered neighbourhoods; the same verdict on
                            # every rank): the segment-sharded run, products on every device
                            ls[r].shard_run(r, W, 10 * S * N // W + 1024, make(r), None, commit="device")
                            fell[r] = True
                        if True:
                            held[r] = (digest(ls[r], scene), digest_products(ls[r]))
                            ls[r].finish(False)
                    else:
                        if mode == "segments":
                            ls[r].shard_run(r, W, 10 * S * N // W + 1024, make(r), None, commit="partition")
                            verdicts[r] = True
                        else:
                            verdicts[r] = ls[r].partition_run(r, W, make(r), None, warm)
                        if verdicts[r]:
                            info = ls[r].partition_info()
                            mine = by_id[info["held"][0]:info["held"][1]]
                            held[r] = (digest(ls[r], scene, mine), mine)
                            ls[r].finish_sharded(False)
                except Exception as e:      # noqa: BLE001
                    errors.append((r, repr(e)))
            th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            ok = not errors and (verdicts == [True] * W or (mode == "replicas" and scattered and verdicts == [False] * W and all(fell)))
            n_rep = ls[0].partition_info()["blocks_rerun"] if ok and not any(fell) else (0 if ok else -1)
            if ok and mode == "replicas":
                ok = all(h == (want, want_prod) for h in held)
            elif ok:
                refl = Line3D("", matchingNeighbors=N)          # the one chain's lists of exactly the views a rank holds
                refl.keep_view_matches(True)
                load(refl, scene)
                refl.prepare()
                refl.match_views()
                ok = all(h[0] == digest(refl, scene, h[1]) for h in held)
                refl.close()
            if ok:
                ok = all(digest_result(l) == want_res for l in ls)
            if ok and mode != "replicas":
                ok = -2 not in [c[0] for c in calls] and -4 not in [c[0] for c in calls]      # no block and no table piece travels
            bad += 0 if ok else 1
            repaired += max(0, n_rep)
            print("scene %2d (%s%d x %d x %d, %d ranks, warm-up %s) %-9s: verdicts %s, %d block(s) re-run warm  %s %s"
                  % (s, "scattered " if scattered else "", V, S, N, W, warm if warm >= 0 else "8 windows", mode, verdicts, n_rep, "ok" if ok else "WRONG", errors or ""), flush=True)
            for l in ls:
                l.close()
    print("soak done: %d scenes x 3 modes, %d blocks repaired, %d violations" % (n, repaired, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
