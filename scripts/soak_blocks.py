#!/usr/bin/env python3
"""Soak of the block mode's safety property over many scenes: whenever the verdict of l3d_match_chain_blocks is "exact", every rank must hold the
ONE chain's kept lists and products byte for byte (and the verdict must be the same on every rank); a "not exact" verdict must commit nothing.
Virtual ranks as threads on one GPU, all-gather through the host.   python scripts/soak_blocks.py [n_scenes]"""
import hashlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from line3d_amd.pipeline import Line3D, load_scene   # noqa: E402
from line3d_amd.synth import make_scene              # noqa: E402
from helpers import thread_exchange as _thread_exchange  # noqa: E402


def digest(l, scene):
    h = hashlib.sha256()
    for v in scene.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes())
        h.update(np.float32(med).tobytes())
    p = l.resident_products()
    for k in ("pot_start", "pot_tgt", "best"):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    return h.hexdigest()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    shapes = [(48, 150, 6, 3, -1), (64, 300, 8, 4, -1), (40, 200, 6, 2, 9), (96, 120, 6, 4, 12), (72, 400, 12, 3, -1)]
    bad = exact = 0
    for s in range(n):
        V, S, N, W, warm = shapes[s % len(shapes)]
        scene = make_scene(V, S, N, seed=4000 + s)
        ref = Line3D("", matchingNeighbors=N)
        ref.keep_view_matches(True)
        load_scene(ref, scene)
        ref.prepare()
        ref.match_views()
        want = digest(ref, scene)
        ref.close()
        make, _calls = _thread_exchange(W)
        ls, verdicts, errors = [], [None] * W, []
        for r in range(W):
            l = Line3D("", matchingNeighbors=N)
            l.keep_view_matches(True)
            load_scene(l, scene)
            l.prepare()
            ls.append(l)

        def run(r):
            try:
                verdicts[r] = ls[r].block_run(r, W, make(r), None, warm)
            except Exception as e:      # noqa: BLE001
                errors.append((r, repr(e)))
        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        ok = not errors and len(set(verdicts)) == 1
        if ok and verdicts[0]:
            exact += 1
            ok = all(digest(l, scene) == want for l in ls)
        elif ok:
            ok = all(l.resident_products() is None for l in ls)
        bad += 0 if ok else 1
        print("scene %2d (%d x %d x %d, %d ranks, warm-up %s): verdict %s  %s %s" % (s, V, S, N, W, warm if warm >= 0 else "8 windows", verdicts, "ok" if ok else "WRONG", errors or ""), flush=True)
        for l in ls:
            l.close()
    print("soak done: %d scenes, %d exact speculations, %d violations" % (n, exact, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
