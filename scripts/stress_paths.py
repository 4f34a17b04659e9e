#!/usr/bin/env python3
"""Race / determinism stress: many scenes, every matching path (resident chain, per-view, native sharded world 1) must give
byte-identical kept lists, medians and affinity matrices, run after run."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("L3D_CHECK_POT", "1")
from line3d_amd.pipeline import Line3D, load_scene  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402
from line3d_amd import distributed as l3dist  # noqa: E402


def digest(l, scene, with_affinity=True):
    h = hashlib.sha256()
    for v in scene.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes())
        h.update(np.float32(med).tobytes())
    if with_affinity:
        l.finish(False)
        h.update(l.affinity()[0].tobytes())
    return h.hexdigest()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bad = 0
    for it in range(reps):
        V, S, N = [(14, 900, 8), (20, 500, 12), (9, 1500, 6), (30, 300, 10)][it % 4]
        sc = make_scene(V, S, N, seed=1000 + it, noise_px=0.5 + (it % 3))
        ref = None
        for mode in ("chain", "chain", "sync", "native", "chain"):
            l = Line3D("", matchingNeighbors=N)
            l.keep_view_matches(True)
            l.set_sync_matching(mode == "sync")
            load_scene(l, sc)
            l.prepare()
            if mode == "native":
                l3dist.match_views_chain_native(l, 0, 1, None, commit=True, n_segments=S, n_neighbors=N)
            else:
                l.match_views()
            d = digest(l, sc)
            l.close()
            if ref is None:
                ref = d
            elif d != ref:
                bad += 1
                print("MISMATCH scene %d mode %s" % (it, mode))
        print("scene %d (%d x %d, N=%d): ok" % (it, V, S, N), flush=True)
    print("stress done, mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
