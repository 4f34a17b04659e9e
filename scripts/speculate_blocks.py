#!/usr/bin/env python3
"""Does a view block that starts COLD a few views early reproduce the chain's kept lists exactly?  ("what would approach 6x".)

matchViews is a chain: a view's verification reads the kept matches of its earlier neighbours.  A rank that owns views [B, E) could start at
B - L with no history at all -- views in front of B - L contribute nothing -- and hope that by view B its kept lists are the true ones; if the
lists of the `window` views in front of B equal the true ones bit for bit, everything from B on is exact by induction (same inputs, same
arithmetic), so the speculation can be VERIFIED by comparing digests with the rank that owns those views.  This script measures, on one GPU,
how many warm-up views L that takes on a scene:

    python scripts/speculate_blocks.py [--views 64 --segments 2000 --neighbors 12] [--starts 16 32 48] [--warmups 0 6 12 18 24]

Truth: the resident chain.  Speculation: the step-wise seam path (match_view_compute / match_view_commit), views in front of B - L committed empty.
Prints, per (B, L), the first view >= B - L from which on every kept list equals the truth, and whether the `window` views in front of B do."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene   # noqa: E402
from line3d_amd.capi import MATCH_DTYPE              # noqa: E402
from line3d_amd.synth import make_scene              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--segments", type=int, default=2000)
    ap.add_argument("--neighbors", type=int, default=12)
    ap.add_argument("--seed", type=int, default=20260)
    ap.add_argument("--starts", type=int, nargs="+", default=[16, 32, 48])
    ap.add_argument("--warmups", type=int, nargs="+", default=[0, 6, 12, 18, 24])
    ap.add_argument("--check", type=int, default=10, help="views behind B that are compared as well")
    a = ap.parse_args()
    V, S, N = a.views, a.segments, a.neighbors
    window = N // 2
    scene = make_scene(V, S, N, seed=a.seed)
    ids = [v["id"] for v in scene.views]
    t = Line3D("", matchingNeighbors=N)
    t.keep_view_matches(True)
    load_scene(t, scene)
    t.prepare()
    t.match_views()
    truth = {i: t.view_matches(i)[0].tobytes() for i in ids}
    t.close()
    empty = np.zeros(0, MATCH_DTYPE)
    for B in a.starts:
        for L in a.warmups:
            first = max(0, B - L)
            l = Line3D("", matchingNeighbors=N)
            load_scene(l, scene)
            l.prepare()
            order, _ = l.match_begin()
            got = {}
            last = min(V, B + a.check)
            for k, vid in enumerate(order):
                vid = int(vid)
                if k >= last:
                    break
                if k < first:
                    l.match_view_commit(vid, empty, None, 1.0)           # nothing known about the views in front of the block
                    continue
                m, med, best = l.match_view_compute(vid, 0, S)
                got[k] = m.tobytes()
                l.match_view_commit(vid, m, best, med)
            l.close()
            same = {k: got[k] == truth[int(order[k])] for k in got}
            exact_from = None
            for k in sorted(same):
                if all(same[q] for q in same if q >= k):
                    exact_from = k
                    break
            boundary_ok = all(same.get(k, False) for k in range(max(first, B - window), B)) if B - window >= first else False
            print("block start %3d, warm-up %2d views (cold start at %3d): kept lists exact from view %s on; the %d views in front of the block exact: %s; views %d..%d exact: %s"
                  % (B, L, first, exact_from, window, boundary_ok, B, last - 1, all(same[k] for k in same if k >= B)), flush=True)


if __name__ == "__main__":
    main()
