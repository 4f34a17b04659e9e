#!/usr/bin/env python3
"""Load balance of k_verify_window over the source segments of a view: candidates per segment m (mean, rms, max) through the per-view seam path
    L3D_TIMING=1 python3 scripts/seg_load_shape.py VIEWS SEGMENTS NEIGHBOURS
A segment's workgroup walks m hypotheses x a window that grows with m: its time goes like m^2, and a launch lasts as long as its longest segment."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in sys.argv[1:4])
sc = make_scene(V, S, N, seed=20260)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc)
l.prepare()
ids, ns = l.match_begin()
ctx = l.context()
for vid, Sv in zip(ids.tolist(), ns.tolist()):
    if l.view_num_to_be_matched(vid) == 0:
        continue
    res = l.match_view_compute(vid, 0, Sv)
    st = ctx.last_stats()
    mean, rms = st[1] / Sv, (st[2] / Sv) ** 0.5
    print("view %3d: %d segments, candidates per segment mean %.0f rms %.0f; sum m^2 / (S mean^2) = %.2f; kept %d" % (vid, Sv, mean, rms, (rms / mean) ** 2, len(res[0])))
    l.match_view_commit(vid, res[0])
l.match_end()
l.close()
