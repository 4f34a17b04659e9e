#!/usr/bin/env python3
"""Window statistics of k_verify_window on a scene of any shape (printed by the library when the context closes):
    L3D_VW_STAMPS=1 python3 scripts/vw_stats_shape.py VIEWS SEGMENTS NEIGHBOURS [seed]
wave-cycle split of the phases, entries walked per hypothesis / inside the d1 window / inside both windows, evaluated pairs and their fate;
plus the distribution of candidates per source segment (m) of a mid-chain view."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in sys.argv[1:4])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20260
sc = make_scene(V, S, N, seed=seed)
l = Line3D("", matchingNeighbors=N)
l.keep_view_matches(True)
load_scene(l, sc)
l.prepare()
l.match_views()
st = l.stats()
if os.environ.get("L3D_VW_STAMPS"):        # the in-kernel stamps are collected by the per-view seam path (l3d_compute_pairwise_matches): one more pass that way
    l.set_sync_matching(True)
    l.match_views()
    l.set_sync_matching(False)
print("shape %dx%dx%d: pairs %.4g raw candidates %.4g kept %.4g" % (V, S, N, st["pairs"], st["raw"], st["kept"]))
mid = sc.views[V // 2]["id"]
m, _ = l.view_matches(mid)
per = np.bincount(m["segID1"], minlength=S)
print("view %d: kept per segment mean %.1f max %d" % (mid, per.mean(), per.max()))
l.close()
