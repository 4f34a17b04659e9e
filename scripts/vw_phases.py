#!/usr/bin/env python3
"""Phase split of k_verify_window (wave cycles: build / setup / scan / drain / final) when every view is verified in W
source-segment slices through the per-view seam: `L3D_VW_STAMPS=1 python scripts/vw_phases.py 8` (printed when the context closes)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = make_scene(16, 2000, 12, seed=20260)
l = Line3D("", matchingNeighbors=12)
load_scene(l, sc); l.prepare()
ids, ns = l.match_begin()
for vid, S in zip(ids.tolist(), ns.tolist()):
    parts, bests = [], []
    med = 1.0
    for r in range(W):
        s0, s1 = S * r // W, S * (r + 1) // W
        res = l.match_view_compute(vid, s0, s1)
        parts.append(res[0]); 
    m = np.concatenate(parts)
    l.match_view_commit(vid, m)
l.match_end()
l.close()
