#!/usr/bin/env python3
"""Kept matches per view (and per rank range) of a synthetic scene: sizing of the sharded chain's slots."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N, seed = (int(x) for x in sys.argv[1:5])
sc = make_scene(V, S, N, seed=seed)
l = Line3D("", matchingNeighbors=N, useCollinearity=False)
l.keep_view_matches(True)
load_scene(l, sc)
l.prepare()
l.match_views()
tot = []
worst = {2: 0, 4: 0, 8: 0}
for v in sc.views:
    m, _ = l.view_matches(v["id"])
    tot.append(len(m))
    for W in worst:
        h = np.bincount(np.minimum(m["segID1"].astype(np.int64) * W // S, W - 1), minlength=W)
        worst[W] = max(worst[W], int(h.max()))
tot = np.array(tot)
print("V %d S %d N %d seed %d: kept per view mean %.0f max %d (view %d) p99 %.0f; worst rank range: %s; mean per range: %s"
      % (V, S, N, seed, tot.mean(), tot.max(), int(tot.argmax()), np.percentile(tot, 99), worst, {W: int(tot.mean() / W) for W in worst}))
print("kept per view by block of 32 views:", [int(tot[i:i + 32].mean()) for i in range(0, V, 32)])
