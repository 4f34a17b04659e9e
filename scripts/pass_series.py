#!/usr/bin/env python3
"""Per-pass times of matchViews over many consecutive passes (clock ramp / drift of a box): python scripts/pass_series.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sc = make_scene(64, 2000, 12, seed=20260)
l = Line3D("", matchingNeighbors=12)
load_scene(l, sc)
l.prepare()
ts = []
for _ in range(n):
    t0 = time.perf_counter(); l.match_views(); ts.append((time.perf_counter() - t0) * 1e3)
for i in range(0, n, 10):
    print("%3d: " % i + " ".join("%.2f" % t for t in ts[i:i + 10]))
l.close()
