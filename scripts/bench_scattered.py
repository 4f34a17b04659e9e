#!/usr/bin/env python3
"""ms per matchViews pass and the per-kernel split on the SCATTERED scene (synth.make_scene_scattered: cameras in no order, ragged views, neighbours chosen by
the library from shared world points -- not mutual, twins under min_baseline):   python scripts/bench_scattered.py [VIEWS SEGMENTS NEIGHBOURS seed passes]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene_worldpoints
from line3d_amd.synth import make_scene_scattered

V, S, N, seed, passes = ([int(x) for x in sys.argv[1:6]] + [48, 1500, 10, 4242, 10][len(sys.argv) - 1:])[:5]
sc = make_scene_scattered(V, S, seed=seed)
l = Line3D("", matchingNeighbors=N)
load_scene_worldpoints(l, sc)
l.prepare()
ctx = l.context()
ts = []
for _ in range(passes + 2):
    t0 = time.perf_counter()
    l.match_views()
    ts.append(time.perf_counter() - t0)
ctx.profile_only(None)
ctx.profile_enable(True)
ctx.profile_reset()
l.match_views()
prof = {k: round(v[1], 3) for k, v in ctx.profile_all().items() if v[0]}
ctx.profile_enable(False)
st = l.stats()
t1 = time.perf_counter(); l.finish(False); tf = time.perf_counter() - t1
print(json.dumps(dict(shape=[V, S, N], path=l.match_path(), ms_per_pass=round(min(ts[2:]) * 1e3, 3), pairs=st["pairs"], raw=st["raw"], kept=st["kept"],
                      g_pairs_per_s=round(st["pairs"] / min(ts[2:]) / 1e9, 2), kernels_ms=prof, finish_ms=round(tf * 1e3, 2), lines=int(l.stats()["lines"]))))
l.close()
