#!/usr/bin/env python3
"""ms per matchViews pass of the resident chain on a synthetic scene of any shape, with the per-kernel split of one bracketed pass:
    python scripts/bench_shape.py VIEWS SEGMENTS NEIGHBOURS [passes] [seed]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in sys.argv[1:4])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 5
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 20260
sc = make_scene(V, S, N, seed=seed)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc)
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
t1 = time.perf_counter(); l.finish(False); tf1 = time.perf_counter() - t1
t1 = time.perf_counter(); l.finish(False); tf2 = time.perf_counter() - t1
best = min(ts[2:])
print(json.dumps(dict(shape=[V, S, N], first_pass_ms=round(ts[0] * 1e3, 2), second_pass_ms=round(ts[1] * 1e3, 2), ms_per_pass=round(best * 1e3, 3),
                      median_ms=round(sorted(ts[2:])[len(ts[2:]) // 2] * 1e3, 3), pairs=st["pairs"], raw=st["raw"], kept=st["kept"],
                      g_pairs_per_s=round(st["pairs"] / best / 1e9, 2), kernels_ms=prof, finish_first_ms=round(tf1 * 1e3, 2), finish_ms=round(tf2 * 1e3, 2),
                      host_split_ms={k: round(st[k] * 1e3, 3) for k in ("t_match", "t_gpu_call", "t_commit", "t_finalize")})))
l.close()
