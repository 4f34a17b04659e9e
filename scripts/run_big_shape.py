#!/usr/bin/env python3
"""One resident matchViews pass (+ the products of performMatching, built in blocks of views) on a big synthetic scene, with the HBM in use:
    python scripts/run_big_shape.py VIEWS SEGMENTS NEIGHBOURS [finish]
configs[4]'s per-view shape at 256 views: python scripts/run_big_shape.py 256 4000 24"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in sys.argv[1:4])
want_finish = len(sys.argv) > 4 and sys.argv[4] == "finish"
hip = C.CDLL("libamdhip64.so")


def hbm_used_gb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return round((total.value - free.value) / 2**30, 2), round(total.value / 2**30, 1)


t0 = time.perf_counter()
sc = make_scene(V, S, N, seed=20260)
t_scene = time.perf_counter() - t0
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc)
l.prepare()
out = dict(shape=[V, S, N], scene_s=round(t_scene, 1), hbm_after_prepare_gb=hbm_used_gb()[0])
passes = []
for _ in range(2):
    t0 = time.perf_counter()
    l.match_views()
    passes.append(round(time.perf_counter() - t0, 3))
st = l.stats()
used, total = hbm_used_gb()
out.update(match_views_s=passes, pairs=st["pairs"], raw=st["raw"], kept=st["kept"], g_pairs_per_s=round(st["pairs"] / min(passes) / 1e9, 2),
           hbm_after_match_views_gb=used, hbm_total_gb=total, kept_arena_gb=round(st["kept"] * 32 / 2**30, 2))
p = l.resident_products() if V * S <= 300000 else None
nv, nd, nh = C.c_int(0), C.c_int(0), C.c_int(0)
npot = C.c_int64(0)
l._chk(l.lib.l3d_line3d_products_sizes(l.h, C.byref(nv), C.byref(nd), C.byref(npot), C.byref(nh)))
out.update(potential_correspondences=npot.value, dense_segments=nd.value)
if want_finish:
    try:
        t0 = time.perf_counter()
        l.finish(False)
        cand, passed = l.context().last_fill_counts()
        out.update(finish_s=round(time.perf_counter() - t0, 3), lines=int(l.stats()["lines"]), edges=int(l.stats()["edges"]), hbm_after_finish_gb=hbm_used_gb()[0],
                   affinity_candidates=cand, affinity_passed=passed)
    except Exception as e:      # noqa: BLE001
        cand, passed = l.context().last_fill_counts()
        out.update(finish_error=str(e)[:300], affinity_candidates=cand, affinity_passed=passed, hbm_after_finish_gb=hbm_used_gb()[0])
print(json.dumps(out))
l.close()
