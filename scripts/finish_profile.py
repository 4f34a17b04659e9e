import json, os, sys, time
sys.path.insert(0, "/root/repo")
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
V, S, N = (int(x) for x in sys.argv[1:4])
sc = make_scene(V, S, N, seed=20260)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc); l.prepare(); l.match_views()
t0 = time.perf_counter(); l.finish(False); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); l.finish(False); t2 = time.perf_counter() - t0
ctx = l.context()
ctx.profile_only(None); ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter(); l.finish(False); t3 = time.perf_counter() - t0
prof = {k: (v[0], round(v[1], 3)) for k, v in ctx.profile_all().items() if v[0]}
ctx.profile_enable(False)
st = l.stats()
print(json.dumps(dict(shape=[V, S, N], finish_first_s=round(t1, 3), finish_s=round(t2, 3), finish_profiled_s=round(t3, 3), fill=list(ctx.last_fill_counts()), t_affinity=st["t_affinity"], t_cluster=st["t_cluster"], edges=st["edges"], lines=st["lines"], kernels_ms=prof)))
l.close()
