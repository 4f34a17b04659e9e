import sys, time, os
sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
sc = make_scene(64, 2000, 12, seed=20260)
l = Line3D("", matchingNeighbors=12)
t0=time.time(); load_scene(l, sc); l.prepare(); print("setup", time.time()-t0)
t0=time.time(); l.match_views(); print("match", time.time()-t0)
for diff in (False, True):
    t0=time.time(); l.finish(diff); dt=time.time()-t0
    A, nn = l.affinity()[0], l.affinity()[1] if len(l.affinity())>1 else None
    print("finish diffusion=%s: %.2f s, lines %d, affinity nnz %d" % (diff, dt, len(l.getResult()), len(A)))
print(l.stats())
