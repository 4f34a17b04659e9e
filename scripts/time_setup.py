import sys, time, os
sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
sc = make_scene(64, 2000, 12, seed=20260)
for rep in range(3):
    l = Line3D("", matchingNeighbors=12)
    t0=time.time(); load_scene(l, sc); t1=time.time(); l.prepare(); t2=time.time()
    l.match_views(); t3=time.time(); l.finish(False); t4=time.time()
    print("addImage x64 %.1f ms  prepare %.1f ms  match %.1f ms  finish %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3))
    l.close()
