"""A/B of the affinity fill on the bench scene: device path (default) against the host enumeration (L3D_AFFINITY_HOST=1,
development aid) -- edge lists, node numbering and final lines must be identical."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

V, S, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 2000, 12)))
sc = make_scene(V, S, N, seed=20260)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc); l.prepare(); l.match_views()
res = {}
for mode in ("host", "device", "device"):
    if mode == "host":
        os.environ["L3D_AFFINITY_HOST"] = "1"
    else:
        os.environ.pop("L3D_AFFINITY_HOST", None)
    for diff in (False, True):
        t0 = time.time(); l.finish(diff); dt = time.time() - t0
        A = l.affinity()
        res[(mode, diff)] = (np.array(A[0]).copy(), A[1], [(tuple(s2), [tuple(np.concatenate(p)) for p in s3]) for s2, s3 in l.getResult()])
        print(mode, "diffusion", diff, "%.1f ms" % (dt * 1e3), "edges", len(A[0]), "nodes", A[1], "lines", len(l.getResult()), flush=True)
for diff in (False, True):
    a, b = res[("host", diff)], res[("device", diff)]
    same_A = a[0].shape == b[0].shape and a[0].tobytes() == b[0].tobytes() and a[1] == b[1]
    print("diffusion", diff, "edge lists identical:", same_A, "| lines identical:", a[2] == b[2])
    assert same_A and a[2] == b[2]
print("OK")
