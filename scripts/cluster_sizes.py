import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
sc = make_scene(64, 2000, 12, seed=20260)
l = Line3D("", matchingNeighbors=12)
load_scene(l, sc); l.prepare(); l.match_views()
for diff in (False, True):
    l.finish(diff)
    n = np.array([len(s2) for s2, s3 in l.getResult()])
    print("diffusion", diff, "lines", len(n), "members: median", int(np.median(n)), "p99", int(np.percentile(n, 99)), "max", int(n.max()), "over 128:", int((n > 128).sum()))
