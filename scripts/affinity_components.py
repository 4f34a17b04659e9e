"""Connected components of the bench scene's affinity graph (would a per-component Felzenszwalb-Huttenlocher parallelise?)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
V, S, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 2000, 12)))
sc = make_scene(V, S, N, seed=20260)
l = Line3D("", matchingNeighbors=N)
load_scene(l, sc); l.prepare(); l.match_views(); l.finish(False)
A, n = l.affinity()
g = coo_matrix((np.ones(len(A)), (A["i"], A["j"])), shape=(n, n)).tocsr()
nc, lab = connected_components(g, directed=False)
sizes = np.bincount(lab)
edges_per = np.bincount(lab[A["i"]], minlength=nc)
print("nodes", n, "edges", len(A), "components", nc, "largest (nodes)", int(sizes.max()), "largest (edges)", int(edges_per.max()),
      "share of edges in the 10 largest: %.3f" % (np.sort(edges_per)[-10:].sum() / len(A)))
