"""Generates tests/golden/*.npz.  Run in the build container (needs oracle/_ref, i.e. /root/reference):
    python tests/golden/make_golden.py
Fixtures are data only (inputs + expected outputs):
  clustering_ref.npz   random edge lists with many tied weights + the labels produced by the REFERENCE's own
                       clustering.cc (oracle/_ref/libclustering_ref.so) -> pins the oracle's and the product's
                       graph segmentation to the reference itself.
  seam_small.npz       marshalled inputs of two views of a seeded scene + the oracle's kept matches / median /
                       collinearity / diffusion outputs (oracle-generated: guards against drift of the oracle and
                       lets the GPU tests check against committed vectors).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402


def ref_clustering(ei, ej, ew, n, c):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libclustering_ref.so"))
    labels = np.zeros(n, np.int32)
    ei, ej, ew = (np.ascontiguousarray(a) for a in (ei.astype(np.int32), ej.astype(np.int32), ew.astype(np.float32)))
    rc = lib.l3dref_clustering(ei.ctypes.data_as(C.c_void_p), ej.ctypes.data_as(C.c_void_p), ew.ctypes.data_as(C.c_void_p),
                               C.c_int(len(ei)), C.c_int(n), C.c_float(c), labels.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return labels


def main():
    rng = np.random.default_rng(2015)
    out = {}
    for k, (n, E, levels) in enumerate([(50, 300, 5), (400, 3000, 3), (1000, 9000, 50), (200, 2000, 1)]):
        ei = rng.integers(0, n, E)
        ej = rng.integers(0, n, E)
        keep = ei != ej
        ei, ej = ei[keep], ej[keep]
        ew = (rng.integers(1, levels + 1, len(ei)) / levels).astype(np.float32)      # many exact ties, some == 1.0
        ei2 = np.concatenate([ei, ej])
        ej2 = np.concatenate([ej, ei])
        ew2 = np.concatenate([ew, ew])                                                # both directions like clusterSegments2D
        for c in (1.0, 0.3):
            out["c%d_%g_i" % (k, c)] = ei2.astype(np.int32)
            out["c%d_%g_j" % (k, c)] = ej2.astype(np.int32)
            out["c%d_%g_w" % (k, c)] = ew2
            out["c%d_%g_n" % (k, c)] = np.int32(n)
            out["c%d_%g_labels" % (k, c)] = ref_clustering(ei2, ej2, ew2, n, c)
    np.savez_compressed(os.path.join(HERE, "clustering_ref.npz"), **out)

    sc = make_scene(8, 120, 6, seed=3)
    o = op.run_scene(sc, 6)
    g = {}
    for v in (0, 4):
        tr = o.trace[v]
        mv = tr["marshal"]
        for key in ("src_segs", "RtKinv_src", "C_src", "tgt_segs", "offsets", "F", "RtKinv", "centers", "P"):
            g["v%d_%s" % (v, key)] = mv[key]
        g["v%d_tbm" % v] = np.array(mv["tbm"], np.int32)
        g["v%d_l2g" % v] = np.array(mv["l2g"], np.uint32)
        g["v%d_scalars" % v] = np.array([mv["k_upper"], mv["k_lower"], mv["spatial_k"], tr["median"]], np.float32)
        g["v%d_in" % v] = tr["in_matches"]
        g["v%d_out" % v] = tr["matches"]
    segs = sc.views[0]["segments"]
    rel = op.collinearity(o.lib, segs, 2.0)
    ii, jj = np.nonzero(np.triu(rel > 0, 1))
    g["coll_segs"], g["coll_i"], g["coll_j"], g["coll_w"] = segs, ii.astype(np.int32), jj.astype(np.int32), rel[jj, ii]
    g["rdd_A"] = o.affinity
    g["rdd_n"] = np.int32(len(o.local2global))
    g["rdd_out"] = op.rdd(o.lib, o.affinity, len(o.local2global), 10)
    g["n_lines"] = np.int32(len(o.result))
    np.savez_compressed(os.path.join(HERE, "seam_small.npz"), **g)
    print("written", {k: os.path.getsize(os.path.join(HERE, k)) for k in os.listdir(HERE) if k.endswith(".npz")})


if __name__ == "__main__":
    main()
