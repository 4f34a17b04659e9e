"""Generates tests/golden/pairwise_ref.npz.  Run in the build container (needs oracle/_spliced/libkernels_spliced.so: `make -C oracle ref_devfn`):
    python tests/golden/make_golden_pairwise.py
Data only: for two small seeded scenes (a helix, cameras that face each other) every non-zero entry of the dense buffers the REFERENCE's own
K_pairwise_matches writes (cudawrapper.cu:538-611 + D_get_triangulation_depth :304-335 compiled from its text by oracle/make_ref_devfn.py; texture
fetches replaced by table reads, D_epipolar_line / D_get_ray_tgt restated over tables) for every view and every neighbour camera: (view, camera, source
segment, target segment, four depths); and the non-zero entries of the collinearity relation K_collinearity (:476-535, same build) gives for the
planted segments of tests/devfn_cases.py::collinear_segments."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op  # noqa: E402
from line3d_amd.synth import make_scene, make_scene_from_poses  # noqa: E402


def scenes():
    return [("helix", make_scene(5, 90, 4, seed=5), 4),
            ("opposing", make_scene_from_poses([(4, 0, 0), (-4, 0.2, 0.3), (0, 0.1, 4), (1.5, 0.2, 0.1)], [(0, 0, 0)] * 4, 70, seed=5), 3)]


def entries(o, lib, reference):
    idx, val = [], []
    for v in sorted(o.trace):
        mv = o.trace[v]["marshal"]
        for cam in range(len(mv["offsets"])):
            buf = op.pairwise_dense_view(lib, mv, cam, reference)
            ys, xs = np.nonzero(np.any(buf != 0, axis=2))
            idx += [(v, cam, int(y), int(x)) for y, x in zip(ys, xs)]
            val.append(buf[ys, xs])
    return np.array(idx, np.int32).reshape(-1, 4), np.concatenate(val).astype(np.float32) if val else np.zeros((0, 4), np.float32)


if __name__ == "__main__":
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_spliced", "libkernels_spliced.so"))
    lib = op.load_lib()
    out = {}
    for name, scene, N in scenes():
        o = op.run_scene(scene, N)
        out[name + "_idx"], out[name + "_val"] = entries(o, lib, ref)
    # K_collinearity (cudawrapper.cu:476-535, the kernel's text with its texture fetches replaced by table reads): the non-zero entries of the relation
    import devfn_cases as dc
    segs = dc.collinear_segments(31)
    S = len(segs)
    rel = np.zeros((S, S), np.float32)
    ref.l3dref_collinearity(rel.ctypes.data_as(C.c_void_p), C.c_int(S), C.c_float(2.5 * 2.5), C.c_int(S), segs.ctypes.data_as(C.c_void_p))
    ys, xs = np.nonzero(rel)
    out["coll_idx"] = np.stack([ys, xs], 1).astype(np.int32)
    out["coll_val"] = rel[ys, xs]
    path = os.path.join(HERE, "pairwise_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", {k: v.shape for k, v in out.items()})
