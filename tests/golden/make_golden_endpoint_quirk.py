#!/usr/bin/env python3
"""Golden inputs for tests/test_pair_bounds_cpu.py::test_bounds_on_the_pairs_that_fooled_them: four (source, target) pairs that level 2 of
k_pair_mask decided AGAINST the exact test before its bounds kept away from segment end points -- three of the 512 x 2000 x 12 bench scene
(seed 20260) that the first version of the ACCEPTS let through, one of 256 x 4000 x 24 that round 1's REJECT dropped.  Found on the GPU by the
diagnostic build -DL3D_BOUND_CHECK (`L3D_LIBRARY=<that build> L3D_PAIR_STATS=1 python scripts/bench_shape.py 512 2000 12 1`).
In all of them an intersection point lands on an end point of a segment within float noise; the point-on-segment tests of D_segment_overlap_2D
(cudawrapper.cu:135-141) then flip on that noise and the overlap comes out as 0 -- or as 3791 -- whatever the intervals are.  Stored: the workgroup's
64 source and 256 target segments (their extents set the margins), the fundamental matrix, the pair's position in the tile, the oracle's verdict
for every pair of the tile.

    python tests/golden/make_golden_endpoint_quirk.py        (oracle only; a few minutes: the scenes have 512 / 256 views)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op          # noqa: E402
from line3d_amd.synth import make_scene   # noqa: E402

CASES = (((512, 2000, 12, 20260), 440, 182, 6, 1693), ((512, 2000, 12, 20260), 450, 466, 11, 1917), ((512, 2000, 12, 20260), 509, 1896, 6, 858),
         ((256, 4000, 24, 20260), 215, 2204, 20, 2839))      # ((views, segments, neighbours, seed), view id, source segment, local camera, target segment)

lib = op.load_lib()
_scenes = {}


def oracle_of(shape):
    if shape not in _scenes:
        V, S, N, seed = shape
        sc = make_scene(V, S, N, seed=seed)
        o = op.OracleLine3D(matching_neighbors=N, use_collinearity=False)
        for v in sc.views:
            o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        o.matched = {}
        o.find_visual_neighbors()
        o.transform_geometry()
        _scenes.clear()                      # (one scene at a time: they are big)
        _scenes[shape] = o
    return _scenes[shape]


out = {}
for i, (shape, vid, src, cam, tgt) in enumerate(CASES):
    o = oracle_of(shape)
    for nb in o.visual_neighbors[vid]:
        o._fundamental(vid, nb)
    mv = o.marshal_view(vid)
    off, w = (int(x) for x in mv["offsets"][cam])
    s0, t0 = (src // 64) * 64, (tgt // 256) * 256
    S = np.ascontiguousarray(mv["src_segs"][s0:s0 + 64], np.float32)
    T = np.ascontiguousarray(mv["tgt_segs"][off + t0:off + min(w, t0 + 256)], np.float32)
    buf = op.pairwise_dense(lib, S, mv["RtKinv_src"], mv["C_src"], T, 0, len(T), cam, mv["F"], mv["RtKinv"], mv["centers"])
    exact = np.any(buf.reshape(len(S), len(T), 4) != 0, axis=2)
    out["src_%d" % i], out["tgt_%d" % i], out["F_%d" % i] = S, T, np.ascontiguousarray(mv["F"][cam], np.float32)
    out["pair_%d" % i] = np.array([src - s0, tgt - t0], np.int32)
    out["exact_%d" % i] = exact
    print("case %d: %s view %d source %d camera %d target %d -> the exact test %s it" % (i, "x".join(str(x) for x in shape[:3]), vid, src, cam, tgt, "KEEPS" if exact[src - s0, tgt - t0] else "rejects"))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "endpoint_quirk_pairs.npz"), **out)
