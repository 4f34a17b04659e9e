"""Extended randomized end-to-end parity soak (development aid): many seeds of tests/test_gpu_pipeline_parity.py's random test."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import l3d_oracle_pipeline as op
from helpers import assert_lines_equal
import test_gpu_pipeline_parity as T
from line3d_amd.synth import make_scene
bad = 0
for seed in range(1000, 1000 + int(sys.argv[1]) if len(sys.argv) > 1 else 1030):
    rng = np.random.default_rng(seed)
    V, S, N = int(rng.integers(6, 14)), int(rng.integers(100, 300)), int(2 * rng.integers(2, 6))
    sc = make_scene(V, S, N, seed=seed, noise_px=float(rng.choice([0.3, 0.5, 1.5])), first_id=int(rng.choice([0, 3, 50])))
    for v in sc.views:
        keep = int(rng.integers(S // 2, S + 1))
        v["segments"] = np.ascontiguousarray(v["segments"][:keep]); v["gt"] = v["gt"][:keep]
    collin, diffusion = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    o = op.run_scene(sc, N, use_collinearity=collin, perform_diffusion=diffusion)
    l = T._run_gpu(sc, N, diffusion=diffusion, collin=collin)
    ok = True
    for v in sorted(o.trace):
        got, med = l.view_matches(v)
        ok &= got.tobytes() == o.trace[v]["matches"].tobytes() and np.float32(med) == np.float32(o.trace[v]["median"])
    A, nn = l.affinity()
    ok &= A.tobytes() == o.affinity.tobytes()
    try:
        assert_lines_equal(l.getResult(), o.result, 1e-4)
    except AssertionError as e:
        ok = False
    print("seed %d: V=%d S=%d N=%d collin=%d diff=%d edges=%d lines=%d %s" % (seed, V, S, N, collin, diffusion, len(A), len(o.result), "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    l.close()
print("mismatches:", bad)
