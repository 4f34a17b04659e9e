import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import l3d_oracle_pipeline as op
from helpers import assert_lines_equal
import test_gpu_pipeline_parity as T
from line3d_amd.synth import make_scene
for (V, S, N) in ((36, 90, 30), (60, 60, 56), (24, 120, 20)):
    sc = make_scene(V, S, N, seed=V * 7 + N)
    o = op.run_scene(sc, N, perform_diffusion=False)
    l = T._run_gpu(sc, N, diffusion=False)
    ok = True
    for v in sorted(o.trace):
        got, med = l.view_matches(v)
        ok &= got.tobytes() == o.trace[v]["matches"].tobytes() and np.float32(med) == np.float32(o.trace[v]["median"])
    A, nn = l.affinity()
    ok &= A.tobytes() == o.affinity.tobytes()
    assert_lines_equal(l.getResult(), o.result, 1e-4)
    print("V=%d S=%d N=%d edges=%d lines=%d %s" % (V, S, N, len(A), len(o.result), "ok" if ok else "MISMATCH"), flush=True)
    l.close()
