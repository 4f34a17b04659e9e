"""Generates tests/golden/rdd_ref.npz.  Run in the build container (needs oracle/_spliced/libkernels_spliced.so: `make -C oracle ref_devfn`):
    python tests/golden/make_golden_rdd.py
Data only: the affinity lists of tests/rdd_cases.py and the result of replicator_dynamics_diffusion (cudawrapper.cu:1131-1191) with the
arithmetic done by the REFERENCE's own kernels -- K_sparseMat_row_normalization and K_sparseMat_diffusion_step compiled from
cudawrapper.cu:717-829 -- inside the oracle's restatement of the host orchestration (sparsematrix.cc: sort orders, start indices; the
iteration loop).  Pins the oracle's two kernel restatements (and, through the bit-exact GPU parity tests, k_rdd_*) to the reference
wherever a test machine lacks oracle/_ref."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op  # noqa: E402
import rdd_cases as rc  # noqa: E402

if __name__ == "__main__":
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_spliced", "libkernels_spliced.so"))
    lib = op.load_lib()
    out = {}
    for k, case in enumerate(rc.CASES):
        A = rc.make_list(**case)
        for iters in (1, 10):
            W = op.rdd_hooked(lib, ref, A, case["n"], iters)
            out["c%d_it%d" % (k, iters)] = np.frombuffer(W.tobytes(), np.uint8).copy()       # (i, j, w) records, bit for bit
        out["c%d_in" % k] = np.frombuffer(A.tobytes(), np.uint8).copy()
    path = os.path.join(HERE, "rdd_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
