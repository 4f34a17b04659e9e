"""Generates tests/golden/devfn_ref.npz.  Run in the build container (needs oracle/_ref/libdevfn_ref.so, i.e. /root/reference
and the NVIDIA runtime headers of the triton wheel; `make -C oracle ref_devfn`):
    python tests/golden/make_golden_devfn.py
Data only: for every texture-free device function of the reference (cudawrapper.cu:56-61,93-99,116-141,165-285,337-344 and the
helper_math.h vector functions they call) the seeded inputs of tests/devfn_cases.py and the outputs of the REFERENCE's own
code compiled from its sources -> pins the oracle's restatements (and, through the bit-exact GPU parity tests, the kernels)
to the reference itself wherever a test machine lacks oracle/_ref.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import devfn_cases as dc  # noqa: E402

SEED, N = 20261, 3000

if __name__ == "__main__":
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdevfn_ref.so"))                    # unmodified reference text
    spliced = C.CDLL(os.path.join(ROOT, "oracle", "_spliced", "libkernels_spliced.so"))       # the three kernel bodies (dc.SPLICED): corroboration
    res = dc.run_reference(ref, spliced, dc.make_inputs(SEED, N))
    out = {"seed": np.int64(SEED), "n": np.int64(N), "sizeof_angle_acos": np.int64(ref.l3dref_sizeof_angle_acos())}
    for name, (ins, o) in res.items():
        for i, a in enumerate(ins):
            out["%s__in%d" % (name, i)] = a
        out["%s__out" % name] = o
    path = os.path.join(HERE, "devfn_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
