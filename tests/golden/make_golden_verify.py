"""Generates tests/golden/verify_ref.npz.  Run in the build container (needs oracle/_spliced/libkernels_spliced.so: `make -C oracle ref_devfn`):
    python tests/golden/make_golden_verify.py
Data only: the confidences the REFERENCE's own K_verify_matches (cudawrapper.cu:614-714, compiled from its text by oracle/make_ref_devfn.py:
everything but the five lines that fetch the source segment from a texture; its callee D_hypothesis_confidence is the reference's body,
D_project_point_tgt a restatement over a table) writes for the seeded candidate lists of tests/verify_cases.py (inputs are regenerated
from the seeds; a digest of them is stored)."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op  # noqa: E402
import verify_cases as vc  # noqa: E402


def digest(case):
    h = hashlib.sha256()
    for k in sorted(case):
        h.update(np.ascontiguousarray(case[k]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


if __name__ == "__main__":
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_spliced", "libkernels_spliced.so"))
    out = {}
    for k, kw in enumerate(vc.CASES):
        case = vc.make_case(**kw)
        out["c%d_conf" % k] = op.verify_case(None, case, ref)
        out["c%d_digest" % k] = digest(case)
    path = os.path.join(HERE, "verify_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
