"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares
(no compute calls here); creating a context without a GPU fails loudly instead of falling back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for f in os.listdir(os.path.join(ROOT, "include")):
        if f.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(l3d_[a-z0-9_A-Z]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from line3d_amd import capi
    lib = capi.load_library()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts():
    from line3d_amd import capi
    assert capi.MATCH_DTYPE.itemsize == 32 and capi.EDGE_DTYPE.itemsize == 12 and capi.HYP_DTYPE.itemsize == 96


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from line3d_amd import capi
    from line3d_amd.pipeline import Line3D
    with pytest.raises(capi.L3DError):
        capi.Context(0)
    with pytest.raises(capi.L3DError):
        Line3D("")


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may reach into oracle/."""
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "line3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(l for l in txt.splitlines() if "oracle" in l and not l.strip().startswith(("//", "#", "*", '"""')))
                assert "l3d_oracle" not in code and "oracle/" not in code, (f, code)


def test_host_clustering_matches_reference_golden():
    """a12 stays on the host: the product's performClustering against labels produced by the reference's own
    clustering.cc (tests/golden/clustering_ref.npz)."""
    import numpy as np
    from line3d_amd import capi
    lib = capi.load_library()
    g = np.load(os.path.join(ROOT, "tests", "golden", "clustering_ref.npz"))
    names = sorted({k.rsplit("_", 1)[0] for k in g.files})
    for nm in names:
        e = np.zeros(len(g[nm + "_i"]), dtype=capi.EDGE_DTYPE)
        e["i"], e["j"], e["w"] = g[nm + "_i"], g[nm + "_j"], g[nm + "_w"]
        n = int(g[nm + "_n"])
        labels = np.zeros(n, np.int32)
        rc = lib.l3d_perform_clustering(e.ctypes.data_as(C.c_void_p), C.c_int(len(e)), C.c_int(n), C.c_float(float(nm.split("_")[1])),
                                        labels.ctypes.data_as(C.c_void_p))
        assert rc == 0 and np.array_equal(labels, g[nm + "_labels"]), nm
