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


def test_struct_layouts(tmp_path):
    """The Python mirrors of the C structs against the header itself: sizes and member offsets printed by a C program."""
    import ctypes
    import subprocess
    from line3d_amd import capi
    assert capi.MATCH_DTYPE.itemsize == 32 and capi.EDGE_DTYPE.itemsize == 12 and capi.HYP_DTYPE.itemsize == 96
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "line3d_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(l3d_match), sizeof(l3d_edge), sizeof(l3d_hypothesis),\n'
                   '  sizeof(l3d_affinity_input), offsetof(l3d_affinity_input, seg_base), offsetof(l3d_affinity_input, n_hyp), offsetof(l3d_affinity_input, hyp),\n'
                   '  offsetof(l3d_affinity_input, coll_w), offsetof(l3d_affinity_input, sigma_a)); return 0; }\n')
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    A = capi.AffinityInput
    assert got == [capi.MATCH_DTYPE.itemsize, capi.EDGE_DTYPE.itemsize, capi.HYP_DTYPE.itemsize, ctypes.sizeof(A), A.seg_base.offset, A.n_hyp.offset,
                   A.hyp.offset, A.coll_w.offset, A.sigma_a.offset], got


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


def test_host_clustering_large_lists_with_ties(oracle_lib):
    """The product orders large edge lists on several threads (bucket + sort, l3d_hostsort.hpp); the result must be the
    stable order of clustering.cc:14 whatever the list size: many equal weights, against the pinned oracle (and the
    reference's own clustering.cc when oracle/_ref is built)."""
    import numpy as np
    import l3d_oracle_pipeline as op
    from line3d_amd import capi
    lib = capi.load_library()
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libclustering_ref.so")
    ref = C.CDLL(ref_path) if os.path.exists(ref_path) else None
    rng = np.random.default_rng(7)
    for n, E, q in ((3000, 40000, 2), (20000, 300000, 3), (50, 70000, 1), (20000, 250000, -1)):
        e = np.zeros(E, dtype=capi.EDGE_DTYPE)
        e["i"], e["j"] = rng.integers(0, n, E), rng.integers(0, n, E)
        if q < 0:     # skewed: most weights inside one bucket of the major key, with repeats
            e["w"] = (0.997 + np.round(rng.random(E) * 0.002, 5)).astype(np.float32)
        else:
            e["w"] = np.where(rng.random(E) < 0.3, 1.0, np.round(rng.random(E), q)).astype(np.float32)
        labels = np.zeros(n, np.int32)
        rc = lib.l3d_perform_clustering(e.ctypes.data_as(C.c_void_p), C.c_int(E), C.c_int(n), C.c_float(1.0), labels.ctypes.data_as(C.c_void_p))
        assert rc == 0
        eo = np.zeros(E, dtype=op.EDGE_DTYPE)
        eo["i"], eo["j"], eo["w"] = e["i"], e["j"], e["w"]
        assert np.array_equal(labels, op.clustering(oracle_lib, eo, n, 1.0)), (n, E)
        if ref is not None:
            want = np.zeros(n, np.int32)
            ei, ej, ew = (np.ascontiguousarray(e[k]) for k in ("i", "j", "w"))
            assert ref.l3dref_clustering(ei.ctypes.data_as(C.c_void_p), ej.ctypes.data_as(C.c_void_p), ew.ctypes.data_as(C.c_void_p),
                                         C.c_int(E), C.c_int(n), C.c_float(1.0), want.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(labels, want), (n, E)


def test_the_environment_is_read_in_one_place_only():
    """Every L3D_* switch is read once, by l3d_ctx_create (l3d::options_from_env in l3d_capi.hip); afterwards switches change only
    through l3d_set_option.  No other getenv call anywhere in the product sources."""
    src = os.path.join(ROOT, "line3d_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".cpp", ".hpp")):
            for i, line in enumerate(open(os.path.join(src, f)), 1):
                code = line.split("//")[0]
                if re.search(r"\bgetenv\s*\(", code):
                    hits.append("%s:%d" % (f, i))
    assert len(hits) == 1 and hits[0].startswith("l3d_capi.hip:"), hits
