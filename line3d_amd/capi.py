"""ctypes binding of the C ABI (include/line3d_amd.h).  Plumbing only: every compute call lands in the
HIP library; if the library or a GPU is missing the constructor raises (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libline3d_amd.so")
if os.environ.get("L3D_LIBRARY"):      # A/B measurements: an alternative build of the library
    LIB_PATH = os.environ["L3D_LIBRARY"]

MATCH_DTYPE = np.dtype([("segID1", "<u4"), ("camID2", "<u4"), ("segID2", "<u4"),
                        ("depths", "<f4", (4,)), ("confidence", "<f4")])
EDGE_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("w", "<f4")])
HYP_DTYPE = np.dtype([("P1", "<f8", (3,)), ("P2", "<f8", (3,)), ("dir", "<f8", (3,)),
                      ("depth_p1", "<f4"), ("depth_p2", "<f4"),
                      ("k_lower", "<f4"), ("k_upper", "<f4"), ("median_depth", "<f4"), ("pad", "<u4")])
assert MATCH_DTYPE.itemsize == 32 and EDGE_DTYPE.itemsize == 12 and HYP_DTYPE.itemsize == 96

_lib = None
_lib_check = None
CHECK_LIB_PATH = os.path.join(_HERE, "libline3d_amd_check.so")


def load_library(crosschecks: bool = False):
    """dlopen libline3d_amd.so (built in-tree by __graft_entry__.build() / make -C line3d_amd/csrc).
    crosschecks=True (tests only): libline3d_amd_check.so, the same sources built with -DL3D_CROSSCHECKS -- the only build in which
    L3D_HOST_BOOKKEEPING / L3D_HOST_CLUSTERING / L3D_MATCH_SYNC force a host-side stage where the device stage would run."""
    global _lib, _lib_check
    if crosschecks:
        if _lib_check is None:
            load_library()
            if not os.path.exists(CHECK_LIB_PATH):
                raise RuntimeError("cross-check library not built: %s (make -C line3d_amd/csrc check)" % CHECK_LIB_PATH)
            lib = C.CDLL(CHECK_LIB_PATH)
            lib.l3d_last_error.restype = C.c_char_p
            lib.l3d_last_error.argtypes = [C.c_void_p]
            lib.l3d_profile_names.restype = C.c_char_p
            lib.l3d_free.argtypes = [C.c_void_p]
            lib.l3d_ctx_destroy.argtypes = [C.c_void_p]
            _lib_check = lib
        return _lib_check
    if _lib is None:
        try:
            # PyTorch-ROCm ships its own HIP runtime; when both live in one process (multi-GPU driver, tests) the
            # framework's copy has to be loaded first or torch.cuda reports no devices.  Plumbing only.
            import torch  # noqa: F401
        except Exception:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.l3d_last_error.restype = C.c_char_p
        lib.l3d_last_error.argtypes = [C.c_void_p]
        lib.l3d_profile_names.restype = C.c_char_p
        lib.l3d_free.argtypes = [C.c_void_p]
        lib.l3d_ctx_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


class L3DError(RuntimeError):
    pass


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class AffinityInput(C.Structure):
    """l3d_affinity_input (include/line3d_amd.h)"""
    _fields_ = [("n_views", C.c_int32), ("seg_base", C.c_void_p), ("view_hyp_begin", C.c_void_p), ("n_hyp", C.c_int32),
                ("hyp", C.c_void_p), ("score", C.c_void_p), ("hyp_dense", C.c_void_p), ("best", C.c_void_p),
                ("pot_start", C.c_void_p), ("pot_tgt", C.c_void_p), ("coll_start", C.c_void_p), ("coll_other", C.c_void_p),
                ("coll_w", C.c_void_p), ("sigma_a", C.c_float)]


class Context:
    """One GPU, one stream, grow-only device arenas (l3d_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.l3d_ctx_create(C.c_int(device), C.byref(h))
        if rc != 0:
            raise L3DError("l3d_ctx_create failed (code %d): no usable MI355X / HIP device -- this package has no CPU fallback" % rc)
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.l3d_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise L3DError("line3d_amd error %d: %s" % (rc, self.lib.l3d_last_error(self.h).decode()))

    # -- measurement --------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.lib.l3d_profile_enable(self.h, C.c_int(1 if on else 0)))

    def profile_only(self, name: str | None):
        """Bracket only this kernel with HIP events (None: all)."""
        self._chk(self.lib.l3d_profile_only(self.h, (name or "").encode()))

    def profile_reset(self):
        self._chk(self.lib.l3d_profile_reset(self.h))

    def profile_get(self, name: str):
        n = C.c_int64(0)
        ms = C.c_double(0)
        self._chk(self.lib.l3d_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_all(self):
        return {k: self.profile_get(k) for k in self.lib.l3d_profile_names().decode().split(";")}

    def test_sq_threshold(self, u: np.ndarray):
        u = np.ascontiguousarray(u, dtype=np.float32)
        a, b = np.zeros_like(u), np.zeros_like(u)
        self._chk(self.lib.l3d_test_sq_threshold(self.h, _p(u), C.c_int(len(u)), _p(a), _p(b)))
        return a, b

    def set_chain_capacities(self, cand_cap: int, arena_cap: int):
        self._chk(self.lib.l3d_set_chain_capacities(self.h, C.c_size_t(cand_cap), C.c_size_t(arena_cap)))

    def set_verify_lds_budget(self, nbytes: int):
        self._chk(self.lib.l3d_set_verify_lds_budget(C.c_size_t(nbytes)))

    def set_pair_pretest(self, on: bool):
        self._chk(self.lib.l3d_set_pair_pretest(self.h, C.c_int(3 if on is True else int(on))))

    def set_verify_mode(self, mode: int):
        self._chk(self.lib.l3d_set_verify_mode(self.h, C.c_int(mode)))

    def set_option(self, name: str, value: int):
        """A diagnostic / A-B switch of the context (l3d_options.hpp); the environment is read once, at context creation."""
        self._chk(self.lib.l3d_set_option(self.h, name.encode(), C.c_int(int(value))))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._chk(self.lib.l3d_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def last_stats(self):
        s = (C.c_double * 4)()
        self._chk(self.lib.l3d_last_stats(self.h, s))
        return list(s)

    def register_segments(self, segs: np.ndarray):
        assert segs.dtype == np.float32 and segs.flags.c_contiguous
        self._keep.append(segs)
        self._chk(self.lib.l3d_register_segments(self.h, _p(segs), C.c_int(len(segs))))

    # -- the three seam functions ---------------------------------------------------------------
    def compute_collinearity(self, segs, collin_s=2.0):
        segs = np.ascontiguousarray(segs, dtype=np.float32)
        oi, oj, ow = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        n = C.c_int(0)
        self._chk(self.lib.l3d_compute_collinearity(self.h, _p(segs), C.c_int(len(segs)), C.c_float(collin_s),
                                                    C.byref(oi), C.byref(oj), C.byref(ow), C.byref(n)))
        k = n.value
        i = np.ctypeslib.as_array(oi, (k,)).copy() if k else np.zeros(0, np.int32)
        j = np.ctypeslib.as_array(oj, (k,)).copy() if k else np.zeros(0, np.int32)
        w = np.ctypeslib.as_array(ow, (k,)).copy() if k else np.zeros(0, np.float32)
        for p in (oi, oj, ow):
            self.lib.l3d_free(p)
        return i, j, w

    def compute_collinearity_batch(self, seg_sets, collin_s=2.0):
        """l3d_compute_collinearity_batch: list of (S_v, 4) arrays -> list of (i, j, w) triplet arrays, one per set."""
        sets = [np.ascontiguousarray(s, dtype=np.float32).reshape(-1, 4) for s in seg_sets]
        n = len(sets)
        ptrs = (C.c_void_p * max(n, 1))(*[s.ctypes.data for s in sets])
        ns = np.array([len(s) for s in sets], np.int32)
        start = np.zeros(n + 1, np.int32)
        oi, oj, ow = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        self._chk(self.lib.l3d_compute_collinearity_batch(self.h, ptrs, _p(ns), C.c_int(n), C.c_float(collin_s),
                                                          C.byref(oi), C.byref(oj), C.byref(ow), _p(start)))
        total = int(start[n])
        i = np.ctypeslib.as_array(oi, (total,)).copy() if total else np.zeros(0, np.int32)
        j = np.ctypeslib.as_array(oj, (total,)).copy() if total else np.zeros(0, np.int32)
        w = np.ctypeslib.as_array(ow, (total,)).copy() if total else np.zeros(0, np.float32)
        for p in (oi, oj, ow):
            self.lib.l3d_free(p)
        return [(i[start[v]:start[v + 1]], j[start[v]:start[v + 1]], w[start[v]:start[v + 1]]) for v in range(n)]

    def compute_pairwise_matches(self, src_segs, RtKinv_src, C_src, tgt_segs, offsets, F, RtKinv, centers, P,
                                 to_be_matched, in_matches, local2global, k_upper, k_lower, sigma_p, sigma_a,
                                 spatial_k, median_depth=1.0, seg_range=None, want_best=False):
        def f32(a):
            return a if (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous) else np.ascontiguousarray(a, dtype=np.float32)
        src_segs, tgt_segs = f32(src_segs), f32(tgt_segs)
        RtKinv_src, C_src, F, RtKinv, centers, P = f32(RtKinv_src), f32(C_src), f32(F), f32(RtKinv), f32(centers), f32(P)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        tbm = np.ascontiguousarray(to_be_matched, dtype=np.int32)
        inm = np.ascontiguousarray(in_matches, dtype=MATCH_DTYPE)
        l2g = np.ascontiguousarray(local2global, dtype=np.uint32)
        S = len(src_segs)
        s0, s1 = (0, S) if seg_range is None else seg_range
        out = C.c_void_p()
        n_out = C.c_int(0)
        med = C.c_float(median_depth)
        bd = C.POINTER(C.c_float)()
        nb = C.c_int(0)
        self._chk(self.lib.l3d_compute_pairwise_matches(
            self.h, _p(src_segs), C.c_int(S), _p(RtKinv_src), _p(C_src), _p(tgt_segs), _p(offsets), C.c_int(len(offsets)),
            _p(F), _p(RtKinv), _p(centers), _p(P), _p(tbm), C.c_int(len(tbm)), _p(inm), C.c_int(len(inm)), _p(l2g),
            C.c_float(k_upper), C.c_float(k_lower), C.c_float(sigma_p), C.c_float(sigma_a), C.c_float(spatial_k),
            C.c_int(s0), C.c_int(s1), C.byref(out), C.byref(n_out), C.byref(med), C.byref(bd), C.byref(nb)))
        n = n_out.value
        res = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            C.memmove(res.ctypes.data, out, n * 32)
        self.lib.l3d_free(out)
        best = np.ctypeslib.as_array(bd, (nb.value * 2,)).copy() if nb.value else np.zeros(0, np.float32)
        if bd:
            self.lib.l3d_free(bd)
        if want_best:
            return res, med.value, best
        return res, med.value

    def replicator_dynamics_diffusion(self, edges, n, iters=10):
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        out = np.zeros(len(edges), dtype=EDGE_DTYPE)
        self._chk(self.lib.l3d_replicator_dynamics_diffusion(self.h, _p(edges), C.c_int(len(edges)), C.c_int(n),
                                                             C.c_int(iters), _p(out)))
        return out

    def similarity_coll3D_batch(self, hyp, pairs, sigma_a):
        hyp = np.ascontiguousarray(hyp, dtype=HYP_DTYPE)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        sim = np.zeros(len(pairs), dtype=np.float32)
        self._chk(self.lib.l3d_similarity_coll3D_batch(self.h, _p(hyp), C.c_int(len(hyp)), _p(pairs), C.c_int(len(pairs)),
                                                       C.c_float(sigma_a), _p(sim)))
        return sim

    def affinity_fill(self, seg_base, view_hyp_begin, hyp, score, hyp_dense, best, pot_start, pot_tgt, coll_start, coll_other, coll_w, sigma_a):
        """l3d_affinity_fill: flat tables -> (edges EDGE_DTYPE, node_hyp int32, number of enumerated candidate pairs)."""
        In = AffinityInput
        arrs = [np.ascontiguousarray(seg_base, np.int32), np.ascontiguousarray(view_hyp_begin, np.int32), np.ascontiguousarray(hyp, HYP_DTYPE),
                np.ascontiguousarray(score, np.float32), np.ascontiguousarray(hyp_dense, np.int32), np.ascontiguousarray(best, np.int32),
                np.ascontiguousarray(pot_start, np.int64), np.ascontiguousarray(pot_tgt, np.int32), np.ascontiguousarray(coll_start, np.int64),
                np.ascontiguousarray(coll_other, np.int32), np.ascontiguousarray(coll_w, np.float32)]
        ptr = [a.ctypes.data_as(C.c_void_p) for a in arrs]
        inp = In(len(arrs[0]) - 1, ptr[0], ptr[1], len(arrs[2]), ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], ptr[7], ptr[8], ptr[9], ptr[10], float(sigma_a))
        edges, nodes = C.c_void_p(), C.c_void_p()
        ne, nn, nc = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.lib.l3d_affinity_fill(self.h, C.byref(inp), C.byref(edges), C.byref(ne), C.byref(nodes), C.byref(nn), C.byref(nc)))
        A = np.zeros(ne.value, dtype=EDGE_DTYPE)
        node_hyp = np.zeros(nn.value, dtype=np.int32)
        if ne.value:
            C.memmove(A.ctypes.data, edges, ne.value * 12)
        if nn.value:
            C.memmove(node_hyp.ctypes.data, nodes, nn.value * 4)
        self.lib.l3d_free(edges)
        self.lib.l3d_free(nodes)
        return A, node_hyp, nc.value

    def chain_kept_list(self, index: int):
        """l3d_chain_kept_list: the kept list of chain view `index` out of the resident arena (MATCH_DTYPE array); partitioned products: the
        views this rank holds, empty for the others"""
        p, n = C.c_void_p(), C.c_int(0)
        self._chk(self.lib.l3d_chain_kept_list(self.h, C.c_int(index), C.byref(p), C.byref(n)))
        out = np.zeros(n.value, dtype=MATCH_DTYPE)
        if n.value:
            C.memmove(out.ctypes.data, p, n.value * 32)
        self.lib.l3d_free(p)
        return out

    def last_fill_counts(self):
        """(candidate pairs enumerated, candidates that passed their threshold) of the last affinity fill on this context, as 64-bit counts"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.l3d_last_fill_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def clustering_edges(self, edges, n_nodes, perform_diffusion=False, iters=10):
        """l3d_clustering_edges: (diffused, symmetrised) edge list in performClustering's stable ascending weight order."""
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        out = np.zeros(len(edges), dtype=EDGE_DTYPE)
        self._chk(self.lib.l3d_clustering_edges(self.h, _p(edges), C.c_int(len(edges)), C.c_int(n_nodes), C.c_int(int(perform_diffusion)),
                                                C.c_int(iters), _p(out)))
        return out

    def clustering_edges_grouped(self, edges, n_nodes, perform_diffusion=False, iters=10):
        """l3d_clustering_edges_grouped: -> (edges grouped by connected component, stable ascending weight inside a group; group_start)."""
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        out = np.zeros(len(edges), dtype=EDGE_DTYPE)
        gs = C.POINTER(C.c_int32)()
        ng = C.c_int(0)
        self._chk(self.lib.l3d_clustering_edges_grouped(self.h, _p(edges), C.c_int(len(edges)), C.c_int(n_nodes), C.c_int(int(perform_diffusion)),
                                                        C.c_int(iters), _p(out), C.byref(gs), C.byref(ng)))
        start = np.ctypeslib.as_array(gs, (ng.value + 1,)).copy() if ng.value else np.zeros(1, np.int32)
        self.lib.l3d_free(gs)
        return out, start

    def perform_clustering_device(self, edges, n_nodes, c=1.0, perform_diffusion=False, iters=10):
        """l3d_perform_clustering_device: [diffusion +] performClustering's merge loop on the device -> (labels (n_nodes,), components with an edge)."""
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        labels = np.full(max(n_nodes, 1), -1, dtype=np.int32)
        nc = C.c_int(0)
        self._chk(self.lib.l3d_perform_clustering_device(self.h, _p(edges), C.c_int(len(edges)), C.c_int(n_nodes), C.c_int(int(perform_diffusion)),
                                                         C.c_int(iters), C.c_float(c), _p(labels), C.byref(nc)))
        return labels[:n_nodes], nc.value

    def fit_clusters(self, group_start, member_hyp, hyp, hyp_cam, Rinv, scale_inv, tneg):
        """l3d_fit_clusters: -> list (one per cluster) of lists of (start (3,), end (3,)) float64."""
        gs = np.ascontiguousarray(group_start, np.int32)
        mh = np.ascontiguousarray(member_hyp, np.int32)
        hy = np.ascontiguousarray(hyp, HYP_DTYPE)
        hc = np.ascontiguousarray(hyp_cam, np.uint32)
        R = np.ascontiguousarray(Rinv, np.float64).reshape(9)
        t = np.ascontiguousarray(tneg, np.float64).reshape(3)
        cnt, segs = C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        n = C.c_int(0)
        ng = len(gs) - 1
        self._chk(self.lib.l3d_fit_clusters(self.h, _p(gs), C.c_int(ng), _p(mh), _p(hy), _p(hc), C.c_int(len(hy)), _p(R), C.c_double(scale_inv), _p(t),
                                            C.byref(cnt), C.byref(segs), C.byref(n)))
        counts = np.ctypeslib.as_array(cnt, (ng,)).copy() if ng else np.zeros(0, np.int32)
        flat = np.ctypeslib.as_array(segs, (n.value * 6,)).copy().reshape(-1, 6) if n.value else np.zeros((0, 6))
        self.lib.l3d_free(cnt)
        self.lib.l3d_free(segs)
        out, k = [], 0
        for c in counts:
            out.append([(flat[k + i, :3].copy(), flat[k + i, 3:].copy()) for i in range(c)])
            k += int(c)
        return out

    def fit_labelled_clusters(self, labels, node_hyp, hyp, hyp_cam, Rinv, scale_inv, tneg):
        """l3d_fit_labelled_clusters: -> (group_start, member_hyp, list per fitted cluster of (start, end))."""
        lab = np.ascontiguousarray(labels, np.int32)
        nh = np.ascontiguousarray(node_hyp, np.int32)
        hy = np.ascontiguousarray(hyp, HYP_DTYPE)
        hc = np.ascontiguousarray(hyp_cam, np.uint32)
        R = np.ascontiguousarray(Rinv, np.float64).reshape(9)
        t = np.ascontiguousarray(tneg, np.float64).reshape(3)
        gs, mh, cnt, segs = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        ng, n = C.c_int(0), C.c_int(0)
        self._chk(self.lib.l3d_fit_labelled_clusters(self.h, _p(lab), _p(nh), C.c_int(len(lab)), _p(hy), _p(hc), C.c_int(len(hy)), _p(R), C.c_double(scale_inv), _p(t),
                                                     C.byref(gs), C.byref(mh), C.byref(ng), C.byref(cnt), C.byref(segs), C.byref(n)))
        g = ng.value
        group_start = np.ctypeslib.as_array(gs, (g + 1,)).copy() if g else np.zeros(1, np.int32)
        members = np.ctypeslib.as_array(mh, (int(group_start[-1]),)).copy() if g and group_start[-1] else np.zeros(0, np.int32)
        counts = np.ctypeslib.as_array(cnt, (g,)).copy() if g else np.zeros(0, np.int32)
        flat = np.ctypeslib.as_array(segs, (n.value * 6,)).copy().reshape(-1, 6) if n.value else np.zeros((0, 6))
        for q in (gs, mh, cnt, segs):
            self.lib.l3d_free(q)
        out, k = [], 0
        for c in counts:
            out.append([(flat[k + i, :3].copy(), flat[k + i, 3:].copy()) for i in range(c)])
            k += int(c)
        return group_start, members, out

    def test_contract_math(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        e = np.zeros(len(x), np.float32)
        ac = np.zeros(len(x), np.float32)
        acd = np.zeros(len(x), np.float64)
        self._chk(self.lib.l3d_test_contract_math(self.h, _p(x), C.c_int(len(x)), _p(e), _p(ac), _p(acd)))
        return e, ac, acd
