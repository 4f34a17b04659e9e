"""oracle/l3d_oracle_pipeline.py -- TEST INFRASTRUCTURE ONLY.

Python restatement of the HOST side of Line3D's hot path (line3D.cc, view.cc), driving the C
restatement of the device arithmetic (oracle/l3d_oracle.c) through ctypes.  Pure-Python loops:
only for small scenes (config-1 size).  Every method cites the reference file:line it follows.

PARITY PIN STATUS: "parity unpinned" by the reference for everything except graph segmentation
(see the header of oracle/l3d_oracle.c).  Eigen (JacobiSVD, inverse) is replaced by numpy.linalg;
the sign of the dominant singular vector in getLineEquation3D is fixed (largest |component| > 0)
because Eigen's sign is an implementation detail the reference never pins.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the reference compiled here (unmodified text behind an extern "C" door) / the kernels with builder-written splices (corroboration only)
REF_DEVFN = os.path.join(_HERE, "_ref", "libdevfn_ref.so")
SPLICED_KERNELS = os.path.join(_HERE, "_spliced", "libkernels_spliced.so")


class Match(C.Structure):
    """L3DMatchingPair, sparsematrix.h:37-65 (active_ dropped: always true on this path)."""
    _fields_ = [("segID1", C.c_uint32), ("camID2", C.c_uint32), ("segID2", C.c_uint32),
                ("depths", C.c_float * 4), ("confidence", C.c_float)]


class Edge(C.Structure):
    """CLEdge, clustering.h:57-61."""
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("w", C.c_float)]


MATCH_DTYPE = np.dtype([("segID1", "<u4"), ("camID2", "<u4"), ("segID2", "<u4"),
                        ("depths", "<f4", (4,)), ("confidence", "<f4")])
EDGE_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("w", "<f4")])

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _f(a):
    return a.ctypes.data_as(_fp)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def load_lib(libm: bool = False):
    name = "libl3d_oracle_libm.so" if libm else "libl3d_oracle.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError("oracle library missing: run `make -C oracle` (%s)" % path)
    lib = C.CDLL(path)
    lib.l3do_similarity_coll3D.restype = C.c_float
    lib.l3do_spatial_uncertainty_k.restype = C.c_double
    lib.l3do_test_expf.restype = C.c_float
    lib.l3do_test_expf.argtypes = [C.c_float]
    lib.l3do_test_acosf.restype = C.c_float
    lib.l3do_test_acosf.argtypes = [C.c_float]
    lib.l3do_test_acos.restype = C.c_double
    lib.l3do_test_acos.argtypes = [C.c_double]
    lib.l3do_spatial_uncertainty_k.argtypes = [_dp, _dp, C.c_double, C.c_double, C.c_double]
    lib.l3do_similarity_coll3D.argtypes = [_dp, _fp, _fp, _dp, _fp, _fp, C.c_float]
    lib.l3do_free.argtypes = [C.c_void_p]
    return lib


# ----------------------------------------------------------------------------------------------
# thin numpy wrappers around the C oracle
# ----------------------------------------------------------------------------------------------
def collinearity(lib, segs: np.ndarray, collin_s: float = 2.0) -> np.ndarray:
    segs = np.ascontiguousarray(segs, dtype=np.float32)
    S = len(segs)
    rel = np.zeros((S, S), dtype=np.float32)
    lib.l3do_collinearity(_f(segs), C.c_int(S), C.c_float(collin_s), _f(rel))
    return rel


def pairwise_dense(lib, src_segs, RtKinv_src, C_src, tgt_segs, offset, width, cam, F, RtKinv, centers):
    buf = np.zeros((len(src_segs), width, 4), dtype=np.float32)
    lib.l3do_pairwise_dense(_f(src_segs), C.c_int(len(src_segs)), _f(RtKinv_src), _f(C_src), _f(tgt_segs),
                            C.c_int(offset), C.c_int(width), C.c_int(cam), _f(F), _f(RtKinv), _f(centers), _f(buf))
    return buf


def compute_pairwise_matches(lib, src_segs, RtKinv_src, C_src, tgt_segs, offsets, F, RtKinv, centers, P,
                             to_be_matched, in_matches, local2global, k_upper, k_lower, sigma_p, sigma_a,
                             spatial_k, median_depth=1.0, seg_range=None, want_stats=False, want_best=False):
    """cudawrapper.cu:858-1128.  Arrays are float32 row-major; in_matches is a MATCH_DTYPE array with
    LOCAL camera ids.  Returns (matches MATCH_DTYPE array, median_depth[, stats])."""
    src_segs = np.ascontiguousarray(src_segs, dtype=np.float32)
    tgt_segs = np.ascontiguousarray(tgt_segs, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    F = np.ascontiguousarray(F, dtype=np.float32)
    RtKinv = np.ascontiguousarray(RtKinv, dtype=np.float32)
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    RtKinv_src = np.ascontiguousarray(RtKinv_src, dtype=np.float32)
    C_src = np.ascontiguousarray(C_src, dtype=np.float32)
    tbm = np.ascontiguousarray(to_be_matched, dtype=np.int32)
    inm = np.ascontiguousarray(in_matches, dtype=MATCH_DTYPE)
    l2g = np.ascontiguousarray(local2global, dtype=np.uint32)
    out = C.POINTER(Match)()
    n_out = C.c_int(0)
    med = C.c_float(median_depth)
    stats = np.zeros(4, dtype=np.float64)
    S = len(src_segs)
    s0, s1 = (0, S) if seg_range is None else seg_range
    best = np.zeros(2 * S + 2, dtype=np.float32)
    n_best = C.c_int(0)
    rc = lib.l3do_compute_pairwise_matches(
        _f(src_segs), C.c_int(S), _f(RtKinv_src), _f(C_src), _f(tgt_segs), _i(offsets), C.c_int(len(offsets)),
        _f(F), _f(RtKinv), _f(centers), _f(P), _i(tbm), C.c_int(len(tbm)),
        inm.ctypes.data_as(C.POINTER(Match)), C.c_int(len(inm)), l2g.ctypes.data_as(C.POINTER(C.c_uint32)),
        C.c_float(k_upper), C.c_float(k_lower), C.c_float(sigma_p), C.c_float(sigma_a), C.c_float(spatial_k),
        C.c_int(s0), C.c_int(s1), C.byref(out), C.byref(n_out), C.byref(med), _d(stats), _f(best), C.byref(n_best))
    assert rc == 0
    n = n_out.value
    res = np.zeros(n, dtype=MATCH_DTYPE)
    if n:
        C.memmove(res.ctypes.data, out, n * C.sizeof(Match))
    lib.l3do_free(out)
    if want_best:
        return res, med.value, best[:2 * n_best.value].copy()
    if want_stats:
        return res, med.value, stats
    return res, med.value


def rdd(lib, edges: np.ndarray, n: int, iters: int = 10) -> np.ndarray:
    edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    out = np.zeros(len(edges), dtype=EDGE_DTYPE)
    lib.l3do_rdd(edges.ctypes.data_as(C.POINTER(Edge)), C.c_int(len(edges)), C.c_int(n), C.c_int(iters),
                 out.ctypes.data_as(C.POINTER(Edge)))
    return out


def set_reference_kernels(lib, ref):
    """Run the oracle's host orchestration with the REFERENCE's own kernels (ref = oracle/_spliced/libkernels_spliced.so: K_collinearity, K_pairwise_matches,
    K_verify_matches, K_sparseMat_row_normalization, K_sparseMat_diffusion_step compiled from cudawrapper.cu's text); ref = None: back to the
    restatements.  Process-wide for that library."""
    if ref is None:
        lib.l3do_set_kernel_hooks(None, None, None, None, None)
        return
    ptr = lambda f: C.cast(f, C.c_void_p)
    lib.l3do_set_kernel_hooks(ptr(ref.l3dref_collinearity), ptr(ref.l3dref_pairwise_matches), ptr(ref.l3dref_verify_matches),
                              ptr(ref.l3dref_sparse_row_normalization), ptr(ref.l3dref_sparse_diffusion_step))


def pairwise_dense_view(lib, mv, cam, reference=None):
    """The dense S x width float4 buffer K_pairwise_matches fills for neighbour `cam` of a marshalled view (marshal_view): the oracle's
    l3do_pairwise_dense, or -- reference = oracle/_spliced/libkernels_spliced.so -- the reference's kernel text with table reads for its texture fetches (l3dref_pairwise_matches)."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    src, tgt, Rs, Cs, F, R, Cn = f(mv["src_segs"]), f(mv["tgt_segs"]), f(mv["RtKinv_src"]), f(mv["C_src"]), f(mv["F"]), f(mv["RtKinv"]), f(mv["centers"])
    off, w, S = int(mv["offsets"][cam][0]), int(mv["offsets"][cam][1]), len(src)
    out = np.zeros((S, w, 4), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    if reference is None:
        lib.l3do_pairwise_dense(p(src), C.c_int(S), p(Rs), p(Cs), p(tgt), C.c_int(off), C.c_int(w), C.c_int(cam), p(F), p(R), p(Cn), p(out))
    else:
        reference.l3dref_pairwise_matches(p(out), C.c_int(w), C.c_int(S), p(Rs), C.c_int(3), C.c_int(off), C.c_int(cam), p(Cs), C.c_int(w), p(src), p(tgt),
                                          p(F), p(R), p(Cn), C.c_int(0), C.c_int(S))
    return out


def verify_case(lib, case, reference=None):
    """K_verify_matches on a packed candidate list (tests/verify_cases.py): the oracle's l3do_verify, or -- reference = oracle/_spliced/libkernels_spliced.so --
    the reference's kernel text with table reads for its texture fetches (l3dref_verify_matches).  Returns the confidences (the .w of matches_data)."""
    md = np.ascontiguousarray(case["matches_data"], np.float32).copy()
    R = len(md)
    f = lambda a, dt=np.float32: np.ascontiguousarray(a, dt)
    dep, mo, co = f(case["matches_depths"]), f(case["match_offsets"], np.int32), f(case["camera_offsets"], np.int32)
    src, Rk, Cs, tgt, P = f(case["src_segs"]), f(case["RtKinv"]), f(case["C_src"]), f(case["tgt_segs"]), f(case["P"])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sp, sa, sk = C.c_float(float(case["sigma_p"])), C.c_float(float(case["sigma_a"])), C.c_float(float(case["spatial_k"]))
    if reference is None:
        lib.l3do_verify(p(md), p(dep), p(mo), p(co), C.c_int(R), p(src), p(Rk), p(Cs), p(tgt), p(P), sp, sa, sk, C.c_int(0), C.c_int(R))
    else:
        reference.l3dref_verify_matches(p(md), p(dep), p(mo), p(co), C.c_int(R), p(src), p(Rk), C.c_int(3), p(Cs), p(tgt), p(P), sp, sa, sk)
    return md[:, 3].copy()


def rdd_hooked(lib, ref, edges: np.ndarray, n: int, iters: int = 10) -> np.ndarray:
    """l3do_rdd_hooked with the two kernels of `ref` (oracle/_spliced/libkernels_spliced.so: the reference's own K_sparseMat_row_normalization and
    K_sparseMat_diffusion_step, cudawrapper.cu:717-829) in place of the oracle's restatements."""
    edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    out = np.zeros(len(edges), dtype=EDGE_DTYPE)
    norm = C.cast(ref.l3dref_sparse_row_normalization, C.c_void_p)
    step = C.cast(ref.l3dref_sparse_diffusion_step, C.c_void_p)
    lib.l3do_rdd_hooked(edges.ctypes.data_as(C.POINTER(Edge)), C.c_int(len(edges)), C.c_int(n), C.c_int(iters),
                        out.ctypes.data_as(C.POINTER(Edge)), norm, step)
    return out


def clustering(lib, edges: np.ndarray, num_nodes: int, c: float = 1.0) -> np.ndarray:
    edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    labels = np.zeros(num_nodes, dtype=np.int32)
    lib.l3do_clustering(edges.ctypes.data_as(C.POINTER(Edge)), C.c_int(len(edges)), C.c_int(num_nodes),
                        C.c_float(c), _i(labels))
    return labels


# ----------------------------------------------------------------------------------------------
class OracleView:
    """L3DView, view.h:40-153 / view.cc."""

    def __init__(self, lib, vid, segments, collin, K, R, t, width, height, unc_upper_px, unc_lower_px):
        self.lib = lib
        self.id = vid
        self.segments = np.ascontiguousarray(segments, dtype=np.float32)
        self.collin = collin                      # {seg: OrderedDict{other: w}} ascending, segments.h:84-97
        self.K = np.array(K, dtype=np.float64).reshape(3, 3)
        self.R = np.array(R, dtype=np.float64).reshape(3, 3)
        self.t = np.array(t, dtype=np.float64).reshape(3)
        self.width, self.height = width, height
        # view.cc:20-21: principal point is the image centre, not K's
        self.pp = (float(np.float32(width) / np.float32(2.0)), float(np.float32(height) / np.float32(2.0)))
        self.unc_upper_px = float(np.float32(unc_upper_px))
        self.unc_lower_px = float(np.float32(unc_lower_px))
        self.median_depth = np.float32(1.0)
        self.store = None                         # the "_raw.bin" match file: None = does not exist
        self.Kinv = np.zeros((3, 3))
        self.RtKinv = np.zeros((3, 3))
        self.C = np.zeros(3)
        self.P = np.zeros((3, 4))
        self._derive()

    def _derive(self):                            # view.cc:24-34 and :243-260
        self.lib.l3do_view_derive(_d(self.K), _d(self.R), _d(self.t), _d(self.Kinv), _d(self.RtKinv), _d(self.C), _d(self.P))
        self.k_upper = np.float32(self.specific_k(self.unc_upper_px))     # view.cc:119-120 (float members)
        self.k_lower = np.float32(self.specific_k(self.unc_lower_px))

    def specific_k(self, dist_px) -> float:       # view.cc:124-147
        return self.lib.l3do_spatial_uncertainty_k(_d(self.RtKinv), _d(self.C), self.pp[0], self.pp[1], float(dist_px))

    def transform(self, Qinv, scale):             # view.cc:227-261
        self.t = self.t * scale
        Rt = np.concatenate([self.R, self.t[:, None]], axis=1) @ Qinv
        self.R = np.ascontiguousarray(Rt[:, :3])
        self.t = np.ascontiguousarray(Rt[:, 3])
        self._derive()

    def baseline(self, other) -> np.float32:      # view.cc:446-449
        return np.float32(np.linalg.norm(self.C - other.C))

    def add_matches(self, matches, remove_old=False, only_best=False):   # view.cc:162-197
        matches = list(matches)
        if only_best:
            best = OrderedDict()
            for m in matches:
                best.setdefault(int(m["segID1"]), []).append(m)
            matches = []
            for seg in sorted(best):
                lst = sorted(best[seg], key=lambda m: -float(m["confidence"]))   # stable, desc
                matches.append(lst[0])
        if self.store is not None and not remove_old:
            self.store = self.store + matches
        else:
            self.store = matches

    def unproject_segment(self, seg_id, d1, d2):  # view.cc:302-342
        out = np.zeros(9, dtype=np.float64)
        self.lib.l3do_unproject_segment(_d(self.RtKinv), _d(self.C), _f(self.segments[seg_id]),
                                        C.c_float(d1), C.c_float(d2), _d(out))
        return out


class OracleLine3D:
    """L3D::Line3D, line3D.h:61-101 / line3D.cc."""

    def __init__(self, matching_neighbors=10, unc_upper=5.0, unc_lower=1.0, sigma_p=3.5, sigma_a=10.0,
                 min_baseline=0.25, use_collinearity=True, libm=False):
        self.lib = load_lib(libm)
        self.matching_neighbors = matching_neighbors
        self.unc_upper = abs(np.float32(unc_upper))            # line3D.cc:18-28
        self.unc_lower = abs(np.float32(unc_lower))
        if self.unc_lower < 1.0:
            self.unc_lower = np.float32(1.0)
        if self.unc_upper <= self.unc_lower:
            self.unc_upper = np.float32(self.unc_lower + np.float32(1.0))
        self.sigma_p = np.float32(sigma_p)
        self.sigma_a = np.float32(sigma_a)
        self.min_baseline = np.float32(min_baseline)
        self.use_collinearity = use_collinearity
        self.views = {}
        self.view_similarities = {}
        self.num_wps, self.common_wps, self.worldpoints2views = {}, {}, {}
        self.computation = False
        self.result = []
        self.trace = {}                                         # per-view kept matches etc. for tests

    # -- addImage (segments supplied: the detector is out of scope) ---------------------------
    def _make_view(self, vid, width, height, segments, K, R, t, cached_collin=None):
        segments = np.ascontiguousarray(segments, dtype=np.float32)
        collin = {}
        if self.use_collinearity and cached_collin is not None:  # line3D.cc:160-168: the L3DSegments of the cache file as it is
            collin = {k: OrderedDict(sorted(v.items())) for k, v in cached_collin.items()}
        elif self.use_collinearity and len(segments):           # segments.h:73-101
            rel = collinearity(self.lib, segments, 2.0)
            S = len(segments)
            ii, jj = np.nonzero(np.triu(rel > 0.0, 1))
            for a, b in zip(ii.tolist(), jj.tolist()):
                # loop order i<j ascending i then j; both directions inserted -> maps end up ascending
                collin.setdefault(a, {})[b] = rel[b, a]
                collin.setdefault(b, {})[a] = rel[b, a]
            collin = {k: OrderedDict(sorted(v.items())) for k, v in collin.items()}
            assert S == rel.shape[0]
        return OracleView(self.lib, vid, segments, collin, K, R, t, width, height, self.unc_upper, self.unc_lower)

    def add_image_fixed_sim(self, vid, width, height, segments, K, R, t, sims):   # line3D.cc:220-342
        if self.computation or vid in self.views or len(sims) == 0:
            return False
        self.views[vid] = self._make_view(vid, width, height, segments, K, R, t)
        for k, s in sims.items():                               # setViewSimilarity, 1938-1946
            if np.float32(s) > np.float32(0.01):
                self.view_similarities.setdefault(vid, {})[k] = np.float32(s)
        return True

    def add_image(self, vid, width, height, segments, K, R, t, worldpoint_ids):   # line3D.cc:95-217
        if self.computation or vid in self.views or len(worldpoint_ids) == 0:
            return False
        self.views[vid] = self._make_view(vid, width, height, segments, K, R, t)
        self._process_worldpoints(vid, worldpoint_ids)
        return True

    def add_image_cached(self, vid, width, height, cache_path, K, R, t, worldpoint_ids):   # line3D.cc:95-217 with :160-168
        from l3d_oracle_sfm import read_segment_cache
        if self.computation or vid in self.views or len(worldpoint_ids) == 0:
            return False
        segments, collin, _ = read_segment_cache(cache_path)
        if len(segments) == 0:
            return False
        self.views[vid] = self._make_view(vid, width, height, segments, K, R, t, cached_collin=collin)
        self._process_worldpoints(vid, worldpoint_ids)
        return True

    def _process_worldpoints(self, view_id, wps):               # line3D.cc:1874-1935
        self.num_wps[view_id] = 0
        cw = self.common_wps
        for wp in wps:
            w2v = self.worldpoints2views.setdefault(wp, {})
            if len(w2v) == 2:
                v1, v2 = sorted(w2v)
                cw.setdefault(v1, {})
                cw.setdefault(v2, {})
                cw[v1][v2] = cw[v1].get(v2, 0) + 1
                cw[v2][v1] = cw[v2].get(v1, 0) + 1
                self.num_wps[v1] = self.num_wps.get(v1, 0) + 1
                self.num_wps[v2] = self.num_wps.get(v2, 0) + 1
            if len(w2v) >= 2:
                for v in sorted(w2v):
                    cw.setdefault(v, {})
                    cw.setdefault(view_id, {})
                    cw[v][view_id] = cw[v].get(view_id, 0) + 1
                    cw[view_id][v] = cw[view_id].get(v, 0) + 1
                self.num_wps[view_id] += 1
            w2v[view_id] = True

    # -- compute3Dmodel, line3D.cc:345-374 ------------------------------------------------------
    def compute3Dmodel(self, perform_diffusion=False):
        if len(self.views) < 4:
            return False
        self.computation = True
        self.matched = {}
        self.potential = {}
        self.result = []
        self.find_visual_neighbors()
        self.transform_geometry()
        self.match_views()
        self.greedy_selection()
        self.cluster_segments_2D(perform_diffusion)
        return True

    def find_visual_neighbors(self):                            # line3D.cc:476-549
        self.visual_neighbors = {}
        for v in sorted(self.common_wps):
            if v in self.view_similarities:
                continue
            for n in sorted(self.common_wps[v]):
                sim = np.float32(2.0) * np.float32(self.common_wps[v][n]) / np.float32(self.num_wps.get(v, 0) + self.num_wps.get(n, 0))
                if sim > 1e-12:
                    self.view_similarities.setdefault(v, {})[n] = np.float32(sim)
        for v in sorted(self.view_similarities):
            vn = []
            for n in sorted(self.view_similarities[v]):
                if n in self.views and self.views[v].baseline(self.views[n]) > self.min_baseline:
                    ok = True
                    for (cam, _s) in vn:
                        if self.views[cam].baseline(self.views[n]) <= self.min_baseline:
                            ok = False
                            break
                    if ok:
                        vn.append((n, self.view_similarities[v][n]))
            vn.sort(key=lambda e: -float(e[1]))                # stable, descending similarity
            if self.matching_neighbors > 0 and len(vn) > self.matching_neighbors:
                vn = vn[:self.matching_neighbors]
            self.visual_neighbors[v] = sorted(cam for cam, _ in vn)

    def transform_geometry(self):                               # line3D.cc:552-617, 1694-1779
        self.fundamentals = {}
        ids = sorted(self.views)
        size = float(len(ids))
        in_pts = [self.views[i].C.copy() for i in ids]
        m = np.zeros(3)
        for p in in_pts:
            m = m + p
        m = m / size
        q = 0.0
        for p in in_pts:
            q += float(np.linalg.norm(p - m))
        q /= size
        q = float(np.sqrt(np.float32(2.0))) / q                # sqrtf(2.0)/q, line3D.cc:581
        out_pts = []
        cog_out = np.zeros(3)
        for p in in_pts:
            t3 = np.array([q * p[0] + (-q * m[0]), q * p[1] + (-q * m[1]), q * p[2] + (-q * m[2])])
            cog_out = cog_out + t3
            out_pts.append(t3)
        cog_out = cog_out / size
        # findSimilarityTransform
        n = len(in_pts)
        scales_sum = 0.0
        for i in range(n):
            d1 = float(np.linalg.norm(in_pts[i] - m))
            d2 = float(np.linalg.norm(out_pts[i] - cog_out))
            scales_sum += d2 / d1
        scale = scales_sum / float(n)
        cog_in = m * scale
        inp = [p * scale for p in in_pts]
        # euclideanTransformation
        inp = [p - cog_in for p in inp]
        outp = [p - cog_out for p in out_pts]
        H = np.zeros((3, 3))
        for i in range(n):
            H = H + np.outer(outp[i], inp[i])
        U, _S, Vt = np.linalg.svd(H)
        Rm = U @ Vt
        if np.linalg.det(Rm) < 0:
            Vt = Vt.copy()
            Vt[2, :] *= -1
            Rm = U @ Vt
        tt = cog_out - Rm @ cog_in
        tt = tt / scale
        # applyTransformation
        Q = np.eye(4)
        Q[:3, :3] = Rm
        Q[:3, 3] = tt * scale
        Qinv = np.linalg.inv(Q)
        self.transf_scale_inv = 1.0 / scale
        self.transf_Rinv = Rm.T.copy()
        self.transf_tneg = -tt
        for i in ids:
            self.views[i].transform(Qinv, scale)

    def inverse_transform(self, P):                             # line3D.cc:1782-1786
        return self.transf_Rinv @ (P * self.transf_scale_inv + self.transf_tneg)

    def _fundamental(self, a, b):                               # line3D.cc:1949-1993
        fa = self.fundamentals.setdefault(a, {})
        if b not in fa:
            va, vb = self.views[a], self.views[b]
            F = np.zeros((3, 3))
            self.lib.l3do_fundamental(_d(va.K), _d(va.R), _d(va.t), _d(vb.K), _d(vb.R), _d(vb.t), _d(F))
            fa[b] = F
            self.fundamentals.setdefault(b, {})[a] = np.ascontiguousarray(F.T)
        return fa[b]

    def match_views(self):                                      # line3D.cc:620-648
        for v in sorted(self.visual_neighbors):
            if len(self.visual_neighbors[v]) == 0:
                continue
            for n in self.visual_neighbors[v]:
                self._fundamental(v, n)
            self.perform_matching(v)

    def marshal_view(self, v):
        """line3D.cc:708-803: the arrays handed to compute_pairwise_matches for view v."""
        nbs = self.visual_neighbors[v]
        N = len(nbs)
        F = np.zeros((N, 3, 3), dtype=np.float32)
        RtKinv = np.zeros((N, 3, 3), dtype=np.float32)
        P = np.zeros((N, 3, 4), dtype=np.float32)
        centers = np.zeros((N, 3), dtype=np.float32)
        offsets = np.zeros((N, 2), dtype=np.int32)
        segs = []
        g2l, l2g, tbm = {}, [], []
        total = 0
        for loc, nb in enumerate(nbs):
            g2l[nb] = loc
            l2g.append(nb)
            if nb not in self.matched.get(v, {}):
                tbm.append(loc)
            F[loc] = self.fundamentals[v][nb].astype(np.float32)
            RtKinv[loc] = self.views[nb].RtKinv.astype(np.float32)
            P[loc] = self.views[nb].P.astype(np.float32)
            centers[loc] = self.views[nb].C.astype(np.float32)
            segs.append(self.views[nb].segments)
            offsets[loc] = (total, len(self.views[nb].segments))
            total += len(self.views[nb].segments)
        view = self.views[v]
        return dict(src_segs=view.segments, RtKinv_src=view.RtKinv.astype(np.float32), C_src=view.C.astype(np.float32),
                    tgt_segs=np.concatenate(segs, axis=0) if segs else np.zeros((0, 4), np.float32),
                    offsets=offsets, F=F, RtKinv=RtKinv, centers=centers, P=P, tbm=tbm, g2l=g2l, l2g=l2g,
                    k_upper=float(view.k_upper), k_lower=float(view.k_lower),
                    spatial_k=float(np.float32(view.specific_k(float(np.float32(2.0) * self.sigma_p)))))

    def existing_localized(self, v, mv):                        # loadAndLocalizeExistingMatches, view.cc:200-224
        view = self.views[v]
        existing = []
        if view.store is not None:
            for m in view.store:
                if int(m["camID2"]) in mv["g2l"]:
                    mm = m.copy()
                    mm["camID2"] = mv["g2l"][int(m["camID2"])]
                    existing.append(mm)
        return np.array(existing, dtype=MATCH_DTYPE) if existing else np.zeros(0, dtype=MATCH_DTYPE)

    def matching_compute(self, v, seg_range=None):              # line3D.cc:698-822 (up to the seam call)
        mv = self.marshal_view(v)
        in_arr = self.existing_localized(v, mv)
        matches, median, stats = compute_pairwise_matches(
            self.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"],
            mv["RtKinv"], mv["centers"], mv["P"], mv["tbm"], in_arr, mv["l2g"], mv["k_upper"], mv["k_lower"],
            float(self.sigma_p), float(self.sigma_a), mv["spatial_k"], median_depth=1.0, seg_range=seg_range,
            want_stats=True)
        return mv, in_arr, matches, median

    def matching_commit(self, v, matches, median):              # line3D.cc:834-884
        view = self.views[v]
        view.median_depth = np.float32(median)
        other = OrderedDict()
        for m in matches:                                       # 838-866
            cam = int(m["camID2"])
            if v in self.visual_neighbors.get(cam, []) and v not in self.matched.get(cam, {}):
                r = np.zeros((), dtype=MATCH_DTYPE)
                r["segID1"] = m["segID2"]
                r["camID2"] = v
                r["segID2"] = m["segID1"]
                r["confidence"] = 0.0
                r["depths"] = (m["depths"][2], m["depths"][3], m["depths"][0], m["depths"][1])
                other.setdefault(cam, []).append(r)
            if getattr(self, "track_potential", True):           # (False: matchViews' kept lists only -- the 512-view golden, whose 7e7 map entries would not fit)
                ref = (v, int(m["segID1"]))
                tgt = (cam, int(m["segID2"]))
                self.potential.setdefault(ref, {})[tgt] = True
                self.potential.setdefault(tgt, {})[ref] = True
        for cam in sorted(other):                               # 868-872
            self.views[cam].add_matches(other[cam])
        for nb in self.visual_neighbors[v]:                     # 875-881
            self.matched.setdefault(v, {})[nb] = True
            if v in self.visual_neighbors.get(nb, []):
                self.matched.setdefault(nb, {})[v] = True
        view.add_matches(list(matches), True, True)             # 884

    def perform_matching(self, v):                              # line3D.cc:698-885
        mv, in_arr, matches, median = self.matching_compute(v)
        self.trace[v] = dict(marshal=mv, in_matches=in_arr, matches=matches.copy(), median=median)
        self.matching_commit(v, matches, median)

    def greedy_selection(self):                                 # line3D.cc:899-965
        self.best_match = {}
        for v in sorted(self.views):
            view = self.views[v]
            local = view.store if view.store is not None else []
            per_seg = OrderedDict()
            for m in local:
                per_seg.setdefault(int(m["segID1"]), []).append(m)
            for seg in sorted(per_seg):
                lst = sorted(per_seg[seg], key=lambda m: -float(m["confidence"]))
                mp = lst[0]
                conf = min(np.float32(mp["confidence"]), np.float32(1.0))
                s3 = view.unproject_segment(seg, float(mp["depths"][0]), float(mp["depths"][1]))
                self.best_match[(v, seg)] = dict(score=np.float32(conf), seg3D=s3,
                                                 depths=np.array([mp["depths"][0], mp["depths"][1]], dtype=np.float32),
                                                 cam=v, seg=seg, tgt=(int(mp["camID2"]), int(mp["segID2"])))

    def _similarity(self, b1, b2) -> np.float32:                # line3D.cc:1600-1681
        v1, v2 = self.views[b1["cam"]], self.views[b2["cam"]]
        c1 = np.array([v1.k_lower, v1.k_upper, v1.median_depth], dtype=np.float32)
        c2 = np.array([v2.k_lower, v2.k_upper, v2.median_depth], dtype=np.float32)
        return np.float32(self.lib.l3do_similarity_coll3D(_d(b1["seg3D"]), _f(b1["depths"]), _f(c1),
                                                          _d(b2["seg3D"]), _f(b2["depths"]), _f(c2),
                                                          C.c_float(float(self.sigma_a))))

    def cluster_segments_2D(self, perform_diffusion):           # line3D.cc:968-1252
        A = []
        g2l, l2g = {}, {}
        used = set()

        def node(key):
            if key not in g2l:
                g2l[key] = len(g2l)
                l2g[g2l[key]] = key
            return g2l[key]

        def add_edge(src, tgt, w):
            a = node(src)
            b = node(tgt)
            A.append((a, b, w))
            A.append((b, a, w))

        half = np.float32(0.5)
        for src in sorted(self.best_match):
            Cs = self.best_match[src]
            pot = self.potential.setdefault(src, {})
            for tgt in sorted(pot):
                if (src, tgt) in used:
                    continue
                used.add((src, tgt))
                used.add((tgt, src))
                if tgt in self.best_match:
                    C2 = self.best_match[tgt]
                    w = np.float32(half * np.float32(Cs["score"] + C2["score"])) * self._similarity(Cs, C2)
                    if w > np.float32(0.25):
                        add_edge(src, tgt, w)
                    tview = self.views.get(tgt[0])
                    if tview is not None and tgt[1] in tview.collin:   # 1065 (NULL deref in the reference if absent)
                        for other in tview.collin[tgt[1]]:
                            tgtc = (tgt[0], other)
                            if (src, tgtc) in used:
                                continue
                            used.add((src, tgtc))
                            used.add((tgtc, src))
                            if tgtc in self.best_match:
                                C3 = self.best_match[tgtc]
                                w = np.float32(half * np.float32(Cs["score"] + C3["score"])) * self._similarity(Cs, C3)
                                if w > np.float32(0.01):
                                    add_edge(src, tgtc, w)
            sview = self.views[src[0]]
            if src[1] in sview.collin:                          # 1141-1214
                for sID, collin_w in sview.collin[src[1]].items():
                    tgt = (src[0], sID)
                    if (src, tgt) in used:
                        continue
                    used.add((src, tgt))
                    used.add((tgt, src))
                    if tgt in self.best_match:
                        C2 = self.best_match[tgt]
                        w = np.float32(np.float32(np.float32(collin_w) * half) * np.float32(Cs["score"] + C2["score"])) * self._similarity(Cs, C2)
                        if w > np.float32(0.01):
                            add_edge(src, tgt, w)
        self.affinity = np.array(A, dtype=EDGE_DTYPE) if A else np.zeros(0, dtype=EDGE_DTYPE)
        self.local2global = l2g
        if len(A) == 0:
            return
        edges = self.affinity
        if perform_diffusion:
            edges = self.perform_diffusion(edges, len(l2g))
        self.affinity_final = edges
        labels = clustering(self.lib, edges, len(l2g), 1.0)
        self.labels = labels
        self.process_clustered_segments(labels, l2g)

    def perform_diffusion(self, A, n):                          # line3D.cc:1255-1303
        W = rdd(self.lib, A, n, 10)
        entries = {}
        for e in W:
            s1, s2, w12 = int(e["i"]), int(e["j"]), np.float32(e["w"])
            w21 = w12
            if s1 in entries.get(s2, {}):
                w21 = entries[s2][s1]
            entries.setdefault(s2, {})
            w = min(w12, w21)
            entries.setdefault(s1, {})[s2] = w
            entries[s2][s1] = w
        out = []
        for a in sorted(entries):
            for b in sorted(entries[a]):
                out.append((a, b, entries[a][b]))
        return np.array(out, dtype=EDGE_DTYPE)

    def process_clustered_segments(self, labels, l2g):          # line3D.cc:1306-1368
        cl2seg, cl2cam = OrderedDict(), {}
        for lid in sorted(l2g):
            cl = int(labels[lid])
            cl2seg.setdefault(cl, []).append(l2g[lid])
            cl2cam.setdefault(cl, set()).add(l2g[lid][0])
        self.result = []
        for cl in sorted(cl2seg):
            if len(cl2cam[cl]) >= 4:
                t3 = OrderedDict()
                for seg in sorted(cl2seg[cl]):                  # std::map keyed by L3DSegment2D
                    if seg in self.best_match:
                        s3 = self.best_match[seg]["seg3D"]
                        t3[seg] = (self.inverse_transform(s3[0:3]), self.inverse_transform(s3[3:6]))
                segs3D = self.align(t3)
                if len(segs3D) > 0:
                    self.result.append((list(t3.keys()), segs3D))

    def align(self, t3):                                        # line3D.cc:1392-1597
        if len(t3) == 0:
            return []
        pts = []
        for (P1, P2) in t3.values():
            pts.append(P1)
            pts.append(P2)
        g = np.array(pts).T                                     # 3 x n
        n = g.shape[1]
        Pc = np.zeros(3)
        for p in pts:
            Pc = Pc + p
        Pc = Pc / float(n)
        Cm = np.eye(n) - (1.0 / float(n)) * np.ones((n, n))
        Scat = g @ Cm @ g.T
        U, S, _ = np.linalg.svd(Scat)
        d = U[:, int(np.argmax(S))].copy()
        d = d / np.linalg.norm(d)
        if d[int(np.argmax(np.abs(d)))] < 0:                    # sign convention (see module docstring)
            d = -d
        # projectToLine
        sortable = []
        min_point = np.zeros(3)
        min_length = 0.0
        max_length = 0.0
        dn2 = float(np.linalg.norm(d)) * float(np.linalg.norm(d))
        for seg_id, (key, (P1, P2)) in enumerate(t3.items()):
            proj1 = Pc + (float(d @ (P1 - Pc)) / dn2) * d
            proj2 = Pc + (float(d @ (P2 - Pc)) / dn2) * d
            loc1 = float(d @ (Pc - proj1))
            if loc1 <= min_length:
                min_length, min_point = loc1, proj1
            if loc1 >= max_length:
                max_length = loc1
            loc2 = float(d @ (Pc - proj2))
            if loc2 <= min_length:
                min_length, min_point = loc2, proj2
            if loc2 >= max_length:
                max_length = loc2
            sortable.append([P1, seg_id, key[0], 0.0])
            sortable.append([P2, seg_id, key[0], 0.0])
        for s in sortable:
            s[3] = np.float32(np.linalg.norm(s[0] - min_point))
        sortable.sort(key=lambda s: float(s[3]))                # stable
        open_cams, open_lines = {}, set()
        opened = False
        start = None
        aligned = []
        for (P, seg_id, cam, _dist) in sortable:
            if seg_id not in open_lines:
                open_lines.add(seg_id)
                open_cams[cam] = open_cams.get(cam, 0) + 1
            else:
                open_lines.discard(seg_id)
                open_cams[cam] -= 1
                if open_cams[cam] == 0:
                    del open_cams[cam]
            if opened and len(open_cams) < 3:
                aligned.append((start, P))
                opened = False
            elif (not opened) and len(open_cams) >= 3:
                start = P
                opened = True
        return aligned


def save_result_txt(o, path):
    """Line3D::save3DLinesAsTXT (line3D.cc:434-473; README.txt:177-185): stream-default formatting = "%g"."""
    with open(path, "w") as f:
        for seg2, seg3 in o.result:
            if len(seg3) == 0:
                continue
            f.write("%d " % len(seg3))
            for P, Q in seg3:
                f.write("%g %g %g %g %g %g " % (P[0], P[1], P[2], Q[0], Q[1], Q[2]))
            f.write("%d " % len(seg2))
            for cam, seg in seg2:
                c = o.views[cam].segments[seg]
                f.write("%d %d %g %g %g %g " % (cam, seg, c[0], c[1], c[2], c[3]))
            f.write("\n")


def save_result_stl(o, path):
    """Line3D::save3DLinesAsSTL (line3D.cc:384-431): one degenerate facet per 3-D segment, "%e"."""
    with open(path, "w") as f:
        f.write("solid lineModel\n")
        for _seg2, seg3 in o.result:
            for P, Q in seg3:
                f.write(" facet normal 1.0e+000 0.0e+000 0.0e+000\n  outer loop\n")
                f.write("   vertex %e %e %e\n" % (P[0], P[1], P[2]))
                f.write("   vertex %e %e %e\n" % (Q[0], Q[1], Q[2]))
                f.write("   vertex %e %e %e\n" % (P[0], P[1], P[2]))
                f.write("  endloop\n endfacet\n")
        f.write("endsolid lineModel\n")


def run_scene(scene, matching_neighbors, perform_diffusion=False, libm=False, use_collinearity=True):
    o = OracleLine3D(matching_neighbors=matching_neighbors, libm=libm, use_collinearity=use_collinearity)
    for v in scene.views:
        o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    o.compute3Dmodel(perform_diffusion)
    return o
