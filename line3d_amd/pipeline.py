"""Python mirror of the reference's operator interface, class L3D::Line3D (line3D.h:61-101), over the
C ABI (l3d_line3d_* in include/line3d_amd.h).  Same calls, argument meaning and defaults
(commons.h:42-61); images are replaced by their detected segments."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import MATCH_DTYPE, EDGE_DTYPE, L3DError, _p


class Line3D:
    def __init__(self, data_directory: str = "", matchingNeighbors: int = 10, uncertainty_t_upper_2D: float = 5.0,
                 uncertainty_t_lower_2D: float = 1.0, sigma_p: float = 3.5, sigma_a: float = 10.0,
                 min_baseline: float = 0.25, useCollinearity: bool = True, verbose: bool = False, device: int = 0, crosschecks: bool = False):
        self.lib = capi.load_library(crosschecks)        # (crosschecks: the test-only build in which the L3D_HOST_* switches exist)
        self.lib.l3d_line3d_last_error.restype = C.c_char_p
        self.lib.l3d_line3d_last_error.argtypes = [C.c_void_p]
        self.lib.l3d_line3d_context.restype = C.c_void_p
        self.lib.l3d_line3d_context.argtypes = [C.c_void_p]
        self.lib.l3d_line3d_destroy.argtypes = [C.c_void_p]
        self.data_directory = data_directory          # kept for signature parity; nothing is written to disk
        h = C.c_void_p()
        rc = self.lib.l3d_line3d_create(C.c_int(device), C.c_int(matchingNeighbors), C.c_float(uncertainty_t_upper_2D),
                                        C.c_float(uncertainty_t_lower_2D), C.c_float(sigma_p), C.c_float(sigma_a),
                                        C.c_float(min_baseline), C.c_int(int(useCollinearity)), C.c_int(int(verbose)), C.byref(h))
        if rc != 0:
            raise L3DError("l3d_line3d_create failed (code %d): no usable MI355X / HIP device -- no CPU fallback" % rc)
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.l3d_line3d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise L3DError("line3d_amd error %d: %s" % (rc, self.lib.l3d_line3d_last_error(self.h).decode()))

    def context(self) -> capi.Context:
        """The pipeline's l3d_ctx as a (non-owning) Context, for profiling."""
        c = capi.Context.__new__(capi.Context)
        c.lib = self.lib
        c.h = C.c_void_p(self.lib.l3d_line3d_context(self.h))
        c._keep = []
        c.close = lambda: None
        return c

    # -- reference interface ------------------------------------------------------------------
    def addImage(self, imageID, width, height, segments, K, R, t, worldpointIDs):
        segs = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1, 4)
        K, R, t = (np.ascontiguousarray(a, dtype=np.float64) for a in (K, R, t))
        wps = np.ascontiguousarray(list(worldpointIDs), dtype=np.uint32)
        rc = self.lib.l3d_line3d_add_image(self.h, C.c_uint32(imageID), C.c_uint(width), C.c_uint(height), _p(segs),
                                           C.c_int(len(segs)), _p(K), _p(R), _p(t), _p(wps), C.c_int(len(wps)))
        return rc == 0          # the reference prints to cerr and returns (line3D.cc:101-127)

    def addImage_cached(self, imageID, width, height, cache_path, K, R, t, worldpointIDs):
        """addImage when the segment cache file exists (line3D.cc:160-168): segments and collinearities from the file."""
        from .io import open_segment_cache, close_segment_cache
        K, R, t = (np.ascontiguousarray(a, dtype=np.float64) for a in (K, R, t))
        wps = np.ascontiguousarray(list(worldpointIDs), dtype=np.uint32)
        cache = open_segment_cache(cache_path)
        try:
            rc = self.lib.l3d_line3d_add_image_cached(self.h, C.c_uint32(imageID), C.c_uint(width), C.c_uint(height), cache,
                                                      _p(K), _p(R), _p(t), _p(wps), C.c_int(len(wps)))
        finally:
            close_segment_cache(cache)
        return rc == 0

    def addImage_ex(self, imageID, width, height, segments, K, R, t, links, maxImgWidth=1920, loadAndStoreSegments=True, fixed_sim=False):
        """addImage / addImage_fixed_sim with the reference's segment-cache behaviour in `data_directory` (line3D.cc:128-199): the cache
        file is removed, read instead of `segments`, or written.  links: world point ids, or {view id: similarity} with fixed_sim."""
        segs = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1, 4)
        K, R, t = (np.ascontiguousarray(a, dtype=np.float64) for a in (K, R, t))
        d = self.data_directory.encode()
        if fixed_sim:
            ids = np.ascontiguousarray(sorted(links), dtype=np.uint32)
            sims = np.ascontiguousarray([links[int(i)] for i in ids], dtype=np.float32)
            rc = self.lib.l3d_line3d_add_image_fixed_sim_ex(self.h, C.c_uint32(imageID), C.c_uint(width), C.c_uint(height), _p(segs), C.c_int(len(segs)),
                                                            _p(K), _p(R), _p(t), _p(ids), _p(sims), C.c_int(len(ids)), C.c_char_p(d), C.c_int(maxImgWidth),
                                                            C.c_int(int(loadAndStoreSegments)))
        else:
            wps = np.ascontiguousarray(list(links), dtype=np.uint32)
            rc = self.lib.l3d_line3d_add_image_ex(self.h, C.c_uint32(imageID), C.c_uint(width), C.c_uint(height), _p(segs), C.c_int(len(segs)),
                                                  _p(K), _p(R), _p(t), _p(wps), C.c_int(len(wps)), C.c_char_p(d), C.c_int(maxImgWidth),
                                                  C.c_int(int(loadAndStoreSegments)))
        return rc == 0

    def addImage_fixed_sim(self, imageID, width, height, segments, K, R, t, viewSimilarity):
        segs = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1, 4)
        K, R, t = (np.ascontiguousarray(a, dtype=np.float64) for a in (K, R, t))
        ids = np.ascontiguousarray(sorted(viewSimilarity), dtype=np.uint32)
        sims = np.ascontiguousarray([viewSimilarity[int(i)] for i in ids], dtype=np.float32)
        rc = self.lib.l3d_line3d_add_image_fixed_sim(self.h, C.c_uint32(imageID), C.c_uint(width), C.c_uint(height), _p(segs),
                                                     C.c_int(len(segs)), _p(K), _p(R), _p(t), _p(ids), _p(sims), C.c_int(len(ids)))
        return rc == 0

    def compute3Dmodel(self, perform_diffusion: bool = False):
        self._chk(self.lib.l3d_line3d_compute3Dmodel(self.h, C.c_int(int(perform_diffusion))))

    def getResult(self):
        """list of (segments2D [(camID, segID)...], segments3D [(P1, P2)...]) -- L3DFinalLine3D (commons.h:215-238)."""
        nl, n3, n2 = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.lib.l3d_line3d_result_sizes(self.h, C.byref(nl), C.byref(n3), C.byref(n2)))
        l3 = np.zeros(nl.value, np.int32)
        l2 = np.zeros(nl.value, np.int32)
        s3 = np.zeros((n3.value, 6), np.float64)
        s2 = np.zeros((n2.value, 2), np.uint32)
        if nl.value:
            self._chk(self.lib.l3d_line3d_get_result(self.h, _p(l3), _p(l2), _p(s3), _p(s2)))
        out, a, b = [], 0, 0
        for k in range(nl.value):
            seg3 = [(s3[a + i, :3].copy(), s3[a + i, 3:].copy()) for i in range(l3[k])]
            seg2 = [(int(s2[b + i, 0]), int(s2[b + i, 1])) for i in range(l2[k])]
            a += l3[k]
            b += l2[k]
            out.append((seg2, seg3))
        return out

    def getSegment2D(self, camID, segID):
        o = (C.c_float * 4)()
        self.lib.l3d_line3d_get_segment2D(self.h, C.c_uint32(camID), C.c_uint32(segID), o)
        return tuple(o)

    # line3D.h:91-95 -- result writers (formats: line3D.cc:384-473, README.txt:177-185)
    def save3DLinesAsSTL(self, filename: str):
        self._chk(self.lib.l3d_line3d_save_result(self.h, filename.encode(), C.c_int(0)))

    def save3DLinesAsTXT(self, filename: str):
        self._chk(self.lib.l3d_line3d_save_result(self.h, filename.encode(), C.c_int(1)))

    def numCameras(self):
        return self.lib.l3d_line3d_num_cameras(self.h)

    def reset(self):
        self._chk(self.lib.l3d_line3d_reset(self.h))

    # -- stages ----------------------------------------------------------------------------------
    def prepare(self):
        self._chk(self.lib.l3d_line3d_prepare(self.h))

    def match_views(self):
        self._chk(self.lib.l3d_line3d_match_views(self.h))

    def finish(self, perform_diffusion=False):
        self._chk(self.lib.l3d_line3d_finish(self.h, C.c_int(int(perform_diffusion))))

    def match_begin(self):
        n = C.c_int(0)
        self._chk(self.lib.l3d_line3d_match_begin(self.h, C.byref(n)))
        ids = np.zeros(n.value, np.uint32)
        ns = np.zeros(n.value, np.int32)
        self._chk(self.lib.l3d_line3d_match_order(self.h, _p(ids), _p(ns)))
        return ids, ns

    def view_num_to_be_matched(self, view_id):
        return self.lib.l3d_line3d_view_num_to_be_matched(self.h, C.c_uint32(view_id))

    def match_view_compute(self, view_id, seg_begin, seg_end):
        out = C.c_void_p()
        n = C.c_int(0)
        med = C.c_float(1.0)
        bd = C.POINTER(C.c_float)()
        nb = C.c_int(0)
        self._chk(self.lib.l3d_line3d_match_view_compute(self.h, C.c_uint32(view_id), C.c_int(seg_begin), C.c_int(seg_end),
                                                         C.byref(out), C.byref(n), C.byref(med), C.byref(bd), C.byref(nb)))
        res = np.zeros(n.value, dtype=MATCH_DTYPE)
        if n.value:
            C.memmove(res.ctypes.data, out, n.value * 32)
        self.lib.l3d_free(out)
        best = np.ctypeslib.as_array(bd, (nb.value * 2,)).copy() if nb.value else np.zeros(0, np.float32)
        if bd:
            self.lib.l3d_free(bd)
        return res, med.value, best

    def match_view_commit(self, view_id, matches, best_depths=None, median=1.0):
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        if best_depths is None:
            self._chk(self.lib.l3d_line3d_match_view_commit(self.h, C.c_uint32(view_id), _p(m), C.c_int(len(m)), None,
                                                            C.c_int(-1), C.c_float(median)))
        else:
            b = np.ascontiguousarray(best_depths, dtype=np.float32)
            self._chk(self.lib.l3d_line3d_match_view_commit(self.h, C.c_uint32(view_id), _p(m), C.c_int(len(m)), _p(b),
                                                            C.c_int(len(b) // 2), C.c_float(median)))

    # -- resident chain sharded over ranks (line3d_amd/distributed.py drives it) -------------------
    def stream_ptr(self) -> int:
        self.lib.l3d_ctx_stream.restype = C.c_void_p
        self.lib.l3d_ctx_stream.argtypes = [C.c_void_p]
        return int(self.lib.l3d_ctx_stream(C.c_void_p(self.lib.l3d_line3d_context(self.h))) or 0)

    def shard_open(self, rank: int, world: int, slot_records: int):
        n = C.c_int(0)
        sb = C.c_size_t(0)
        self._chk(self.lib.l3d_line3d_shard_open(self.h, C.c_int(rank), C.c_int(world), C.c_int(slot_records), C.byref(n), C.byref(sb)))
        return n.value, sb.value

    def shard_view_verified(self, k: int) -> bool:
        return self.lib.l3d_line3d_shard_view_verified(self.h, C.c_int(k)) == 1

    def shard_enqueue(self, k: int, send_slot_ptr: int, gathered_ptr: int):
        self._chk(self.lib.l3d_line3d_shard_enqueue(self.h, C.c_int(k), C.c_void_p(send_slot_ptr), C.c_void_p(gathered_ptr)))

    def shard_mark(self, k: int):
        self._chk(self.lib.l3d_line3d_shard_mark(self.h, C.c_int(k)))

    def shard_fetch(self, k: int):
        self._chk(self.lib.l3d_line3d_shard_fetch(self.h, C.c_int(k)))

    def shard_run(self, rank: int, world: int, slot_records: int, exchange: str = "local", exchange_user=None, commit: bool = True):
        """The whole sharded chain as one native call (l3d_shard_chain_run).  exchange: "rccl" (exchange_user = a ctypes
        l3d_rccl_link), "local" (world 1) or "replay" (exchange_user = device address of recorded gathered blocks).
        commit: False / 0 = compute and exchange only, True / 1 = host bookkeeping on this rank, 2 (or "device") = matchViews' products built on
        this rank's device from the gathered slots, 3 (or "partition") = as 2, but this rank keeps the records and builds the rows of its block of
        views only (l3d_shard_chain_partition: exact without speculation; finish_sharded() follows on every rank).
        Returns (device address of the gathered blocks, slot_bytes)."""
        if commit == "device":
            commit = 2
        if commit == "partition":
            commit = 3
        if callable(exchange):       # tests: a Python exchange (called on this thread by the enqueue loop), e.g. to inject a failure
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
            fn = self._exchange_keepalive = proto(exchange)
        else:
            fn = {"rccl": self.lib.l3d_exchange_rccl, "local": self.lib.l3d_exchange_local, "replay": self.lib.l3d_exchange_replay}[exchange]
        user = C.c_void_p(exchange_user) if isinstance(exchange_user, int) else (C.c_void_p(C.addressof(exchange_user)) if exchange_user is not None else None)
        g = C.c_void_p(0)
        sb = C.c_size_t(0)
        self._chk(self.lib.l3d_line3d_shard_run(self.h, C.c_int(rank), C.c_int(world), C.c_int(slot_records), C.cast(fn, C.c_void_p), user,
                                                C.c_int(int(commit)), C.byref(g), C.byref(sb)))
        return g.value, sb.value

    def block_run(self, rank: int, world: int, exchange="local", exchange_user=None, warmup_views: int = -1) -> bool:
        """matchViews with the views sharded over the ranks in blocks (l3d_line3d_block_run): True = the speculation was exact and this object
        holds matchViews' products; False = it was not (same answer on every rank): run shard_run instead.  exchange as for shard_run."""
        if callable(exchange):
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
            fn = self._exchange_keepalive = proto(exchange)
        else:
            fn = {"rccl": self.lib.l3d_exchange_rccl, "local": self.lib.l3d_exchange_local}[exchange]
        user = C.c_void_p(exchange_user) if isinstance(exchange_user, int) else (C.c_void_p(C.addressof(exchange_user)) if exchange_user is not None else None)
        verdict = C.c_int(1)
        self._chk(self.lib.l3d_line3d_block_run(self.h, C.c_int(rank), C.c_int(world), C.c_int(warmup_views), C.cast(fn, C.c_void_p), user, C.byref(verdict)))
        return verdict.value == 0

    def partition_run(self, rank: int, world: int, exchange="local", exchange_user=None, warmup_views: int = -1) -> bool:
        """matchViews sharded by blocks of views with nothing replicated (l3d_line3d_partition_run): this object then holds its block's share of the
        kept records and of matchViews' products; finish_sharded() -- on every rank -- completes compute3Dmodel.  exchange as for block_run."""
        if callable(exchange):
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
            fn = self._exchange_keepalive = proto(exchange)
        else:
            fn = {"rccl": self.lib.l3d_exchange_rccl, "local": self.lib.l3d_exchange_local}[exchange]
        user = C.c_void_p(exchange_user) if isinstance(exchange_user, int) else (C.c_void_p(C.addressof(exchange_user)) if exchange_user is not None else None)
        verdict = C.c_int(1)
        self._chk(self.lib.l3d_line3d_partition_run(self.h, C.c_int(rank), C.c_int(world), C.c_int(warmup_views), C.cast(fn, C.c_void_p), user, C.byref(verdict)))
        return verdict.value == 0

    def finish_sharded(self, perform_diffusion=False):
        """the rest of compute3Dmodel after partition_run, on every rank (collective; the exchange of the run)"""
        self._chk(self.lib.l3d_line3d_finish_sharded(self.h, C.c_int(int(perform_diffusion)), None, None))

    def partition_info(self):
        """what this rank's share covers (views of the dense map): dict(rank, world, own, rows, held, n_pot_all, recovery_rounds, blocks_rerun)"""
        info = (C.c_int * 10)()
        npot = C.c_int64(0)
        self._chk(self.lib.l3d_partition_info(C.c_void_p(self.lib.l3d_line3d_context(self.h)), info, C.byref(npot)))
        return dict(rank=info[0], world=info[1], own=(info[2], info[3]), rows=(info[4], info[5]), held=(info[6], info[7]), n_pot_all=npot.value, recovery_rounds=info[8],
                    blocks_rerun=info[9])

    def shard_close(self, committed: bool):
        self._chk(self.lib.l3d_line3d_shard_close(self.h, C.c_int(int(committed))))

    def match_end(self):
        self._chk(self.lib.l3d_line3d_match_end(self.h))

    # -- inspection ------------------------------------------------------------------------------
    def set_sync_matching(self, on=True):
        self._chk(self.lib.l3d_line3d_set_sync_matching(self.h, C.c_int(int(on))))

    def match_path(self) -> int:
        """l3d_line3d_match_path: 0 resident chain, 1 chain + host bookkeeping, 2 per-view seam calls on request, 3 per-view because the schedule is not static"""
        return int(self.lib.l3d_line3d_match_path(self.h))

    def keep_view_matches(self, on=True):
        self._chk(self.lib.l3d_line3d_keep_view_matches(self.h, C.c_int(int(on))))

    def view_matches(self, view_id):
        p = C.c_void_p()
        n = C.c_int(0)
        med = C.c_float(0)
        self._chk(self.lib.l3d_line3d_view_matches(self.h, C.c_uint32(view_id), C.byref(p), C.byref(n), C.byref(med)))
        res = np.zeros(n.value, dtype=MATCH_DTYPE)
        if n.value:
            C.memmove(res.ctypes.data, p, n.value * 32)
        return res, med.value

    def affinity(self):
        p = C.c_void_p()
        nnz, nn = C.c_int(0), C.c_int(0)
        self._chk(self.lib.l3d_line3d_affinity(self.h, C.byref(p), C.byref(nnz), C.byref(nn)))
        res = np.zeros(nnz.value, dtype=EDGE_DTYPE)
        if nnz.value:
            C.memmove(res.ctypes.data, p, nnz.value * 12)
        return res, nn.value

    def resident_products(self):
        """The device-resident products of the last match_views (and, after finish, the hypothesis table), copied to the host:
        None when matchViews ran with host bookkeeping.  dict(seg_base, pot_start, pot_tgt, best, hyp, score)."""
        nv, nd, nh = C.c_int(0), C.c_int(0), C.c_int(0)
        npot = C.c_int64(0)
        self._chk(self.lib.l3d_line3d_products_sizes(self.h, C.byref(nv), C.byref(nd), C.byref(npot), C.byref(nh)))
        if nv.value == 0:
            return None
        seg_base = np.zeros(nv.value + 1, np.int32)
        pot_start = np.zeros(nd.value + 1, np.int64)
        pot_tgt = np.zeros(max(1, npot.value), np.int32)
        best = np.zeros(nd.value, MATCH_DTYPE)
        hyp = np.zeros(max(1, nh.value), capi.HYP_DTYPE)
        score = np.zeros(max(1, nh.value), np.float32)
        self._chk(self.lib.l3d_line3d_products_get(self.h, _p(seg_base), _p(pot_start), _p(pot_tgt), _p(best), _p(hyp) if nh.value else None,
                                                   _p(score) if nh.value else None))
        return dict(seg_base=seg_base, pot_start=pot_start, pot_tgt=pot_tgt[:npot.value], best=best, hyp=hyp[:nh.value], score=score[:nh.value])

    def stats(self):
        s = (C.c_double * 12)()
        self._chk(self.lib.l3d_line3d_stats(self.h, s))
        keys = ["pairs", "raw", "kept", "hypotheses", "t_match", "t_gpu_call", "t_commit", "t_finalize", "t_affinity",
                "t_cluster", "edges", "lines"]
        return dict(zip(keys, list(s)))


def load_scene(l3d: Line3D, scene):
    for v in scene.views:
        ok = l3d.addImage_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        assert ok


def load_scene_worldpoints(l3d: Line3D, scene):
    """views that carry the world points they see (synth.make_scene_scattered): Line3D::addImage, neighbours chosen by the library"""
    for v in scene.views:
        ok = l3d.addImage(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["worldpoints"])
        assert ok
