"""Dependency-free reader of the reference's TXT result format (README.txt:177-185, written by
Line3D::save3DLinesAsTXT, line3D.cc:434-473): one 3-D line per text line,

    n P1x P1y P1z Q1x Q1y Q1z ... m camID1 segID1 p1x p1y q1x q1y ...

so that results of this implementation and of any external Line3D run can be diffed; and the segment cache files of
Line3D::addImage over the C ABI (l3d_segment_cache_*, line3d_amd/csrc/l3d_segcache.cpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np


def load_txt(path: str):
    """-> list of (segments2D [(camID, segID, (x1, y1, x2, y2))...], segments3D [(P (3,), Q (3,))...])."""
    out = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            n = int(tok[0])
            pos = 1
            seg3 = []
            for _ in range(n):
                v = np.array([float(x) for x in tok[pos:pos + 6]])
                seg3.append((v[:3], v[3:]))
                pos += 6
            m = int(tok[pos])
            pos += 1
            seg2 = []
            for _ in range(m):
                seg2.append((int(tok[pos]), int(tok[pos + 1]), tuple(float(x) for x in tok[pos + 2:pos + 6])))
                pos += 6
            if pos != len(tok):
                raise ValueError("malformed line: %d tokens, %d consumed" % (len(tok), pos))
            out.append((seg2, seg3))
    return out


class SegmentCache:
    """One "segments_<id>_<w>x<h>_coll<0|1>.bin" (line3D.cc:143-150): segments (S,4) float32 and the directed
    collinearity entries (i, j, w) of segment2collinearities_, ascending (i, j)."""

    def __init__(self, segments, ci, cj, cw, library_version):
        self.segments, self.ci, self.cj, self.cw, self.library_version = segments, ci, cj, cw, library_version


def _cache_lib():
    from .capi import load_library
    lib = load_library()
    lib.l3d_segment_cache_last_error.restype = C.c_char_p
    lib.l3d_segment_cache_last_error.argtypes = [C.c_void_p]
    lib.l3d_segment_cache_free.argtypes = [C.c_void_p]
    for f in ("num_segments", "num_collinearities", "library_version"):
        getattr(lib, "l3d_segment_cache_" + f).argtypes = [C.c_void_p]
    return lib


def segment_cache_filename(image_id: int, width: int, height: int, use_collinearity: bool = True) -> str:
    buf = C.create_string_buffer(128)
    if _cache_lib().l3d_segment_cache_filename(C.c_uint32(image_id), C.c_uint(width), C.c_uint(height), C.c_int(int(use_collinearity)), buf, C.c_size_t(128)) != 0:
        raise RuntimeError("segment cache file name does not fit")
    return buf.value.decode()


def open_segment_cache(path: str):
    """-> opaque handle for Line3D.addImage_cached (free with close_segment_cache); raises with the reader's message."""
    lib = _cache_lib()
    h = C.c_void_p()
    rc = lib.l3d_segment_cache_read(path.encode(), C.byref(h))
    if rc != 0:
        msg = lib.l3d_segment_cache_last_error(h).decode() if h else "cannot read %s" % path
        if h:
            lib.l3d_segment_cache_free(h)
        raise RuntimeError(msg)
    return h


def close_segment_cache(h):
    _cache_lib().l3d_segment_cache_free(h)


def read_segment_cache(path: str) -> SegmentCache:
    lib = _cache_lib()
    h = open_segment_cache(path)
    try:
        n, nc = lib.l3d_segment_cache_num_segments(h), lib.l3d_segment_cache_num_collinearities(h)
        segs = np.zeros((n, 4), np.float32)
        ci, cj, cw = np.zeros(nc, np.int32), np.zeros(nc, np.int32), np.zeros(nc, np.float32)
        lib.l3d_segment_cache_get(h, *(a.ctypes.data_as(C.c_void_p) for a in (segs, ci, cj, cw)))
        return SegmentCache(segs, ci, cj, cw, lib.l3d_segment_cache_library_version(h))
    finally:
        close_segment_cache(h)


def write_segment_cache(path: str, segments, ci=(), cj=(), cw=(), library_version: int = 12):
    """ci/cj/cw: DIRECTED collinearity entries (both (i,j) and (j,i), as the L3DSegments constructor inserts them)."""
    segs = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1, 4)
    ci, cj = (np.ascontiguousarray(a, dtype=np.int32) for a in (ci, cj))
    cw = np.ascontiguousarray(cw, dtype=np.float32)
    rc = _cache_lib().l3d_segment_cache_write(path.encode(), segs.ctypes.data_as(C.c_void_p), C.c_int(len(segs)),
                                              ci.ctypes.data_as(C.c_void_p), cj.ctypes.data_as(C.c_void_p), cw.ctypes.data_as(C.c_void_p),
                                              C.c_int(len(ci)), C.c_int(library_version))
    if rc != 0:
        raise RuntimeError("cannot write segment cache %s (rc %d)" % (path, rc))
