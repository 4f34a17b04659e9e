"""SfM front ends over the C ABI: VisualSfM NVM and bundler files -> what Line3D::addImage needs, and the
drivers' flow (main_vsfm.cpp / main_bundler.cpp) with segments supplied instead of images."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .capi import load_library


class SfmScene:
    """cameras: list of dicts {name, focal, dist (2,), R (3,3), t (3,), worldpoints (uint32 array)}."""

    def __init__(self, cameras, n_points):
        self.cameras = cameras
        self.n_points = n_points


def _read(path: str, fn_name: str) -> SfmScene:
    lib = load_library()
    lib.l3d_sfm_last_error.restype = C.c_char_p
    lib.l3d_sfm_last_error.argtypes = [C.c_void_p]
    lib.l3d_sfm_camera_name.restype = C.c_char_p
    lib.l3d_sfm_camera_name.argtypes = [C.c_void_p, C.c_int]
    lib.l3d_sfm_free.argtypes = [C.c_void_p]
    h = C.c_void_p()
    rc = getattr(lib, fn_name)(path.encode(), C.byref(h))
    try:
        if rc != 0:
            raise RuntimeError(lib.l3d_sfm_last_error(h).decode() if h else "cannot read %s" % path)
        cams = []
        for i in range(lib.l3d_sfm_num_cameras(h)):
            focal = C.c_double(0)
            dist = (C.c_double * 2)()
            R = (C.c_double * 9)()
            t = (C.c_double * 3)()
            nw = C.c_int(0)
            lib.l3d_sfm_camera(h, C.c_int(i), C.byref(focal), dist, R, t, C.byref(nw))
            w = np.zeros(nw.value, dtype=np.uint32)
            if nw.value:
                lib.l3d_sfm_camera_worldpoints(h, C.c_int(i), w.ctypes.data_as(C.c_void_p))
            cams.append(dict(name=lib.l3d_sfm_camera_name(h, C.c_int(i)).decode(), focal=focal.value, dist=np.array(list(dist)),
                             R=np.array(list(R)).reshape(3, 3), t=np.array(list(t)), worldpoints=w))
        return SfmScene(cams, lib.l3d_sfm_num_points(h))
    finally:
        if h:
            lib.l3d_sfm_free(h)


def read_nvm(path: str) -> SfmScene:
    """main_vsfm.cpp:121-223"""
    return _read(path, "l3d_sfm_read_nvm")


def read_bundler(path: str) -> SfmScene:
    """main_bundler.cpp:110-204 (bundle.rd.out)"""
    return _read(path, "l3d_sfm_read_bundler")


def intrinsics(focal: float, width: int, height: int) -> np.ndarray:
    """main_vsfm.cpp:232-241"""
    K = (C.c_double * 9)()
    load_library().l3d_sfm_intrinsics(C.c_double(focal), C.c_uint(width), C.c_uint(height), K)
    return np.array(list(K)).reshape(3, 3)


def result_basename(max_width=-1, neighbors=10, min_uncertainty=1.0, max_uncertainty=5.0, sigma_p=3.5, sigma_a=10.0,
                    collinearity=True, diffusion=False) -> str:
    """The drivers' output name (main_vsfm.cpp:289-313), numbers in stream-default formatting."""
    n = "N_ALL__" if neighbors < 0 else "N_%d__" % neighbors
    return ("line3D_result__W_%d__%stL_%g__tU_%g__sigmaP_%g__sigmaA_%g__%s__%s"
            % (max_width, n, min_uncertainty, max_uncertainty, sigma_p, sigma_a,
               "COLLIN" if collinearity else "NO_COLLIN", "DIFFUSION" if diffusion else "NO_DIFFUSION"))


def reconstruct(scene: SfmScene, segments, image_sizes, out_dir=None, neighbors=10, diffusion=False, device=0, **line3d_kwargs):
    """The drivers' flow (main_vsfm.cpp:226-325) with detected segments in place of images: K from focal and image size,
    addImage with the world point lists, compute3Dmodel, optional STL + TXT output under the drivers' file name.
    segments[i]: (S,4) float32 of camera i (undistorted image coordinates); image_sizes[i]: (width, height).
    segments may also be a DIRECTORY: the reference's data directory (main_vsfm.cpp:108-116, "<image folder>/L3D_data"),
    whose segment caches "segments_<id>_<w>x<h>_coll<0|1>.bin" of an earlier run are replayed (line3D.cc:143-168)."""
    import os
    from .pipeline import Line3D
    from .io import segment_cache_filename
    l3d = Line3D("", matchingNeighbors=neighbors, device=device, **line3d_kwargs)
    for i, cam in enumerate(scene.cameras):
        if np.any(np.abs(cam["dist"]) > 1e-12):
            raise RuntimeError("camera %d has lens distortion: undistort the image before detecting segments (out of scope here)" % i)
        w, h = image_sizes[i]
        if isinstance(segments, (str, os.PathLike)):
            path = os.fspath(segments) + segment_cache_filename(i, w, h, line3d_kwargs.get("useCollinearity", True))
            if not os.path.exists(path):
                raise RuntimeError("no segment cache %s (the detector is out of scope: segments must come from a cache or the caller)" % path)
            l3d.addImage_cached(i, w, h, path, intrinsics(cam["focal"], w, h), cam["R"], cam["t"], cam["worldpoints"])
            continue
        l3d.addImage(i, w, h, segments[i], intrinsics(cam["focal"], w, h), cam["R"], cam["t"], cam["worldpoints"])
    l3d.compute3Dmodel(diffusion)
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        base = os.path.join(out_dir, result_basename(neighbors=neighbors, diffusion=diffusion))
        l3d.save3DLinesAsSTL(base + ".stl")
        l3d.save3DLinesAsTXT(base + ".txt")
    return l3d
