"""Test oracle (NOT product code) -- pure-Python restatement of the reference drivers' SfM readers:
VisualSfM NVM (main_vsfm.cpp:121-223) and bundler bundle.rd.out (main_bundler.cpp:110-204).
Parity status: unpinned by the reference (it ships no test data); pinned by files written from known cameras
(tests/helpers.py) and by agreement with the C++ readers of the product (line3d_amd/csrc/l3d_sfm.cpp)."""
import numpy as np


def _f32(x):
    return float(np.float32(x))


def read_nvm(path):
    with open(path) as f:
        lines = f.read().split("\n")
    pos = 2                                                   # main_vsfm.cpp:125-126: two ignored lines
    n = int(lines[pos].split()[0]); pos += 1
    if n == 0:
        raise RuntimeError("No aligned cameras in NVM file!")
    cams = []
    for _ in range(n):
        tok = lines[pos].split(); pos += 1
        name, fl, q3, q0, q1, q2, cx, cy, cz, d = tok[0], *map(float, tok[1:10])       # :153-155 (file order w x y z)
        R = np.array([[1.0 - 2.0 * q1 * q1 - 2.0 * q2 * q2, 2.0 * q0 * q1 - 2.0 * q2 * q3, 2.0 * q0 * q2 + 2.0 * q1 * q3],
                      [2.0 * q0 * q1 + 2.0 * q2 * q3, 1.0 - 2.0 * q0 * q0 - 2.0 * q2 * q2, 2.0 * q1 * q2 - 2.0 * q0 * q3],
                      [2.0 * q0 * q2 - 2.0 * q1 * q3, 2.0 * q1 * q2 + 2.0 * q0 * q3, 1.0 - 2.0 * q0 * q0 - 2.0 * q1 * q1]])   # :162-173
        Cc = (cx, cy, cz)
        t = np.array([((-R[r, 0]) * Cc[0] + (-R[r, 1]) * Cc[1]) + (-R[r, 2]) * Cc[2] for r in range(3)])                    # :176-177
        cams.append(dict(name=name, focal=_f32(fl), dist=np.array([_f32(d), 0.0]), R=R, t=t, worldpoints=[]))
    pos += 1                                                  # :184 ignored line
    npts = int(lines[pos].split()[0]); pos += 1
    for i in range(npts):
        tok = lines[pos].split(); pos += 1
        nv = int(tok[6])
        for j in range(nv):
            cam = int(tok[7 + 4 * j])
            cams[cam]["worldpoints"].append(i)                # :193-215
    for c in cams:
        c["worldpoints"] = np.array(c["worldpoints"], dtype=np.uint32)
    return cams, npts


def read_bundler(path):
    with open(path) as f:
        lines = f.read().split("\n")
    n, npts = map(int, lines[1].split()[:2])                  # main_bundler.cpp:114-120
    if n == 0 or npts == 0:
        raise RuntimeError("No cameras and/or points in bundle file!")
    pos = 2
    cams = []
    for i in range(n):
        fl, d1, d2 = map(float, lines[pos].split()[:3]); pos += 1
        R = np.array([[float(x) for x in lines[pos + r].split()[:3]] for r in range(3)]); pos += 3
        R[1:] *= -1.0                                         # :158-160
        t = np.array([float(x) for x in lines[pos].split()[:3]]); pos += 1
        t[1:] *= -1.0                                         # :172-174
        cams.append(dict(name="%08d" % i, focal=_f32(fl), dist=np.array([_f32(d1), _f32(d2)]), R=R, t=t, worldpoints=[]))
    for i in range(npts):
        tok = lines[pos + 2].split(); pos += 3                # :183-185
        nv = int(tok[0])
        for j in range(nv):
            cams[int(tok[1 + 4 * j])]["worldpoints"].append(i)
    for c in cams:
        c["worldpoints"] = np.array(c["worldpoints"], dtype=np.uint32)
    return cams, npts


def intrinsics(focal, width, height):                         # main_vsfm.cpp:232-241
    px, py, f = _f32(np.float32(width) / np.float32(2.0)), _f32(np.float32(height) / np.float32(2.0)), _f32(focal)
    return np.array([[f, 0.0, px], [0.0, f, py], [0.0, 0.0, 1.0]])


# ---------------------------------------------------------------------------------------------------------------------
# Segment cache "segments_<id>_<w>x<h>_coll<0|1>.bin" (line3D.cc:143-168): boost::archive::binary_oarchive of one
# L3DSegments (serialization.h:49-69; segments.h:124-131: collinearity map, then DataArray<float>*; dataArray.h:296-318).
# boost is absent here and unpinned by the reference: the layout is restated from boost.serialization's headers for archive
# library versions >= 9 (x86-64), written independently of the product's C++ (l3d_segcache.cpp) so that the two can be
# checked against each other.  Parity unpinned: no file written by boost exists here.
#   basic_binary_oarchive::init            u64 len + "serialization::archive", u16 library version
#   basic_binary_oprimitive::init          u8 sizeof(int, long, float, double), i32 1
#   basic_oarchive::save_object            first object of a class: u8 tracking, u32 version (class_id_optional unwritten)
#   collections_save_imp.hpp               u64 count, u32 item_version, items
#   basic_oarchive::save_pointer           i16 class id (-1 null); first of class: u8 tracking, u32 version; u32 object id
import struct

_SIG = b"serialization::archive"


def filename_segment_cache(image_id, width, height, use_collinearity):          # line3D.cc:143-148
    return "/segments_%d_%dx%d_coll%d.bin" % (image_id, width, height, 1 if use_collinearity else 0)


def write_segment_cache(path, segments, coll, library_version=12):
    """segments: (S,4) float32; coll: dict i -> dict j -> w (segment2collinearities_)."""
    segments = np.asarray(segments, np.float32).reshape(-1, 4)
    out = [struct.pack("<Q", len(_SIG)), _SIG, struct.pack("<H", library_version), struct.pack("<BBBBi", 4, 8, 4, 8, 1)]
    pre = struct.pack("<BI", 0, 0)
    out += [pre, pre, struct.pack("<QI", len(coll), 0)]                       # L3DSegments, outer map
    first_outer = first_inner = True
    for i in sorted(coll):
        if first_outer:
            out.append(pre)                                                    # pair<const unsigned, map>
        out.append(struct.pack("<I", i))
        if first_outer:
            out.append(pre)                                                    # inner map
        first_outer = False
        out.append(struct.pack("<QI", len(coll[i]), 0))
        for j in sorted(coll[i]):
            if first_inner:
                out.append(pre); first_inner = False                           # pair<const unsigned, float>
            out.append(struct.pack("<If", j, float(np.float32(coll[i][j]))))
    n_classes_before = 2 if not coll else 5
    out.append(struct.pack("<hBII", n_classes_before, 1, 0, 0))                # class id, tracked, version, object id
    width, real_width = 4, 8                                                   # 16-byte rows padded to 32 (dataArray.h:74-84)
    out.append(struct.pack("<IIIQQQQ", width, len(segments), real_width, real_width * 4, real_width, 0, 0))
    rows = np.zeros((len(segments), real_width), np.float32)
    rows[:, :4] = segments
    out.append(rows.tobytes())
    with open(path, "wb") as f:
        f.write(b"".join(out))


def read_segment_cache(path):
    """-> (segments (S,4) float32, coll dict i -> dict j -> w, library version)"""
    b = open(path, "rb").read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, b, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def preamble():
        t, ver = take("BI")
        if t > 1 or ver != 0:
            raise ValueError("unexpected class preamble")
        return t

    if take("Q") != len(_SIG) or b[pos:pos + len(_SIG)] != _SIG:
        raise ValueError("not a boost binary archive")
    pos += len(_SIG)
    lib = take("H")
    if lib < 9:
        raise ValueError("unsupported archive library version %d" % lib)
    if take("BBBBi") != (4, 8, 4, 8, 1):
        raise ValueError("foreign native sizes")
    if preamble() or preamble():
        raise ValueError("tracked top-level object")
    n_outer, iv = take("QI")
    coll, seen_pair, seen_inner, seen_ipair = {}, False, False, False
    for _ in range(n_outer):
        if not seen_pair:
            preamble(); seen_pair = True
        i = take("I")
        if not seen_inner:
            preamble(); seen_inner = True
        n_inner, iv = take("QI")
        row = coll.setdefault(i, {})
        for _ in range(n_inner):
            if not seen_ipair:
                preamble(); seen_ipair = True
            j, w = take("If")
            row[j] = np.float32(w)
    cid = take("h")
    segs = np.zeros((0, 4), np.float32)
    if cid != -1:
        if preamble():
            take("I")                                                          # object id
        width, height, real_width, pitch, stride, _pg, _sg = take("IIIQQQQ")
        if width != 4 or stride != real_width or pitch != 4 * real_width:
            raise ValueError("not a 4-column float array")
        rows = np.frombuffer(b, np.float32, real_width * height, pos).reshape(height, real_width)
        pos += 4 * real_width * height
        segs = rows[:, :4].copy()
    if pos != len(b):
        raise ValueError("trailing bytes")
    return segs, coll, lib
