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
