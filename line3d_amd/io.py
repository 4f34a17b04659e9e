"""Dependency-free reader of the reference's TXT result format (README.txt:177-185, written by
Line3D::save3DLinesAsTXT, line3D.cc:434-473): one 3-D line per text line,

    n P1x P1y P1z Q1x Q1y Q1z ... m camID1 segID1 p1x p1y q1x q1y ...

so that results of this implementation and of any external Line3D run can be diffed."""
from __future__ import annotations

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
