#!/usr/bin/env python3
"""Soak: compute3Dmodel's tail on the device (default) against the host merge loop + grouping (L3D_HOST_CLUSTERING=1) over many scenes,
with and without diffusion: same affinity list, same lines, bit for bit.   python scripts/soak_device_host_finish.py [scenes]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402


def digest(l):
    h = hashlib.sha256(np.ascontiguousarray(l.affinity()[0]).tobytes())
    n = 0
    for seg2, seg3 in l.getResult():
        h.update(np.asarray(seg2, dtype=np.int64).tobytes())
        h.update(np.asarray([np.concatenate(p) for p in seg3], dtype=np.float64).tobytes())
        n += 1
    return h.hexdigest(), n


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    bad = 0
    for s in range(scenes):
        V, S, N = [(12, 600, 6), (20, 1500, 10), (16, 2500, 12), (10, 300, 4)][s % 4]
        sc = make_scene(V, S, N, seed=9000 + s)
        l = Line3D("", matchingNeighbors=N, crosschecks=True)       # (the cross-check build: the only one with the L3D_HOST_CLUSTERING switch)
        load_scene(l, sc)
        l.prepare()
        l.match_views()
        for diff in (False, True):
            l.context().set_option("L3D_HOST_CLUSTERING", 0)
            l.finish(diff)
            a = digest(l)
            l.context().set_option("L3D_HOST_CLUSTERING", 1)
            l.finish(diff)
            b = digest(l)
            l.context().set_option("L3D_HOST_CLUSTERING", 0)
            if a != b:
                bad += 1
            print("scene %d (%d x %d x %d) diffusion=%d: %d lines %s" % (s, V, S, N, diff, a[1], "ok" if a == b else "DIFFERENT"))
        l.close()
    print("soak done: %d mismatches" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
