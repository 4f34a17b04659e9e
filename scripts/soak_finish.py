#!/usr/bin/env python3
"""Soak test of the threaded finishing stages: the same scene finished many times (with and without diffusion) must give the
same affinity list and the same lines every time; matching itself is re-run between rounds."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3d_amd.pipeline import Line3D, load_scene  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402


def digest(l):
    A = l.affinity()[0]
    h = hashlib.sha256(np.ascontiguousarray(A).tobytes())
    for seg2, seg3 in l.getResult():
        h.update(np.asarray(seg2, dtype=np.int64).tobytes())
        h.update(np.asarray([np.concatenate(p) for p in seg3], dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    V, S, N = 24, 2000, 12
    sc = make_scene(V, S, N, seed=4242)
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, sc)
    l.prepare()
    ref = {}
    bad = 0
    for r in range(rounds):
        if r % 5 == 0:
            l.match_views()
        for diff in (False, True):
            l.finish(diff)
            d = digest(l)
            if diff not in ref:
                ref[diff] = d
            elif ref[diff] != d:
                bad += 1
                print("round %d diffusion=%s: digest differs" % (r, diff))
    print("soak done: %d rounds, %d mismatches, lines %d" % (rounds, bad, len(l.getResult())))
    l.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
