#!/usr/bin/env python3
"""Soak: the sharded run with commit on the device (every virtual rank of a recorded world-W job replayed through the native loop) against
the single-GPU chain over several scenes and world sizes: kept lists, medians, products and lines must be identical.
    python scripts/soak_sharded_device_commit.py [scenes]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from line3d_amd.pipeline import Line3D, load_scene  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402


def digest(l, scene):
    h = hashlib.sha256()
    for v in scene.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes()); h.update(np.float32(med).tobytes())
    p = l.resident_products()
    for k in ("seg_base", "pot_start", "pot_tgt", "best", "hyp", "score"):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    for seg2, seg3 in l.getResult():
        h.update(np.asarray(seg2, dtype=np.int64).tobytes())
        h.update(np.asarray([np.concatenate(q) for q in seg3], dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda", 0)
    bad = 0
    for s in range(scenes):
        V, S, N, W = [(16, 700, 8, 2), (20, 1200, 10, 5), (12, 2000, 12, 8), (24, 500, 6, 3)][s % 4]
        scene = make_scene(V, S, N, seed=7100 + s)

        def mk():
            l = Line3D("", matchingNeighbors=N)
            l.keep_view_matches(True)
            load_scene(l, scene)
            l.prepare()
            return l
        ref = mk()
        ref.match_views(); ref.finish(False)
        want = digest(ref, scene)
        ref.close()
        slot = 10 * S * N // W + 1024
        ls = [mk() for _ in range(W)]
        n_views, slot_bytes = [l.shard_open(r, W, slot) for r, l in enumerate(ls)][0]
        gathered = torch.zeros(n_views * W * slot_bytes, dtype=torch.uint8, device=dev)
        send = [torch.zeros(n_views * slot_bytes, dtype=torch.uint8, device=dev) for _ in range(W)]
        torch.cuda.synchronize()
        for k in range(n_views):
            for r, l in enumerate(ls):
                l.shard_enqueue(k, send[r].data_ptr() + k * slot_bytes, gathered.data_ptr())
            torch.cuda.synchronize()
            if ls[0].shard_view_verified(k):
                for r in range(W):
                    gathered[(k * W + r) * slot_bytes:(k * W + r + 1) * slot_bytes].copy_(send[r][k * slot_bytes:(k + 1) * slot_bytes])
            torch.cuda.synchronize()
            for l in ls:
                l.shard_mark(k)
        for l in ls:
            l.shard_close(False)
        ok = True
        for r, l in enumerate(ls):
            l.shard_run(r, W, slot, "replay", gathered.data_ptr(), commit="device")
            torch.cuda.synchronize()
            l.finish(False)
            if digest(l, scene) != want:
                ok = False
                print("  rank %d differs" % r)
            l.close()
        bad += not ok
        print("scene %d (%d x %d x %d, world %d): %s" % (s, V, S, N, W, "ok" if ok else "DIFFERENT"))
    print("soak done: %d mismatches" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
