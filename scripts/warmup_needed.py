#!/usr/bin/env python3
"""How long a warm-up a cold-started block needs on a scene before its speculation holds -- asked of the library's own check:
    python3 scripts/warmup_needed.py VIEWS SEGMENTS NEIGHBOURS RANKS warmup [warmup ...]          (L3D_WARMUP_MODE=partition: l3d_line3d_partition_run)
RANKS virtual ranks (threads, all-gather through the host) run l3d_line3d_block_run with each warm-up (in views); reported: rounds of warm
re-runs and blocks re-run (0 = every rank's check passed at once).  No kept list goes to the host: digests on the device decide."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from line3d_amd.pipeline import Line3D, load_scene   # noqa: E402
from line3d_amd.synth import make_scene              # noqa: E402
from helpers import thread_exchange                   # noqa: E402

V, S, N, W = (int(x) for x in sys.argv[1:5])
warmups = [int(x) for x in sys.argv[5:]]
scene = make_scene(V, S, N, seed=20260)
ls = []
for r in range(W):
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, scene)
    l.prepare()
    ls.append(l)
for warm in warmups:
    make, calls = thread_exchange(W)
    verdicts, errors = [None] * W, []

    def run(r):
        try:
            verdicts[r] = (ls[r].partition_run if os.environ.get("L3D_WARMUP_MODE") == "partition" else ls[r].block_run)(r, W, make(r), None, warm)
        except Exception as e:      # noqa: BLE001
            errors.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    info = ls[0].partition_info() if not errors else {}
    print(json.dumps(dict(shape=[V, S, N], ranks=W, warmup_views=warm, warmup_windows=round(warm / (N / 2), 2), verdicts=verdicts, rounds=info.get("recovery_rounds"),
                          blocks_rerun=info.get("blocks_rerun"), seconds=round(time.perf_counter() - t0, 2), errors=errors)), flush=True)
for l in ls:
    l.close()
