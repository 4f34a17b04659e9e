#!/usr/bin/env python3
"""BASELINE configs[4]'s per-view shape as a PARTITIONED job, validated on one GPU:
    python3 scripts/validate_partition_big.py ref  VIEWS SEGMENTS NEIGHBOURS out.json       # the one chain on one GPU: per-view sha256 of every kept list, affinity list, lines
    python3 scripts/validate_partition_big.py part VIEWS SEGMENTS NEIGHBOURS ref.json [W] [warm-up views]
    python3 scripts/validate_partition_big.py seg  VIEWS SEGMENTS NEIGHBOURS ref.json [W]                   # the segment-sharded run, partitioned (l3d_shard_chain_partition)
`part`: W virtual ranks (threads of one process, all-gather through the host) run l3d_line3d_partition_run + finish_sharded; every rank hashes the kept
lists of ITS block's views straight out of its arena (together: every view of the scene), and its affinity list / lines after the collective finish.
Everything must equal the `ref` run byte for byte.  The ranks' CHAINS run one after the other (a rank starts when its predecessor reaches the first
collective and has released its chain scratch): eight chains' candidate rings at once would not fit the one GPU the eight ranks share here -- on a
node every rank has its own 288 GB.  Reports the HBM in use at the end and the peak seen."""
import ctypes as C
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from line3d_amd.pipeline import Line3D, load_scene   # noqa: E402
from line3d_amd.synth import make_scene              # noqa: E402

hip = C.CDLL("libamdhip64.so")
try:                                    # (the kept lists of 256 dense views are tens of GB: a fast digest when there is one -- both modes of one validation use the same)
    import xxhash

    def list_digest(b):
        return xxhash.xxh3_128_hexdigest(b)
except ImportError:                     # pragma: no cover
    def list_digest(b):
        return hashlib.sha256(b).hexdigest()


def hbm_used_gb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return round((total.value - free.value) / 2**30, 2)


def result_digest(l):
    h = hashlib.sha256()
    A, n = l.affinity()
    h.update(A.tobytes())
    lines = l.getResult()
    for seg2, seg3 in lines:
        h.update(np.array(seg2, np.int64).tobytes())
        for P, Q in seg3:
            h.update(np.asarray(P, np.float64).tobytes()); h.update(np.asarray(Q, np.float64).tobytes())
    return dict(affinity_entries=int(len(A)), nodes=int(n), lines=len(lines), sha256=h.hexdigest())


def main():
    mode = sys.argv[1]
    V, S, N = (int(x) for x in sys.argv[2:5])
    path = sys.argv[5]
    scene = make_scene(V, S, N, seed=20260)
    if mode == "ref":
        l = Line3D("", matchingNeighbors=N)
        load_scene(l, scene)
        l.prepare()
        t0 = time.perf_counter(); l.match_views(); t_match = time.perf_counter() - t0
        ctx = l.context()
        kept = []
        for k in range(V):
            m = ctx.chain_kept_list(k)
            kept.append([int(len(m)), list_digest(m.tobytes())])
        used = hbm_used_gb()
        t0 = time.perf_counter(); l.finish(False); t_fin = time.perf_counter() - t0
        out = dict(shape=[V, S, N], match_views_s=round(t_match, 3), finish_s=round(t_fin, 3), kept=kept, kept_total=sum(k[0] for k in kept), hbm_after_match_views_gb=used,
                   hbm_after_finish_gb=hbm_used_gb(), result=result_digest(l), fill_counts=list(ctx.last_fill_counts()))
        json.dump(out, open(path, "w"))
        print(json.dumps({k: v for k, v in out.items() if k != "kept"}))
        l.close()
        return 0
    ref = json.load(open(path))
    assert ref["shape"] == [V, S, N]
    W = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    warm = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    from helpers import thread_exchange
    make, calls = thread_exchange(W, on_device=(mode == "seg"), timeout=900.0)
    gates = [threading.Event() for _ in range(W + 1)]
    gates[0].set()
    peak = [0.0]

    def gated(r):
        inner = make(r)
        first = [True]

        def exchange(user, view, send, recv, slot_bytes, w, stream):
            if first[0]:
                first[0] = False
                peak[0] = max(peak[0], hbm_used_gb())
                gates[r + 1].set()                      # this rank's chain is done and its scratch released: the next rank may start
            return inner(user, view, send, recv, slot_bytes, w, stream)
        return exchange
    ls = []
    seg = mode == "seg"
    slot_records = int(float(os.environ.get("L3D_VALIDATE_SLOT", "2.2")) * max(k[0] for k in ref["kept"]) / W) + 4096 if seg else 0      # (a rank's segment range of the densest view: ranges are not equally rich)
    for r in range(W):
        l = Line3D("", matchingNeighbors=N)
        load_scene(l, scene)
        l.prepare()
        if seg:                                              # this rank's arena: its block and 2 x reach (= the neighbour count on these scenes) either side
            b0, b1 = (V * r) // W, (V * (r + 1)) // W
            l.context().set_chain_capacities(0, int(1.02 * sum(ref["kept"][k][0] for k in range(max(0, b0 - N), min(V, b1 + N)))) + 1000000)
        elif os.environ.get("L3D_VALIDATE_ARENA"):          # a fixed arena per rank instead of growth by doubling (eight ranks share ONE GPU here): what the rank's
            # views hold in the reference run -- its block, `reach2` views taken over in front (a warm re-run) and `reach2` views behind -- + 2 %
            reach2 = N
            b0, b1 = (V * r) // W, (V * (r + 1)) // W
            lo, hi = max(0, b0 - max(reach2, warm if warm >= 0 else 3 * N)), min(V, b1 + reach2)
            l.context().set_chain_capacities(0, int(1.02 * sum(ref["kept"][k][0] for k in range(lo, hi))) + 1000000)
        ls.append(l)
    out = dict(shape=[V, S, N], world=W, mode=mode, warmup_views=warm, slot_records=slot_records, hbm_after_prepare_gb=hbm_used_gb())
    bad, errors, infos, res = [], [], [None] * W, [None] * W
    t_run = [0.0] * W

    def run(r):
        try:
            if not seg:
                gates[r].wait()
            t0 = time.perf_counter()
            if seg:         # (all ranks at once: every rank works on its 1/W of every view's source segments)
                ls[r].shard_run(r, W, slot_records, make(r), None, commit="partition")
                ok = True
            else:
                ok = ls[r].partition_run(r, W, gated(r), None, warm)
            t_run[r] = time.perf_counter() - t0
            assert ok, "verdict 1"
            infos[r] = ls[r].partition_info()
            ctx = ls[r].context()
            for k in range(infos[r]["own"][0], infos[r]["own"][1]):      # (the dense map and the chain are both in view order on these scenes)
                m = ctx.chain_kept_list(k)
                if [int(len(m)), list_digest(m.tobytes())] != ref["kept"][k]:
                    bad.append("rank %d view %d: kept list differs (%d vs %d records)" % (r, k, len(m), ref["kept"][k][0]))
            peak[0] = max(peak[0], hbm_used_gb())
            ls[r].finish_sharded(False)
            res[r] = result_digest(ls[r])
            if res[r] != ref["result"]:
                bad.append("rank %d: result differs: %r vs %r" % (r, res[r], ref["result"]))
        except Exception as e:      # noqa: BLE001
            errors.append((r, repr(e)))
            gates[min(W, r + 1)].set()
            make.abort()                                # (the other ranks' next exchange fails instead of waiting for this one)
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    out.update(wall_s=round(time.perf_counter() - t0, 2), partition_run_s=[round(t, 2) for t in t_run], errors=errors, mismatches=bad[:10], n_mismatches=len(bad),
               hbm_peak_seen_gb=peak[0], hbm_at_end_gb=hbm_used_gb(), infos=infos, result=res[0], ref_result=ref["result"],
               fill_counts=list(ls[0].context().last_fill_counts()) if not errors else None, exchanges=[c[0] for c in calls][:80])
    print(json.dumps(out))
    for l in ls:
        l.close()
    return 1 if (bad or errors) else 0


if __name__ == "__main__":
    sys.exit(main())
