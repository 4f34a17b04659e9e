#!/usr/bin/env python3
"""Per-rank critical path of the sharded resident chain at world size W, measured on ONE GPU.

Pass 1 runs W virtual ranks (W pipelines on the same GPU, the all-gather emulated with device copies) and records
every view's gathered slots.  Pass 2 replays rank R alone: its own kernels run for real, the all-gather of view k is
replaced by one device copy of the recorded gathered block (a lower bound for the collective), rank 0 also does the
host bookkeeping.  ms/view of pass 2 x (views of the W-GPU job) estimates the W-GPU wall time without a W-GPU node.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--segments", type=int, default=2000)
    ap.add_argument("--neighbors", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ahead", type=int, default=12)
    ap.add_argument("--no-commit", action="store_true", help="rank 0 without the host hand-over (what a non-committing rank with this range would take)")
    ap.add_argument("--device-commit", action="store_true", help="the replayed rank builds matchViews' products on its device from the gathered slots (no host hand-over)")
    ap.add_argument("--partition", action="store_true", help="the replayed rank of a PARTITIONED job (l3d_shard_chain_partition): it retires its block of views only and builds its share of the "
                                                             "products; pass 1 records without host bookkeeping")
    ap.add_argument("--profile", action="store_true", help="after the timed replays: one more with every kernel bracketed, per-kernel ms of the rank")
    ap.add_argument("--cand-cap", type=int, default=0, help="candidate capacity of every rank's chain (0: the library's first guess; a recording whose guess overflows cannot be replayed)")
    ap.add_argument("--slot-records", type=int, default=0, help="kept matches one rank may produce for one view (0: distributed.default_slot_records; dense scenes need more)")
    args = ap.parse_args()
    import torch
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    from line3d_amd.distributed import default_slot_records
    W = args.world
    scene = make_scene(args.views, args.segments, args.neighbors, seed=20260)
    dev = torch.device("cuda", 0)
    slot_records = args.slot_records or default_slot_records(args.segments, args.neighbors, W)

    def mk():
        l = Line3D("", matchingNeighbors=args.neighbors)
        load_scene(l, scene)
        l.prepare()
        if args.cand_cap:
            l.context().set_chain_capacities(args.cand_cap, 0)
        return l

    # pass 1: record
    ls = [mk() for _ in range(W)]
    geo = [l.shard_open(r, W, slot_records) for r, l in enumerate(ls)]
    n_views, slot_bytes = geo[0]
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
        if not args.partition:
            ls[0].shard_fetch(k)
    for r, l in enumerate(ls):
        l.shard_close(r == 0 and not args.partition)
    kept = ls[0].stats()["kept"]
    for l in ls[1:]:
        l.close()
    recorded = gathered.clone()
    del send

    # pass 2: replay one rank through the native loop (l3d_shard_chain_run + the replay exchange)
    l = ls[0]
    R = args.rank
    for rep in range(args.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l.shard_run(R, W, slot_records, "replay", recorded.data_ptr(),
                    commit=("partition" if args.partition else "device" if args.device_commit else (R == 0 and not args.no_commit)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep:
            print("world %d rank %d: %d views x %d segs: %.2f ms (%.1f us/view), kept %d (recorded %d), slot %d KB"
                  % (W, R, n_views, args.segments, dt * 1e3, dt / n_views * 1e6, int(l.stats()["kept"]), int(kept), slot_bytes // 1024))
    if args.profile:       # one more replay with every kernel bracketed by HIP events (no graphs then): where the rank's time goes
        ctx = l.context()
        ctx.profile_only(None); ctx.profile_enable(True); ctx.profile_reset()
        l.shard_run(R, W, slot_records, "replay", recorded.data_ptr(),
                    commit=("partition" if args.partition else "device" if args.device_commit else (R == 0 and not args.no_commit)))
        torch.cuda.synchronize()
        print("kernels_ms", {k: round(v[1], 3) for k, v in ctx.profile_all().items() if v[0]})
        ctx.profile_enable(False)
    l.close()


if __name__ == "__main__":
    main()
