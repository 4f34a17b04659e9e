"""View sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

matchViews is a dependency chain over views (the verified matches of a view are candidates of its later
neighbours), so views are NOT distributed; instead every rank computes the same view at the same time on a
1/world slice of its SOURCE segments -- the verification of a source segment only reads candidates of that
segment, so the slices are independent -- and the per-view kept lists (32-byte records, a few MB) are
all-gathered before the replicated host bookkeeping (commit).  The concatenation in rank order is the
(segment, camera, target)-sorted list of the unsharded run, bit for bit.
"""
from __future__ import annotations

import numpy as np

from .capi import MATCH_DTYPE


def seg_range(S: int, rank: int, world: int):
    return (S * rank) // world, (S * (rank + 1)) // world


def allgather_bytes(payload: np.ndarray, dist, device=None):
    """Variable-length all-gather of a uint8 array: counts first, then one padded all_gather."""
    import torch
    n = torch.tensor([payload.size], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload.size:
        buf[: payload.size] = torch.from_numpy(payload).to(device) if device is not None else torch.from_numpy(payload)
    outs = [torch.zeros_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, buf)
    return [o[:c].cpu().numpy() for o, c in zip(outs, counts)]


def pack(matches: np.ndarray, best: np.ndarray) -> np.ndarray:
    head = np.array([len(matches), len(best)], dtype=np.int64).view(np.uint8)
    return np.concatenate([head, matches.view(np.uint8).ravel(), best.view(np.uint8).ravel()])


def unpack(buf: np.ndarray):
    nm, nb = np.frombuffer(buf[:16].tobytes(), dtype=np.int64)
    m = np.frombuffer(buf[16:16 + 32 * nm].tobytes(), dtype=MATCH_DTYPE)
    b = np.frombuffer(buf[16 + 32 * nm:16 + 32 * nm + 4 * nb].tobytes(), dtype=np.float32)
    return m, b


def match_views_sharded(l3d, rank: int, world: int, dist, compute=None, device=None):
    """Line3D::matchViews (line3D.cc:620-648) with every view's source segments sharded over `world` ranks.
    `compute(view_id, s0, s1) -> (matches, median, best_depths)` defaults to the HIP path of `l3d`; the CPU
    tests inject the oracle here to exercise the protocol under gloo."""
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    if compute is None:
        compute = l3d.match_view_compute
    ids, ns = l3d.match_begin()
    for vid, S in zip(ids.tolist(), ns.tolist()):
        if l3d.view_num_to_be_matched(vid) == 0:
            # cudawrapper.cu:877-878: nothing is computed, every rank holds the identical list already
            m, med, _ = compute(vid, 0, S)
            l3d.match_view_commit(vid, m, None, 1.0)
            continue
        s0, s1 = seg_range(S, rank, world)
        m, _med, best = compute(vid, s0, s1)
        parts = [unpack(b) for b in allgather_bytes(pack(m, best), dist, device)]
        allm = np.concatenate([p[0] for p in parts]) if parts else m
        allb = np.concatenate([p[1] for p in parts]) if parts else best
        l3d.match_view_commit(vid, allm, allb)
    l3d.match_end()


def default_slot_records(n_segments: int, n_neighbors: int, world: int) -> int:
    """Kept matches one rank may produce for one view.  The synthetic scenes keep 1.7 (64 views) to 3.5 (512 views: the later
    turns of the helix keep more) matches per (source segment, neighbour) on average, the densest view of the 512-view scene
    5.8, and a rank's segment range up to 6.5 (19 344 records at 8 ranks): 10 per (segment, neighbour) leaves room.  An
    overflow is a verdict every rank reads out of the slot headers, never silent: the native run reopens with larger slots
    and remembers the size for later passes."""
    return max(1024, (10 * n_segments * n_neighbors) // world + 1024)


def match_views_chain_sharded(l3d, rank: int, world: int, dist, commit: bool = True, slot_records: int | None = None,
                              n_segments: int = 2000, n_neighbors: int = 12, ahead: int = 12):
    """Line3D::matchViews as the device-resident chain with every view's source segments sharded over `world` ranks
    (one process per GPU).  Per view: the library enqueues this rank's kernels up to its kept-list slot, the slots are
    all-gathered with RCCL (`all_gather_into_tensor`, enqueued on the library's stream: no host synchronisation), later
    views read the gathered slots on the device.  Ranks with commit=True trail behind and do the host bookkeeping
    (the concatenation of the ranks' lists in rank order is the sorted unsharded list)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    if slot_records is None:
        slot_records = default_slot_records(n_segments, n_neighbors, world)
    n_views, slot_bytes = l3d.shard_open(rank, world, slot_records)
    ext = torch.cuda.ExternalStream(l3d.stream_ptr(), device=dev)
    failure = None          # a rank that hits an error keeps feeding the collectives so that all ranks stay in lock step
    try:
        with torch.cuda.stream(ext):
            gathered = torch.zeros(n_views * world * slot_bytes, dtype=torch.uint8, device=dev)
            send = torch.zeros(n_views * slot_bytes, dtype=torch.uint8, device=dev)
            gbase, sbase = gathered.data_ptr(), send.data_ptr()
            fetched = 0

            def guarded(fn, *a):
                nonlocal failure
                if failure is None:
                    try:
                        fn(*a)
                    except Exception as e:      # noqa: BLE001
                        failure = e

            for k in range(n_views):
                guarded(l3d.shard_enqueue, k, sbase + k * slot_bytes, gbase)
                if l3d.shard_view_verified(k):
                    out = gathered[k * world * slot_bytes:(k + 1) * world * slot_bytes]
                    if dist is not None:
                        dist.all_gather_into_tensor(out, send[k * slot_bytes:(k + 1) * slot_bytes])
                    else:
                        out.copy_(send[k * slot_bytes:(k + 1) * slot_bytes], non_blocking=True)
                guarded(l3d.shard_mark, k)
                if commit:
                    while fetched <= k - ahead and failure is None:
                        guarded(l3d.shard_fetch, fetched)
                        fetched += 1
            if commit:
                while fetched < n_views and failure is None:
                    guarded(l3d.shard_fetch, fetched)
                    fetched += 1
            ext.synchronize()
            if dist is not None:                # every rank learns whether any rank failed (e.g. a slot overflow seen by rank 0)
                flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) and failure is None:
                    failure = RuntimeError("another rank reported a failure of the sharded chain")
    finally:
        l3d.shard_close(commit and failure is None)
    if failure is not None:
        raise failure
    return n_views


# ---- the sharded chain as one native call, RCCL called directly -------------------------------------------------------

class RcclLink:
    """An RCCL communicator of the job's ranks for the library's own stream, created on the RCCL build the process
    already has loaded (PyTorch-ROCm's librccl.so): the unique id travels through torch.distributed, the data path
    (`ncclAllGather` per view, enqueued by libline3d_amd on its stream) never touches the interpreter."""

    def __init__(self, rank: int, world: int, dist, device_index: int):
        import ctypes as C
        import os
        import torch
        if world == 1:      # a single-rank communicator needs no network: keep RCCL's bootstrap away from interface probing
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = C.CDLL(path)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        uid = UniqueId()
        ok = 1
        if rank == 0:
            ok = 0 if self.lib.ncclGetUniqueId(C.byref(uid)) else 1
        if dist is not None and world > 1:
            # [ok flag | 128-byte id] from rank 0: a failure there reaches every rank instead of leaving them in the broadcast
            t = torch.zeros(129, dtype=torch.uint8)
            if rank == 0:
                t[0] = ok
                t[1:] = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", device_index))
            dist.broadcast(t, 0)
            t = t.cpu()
            ok = int(t[0])
            C.memmove(C.byref(uid), t[1:].numpy().tobytes(), 128)
        if not ok:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        torch.cuda.set_device(device_index)
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), C.c_int(world), uid, C.c_int(rank))
        if rc:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)

        class Link(C.Structure):            # == l3d_rccl_link (include/line3d_amd.h)
            _fields_ = [("comm", C.c_void_p), ("all_gather", C.c_void_p)]

        self.link = Link(self.comm.value, C.cast(self.lib.ncclAllGather, C.c_void_p).value)

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def match_views_chain_native(l3d, rank: int, world: int, link: "RcclLink | None", commit=True,
                             slot_records: int | None = None, n_segments: int = 2000, n_neighbors: int = 12):
    """Line3D::matchViews as the sharded resident chain in ONE native call (l3d_shard_chain_run): per view the library
    enqueues this rank's kernels, the RCCL all-gather of the ranks' kept-list slots and a completion event on its own
    stream; a host thread of the library trails behind with the bookkeeping on the ranks that commit (commit=True);
    commit="device": no rank hands lists to the host -- this rank builds matchViews' products on its device from the gathered slots;
    commit="partition": as "device", but the rank keeps the records and builds the rows of its block of views only (l3d_shard_chain_partition:
    the partitioned job without speculation; l3d.finish_sharded() follows on every rank)."""
    if slot_records is None:
        slot_records = default_slot_records(n_segments, n_neighbors, world)
    if world == 1 and link is None:
        return l3d.shard_run(rank, world, slot_records, "local", None, commit)
    return l3d.shard_run(rank, world, slot_records, "rccl", link.link, commit)


def match_views_blocks(l3d, rank: int, world: int, link: "RcclLink | None", warmup_views: int = -1) -> bool:
    """Line3D::matchViews with the VIEWS sharded over the ranks in blocks (l3d_line3d_block_run): every rank runs the full-width single-GPU
    chain on its block of views + a warm-up in front of it, started cold; the ranks verify the speculation with digests of the kept lists and,
    when it holds, all-gather their blocks, build their own block's rows of matchViews' products and all-gather the pieces.  Four data collectives per pass (+ three of status words) instead of one per view.
    False: the speculation did not hold on this scene (identical on every rank): call match_views_chain_native."""
    if world == 1 and link is None:
        return l3d.block_run(rank, world, "local", None, warmup_views)
    return l3d.block_run(rank, world, "rccl", link.link, warmup_views)


def match_views_blocks_stepwise(l3d, rank: int, world: int, dist, warmup_views: int, window: int, compute=None, recover: bool = True, info=None) -> bool:
    """The block protocol of l3d_match_chain_blocks spelled out over the step-wise interface (match_begin / match_view_compute /
    match_view_commit) and torch.distributed object collectives: the reference form of the protocol, what the CPU test drives under gloo with the
    oracle as `compute`.  Rank r computes views [B_r - warmup_views, B_{r+1}) of the processing order, the views in front committed EMPTY (a cold
    start); digests of every computed kept list are all-gathered; rank r MISSES when the `window` views in front of B_r are not equal in r's warm-up
    and in r-1's lists.  recover (round 5): every rank that missed takes over its predecessor's last `window` views and re-runs its block warm from
    them -- all missed blocks at once --, the digests are exchanged again, until nobody misses (rank j is exact after round j at the latest);
    recover = False: any miss returns False (nothing committed, the same on every rank).  Nobody misses: the blocks are all-gathered and every
    rank replays ALL commits in order from the gathered lists (returns True, state = the unsharded run's).  info (a dict): rounds, blocks re-run."""
    import hashlib
    if compute is None:
        compute = l3d.match_view_compute
    ids, ns = l3d.match_begin()
    ids, ns = ids.tolist(), ns.tolist()
    n = len(ids)
    begin = lambda r: (n * r) // world          # noqa: E731
    own0, own1 = begin(rank), begin(rank + 1)
    empty = np.zeros(0, MATCH_DTYPE)
    mine = {}

    def run_from(first, taken):
        """this rank's chain over [first, own1): nothing known in front of `first` but the views in `taken` (k -> (list bytes, best bytes or None))"""
        l3d.match_begin()
        for k in range(own1):
            vid, S = ids[k], ns[k]
            if k in taken:
                mb, bb = taken[k]
                m = np.frombuffer(mb, dtype=MATCH_DTYPE).copy()
                l3d.match_view_commit(vid, m, None if bb is None else np.frombuffer(bb, dtype=np.float32).copy(), 1.0) if bb is None else \
                    l3d.match_view_commit(vid, m, np.frombuffer(bb, dtype=np.float32).copy())
                mine[k] = (mb, bb)
                continue
            if k < first:
                l3d.match_view_commit(vid, empty, None, 1.0)              # nothing is known about the views in front of the cold start
                continue
            m, med, best = compute(vid, 0, S)
            early = l3d.view_num_to_be_matched(vid) == 0                  # cudawrapper.cu:877-878: the list comes back as it is, nothing was computed
            mine[k] = (np.ascontiguousarray(m).tobytes(), None if early else np.ascontiguousarray(best, dtype=np.float32).tobytes())
            l3d.match_view_commit(vid, m, None if early else best, 1.0 if early else med)

    run_from(0 if rank == 0 else max(0, own0 - warmup_views), {})
    rounds = reruns = 0
    while True:
        digests = {k: hashlib.sha256(v[0]).hexdigest() for k, v in mine.items()}
        tables = [None] * world
        dist.all_gather_object(tables, digests)
        miss = [False] * world
        for r in range(1, world):
            b = begin(r)
            lo = max(0, b - window)
            if b - window < begin(r - 1) and not recover:
                miss[r] = True
            miss[r] = miss[r] or any(k not in tables[r] or tables[r].get(k) != tables[r - 1].get(k) for k in range(lo, b))
        if not any(miss):
            break
        if not recover or rounds >= world or any(begin(r) - window < begin(r - 1) for r in range(1, world)):
            return False
        tails = [None] * world
        dist.all_gather_object(tails, {k: mine[k] for k in range(max(0, own1 - window), own1)} if rank + 1 < world and miss[rank + 1] else {})
        if miss[rank]:
            mine.clear()
            run_from(own0, dict(tails[rank - 1]))
        rounds += 1
        reruns += sum(miss)
    if info is not None:
        info.update(rounds=rounds, blocks_rerun=reruns)
    blocks = [None] * world
    dist.all_gather_object(blocks, {k: mine[k] for k in range(own0, own1)})
    l3d.match_begin()                                                   # the one chain's state, replayed from the gathered lists
    for k in range(n):
        r = min(world - 1, (k * world) // max(1, n))
        while r + 1 < world and begin(r + 1) <= k:
            r += 1
        while r > 0 and begin(r) > k:
            r -= 1
        mb, bb = blocks[r][k]
        m = np.frombuffer(mb, dtype=MATCH_DTYPE).copy()
        if bb is None:
            l3d.match_view_commit(ids[k], m, None, 1.0)
        else:
            l3d.match_view_commit(ids[k], m, np.frombuffer(bb, dtype=np.float32).copy())
    l3d.match_end()
    return True
