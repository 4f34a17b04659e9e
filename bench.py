#!/usr/bin/env python3
"""bench.py -- Line3D matching hot path (matchViews: stage 1 pair test + stage 2 verification + filter +
host bookkeeping) on synthetic helix scenes.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full matchViews pass over the scene.  Workload: BASELINE.json configs[1] per GPU --
64*N views x 2000 segments, 12 neighbours (N=1: config 2 itself; N=8: config 3) -> weak scaling.
For N>1 every rank holds the scene (it is small).  From 4 ranks on the VIEWS are sharded in blocks: every rank runs the full-width
single-GPU chain on its block + a warm-up started cold, the speculation is verified with digests of the kept lists and a block
that missed is re-run warm; the blocks and the pieces of matchViews' products are all-gathered over RCCL (DESIGN.md section 6 ii).
With 2-3 ranks every rank computes a 1/N source-segment range of each view and the per-view kept lists are all-gathered (6 i).
No rank hands lists to the host in either mode.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def usable_cpus():
    """CPUs this process may actually use: affinity mask and cgroup quota (a container can see 256 CPUs and own 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views-per-gpu", type=int, default=64)
    ap.add_argument("--segments", type=int, default=2000)
    ap.add_argument("--neighbors", type=int, default=12)
    ap.add_argument("--seed", type=int, default=20260)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-segments", type=int, default=800)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip the (untimed) rest of compute3Dmodel after the timed passes")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold measurement (first matchViews + first finish of the fresh object)")
    ap.add_argument("--partition", choices=["segments", "blocks"], default=None,
                    help="matchViews as a PARTITIONED job (BASELINE configs[4]: nothing replicated; DESIGN.md section 6 iii / iv) instead of the default multi-GPU modes: "
                         "`segments` = l3d_shard_chain_partition (source segments sharded, exact on every scene), `blocks` = l3d_match_chain_partition (views sharded, "
                         "speculated + verified).  A step is the partitioned matchViews of every rank; the chain scratch is kept between steps (L3D_PART_RELEASE=0: "
                         "a one-shot job releases it).  Works at --gpus 1 too (a communicator of one rank).")
    return ap.parse_args()


def cpu_baseline(scene, n_neighbors, sample_segments, lists=None):
    """The oracle (scalar C restatement of the reference formulation, 1 thread) on a bounded sample of the same workload: a
    MID-CHAIN view of the scene (half of its neighbours still to be matched, the other half already matched: their kept
    matches towards it -- taken from `lists`, the GPU run's kept lists -- come back as existing matches, line3D.cc:806,838-872),
    the first `sample_segments` source segments, stage 1 + sort + stage 2 + filter.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import l3d_oracle_pipeline as op
    V = len(scene.views)
    vid = scene.views[V // 2]["id"] if lists is not None else scene.views[0]["id"]
    o = op.OracleLine3D(matching_neighbors=n_neighbors, use_collinearity=False)
    for v in scene.views:
        o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    o.computation = True
    o.matched, o.potential = {}, {}
    o.find_visual_neighbors()
    o.transform_geometry()
    for n in o.visual_neighbors[vid]:
        o._fundamental(vid, n)
    existing = np.zeros(0, dtype=op.MATCH_DTYPE)
    if lists is not None:
        for a in range(vid):                              # matched_ after the earlier views (line3D.cc:875-881; ids are 0..V-1 in order)
            for nb in o.visual_neighbors[a]:
                o.matched.setdefault(a, {})[nb] = True
                if a in o.visual_neighbors.get(nb, []):
                    o.matched.setdefault(nb, {})[a] = True
    mv = o.marshal_view(vid)
    if lists is not None:
        ex = []
        for a in range(vid):
            m = lists[a]
            sel = m[m["camID2"] == vid]
            if len(sel) and a in mv["g2l"]:
                r = np.zeros(len(sel), dtype=op.MATCH_DTYPE)
                r["segID1"], r["segID2"], r["camID2"] = sel["segID2"], sel["segID1"], mv["g2l"][a]
                r["depths"] = sel["depths"][:, [2, 3, 0, 1]]
                ex.append(r)
        if ex:
            existing = np.concatenate(ex)
    t0 = time.time()
    m, med, stats = op.compute_pairwise_matches(
        o.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
        mv["centers"], mv["P"], mv["tbm"], existing, mv["l2g"], mv["k_upper"], mv["k_lower"],
        3.5, 10.0, mv["spatial_k"], seg_range=(0, sample_segments), want_stats=True)
    dt = time.time() - t0
    # the same formulation on all host threads: one source-segment range of the same view per thread (the C oracle releases
    # the GIL), bounded to ~the same wall time
    import threading
    nthreads = max(1, min(usable_cpus(), 64))
    S = len(mv["src_segs"])
    per = max(1, min(S // nthreads, max(1, sample_segments // 2)))
    res = [None] * nthreads

    def work(i):
        s0 = (i * per) % max(1, S - per + 1)
        _m, _med, st = op.compute_pairwise_matches(
            o.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
            mv["centers"], mv["P"], mv["tbm"], existing, mv["l2g"], mv["k_upper"], mv["k_lower"],
            3.5, 10.0, mv["spatial_k"], seg_range=(s0, s0 + per), want_stats=True)
        res[i] = st[3]

    all_threads = None
    if nthreads > 1 and per * nthreads <= 8 * S:
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t1 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dta = time.time() - t1
        all_threads = dict(value=float(sum(r for r in res if r)) / dta, cores=nthreads, seconds=dta,
                           sample="%d threads (usable CPUs of this container) x %d source segments of view %d" % (nthreads, per, vid))
    port = dict(value=stats[3] / dt, cores=1, seconds=dt, verify_iterations_per_s=stats[2] / dt,
                note="oracle/l3d_oracle.c: the scalar C restatement of the same formulation")
    sample = ("view %d of %d of the bench scene (mid-chain: %d cameras still to match, %d existing reverse matches from the %d already "
              "matched), first %d of %d source segments: %d pairs, %d raw candidates, %.3g verify inner iterations, %.1f s on 1 thread"
              % (vid, V, len(mv["tbm"]), len(existing), len(mv["l2g"]) - len(mv["tbm"]), sample_segments, len(mv["src_segs"]), int(stats[3]),
                 int(stats[0]), stats[2], dt))
    out = dict(value=port["value"], unit="segment-pair affinities/s", cores=1, kind="port", all_threads=all_threads,
               sample=sample + " (oracle/l3d_oracle.c, reference formulation)", seconds=dt, verify_iterations_per_s=port["verify_iterations_per_s"])
    # beside it, half of the same sample with the reference's KERNEL TEXT: K_pairwise_matches and K_verify_matches assembled from cudawrapper.cu's lines
    # with builder-written table reads in place of the texture fetches (oracle/_spliced/libkernels_spliced.so -- corroboration, NOT "the reference
    # compiled here", hence not the headline of this object) inside the oracle's host code
    ref_path = os.path.join(ROOT, "oracle", "_spliced", "libkernels_spliced.so")
    if os.path.exists(ref_path):
        import ctypes as C
        ref = C.CDLL(ref_path)
        if hasattr(ref, "l3dref_pairwise_matches") and hasattr(ref, "l3dref_verify_matches"):
            half = max(1, sample_segments // 2)                  # (those kernels run at about a third of the port's rate: same wall time)
            ol = op.OracleLine3D(matching_neighbors=n_neighbors, use_collinearity=False, libm=True)
            try:
                op.set_reference_kernels(ol.lib, ref)
                t2 = time.time()
                _m, _med, st2 = op.compute_pairwise_matches(
                    ol.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
                    mv["centers"], mv["P"], mv["tbm"], existing, mv["l2g"], mv["k_upper"], mv["k_lower"],
                    3.5, 10.0, mv["spatial_k"], seg_range=(0, half), want_stats=True)
                dt2 = time.time() - t2
            finally:
                op.set_reference_kernels(ol.lib, None)
            out["reference_kernel_text_spliced"] = dict(
                value=st2[3] / dt2, cores=1, seconds=dt2, verify_iterations_per_s=st2[2] / dt2,
                sample="K_pairwise_matches + K_verify_matches from cudawrapper.cu's lines (g++ -O2; texture fetches replaced by builder-written table reads, "
                       "launch variables supplied by the builder: oracle/_spliced/libkernels_spliced.so) inside the oracle's restatement of "
                       "compute_pairwise_matches' host code, 1 thread, first %d source segments of view %d (%d pairs, %.1f s)" % (half, vid, int(st2[3]), dt2))
    return out


def profile_for_shape(kind, segments, neighbors):
    """profiles/<round>_<kind>.json of the newest round that has one: per-kernel PMC figures of this same command (scripts/measure_round.sh),
    stamped with the commit they were measured on -- NOT measured in this run."""
    # A per-launch instruction count / byte count is a property of the per-view SHAPE (segments, neighbours): only a summary collected on
    # this run's shape is used (`_shape` = [views, segments, neighbours] stamped by scripts/make_*.py; absent = the default 64 x 2000 x 12);
    # with none, roofline.frac is null and says why -- never a fraction from another shape's counters.
    import glob
    import re
    found = []
    for q in glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % kind)):
        m = re.match(r"r(\d+)_(?:[A-Za-z0-9]+_)?%s\.json$" % kind, os.path.basename(q))
        if not m:
            continue
        try:
            d = json.load(open(q))
        except (OSError, ValueError):
            continue
        shp = d.get("_shape", [64, 2000, 12])
        if (int(shp[1]), int(shp[2])) == (segments, neighbors):
            found.append((int(m.group(1)), q, d))
    if not found:
        return {}, None
    # within a round the FINAL summary is the un-suffixed r<N>_<kind>.json; named ones (r<N>_v2_<kind>.json: mid-round) only when there is no final
    _r, q, d = max(found, key=lambda t: (t[0], os.path.basename(t[1]) == "r%d_%s.json" % (t[0], kind), os.path.getmtime(t[1]), t[1]))
    return d, "profiles/%s @%s shape %s" % (os.path.basename(q), d.get("_commit", "unstamped"), "x".join(str(x) for x in d.get("_shape", [64, 2000, 12])))


def workload_name(n_gpus, V, S, N):
    """which BASELINE.json config the shape is"""
    if (S, N) == (2000, 12):
        if n_gpus == 1 and V == 64:
            return "BASELINE configs[1]"
        if V == 512:
            return "BASELINE configs[2]" + ("" if n_gpus == 8 else " (its 512 views on %d GPU%s)" % (n_gpus, "" if n_gpus == 1 else "s"))
        return "BASELINE configs[1] grown to %d views per GPU" % (V // max(1, n_gpus))
    if (S, N) == (4000, 24):
        return "BASELINE configs[4] per-view shape (4000 segments, 24 neighbours) at %d of its 2048 views" % V
    return "custom shape"


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_sharded = os.environ.get("L3D_BENCH_FORCE_SHARDED") == "1"     # exercise the multi-GPU code path with world = 1
    if world > 1 or (force_sharded and "RANK" in os.environ):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    from line3d_amd import distributed as l3dist

    V = args.views_per_gpu * n_gpus
    scene = make_scene(V, args.segments, args.neighbors, seed=args.seed)
    t0 = time.perf_counter()
    l3d = Line3D("", matchingNeighbors=args.neighbors, device=local_rank)
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    load_scene(l3d, scene)               # addImage_fixed_sim per view (host only: the segments go to HBM in prepare)
    t_add = time.perf_counter() - t0
    t0 = time.perf_counter()
    l3d.prepare()                        # neighbours, scene normalisation, residency of all segment arrays, collinearity of all views
    t_prepare = time.perf_counter() - t0
    t_setup = t_add + t_prepare
    ctx = l3d.context()
    cold = None
    if dist is None and not args.no_cold:
        # what ONE compute3Dmodel of a fresh object costs (line3D.cc:345-374): the first matchViews and the first finish of the process,
        # before any warm-up (the timed passes below are passes 2.. of the same object)
        t0 = time.perf_counter()
        l3d.match_views()
        t_m = time.perf_counter() - t0
        t0 = time.perf_counter()
        l3d.finish(False)
        t_f = time.perf_counter() - t0
        cold = dict(create_s=t_create, add_images_s=t_add, prepare_s=t_prepare, first_match_views_s=t_m, first_finish_s=t_f,
                    compute3Dmodel_total_s=t_prepare + t_m + t_f,
                    note="fresh process, fresh object, no warm-up: compute3Dmodel = prepare + matchViews + finish (no diffusion); create = context, "
                         "streams and the start of the code-object loading that overlaps add_images")

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # multi-GPU modes, best first; a failure on any rank moves ALL ranks to the next mode (agreed with an all-reduce so that
    # nobody is left waiting in a collective)
    MODES = ["blocks of views: every rank the full-width single-GPU chain on its 1/N of the views + a warm-up of 8 neighbour windows in front of it, started cold; "
             "the speculation verified with digests of the kept lists (one all-gather); a block that missed is re-run warm from its predecessor's lists (all missed "
             "blocks at once, then digests again); the blocks all-gathered, every rank builds its own block's rows of matchViews' products and the pieces are "
             "all-gathered -- no per-view collective; falls through to the next mode only when a block is shorter than the neighbour window",
             "native: resident chain, source segments sharded, RCCL all-gather of per-view kept slots enqueued by the library on its own stream, "
             "matchViews' products built on every rank's device from the gathered slots (no host bookkeeping)",
             "resident chain, source segments sharded, all-gather of per-view kept slots through torch.distributed on the library's stream",
             "per-view seam call, source segments sharded, all-gather of kept lists through the host"]
    sharded_mode = {"i": 0, "link": None}
    if dist is not None:
        try:
            sharded_mode["link"] = l3dist.RcclLink(rank, world, dist, local_rank)
        except Exception as e:      # noqa: BLE001
            print("rank %d: RCCL link failed (%r)" % (rank, e), file=sys.stderr)
        import torch
        flag = torch.tensor([0 if sharded_mode["link"] is not None else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            sharded_mode["i"] = 2                       # (no RCCL link: neither of the native modes)
        elif world < 4:
            sharded_mode["i"] = 1                       # (blocks of views pay a warm-up per rank: worth it from 4 ranks on)
        if os.environ.get("L3D_BENCH_MODE"):            # testing: start at a given multi-GPU mode (0 blocks, 1 native, 2 torch-driven, 3 per-view)
            sharded_mode["i"] = max(sharded_mode["i"] if int(flag.item()) else 0, int(os.environ["L3D_BENCH_MODE"]))

    class SpeculationFailed(RuntimeError):
        pass

    def step_once(mode):
        if mode == 0:       # views sharded in blocks; False = the verification failed (the same on every rank): next mode, for good
            if not l3dist.match_views_blocks(l3d, rank, world, sharded_mode["link"]):
                raise SpeculationFailed("the cold-started blocks did not reproduce the chain on this scene")
            return
        mode -= 1
        if mode == 0:       # no rank hands lists to the host: every rank builds matchViews' products on its device from the gathered slots
            l3dist.match_views_chain_native(l3d, rank, world, sharded_mode["link"], commit="device",
                                            n_segments=args.segments, n_neighbors=args.neighbors)
        elif mode == 1:
            l3dist.match_views_chain_sharded(l3d, rank, world, dist, commit=(rank == 0),
                                             n_segments=args.segments, n_neighbors=args.neighbors)
        else:
            l3dist.match_views_sharded(l3d, rank, world, dist)

    part_link = None
    if args.partition:
        part_link = sharded_mode["link"] if dist is not None else l3dist.RcclLink(0, 1, None, local_rank)
        if part_link is None:
            raise RuntimeError("--partition needs the RCCL link")
        ctx.set_option("L3D_PART_RELEASE", 0)
        args.no_extras = True
        part_slot = max(l3dist.default_slot_records(args.segments, args.neighbors, world), 65536 if args.segments * args.neighbors >= 48000 else 0)

    def step():
        if args.partition == "segments":
            l3d.shard_run(rank, world, part_slot, "rccl", part_link.link, commit="partition")
            return
        if args.partition == "blocks":
            if not l3d.partition_run(rank, world, "rccl", part_link.link, -1):
                raise RuntimeError("partitioned blocks: a block is shorter than the neighbour window (use --partition segments)")
            return
        if dist is None:
            l3d.match_views()
            return
        import torch
        while True:
            failed = 0
            try:
                step_once(sharded_mode["i"])
            except Exception as e:      # noqa: BLE001
                failed = 1
                print("rank %d: mode %d failed (%r)" % (rank, sharded_mode["i"], e), file=sys.stderr)
            flag = torch.tensor([failed], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if not int(flag.item()):
                return
            if sharded_mode["i"] == len(MODES) - 1:
                raise RuntimeError("every multi-GPU mode failed")
            sharded_mode["i"] += 1

    if dist is not None and args.warmup == 0 and not args.partition:
        step()          # (untimed: the multi-GPU mode is settled -- a mode that fails moves all ranks to the next one -- before anything is timed, warm-up or not)
    for _ in range(args.warmup):
        step()
    prof_all, dominant = {}, None
    if not args.no_profile:
        # one extra untimed pass with every kernel bracketed by HIP events: the per-kernel split and the dominant kernel;
        # in the timed region only that kernel is bracketed, so the measured throughput is (nearly) undisturbed
        ctx.profile_only(None)
        ctx.profile_enable(True)
        ctx.profile_reset()
        step()
        prof_all = {k: v for k, v in ctx.profile_all().items() if v[0]}
        dominant = max(prof_all.items(), key=lambda kv: kv[1][1])[0] if prof_all else None
        ctx.profile_only(dominant)
        ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = l3d.stats()
    prof = {} if (args.no_profile or dominant is None) else {dominant: ctx.profile_get(dominant)}
    ctx.profile_enable(False)
    ctx.profile_only(None)

    # whole-job pairs per step: every rank evaluates its 1/N share of every view's pairs
    pairs_local = st["pairs"]
    pairs_total = pairs_local
    raw_local = st["raw"]
    if dist is not None:
        import torch
        t = torch.tensor([pairs_local, raw_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        pairs_total, raw_total = float(t[0].item()), float(t[1].item())
    else:
        raw_total = raw_local

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = pairs_total * args.steps / dt
        # dominant kernel = the one with the largest summed duration on this rank
        roof = None
        traffic_json, traffic_src = profile_for_shape("traffic", args.segments, args.neighbors)
        valu_json, valu_src = profile_for_shape("valu", args.segments, args.neighbors)
        if traffic_json and valu_json and traffic_json.get("_commit") != valu_json.get("_commit"):
            # both summaries are passes of ONE measure_round.sh run: counters of two commits are not mixed in one line
            traffic_json, traffic_src = {}, "dropped: %s is not of the commit of %s" % (traffic_src, valu_src)
        # wave64 VALU instructions/s the chip can issue: 256 CUs x 4 SIMD-32s x 2.4 GHz, one wave64 instruction per SIMD every 2 cycles
        # (MI355X_MICROARCH.md, constants table: `v_fma_f32` (wave64) 2 cyc; 4 is what ONE wave alone sustains).  Equivalent to the 157.3 TFLOP/s
        # FP32 vector peak (64 lanes x 2 flop per FMA).  Transcendental / rcp / sqrt instructions cost twice that; the count below is unweighted,
        # profiles/r4_stalls.json carries the weighted figure (`valu_pipe_frac_trans_weighted`).
        PEAK_ISSUE = 256 * 4 * 2.4e9 / 2.0

        def kernel_roof(name, launches, ms, passes):
            # algorithmic HBM bytes per launch: one launch = one view.  The views this rank launched per pass: all of the
            # scene on one GPU, its block + warm-up in the block mode (then `raw` counts those views, `pairs` its own block: approximate per launch)
            nv = max(1, int(round(launches / max(1, passes)))) if launches else max(1, len(scene.views))
            R_per_launch = raw_local / nv
            pairs_per_launch = pairs_local / nv
            n_tbm = args.neighbors / 2.0
            alg = {
                "pair_mask": 16.0 * args.segments * (1 + n_tbm) + pairs_per_launch / 8.0,   # segments read once + 1 bit per pair
                "verify_window": 28.0 * R_per_launch,        # candidate meta 8 + depths 16 read, confidence 4 written
                "verify": 44.0 * R_per_launch,
                "pair_fill": pairs_per_launch / 8.0 + 24.0 * R_per_launch,
                "exist": 2 * 32.0 * st["kept"] / nv * n_tbm,  # the sources' kept lists are read twice (count, scatter)
            }.get(name, 0.0)
            avg_ms = ms / max(1, launches)
            hbm_achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic = traffic_json.get(name, {}).get("hbm_bytes_per_launch")
            hbm = dict(bound="hbm", achieved=hbm_achieved, peak=8000.0, unit="GB/s", frac=hbm_achieved / 8000.0, algorithmic_bytes_per_launch=alg, traffic=traffic,
                       traffic_source=traffic_src, note="SURVEY 8d's 4.5 B per pair x the pairs of one launch / its measured duration: the path's inputs are a few "
                                                        "hundred KB per view and stay in L2 / LDS, so this fraction is tiny by nature")
            # the bound of this path is FP32 VALU issue.  achieved = wave-level VALU instructions per launch (PMC, committed profile) /
            # the launch duration measured live in THIS run; peak = the chip's issue rate
            vi = valu_json.get(name, {}).get("valu_wave_insts_per_launch")
            out = dict(bound="valu", kernel=name, launches=launches, avg_launch_ms=avg_ms, hbm=hbm, traffic=traffic)
            if vi and avg_ms > 0:
                ach = vi / (avg_ms * 1e-3)
                out.update(achieved=ach / 1e9, peak=PEAK_ISSUE / 1e9, unit="G wave-instructions/s", frac=ach / PEAK_ISSUE, valu_wave_insts_per_launch=vi, source=valu_src,
                           note="issue fraction: SQ_INSTS_VALU per launch (rocprofv3 --pmc pass of this command, committed under profiles/ at the commit "
                                "named in `source`) / launch duration by HIP events in this run / (256 CUs x 4 SIMD-32s x 2.4 GHz / 2 cycles per wave64 instruction)")
            else:
                out.update(achieved=None, peak=PEAK_ISSUE / 1e9, unit="G wave-instructions/s", frac=None,
                           note="no committed PMC profile of this shape (%d segments, %d neighbours) under profiles/: the issue fraction is not computed from "
                                "another shape's counters (scripts/measure_round.sh with L3D_SHAPE / L3D_BENCH_ARGS collects one); roofline.hbm below is live"
                                % (args.segments, args.neighbors))
            # reference-formulation flops over the measured duration against the 157.3 TFLOP/s vector peak: an EQUIVALENT rate (the kernels run the
            # reference's arithmetic only on the pairs their conservative filters / depth windows cannot exclude)
            ref_flops = {"pair_mask": 370.0 * pairs_per_launch, "pair_fill": 360.0 * R_per_launch}.get(name)
            if ref_flops and avg_ms > 0:
                tf = ref_flops / (avg_ms * 1e-3) / 1e12
                out["fp32_equivalent"] = dict(reference_formulation_flops_per_launch=ref_flops, equivalent_tflops=tf, peak_tflops=157.3, frac=tf / 157.3,
                                              note="reference-formulation flops / measured launch duration; not a count of executed instructions")
            return out

        if prof:
            name, (launches, ms) = max(prof.items(), key=lambda kv: kv[1][1])
            roof = kernel_roof(name, launches, ms, args.steps)
            roof["kernels_ms"] = {k: round(v[1], 3) for k, v in prof_all.items()}
            roof["kernels_note"] = ("kernels_ms = per-kernel time of one untimed pass with every kernel bracketed (two streams: the times overlap); in the timed "
                                    "region only the dominant kernel carries HIP events")
            # the whole pass against the issue peak: summed VALU instructions of all kernels of one pass (committed PMC profile) over the wall time of a
            # pass (this run) and over the time the GPU is busy (the longer of the two streams' kernel sums of the bracketed pass: an upper bound of busy)
            tot_vi = 0.0
            # (the products' two brackets cover several kernels each: their instructions come from the profile's own launch counts -- its PMC pass is ONE matchViews pass)
            groups = {"prod_keys": ("k_prodv_pair_counts", "k_prodv_pair_transpose", "k_prode_counts", "k_prode_transpose", "k_prod_early_rt", "k_prod_keys_early", "k_prod_best", "k_prod_median", "prod_keys"),
                      "prod_rows": ("k_prodv_rows", "k_prodt_row_starts")}
            for kname, (kl, _kms) in prof_all.items():
                if kname in groups and kname not in valu_json:
                    for g in groups[kname]:
                        e = valu_json.get(g, {})
                        tot_vi += e.get("valu_wave_insts_per_launch", 0.0) * e.get("launches", 0)
                    continue
                tot_vi += valu_json.get(kname, {}).get("valu_wave_insts_per_launch", 0.0) * kl
            if tot_vi > 0:
                km = roof["kernels_ms"]
                s1 = sum(km.get(k, 0.0) for k in ("pair_mask", "row_count", "pair_fill", "tgt_rays"))
                s2 = sum(km.get(k, 0.0) for k in ("scan", "cand_move", "exist", "verify_window", "verify", "seg_post", "kept_write", "prod_keys", "prod_sort", "prod_rows"))
                busy_ms = max(s1, s2)
                roof["pass"] = dict(valu_wave_insts=tot_vi, issue_frac_of_wall=tot_vi / (ms_per_step * 1e-3) / PEAK_ISSUE,
                                    issue_frac_of_gpu_busy=(tot_vi / (busy_ms * 1e-3) / PEAK_ISSUE) if busy_ms > 0 else None, gpu_busy_ms=busy_ms,
                                    source=valu_src, note="sum over kernels of (VALU wave instructions per launch x launches of one pass) / (wall time of a pass | the longer "
                                                          "stream's kernel time of the bracketed pass) / issue peak")
            # k_pair_mask and k_verify_window take about the same time per pass: which one is named dominant changes from run to run.
            # The other one, from the untimed bracketed pass (one launch per view), so that both are always in the line
            others = sorted(((k, v) for k, v in prof_all.items() if k != name), key=lambda kv: -kv[1][1])
            if others:
                k2, (l2, m2) = others[0]
                roof["runner_up"] = kernel_roof(k2, l2, m2, 1)
                roof["runner_up"]["note"] = "second kernel by time, from the untimed pass with every kernel bracketed"
        out = dict(metric="segment-pair affinities/s", value=value, unit="segment-pair affinities/s", n_gpus=n_gpus,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="%s: %d views x %d segments, N=%d neighbours, matchViews (stage 1+2+filter+products of performMatching)"
                                        % (workload_name(n_gpus, V, args.segments, args.neighbors), V, args.segments, args.neighbors),
                               views=V, segments=args.segments, neighbors=args.neighbors, seed=args.seed,
                               parallelism=(("x%d: partitioned job, %s sharded, nothing replicated (DESIGN.md section 6 %s)" % (n_gpus, "source segments" if args.partition == "segments" else "blocks of views",
                                                                                                                               "iv" if args.partition == "segments" else "iii")) if args.partition
                                            else ("x%d: " % n_gpus + MODES[sharded_mode["i"]]) if dist is not None else "single GPU")),
                   views_per_s=V * args.steps / dt, pairs_per_step=pairs_total, raw_candidates_per_step=raw_total,
                   kept_per_step=st["kept"], setup_s=t_setup, cold=cold,
                   host_split_s=dict(gpu_call=st["t_gpu_call"], commit=st["t_commit"], finalize=st["t_finalize"], match=st["t_match"]))
        if roof:
            out["roofline"] = roof
            km = roof["kernels_ms"]
            s1 = sum(km.get(k, 0.0) for k in ("pair_mask", "row_count", "pair_fill"))
            s2 = sum(km.get(k, 0.0) for k in ("cand_move", "exist", "verify_window", "verify", "seg_post", "kept_write"))
            if s1 > 0 and s2 > 0:      # stage-1-only and stage-2-only rates (kernel time of one pass; the stages overlap on two streams)
                out["stage_rates"] = dict(stage1_pairs_per_s=pairs_local / (s1 * 1e-3), stage1_kernel_ms=round(s1, 3),
                                          stage2_candidates_per_s=raw_local / (s2 * 1e-3), stage2_kernel_ms=round(s2, 3),
                                          note="per-pass kernel time of each stage measured with every kernel bracketed (untimed pass)")
        if args.partition:
            out["rest_of_compute3Dmodel"] = dict(skipped="partitioned job: the finish is the collective one (l3d_line3d_finish_sharded), not part of this line")
        if dist is None and not args.no_extras:
            # the rest of compute3Dmodel on the same scene, untimed w.r.t. `value` (affinity edges/s, diffusion):
            # greedy selection + affinity fill (device) + [diffusion +] edge order (device) + union-find + line fit (host).
            # Like the timed passes it is reported warm (second call); the first call, which also grows the device arenas, beside it.
            ex = {}
            for diff in (False, True):
                t1 = time.perf_counter()
                l3d.finish(diff)
                t_first = time.perf_counter() - t1
                if not diff and cold:
                    t_first = cold["first_finish_s"]       # (the real first call was the cold one)
                t1 = time.perf_counter()
                l3d.finish(diff)
                st2 = l3d.stats()
                ex["diffusion" if diff else "no_diffusion"] = dict(finish_s=time.perf_counter() - t1, first_call_s=t_first, affinity_s=st2["t_affinity"], cluster_s=st2["t_cluster"],
                                                                   affinity_edges=st2["edges"], lines=st2["lines"],
                                                                   affinity_edges_per_s=st2["edges"] / st2["t_affinity"] if st2["t_affinity"] > 0 else None)
            try:                       # a11 alone: 10 iterations of row-normalise + positional product on the affinity list
                l3d.finish(False)
                A, n_nodes = l3d.affinity()[:2]
                ctx.replicator_dynamics_diffusion(A, n_nodes)
                t1 = time.perf_counter()
                ctx.replicator_dynamics_diffusion(A, n_nodes)
                t_rdd = time.perf_counter() - t1
                ex["rdd"] = dict(seconds=t_rdd, entries=int(len(A)), iterations=10, iterations_per_s=10.0 / t_rdd,
                                 entry_updates_per_s=10.0 * len(A) / t_rdd, note="upload + device radix sorts + sparse build + 21 launches + download")
            except Exception as e:     # noqa: BLE001
                ex["rdd"] = dict(error=str(e))
            out["rest_of_compute3Dmodel"] = ex
        if not args.no_cpu_baseline:
            lists = None
            if dist is None:                       # the kept lists of the earlier views feed the oracle's mid-chain sample (checker input, untimed)
                l3d.keep_view_matches(True)
                l3d.match_views()
                lists = {v["id"]: l3d.view_matches(v["id"])[0] for v in scene.views[: len(scene.views) // 2]}
            out["cpu_baseline"] = cpu_baseline(scene, args.neighbors, args.cpu_sample_segments, lists)
    l3d.close()
    if part_link is not None and part_link is not sharded_mode["link"]:
        part_link.close()                         # (--partition at --gpus 1: a communicator of one rank of its own)
    if sharded_mode["link"] is not None:
        sharded_mode["link"].close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)        # the ONE line
    if dist is not None:
        # RCCL prints a version banner to stdout from an exit handler: leave without running those (everything of ours is closed and flushed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
