#!/usr/bin/env python3
"""ONE rank's share of a partitioned job that does not fit one GPU, exercised on one GPU at the job's real view count:

    python3 scripts/run_rank_share.py VIEWS SEGMENTS NEIGHBOURS out.json [--world 8] [--ranks 3 4] [--window 64]

BASELINE configs[4] is 2048 views x 4000 segments x 24 neighbours on 8 ranks.  For every rank r of --ranks: the segment-sharded chain at world 1 (every
segment of every view is local) over ALL the views, with the keep set of rank r of a --world job (l3d_shard_chain_partition through the options
part_vrank / part_vworld): the chain retires only what that rank would hold, builds the rows of its block of the products and runs its share of the
sharded affinity fill and the finish.  Reported: seconds per stage, kept ratio, peak HBM (sampled) against scripts/memory_plan.py --mode segpart at the
measured kept ratio, and -- with two ranks -- the sha256 of the kept lists of the views both ranks hold (two different keep sets, one chain: they must
agree).  --window N: the A/B exactness switches (exact pair test alone, all-pairs verification) on the sub-scene of N views in the middle."""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = C.CDLL("libamdhip64.so")


def hbm_used_gb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return (total.value - free.value) / 2**30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("views", type=int); ap.add_argument("segments", type=int); ap.add_argument("neighbors", type=int); ap.add_argument("out")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--slot-records", type=int, default=0)
    ap.add_argument("--turn-period", type=int, default=5, help="synth.make_scene(turn_period=...): the helix repeats every that many turns (0: the default generator, whose radius grows with "
                    "every turn -- at 2048 views the scene then keeps 10-17 M matches per view and a rank's share exceeds 2^32 records)")
    ap.add_argument("--cand-cap", type=int, default=0)
    ap.add_argument("--arena-records", type=int, default=0, help="records of the rank's compact arena (0: the library's first guess, grown by capacity verdicts -- each one re-runs the chain)")
    a = ap.parse_args()
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    from line3d_amd.distributed import default_slot_records
    V, S, N, W = a.views, a.segments, a.neighbors, a.world
    t0 = time.perf_counter()
    scene = make_scene(V, S, N, seed=20260, turn_period=a.turn_period)
    out = dict(shape=[V, S, N], world=W, turn_period=a.turn_period, scene_s=round(time.perf_counter() - t0, 2), ranks={})
    slot = a.slot_records or max(default_slot_records(S, N, 1), int(0.06 * S * S * N / 2 * 0.6) + 65536)   # world 1: a slot holds a whole view's kept list
    digests = {}
    for r in a.ranks:
        peak = [0.0]
        stop = threading.Event()

        def sampler():
            while not stop.is_set():
                peak[0] = max(peak[0], hbm_used_gb())
                time.sleep(0.05)
        th = threading.Thread(target=sampler, daemon=True)
        l = Line3D("", matchingNeighbors=N)
        load_scene(l, scene)
        ctx = l.context()
        ctx.set_option("L3D_RESERVE_HINT", 0)          # (a partitioned job: the finishing stages' arenas are a share's, not the scene's)
        t0 = time.perf_counter(); l.prepare(); t_prep = time.perf_counter() - t0
        ctx.set_option("L3D_PART_VRANK", r); ctx.set_option("L3D_PART_VWORLD", W)
        if a.cand_cap or a.arena_records:
            ctx.set_chain_capacities(a.cand_cap, a.arena_records)
        base = hbm_used_gb()
        th.start()
        t0 = time.perf_counter()
        try:
            l.shard_run(0, 1, slot, "local", None, commit="partition")
        except Exception:
            stop.set()
            raise
        t_run = time.perf_counter() - t0
        info = l.partition_info()
        st = l.stats()
        peak_chain = peak[0]
        # (kept lists are hashed on the host: only the views another rank of --ranks holds too -- a rank holds its block and 2 x reach = N views either side)
        def held_range(q):
            return range(max(0, (V * q) // W - N), min(V, (V * (q + 1)) // W + N))
        others = set()
        for q in a.ranks:
            if q != r:
                others |= set(held_range(q))
        held = [k for k in range(info["held"][0], info["held"][1]) if k in others]
        t0 = time.perf_counter()
        d = {}
        n_held = 0
        for k in held:
            m = ctx.chain_kept_list(k)
            d[k] = [int(len(m)), hashlib.sha256(m.tobytes()).hexdigest()]
            n_held += len(m)
        t_dig = time.perf_counter() - t0
        digests[r] = d
        t0 = time.perf_counter()
        try:
            l.finish_sharded(False)
        finally:
            stop.set()
        t_fin = time.perf_counter() - t0
        th.join()
        A, n_nodes = l.affinity()
        lines = l.getResult()
        n_held_views = max(1, info["held"][1] - info["held"][0])
        kept_ratio = (st["kept"] / n_held_views) / max(1.0, st["raw"] / V)       # (the records this rank retired are those of the views it holds; the candidates are all views')
        # stage-1 candidates per segment pair as measured: candidates verified = stage-1 + reverse matches (about half of the kept lists)
        rho = (st["raw"] / V) * (1.0 - kept_ratio / 2.0) / (float(S) * S * (N // 2))
        plan = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "memory_plan.py"), "--views", str(V), "--segments", str(S), "--neighbors", str(N),
                                                    "--world", str(W), "--mode", "segpart", "--chain-world", "1", "--kept", "%.4f" % kept_ratio, "--rho", "%.4f" % rho, "--json"]).decode())
        out["ranks"][r] = dict(prepare_s=round(t_prep, 2), chain_and_products_s=round(t_run, 2), finish_sharded_s=round(t_fin, 2), digests_s=round(t_dig, 2), partition=info,
                               pairs=st["pairs"], candidates=st["raw"], kept_records_retired=st["kept"], views_held=n_held_views, kept_ratio=round(kept_ratio, 4), rho_measured=round(rho, 4), kept_records_hashed=n_held,
                               g_pairs_per_s=round(st["pairs"] / t_run / 1e9, 2), hbm_after_prepare_gb=round(base, 2), hbm_peak_gb=round(peak[0], 2), hbm_peak_chain_gb=round(peak_chain, 2),
                               plan_peak_gb=plan["peak_gb"], plan=plan, peak_over_plan=round(peak[0] / plan["peak_gb"], 3),
                               affinity_entries_of_the_share=int(len(A)), lines_of_the_share=len(lines), match_path=l.match_path())
        print(json.dumps({r: {k: v for k, v in out["ranks"][r].items() if k != "plan"}}), flush=True)
        l.close()
    if len(a.ranks) >= 2:
        r0, r1 = a.ranks[0], a.ranks[1]
        shared = sorted(set(digests[r0]) & set(digests[r1]))
        out["shared_views"] = dict(ranks=[r0, r1], n=len(shared), first=shared[0] if shared else None, last=shared[-1] if shared else None,
                                   equal=all(digests[r0][k] == digests[r1][k] for k in shared), records=sum(digests[r0][k][0] for k in shared))
    if a.window > 0:
        # the exactness switches on the sub-scene of the middle views (ids renumbered from 0; neighbours outside the window drop out)
        import numpy as np  # noqa: F401
        mid = V // 2
        sub_views = []
        lo = mid - a.window // 2
        for j, v in enumerate(scene.views[lo:lo + a.window]):
            w = dict(v); w["id"] = j
            w["sims"] = {i - lo: s for i, s in v["sims"].items() if lo <= i < lo + a.window}
            sub_views.append(w)
        res = {}
        for name, kw in (("all shortcuts", {}), ("exact pair test alone", dict(pretest=0)), ("all-pairs verification", dict(verify_mode=1))):
            l = Line3D("", matchingNeighbors=N)
            l.keep_view_matches(True)
            if "pretest" in kw:
                l.context().set_pair_pretest(kw["pretest"])
            if "verify_mode" in kw:
                l.context().set_verify_mode(kw["verify_mode"])
            for v in sub_views:
                l.addImage_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
            l.prepare()
            t0 = time.perf_counter(); l.match_views(); dt = time.perf_counter() - t0
            h = hashlib.sha256()
            for v in sub_views:
                h.update(l.view_matches(v["id"])[0].tobytes())
            s2 = l.stats()
            res[name] = dict(kept_lists_sha256=h.hexdigest()[:16], candidates=int(s2["raw"]), kept=int(s2["kept"]), match_views_s=round(dt, 2))
            l.close()
        ref = res["all shortcuts"]
        out["window"] = dict(views=[lo, lo + a.window], identical=all((o["kept_lists_sha256"], o["candidates"], o["kept"]) == (ref["kept_lists_sha256"], ref["candidates"], ref["kept"]) for o in res.values()), variants=res)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "ranks"}))
    return 0


if __name__ == "__main__":
    try:
        rc = main()
    except BaseException:
        import traceback
        traceback.print_exc()
        rc = 1
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(rc)            # (no lingering helper thread keeps a failed run -- and its GPU box -- alive)
