#!/usr/bin/env python3
"""Per-rank HBM plan of a multi-GPU matchViews + finish, from the allocation formulas of the code (file:function cited per row):

    python scripts/memory_plan.py [--views 2048 --segments 4000 --neighbors 24 --world 8 --kept 0.025 0.25 0.48] [--mode partition|segments]

--mode partition (default; round 5): l3d_match_chain_partition + l3d_affinity_fill_sharded -- views sharded in blocks, NOTHING replicated: a rank holds
  the kept records of its block, of the warm-up in front of it and of 2 x reach views either side, its rows of the products, its sources' candidates.
--mode segments: l3d_shard_chain_run (source segments of every view sharded; ring of gathered slots, compact arena REPLICATED on every rank).
--mode segpart: l3d_shard_chain_partition + l3d_shard_chain_run + l3d_affinity_fill_sharded -- the segments of every view sharded (exact without
  speculation), every rank RETIRES only its block of views + 2 x reach either side: the ring of gathered slots, then the partition's share of everything.

`kept` = kept matches / candidates verified (config 2 measures 2.5 %; the synthetic box scene at 4000 x 24 keeps 30 % at 40 views and 48 % at 256:
profiles/r5_256x4000x24_one_gpu_finish.json).  Stage-1 candidates per view: rho * S^2 * n_tbm, rho = 0.065 (measured on config 2);
candidates verified = those + the reverse matches (about half of the neighbours' kept lists): cand = raw / (1 - kept / 2).
Prints a markdown table and the largest index every 32-bit field has to hold."""
import argparse

GB = 1 << 30


RHO = [0.065]


def model(V, S, N, kept_ratio, rho=None):
    rho = RHO[0] if rho is None else rho
    n_tbm = N // 2                                     # interior views of a +-N/2 neighbourhood still match half of their neighbours
    raw_view = rho * S * S * n_tbm                     # stage-1 candidates of one view
    cand_view = raw_view / (1.0 - kept_ratio / 2.0)    # + reverse matches
    kept_view = kept_ratio * cand_view
    return n_tbm, raw_view, cand_view, kept_view


def plan_partition(V, S, N, W, kept_ratio, warmup_windows=4):
    n_tbm, raw_view, cand_view, kept_view = model(V, S, N, kept_ratio)
    reach = window = N // 2
    check = max(window, 2 * reach)
    block = -(-V // W)
    chain_views = block + max(warmup_windows * window, check) + 2 * reach     # what an interior rank's chain computes (or takes over when it is re-run warm)
    held_views = block + check + 2 * reach                                     # ... of which these are exact and kept
    row_views = block + 2 * reach
    n_tgt = N * S
    nd = V * S
    W64 = 4 * ((S + 255) // 256)
    local_kept = chain_views * kept_view
    cand_cap = 1.25 * cand_view
    # candidates of the affinity fill and what passes: measured on the box scene (256 x 4000 x 24: 1.50 G candidates, 10.5 M passed; config 2: 1.74 M, 0.49 M)
    fill_cand_view, passed_view = (5.9e6, 4.2e4) if S >= 3000 else (2.8e4, 7.7e3)
    passed_all = passed_view * V
    n_hyp = nd * 0.98
    rows = []
    add = lambda phase, name, b, where: rows.append((phase, name, b, where))
    add("all", "segments + neighbour tiles, resident (every rank holds the scene: it is small)", V * (S + n_tgt) * 16, "line3d_host_views.cpp:prepare -> l3d_register_segments_batch")
    add("all", "camera tables, best depth pairs + positions, result records of all views", V * (N * 144 + 48 + n_tbm * 4 + window * 8) + V * S * 12 + V * 24, "l3d_chain_common.hip:chain_plan_views, chain_assign_arenas")
    add("all", "kept arena: 32 B x the records of block + warm-up + 2 x reach either side (%d of %d views, x 1.3 growth slack)" % (chain_views, V), 1.3 * local_kept * 32, "l3d_chain.hip:run_chain (ch_kept)")
    add("chain", "viewing rays of every target / own end point (released after the chain)", V * (n_tgt + S) * 32, "chain_upload_tables (k_tgt_rays)")
    add("chain", "bit rows (ring of 15 views), row counters + row starts of all views (released)", 15 * n_tbm * S * W64 * 8 + 3 * V * S * N * 4, "chain_assign_arenas")
    add("chain", "candidate store + window scratch + stage-1 ring of 15 views (released)", cand_cap * 44 + 15 * cand_cap * 24, "chain_reserve_candidates")
    add("products", "transposed build (round 6): rebuilt side words + run tables + transposed entries of a block's lists (4 + ~0.1 + 4 B per record), column starts, bounded (released)", min(1 << 30, local_kept) * 8.2 + row_views * N * (S + 1) * 4, "l3d_products.hip:build_products (transposed)")
    add("after", "potential correspondences of the rows held (%d views): 4 B per entry, counted before they are written (~1.1 per record of those views) + row starts of all segments" % row_views, 1.1 * row_views * kept_view * 4 * 1.25 + nd * 8, "build_products (pot_tgt, pot_start)")
    add("after", "best references of all segments, hypotheses (96 B) + scores + indices of the views held (%d)" % held_views, nd * 12 + held_views * S * 112, "l3d_products_hypotheses")
    add("fill", "flags (1 B per local table entry), decision words (2^26 x 8 B), one block of candidates (2^27 x 20 B)", 2 * row_views * kept_view + (1 << 26) * 8 + (1 << 27) * 20, "l3d_affinity.hip:affinity_fill_core")
    add("fill", "first-touch minima of all hypotheses, own + gathered (8 B x (W + 2)), passed candidates own + gathered (12 B x 2)", n_hyp * 8 * (W + 2) + passed_all * 24, "l3d_affinity_fill_sharded")
    add("finish", "affinity list (24 B per passed candidate) + its clustering copies (x 4), hypothesis table of the scene (96 B)", passed_all * 24 * 5 + n_hyp * 100, "affinity_number_edges, l3d_rdd.hip, l3d_linefit.hip")
    idx = dict(arena_records_per_rank=1.3 * local_kept, table_entries_per_rank=2 * local_kept, fill_candidates_per_rank=fill_cand_view * block, fill_block_candidates=float(1 << 27),
               affinity_entries_job=2 * passed_all, hypotheses_job=n_hyp, kept_records_job=kept_view * V)
    return rows, idx, dict(chain_views=chain_views, held_views=held_views, kept_view=kept_view, cand_view=cand_view)


def plan_segments(V, S, N, W, kept_ratio):
    n_tbm, raw_view, cand_view, kept_view = model(V, S, N, kept_ratio)
    window = N // 2
    n_tgt = N * S
    kept_total = kept_view * V
    nd = V * S
    W64 = 4 * ((S + 255) // 256)
    rows = []
    add = lambda name, b, where: rows.append(("all", name, b, where))
    add("segments + neighbour tiles, resident (every rank holds the scene)", V * (S + n_tgt) * 16, "line3d_host_views.cpp:prepare -> l3d_register_segments_batch")
    add("camera tables of all views", V * (N * 144 + 48 + n_tbm * 4 + window * 8), "l3d_chain_common.hip:chain_plan_views")
    add("viewing rays of every target / own end point", V * (n_tgt + S) * 32, "l3d_chain_common.hip:chain_upload_tables (k_tgt_rays)")
    add("bit rows: ring of 10 views", 10 * n_tbm * S * W64 * 8, "l3d_chain_common.hip:chain_assign_arenas (mask_ring)")
    add("row counters + stage-1 row starts", 2 * V * S * N * 4, "chain_assign_arenas (ch_rowcnt, ch_rowA)")
    add("best depth pairs + best positions per segment", V * S * 12, "chain_assign_arenas (ch_best, ch_bestpos)")
    cand_cap = 1.25 * cand_view / W
    add("candidate store + window scratch (per launch)", cand_cap * 44, "chain_reserve_candidates")
    add("stage-1 candidate ring (10 views ahead)", 10 * cand_cap * 24, "chain_reserve_candidates (ring)")
    slot_records = max(10 * S * N // W, int(1.25 * kept_view / W) + 1024)
    slot_bytes = 32 + (S // W + 1) * 12 + slot_records * 32
    ring = window + 18
    add("send + gathered slots: ring of %d views x %d ranks" % (ring, W), ring * (W + 1) * slot_bytes, "l3d_chain_sharded.hip:l3d_shard_chain_run (ring mode)")
    add("compact kept arena, REPLICATED: 32 B x kept matches of the run (%.2f G records)" % (kept_total / 1e9), kept_total * 32, "k_shard_retire -> ch_kept")
    add("products: key blocks (bounded)", min(1 << 28, 2 * kept_total + 1) * 24, "l3d_products.hip:build_products (ProdBlock)")
    add("products, REPLICATED: potential correspondences, 4 B x 2 per kept match (bound) + row starts", 2 * kept_total * 4 + nd * 8, "build_products (pot_tgt, pot_start)")
    add("products: best references, medians, hypothesis table (96 B per segment)", nd * (8 + 8 + 4 + 4 + 96), "l3d_products.hip:l3d_products_hypotheses")
    return rows, dict(arena_records_per_rank=kept_total, kept_records_job=kept_total), dict(kept_view=kept_view, cand_view=cand_view)


def plan_segpart(V, S, N, W, kept_ratio, chain_world=0):
    CW = chain_world or W          # (scripts/run_rank_share.py runs ONE rank's share at world 1: its chain holds whole views, not 1/W of each)
    n_tbm, raw_view, cand_view, kept_view = model(V, S, N, kept_ratio)
    reach = window = N // 2
    block = -(-V // W)
    held_views = min(V, block + 4 * reach)
    row_views = min(V, block + 2 * reach)
    n_tgt = N * S
    nd = V * S
    W64 = 4 * ((S + 255) // 256)
    local_kept = held_views * kept_view
    fill_cand_view, passed_view = (5.9e6, 4.2e4) if S >= 3000 else (2.8e4, 7.7e3)
    passed_all = passed_view * V
    n_hyp = nd * 0.98
    rows = []
    add = lambda phase, name, b, where: rows.append((phase, name, b, where))
    add("all", "segments + neighbour tiles, resident (every rank holds the scene: it is small)", V * (S + n_tgt) * 16, "line3d_host_views.cpp:prepare -> l3d_register_segments_batch")
    add("all", "camera tables, best depth pairs + positions of all views", V * (N * 144 + 48 + n_tbm * 4 + window * 8) + V * S * 12, "l3d_chain_common.hip:chain_plan_views, chain_assign_arenas")
    add("all", "kept arena: 32 B x the records of block + 2 x reach either side (%d of %d views, + 12 %%)" % (held_views, V), 1.12 * local_kept * 32, "k_shard_retire (keep flags) -> ch_kept")
    slot_records = int((2.2 if CW > 1 else 1.3) * kept_view / CW) + 1024           # (a rank's range of segments holds up to twice its share)
    if slot_records >= 65536:      # (slots of a dense scene carry side words and run tables; the retire kernel files both with the records: no rebuild by the products)
        add("all", "side words of the kept arena (4 B per record: local camera << 16 | target) + run tables of the views held ((N + 1) x S ints each)", 1.12 * local_kept * 4 + held_views * (N + 1) * S * 4,
            "k_shard_retire -> ch_keptcam, ch_rt (round 6)")
    add("chain", "viewing rays of every target / own end point (released after the chain)", V * (n_tgt + S) * 32, "chain_upload_tables (k_tgt_rays)")
    add("chain", "bit rows (ring of 10 views), row counters + row starts of all views (released)", 10 * n_tbm * S * W64 * 8 / CW + 2 * V * S * N * 4, "chain_assign_arenas")
    cand_cap = 1.25 * cand_view / CW
    add("chain", "candidate store + window scratch + stage-1 ring of 10 views, 1/W of every view (released)", cand_cap * 44 + 10 * cand_cap * 24, "chain_reserve_candidates")
    slot_bytes = 32 + (S // CW + 1) * 12 + slot_records * (36 if slot_records >= 65536 else 32)      # (+ the side array of target cameras on dense scenes)
    ring = window + 18
    add("chain", "send + gathered slots: ring of %d views x %d ranks, %.0f MB slots (released)" % (ring, CW, slot_bytes / 2**20), ring * (CW + 1) * slot_bytes, "l3d_chain_sharded.hip:l3d_shard_chain_run (ring mode)")
    add("products", "transposed build (round 6): transposed entries + their staging copy of a block's lists (4 + 4 B per record; slots without side words: + 4.1 B of rebuilt words and run tables), column starts, bounded (released)",
        min(1 << 30, local_kept) * (8.0 if slot_records >= 65536 else 12.2) + row_views * N * (S + 1) * 4, "l3d_products.hip:build_products (transposed)")
    add("after", "potential correspondences of the rows held (%d views): 4 B per entry, counted before they are written (~1.1 per record of those views) + row starts of all segments" % row_views, 1.1 * row_views * kept_view * 4 * 1.25 + nd * 8, "build_products (pot_tgt, pot_start)")
    add("after", "best references of all segments, hypotheses (96 B) + scores + indices of the views held (%d)" % held_views, nd * 12 + held_views * S * 112, "l3d_products_hypotheses")
    add("fill", "flags (1 B per local table entry), decision words (2^26 x 8 B), one block of candidates (2^27 x 20 B)", 2 * row_views * kept_view + (1 << 26) * 8 + (1 << 27) * 20, "l3d_affinity.hip:affinity_fill_core")
    add("fill", "first-touch minima of all hypotheses, own + gathered (8 B x (W + 2)), passed candidates own + gathered (12 B x 2)", n_hyp * 8 * (W + 2) + passed_all * 24, "l3d_affinity_fill_sharded")
    add("finish", "affinity list (24 B per passed candidate) + its clustering copies (x 4), hypothesis table of the scene (96 B)", passed_all * 24 * 5 + n_hyp * 100, "affinity_number_edges, l3d_rdd.hip, l3d_linefit.hip")
    idx = dict(arena_records_per_rank=1.12 * local_kept, table_entries_per_rank=2 * local_kept, fill_candidates_per_rank=fill_cand_view * block, fill_block_candidates=float(1 << 27),
               affinity_entries_job=2 * passed_all, hypotheses_job=n_hyp, kept_records_job=kept_view * V)
    return rows, idx, dict(chain_views=V, held_views=held_views, kept_view=kept_view, cand_view=cand_view)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2048)
    ap.add_argument("--segments", type=int, default=4000)
    ap.add_argument("--neighbors", type=int, default=24)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--kept", type=float, nargs="+", default=[0.025, 0.25, 0.48])
    ap.add_argument("--mode", default="partition", choices=["partition", "segments", "segpart"])
    ap.add_argument("--chain-world", type=int, default=0, help="segpart only: the world size the CHAIN runs at (0 = --world); 1 = one rank's share exercised on one GPU")
    ap.add_argument("--rho", type=float, default=0.065, help="stage-1 candidates per segment pair (0.065: config 2; the 2048-view scene measures 0.11)")
    ap.add_argument("--json", action="store_true", help="the plan of the FIRST --kept value as one JSON object (scripts/run_rank_share.py compares it with a measured peak)")
    a = ap.parse_args()
    RHO[0] = a.rho
    fn = dict(partition=plan_partition, segments=plan_segments, segpart=plan_segpart)[a.mode]
    plans = [fn(a.views, a.segments, a.neighbors, a.world, k, a.chain_world) if a.mode == "segpart" else fn(a.views, a.segments, a.neighbors, a.world, k) for k in a.kept]
    if a.json:
        import json
        p = plans[0]
        g = lambda *ph: sum(r[2] for r in p[0] if r[0] in ph)
        phases = dict(chain=g("all", "chain"), products=g("all", "products", "after"), fill=g("all", "after", "fill"), finish=g("all", "after", "finish"))
        print(json.dumps(dict(mode=a.mode, kept=a.kept[0], peak_gb=round(max(phases.values()) / GB, 2), phases_gb={k: round(v / GB, 2) for k, v in phases.items()},
                              rows=[dict(phase=r[0], what=r[1], gb=round(r[2] / GB, 3)) for r in p[0]], fields={k: float(v) for k, v in p[1].items()})))
        return
    print("| per rank (%s), %d views x %d segments x %d neighbours, %d ranks | " % (a.mode, a.views, a.segments, a.neighbors, a.world) + " | ".join("kept %.1f %%" % (100 * k) for k in a.kept) + " | where |")
    print("|---|" + "---|" * (len(a.kept) + 1))
    for i, (phase, name, _b, where) in enumerate(plans[0][0]):
        print("| [%s] %s | " % (phase, name) + " | ".join("%.2f GB" % (p[0][i][2] / GB) for p in plans) + " | `%s` |" % where)
    # phases: what is live together.  "all" always; "chain" during matchViews only; "products" while the table is built; "after" from then on; "fill", "finish" later
    def peak(p):
        g = lambda *ph: sum(r[2] for r in p[0] if r[0] in ph)
        return max(g("all", "chain"), g("all", "products", "after"), g("all", "after", "fill"), g("all", "after", "finish"))
    print("| **peak over the phases** (chain / products / fill / finish) | " + " | ".join("**%.1f GB**" % (peak(p) / GB) for p in plans) + " | of 288 GB per rank |")
    print()
    print("| largest value a 32-bit field has to hold | " + " | ".join("kept %.1f %%" % (100 * k) for k in a.kept) + " | limit |")
    print("|---|" + "---|" * (len(a.kept) + 1))
    limits = dict(arena_records_per_rank=(2**32, "ChainResult.kept_base, unsigned"), table_entries_per_rank=(2**63, "pot_start is 64-bit"), fill_candidates_per_rank=(2**63, "64-bit positions; blocks of sources"),
                  fill_block_candidates=(2**30, "int, per block of sources"), affinity_entries_job=(2**31, "l3d_edge lists: int nnz"), hypotheses_job=(2**31, "int"), kept_records_job=(2**63, "never indexed as a whole"))
    for key in plans[0][1]:
        lim, what = limits.get(key, (2**31, "int"))
        print("| %s | " % key + " | ".join("%.3g%s" % (p[1][key], " (!)" if p[1][key] >= lim else "") for p in plans) + " | %s: %s |" % ("2^%d" % (lim.bit_length() - 1), what))
    for k, p in zip(a.kept, plans):
        print("kept %.1f %%: %.2f M candidates and %.2f M kept matches per view" % (100 * k, p[2]["cand_view"] / 1e6, p[2]["kept_view"] / 1e6) + (", a rank's chain covers %d views, holds %d" % (p[2]["chain_views"], p[2]["held_views"]) if "chain_views" in p[2] else ""))


if __name__ == "__main__":
    main()
