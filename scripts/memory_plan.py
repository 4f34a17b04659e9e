#!/usr/bin/env python3
"""Per-rank HBM plan of the sharded resident chain, from the allocation formulas of the code (file:function cited per row):

    python scripts/memory_plan.py [--views 2048 --segments 4000 --neighbors 24 --world 8 --kept 0.025 0.25]

Prints a markdown table.  `kept` = kept matches / raw candidates (config 2 measures 2.5 %; the dense synthetic
box scene at 4000 x 24 keeps 25-40 %).  Raw candidates per view: rho * S^2 * n_tbm with rho = 0.065 (measured on config 2)."""
import argparse

GB = 1 << 30


def plan(V, S, N, W, kept_ratio, rho=0.065):
    n_tbm = N // 2                                     # interior views of a +-N/2 neighbourhood still match half of their neighbours
    window = N // 2                                    # the schedule's reach: a view reads the kept lists of its N/2 predecessors
    n_tgt = N * S
    raw_view = rho * S * S * n_tbm                     # stage-1 candidates of one view (all ranks)
    kept_view = kept_ratio * raw_view
    kept_total = kept_view * V
    nd = V * S
    W64 = 4 * ((S + 255) // 256)
    rows = []
    add = lambda name, b, where: rows.append((name, b, where))
    add("segments + neighbour tiles, resident (every rank holds the scene)", V * (S + n_tgt) * 16, "line3d_host_views.cpp:prepare -> l3d_register_segments_batch")
    add("camera tables of all views", V * (N * 144 + 48 + n_tbm * 4 + window * 8), "l3d_chain_common.hip:chain_plan_views")
    add("viewing rays of every target / own end point", V * (n_tgt + S) * 32, "l3d_chain_common.hip:chain_upload_tables (k_tgt_rays)")
    add("bit rows: ring of 10 views (was one slice per view: %.1f GB)" % (V * n_tbm * S * W64 * 8 / GB), 10 * n_tbm * S * W64 * 8, "l3d_chain_common.hip:chain_assign_arenas (mask_ring)")
    add("row counters + stage-1 row starts", 2 * V * S * N * 4, "chain_assign_arenas (ch_rowcnt, ch_rowA)")
    add("best depth pairs + best positions per segment", V * S * 12, "chain_assign_arenas (ch_best, ch_bestpos)")
    cand_cap = max(S / W * n_tbm * S * 0.12 + 65536, 1.25 * (raw_view / W) * (1 + kept_ratio * 2))        # first guess / what a dense scene grows it to
    add("candidate store + window scratch (per launch)", cand_cap * 44, "chain_reserve_candidates")
    add("stage-1 candidate ring (10 views ahead)", 10 * cand_cap * 24, "chain_reserve_candidates (ring)")
    slot_records = max(10 * S * N // W, int(1.25 * kept_view / W) + 1024)
    slot_bytes = 32 + (S // W + 1) * 12 + slot_records * 32
    ring = window + 18
    add("send slots: ring of %d views" % ring, ring * slot_bytes, "l3d_chain_sharded.hip:l3d_shard_chain_run")
    add("gathered slots: ring of %d views x %d ranks (all views: %.1f GB)" % (ring, W, V * W * slot_bytes / GB), ring * W * slot_bytes, "l3d_shard_chain_run (ring mode)")
    add("slot headers + arena offsets of all views", V * W * 32 + V * 24, "l3d_shard_chain_run (ch_hdr)")
    add("compact kept arena: 32 B x kept matches of the run (%.2f G records)" % (kept_total / 1e9), kept_total * 32, "k_shard_retire -> ch_kept")
    block_keys = min(1 << 28, 2 * kept_total + 1)
    add("products: key blocks (2 x 8 B keys + flag + position per slot, bounded)", block_keys * 24, "l3d_products.hip:build_products (ProdBlock)")
    add("products: potential correspondences, 4 B x 2 per kept match (bound) + row starts", 2 * kept_total * 4 + nd * 8, "build_products (pot_tgt, pot_start)")
    add("products: best references, medians, hypothesis table (96 B per segment)", nd * (8 + 8 + 4 + 4 + 96), "l3d_products.hip:l3d_products_hypotheses")
    return rows, dict(kept_total=kept_total, slot_bytes=slot_bytes, cand_cap=cand_cap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2048)
    ap.add_argument("--segments", type=int, default=4000)
    ap.add_argument("--neighbors", type=int, default=24)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--kept", type=float, nargs="+", default=[0.025, 0.25])
    a = ap.parse_args()
    plans = [plan(a.views, a.segments, a.neighbors, a.world, k) for k in a.kept]
    print("| per rank, %d views x %d segments x %d neighbours, %d ranks | " % (a.views, a.segments, a.neighbors, a.world) + " | ".join("kept %.1f %%" % (100 * k) for k in a.kept) + " | where |")
    print("|---|" + "---|" * (len(a.kept) + 1))
    for i, (name, _b, where) in enumerate(plans[0][0]):
        names = [p[0][i][0] for p in plans]
        label = name if len(set(names)) == 1 else " / ".join(names)
        print("| %s | " % label + " | ".join("%.2f GB" % (p[0][i][1] / GB) for p in plans) + " | `%s` |" % where)
    print("| **sum** | " + " | ".join("**%.1f GB**" % (sum(r[1] for r in p[0]) / GB) for p in plans) + " | of 288 GB |")
    for k, p in zip(a.kept, plans):
        print("kept %.1f %%: %.2f G kept matches in the run, slots of %.2f MB, candidate capacity %.2f M records" % (100 * k, p[1]["kept_total"] / 1e9, p[1]["slot_bytes"] / 1e6, p[1]["cand_cap"] / 1e6))


if __name__ == "__main__":
    main()
