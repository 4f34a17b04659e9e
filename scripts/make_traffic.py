#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass):
   make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/<round>_traffic.json
hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-B
requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section), hence the factor 2."""
import collections
import csv
import json
import sys

NAMES = {"k_pair_mask": "pair_mask", "k_row_count": "row_count", "k_scan": "scan", "k_scan_kept_chain": "scan", "k_scan_kept_slot": "scan",
         "k_pair_fill": "pair_fill", "k_exist_count": "exist", "k_exist_scatter": "exist_scatter", "k_exist_sort_runs": "exist_sort_runs",
         "k_verify_window": "verify_window", "k_verify_window_gb": "verify_window", "k_verify": "verify", "k_seg_post": "seg_post", "k_kept_write_chain": "kept_write",
         "k_kept_write": "kept_write", "k_raw_stats": "raw_stats", "k_cand_move": "cand_move", "k_place": "cand_move", "k_pack_view": "pack_view", "k_slot_write": "kept_write", "k_collinearity": "collinearity", "k_collinearity_fill": "collinearity_fill", "k_tgt_rays": "tgt_rays", "k_prod_keys": "prod_keys"}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("l3d::", "").split("<")[0]
    return NAMES.get(n, n)


def load(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(1, len(launches[k])), len(launches[k])) for k in tot}


if __name__ == "__main__":
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, n2 = w.get(k, (0.0, 0))
        out[k] = dict(FETCH_SIZE_KB_per_launch=fk, WRITE_SIZE_KB_per_launch=wk, hbm_bytes_per_launch=(2 * fk + wk) * 1024, launches=max(n, n2),
                      note="(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, per-launch average")
    out["_commit"] = __import__("os").environ.get("L3D_COMMIT", "unstamped")      # (the GPU box has no .git: the caller passes the commit it sent)
    # the shape the counters were collected on, "views,segments,neighbours" (L3D_SHAPE; absent: bench.py's default 64,2000,12): per-launch instruction
    # counts and bytes are properties of the per-view shape (S, N) -- bench.py only prices a run against a profile of ITS shape
    out["_shape"] = [int(x) for x in __import__("os").environ.get("L3D_SHAPE", "64,2000,12").split(",")]
    json.dump(out, sys.stdout, indent=1)
