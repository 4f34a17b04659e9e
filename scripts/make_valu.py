#!/usr/bin/env python3
"""VALU instructions per launch per kernel from a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES pass:
   make_valu.py <counter_collection.csv> > profiles/<round>_valu.json"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import collections
import csv

from make_traffic import short  # noqa: E402  (same kernel-name mapping)

tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
out = {k: dict(valu_wave_insts_per_launch=tot[k].get("SQ_INSTS_VALU", 0.0) / max(1, len(launches[k])),
               waves_per_launch=tot[k].get("SQ_WAVES", 0.0) / max(1, len(launches[k])), launches=len(launches[k]),
               note="SQ_INSTS_VALU / SQ_WAVES summed over all shader engines, per-launch average")
       for k in sorted(tot)}
out["_commit"] = __import__("os").environ.get("L3D_COMMIT", "unstamped")      # (the GPU box has no .git: the caller passes the commit it sent)
# the shape the counters were collected on, "views,segments,neighbours" (L3D_SHAPE; absent: bench.py's default 64,2000,12): per-launch instruction
# counts and bytes are properties of the per-view shape (S, N) -- bench.py only prices a run against a profile of ITS shape
out["_shape"] = [int(x) for x in __import__("os").environ.get("L3D_SHAPE", "64,2000,12").split(",")]
json.dump(out, sys.stdout, indent=1)
