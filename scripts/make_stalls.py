#!/usr/bin/env python3
"""Per-kernel stall attribution from rocprofv3 --pmc passes (scripts/measure_stalls.sh):
   make_stalls.py <counter_collection.csv>... > profiles/<round>_stalls.json

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
WAIT_ANY (wave parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES, disjoint.
A wave64 VALU instruction occupies its SIMD for 2 cycles (transcendentals 4): `valu_pipe_frac` = the SIMDs' VALU issue time over the
time the kernel keeps the chip (SQ_BUSY_CYCLES is per shader engine; GRBM_GUI_ACTIVE / 8 XCDs when collected).  rocprofv3 serialises
dispatches while collecting, so every row is the kernel alone on the chip."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic import short  # noqa: E402

tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        tot[k][c] += float(r["Counter_Value"])
        launches[k][c].add(r["Dispatch_Id"])

N_SIMD = 256 * 4
out = {}
for k in sorted(tot):
    per = {c: tot[k][c] / max(1, len(launches[k][c])) for c in tot[k]}
    row = dict(launches=max(len(s) for s in launches[k].values()), counters_per_launch={c: round(v, 1) for c, v in sorted(per.items())})
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        f = lambda c: round(per.get(c, 0.0) / wc, 4)        # noqa: E731
        split = dict(issuing=f("SQ_ACTIVE_INST_ANY"), parked_waitcnt_or_barrier=f("SQ_WAIT_ANY"), issue_stall=f("SQ_WAIT_INST_ANY"))
        issue = dict(valu=f("SQ_ACTIVE_INST_VALU"), lds=f("SQ_ACTIVE_INST_LDS"), vmem=f("SQ_ACTIVE_INST_VMEM"), scalar=f("SQ_ACTIVE_INST_SCA"),
                     flat=f("SQ_ACTIVE_INST_FLAT"), misc=f("SQ_ACTIVE_INST_MISC"))
        row["wave_cycle_split"] = split
        row["issuing_split_of_wave_cycles"] = issue
        row["lds_issue_stall_of_wave_cycles"] = f("SQ_WAIT_INST_LDS")
        row["top_stall"] = max((("parked at s_waitcnt / barrier (memory or LDS latency, barriers)", split["parked_waitcnt_or_barrier"]),
                                ("issue stall (pipe busy / dependency)", split["issue_stall"])), key=lambda kv: kv[1])[0]
    vi, tr = per.get("SQ_INSTS_VALU", 0.0), per.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    gui = per.get("GRBM_GUI_ACTIVE", 0.0)
    if vi > 0 and gui > 0:
        cyc = gui / 8.0                                          # summed over the 8 XCDs
        row["chip_cycles_per_launch"] = round(cyc, 1)
        row["valu_pipe_frac"] = round(vi * 2.0 / (cyc * N_SIMD), 4)
        row["valu_pipe_frac_trans_weighted"] = round((vi + tr) * 2.0 / (cyc * N_SIMD), 4)
    if per.get("SQ_INSTS_LDS", 0.0) > 0:
        row["lds_bank_conflict_cycles_per_lds_inst"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_INSTS_LDS"], 3)
        if per.get("SQ_INST_LEVEL_LDS", 0.0) > 0:
            row["avg_lds_latency_cycles"] = round(per["SQ_INST_LEVEL_LDS"] / per["SQ_INSTS_LDS"], 1)
    rd = per.get("SQ_INSTS_VMEM_RD", 0.0)
    if rd > 0 and per.get("SQ_INST_LEVEL_VMEM", 0.0) > 0:
        row["avg_vmem_latency_cycles"] = round(per["SQ_INST_LEVEL_VMEM"] / (rd + per.get("SQ_INSTS_VMEM_WR", 0.0)), 1)
    w = per.get("SQ_WAVES", 0.0)
    if w > 0 and vi > 0:
        row["valu_insts_per_wave"] = round(vi / w, 1)
    out[k] = row
out["_commit"] = os.environ.get("L3D_COMMIT", "unstamped")
out["_shape"] = [int(x) for x in os.environ.get("L3D_SHAPE", "64,2000,12").split(",")]      # views, segments, neighbours of the profiled run
out["_note"] = ("per-launch averages; *_CYCLES of the SQ in quad-cycles summed over waves; rocprofv3 serialises dispatches in counter mode: each kernel alone. "
                "valu_pipe_frac = 2 cycles x SQ_INSTS_VALU / (chip cycles x 1024 SIMDs)")
json.dump(out, sys.stdout, indent=1)
