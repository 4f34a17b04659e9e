#!/usr/bin/env python3
"""Per-queue timeline statistics of a rocprofv3 kernel_trace.csv (last `n` dispatches): busy time, idle gaps between
consecutive kernels of the same queue, per-kernel averages -- to see how the chain stream and the stage-1 stream interleave."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
byq = collections.defaultdict(list)
for r in rows:
    byq[r[qkey] if qkey else "all"].append(r)
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
print("span %.2f ms" % ((t1 - t0) / 1e6))
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    print("queue %s: %d kernels, busy %.2f ms, gaps: total %.2f ms, median %.1f us" % (q, len(rs), busy / 1e6, sum(pos) / 1e6, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0))
    acc = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r["Kernel_Name"].split("(")[0][-28:]
        acc[k][0] += 1
        acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:9]:
        print("    %-28s %4d x %7.1f us" % (k, c, d / c / 1e3))
