#!/usr/bin/env python3
"""Per-kernel averages and idle gaps over the LAST `n` kernel dispatches of a rocprofv3 kernel_trace.csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    acc[k][0] += 1
    acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
# union of busy intervals (two streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms, GPU busy (union) %.2f ms, sum of kernel durations %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6, sum(v[1] for v in acc.values()) / 1e6))
for k, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s calls %4d  avg %7.2f us  total %7.2f ms" % (k, c, d / c / 1e3, d / 1e6))
