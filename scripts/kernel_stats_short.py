#!/usr/bin/env python3
"""A rocprofv3 kernel_stats.csv condensed: short kernel name, calls, average and total time.  python scripts/kernel_stats_short.py file.csv [skip-regex]"""
import csv, re, sys
skip = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if skip and skip.search(n):
        continue
    m = re.search(r"(k_\w+|__amd_rocclr_\w+|radix_sort_\w+|merge_sort_\w+|scan\w*|wrapped_\w+)", n)
    print("%-44s calls %5s  avg %10.1f us  total %9.2f ms" % ((m.group(1) if m else n)[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
