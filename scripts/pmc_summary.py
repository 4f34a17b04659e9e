#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 --pmc counter_collection.csv (sums over dimensions)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = max(1, len(cnt[k]))
    print(k, "launches", n, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
