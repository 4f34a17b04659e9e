#!/usr/bin/env python3
"""HIP API calls longer than a threshold from a rocprofv3 --hip-trace csv: slow_hip_calls.py <dir> [ms]"""
import csv, glob, sys
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for path in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d >= thr:
            print("%10.2f ms  +%9.2f ms  tid %s  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Thread_Id", "?"), r["Function"]))
