#!/bin/bash
# scripts/kernel_times_cmd.sh <tag> "<kernel regex>" <python script + arguments...> -- rocprofv3 --kernel-trace --stats over any python3 command of this repo:
# calls, average microseconds and totals of the kernels that match (python csv: kernel names contain commas)
tag=$1; rx=$2; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/kt_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 "$@" > gpurun_out/kt_$tag.out 2> gpurun_out/kt_$tag.err
f=$(find gpurun_out/kt_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$rx" <<'PY'
import csv, re, sys
rx = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if rx.search(r["Name"]):
        print("%-60s calls %5s  avg %10.1f us  total %9.2f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf gpurun_out/kt_$tag
