#!/bin/bash
# scripts/kernel_times.sh <tag> "<kernel regex>" V S N [env assignments...] -- rocprofv3 --kernel-trace --stats over scripts/bench_shape.py V S N 1: calls and average
# microseconds of the kernels that match (python csv: kernel names contain commas)
tag=$1; rx=$2; V=$3; S=$4; N=$5; shift 5
export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rm -rf gpurun_out/kt_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 scripts/bench_shape.py $V $S $N 1 > gpurun_out/kt_$tag.json 2> gpurun_out/kt_$tag.err
f=$(find gpurun_out/kt_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$rx" <<'PY'
import csv, re, sys
rx = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if rx.search(r["Name"]):
        print("%-60s calls %5s  avg %10.1f us  total %9.2f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf gpurun_out/kt_$tag
