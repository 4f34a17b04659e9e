#!/bin/bash
# scripts/pmc_shape.sh <tag> "<counters>" [env assignments...] -- one rocprofv3 --pmc pass over scripts/bench_shape.py 64 2000 12 (1 pass),
# per-kernel averages printed and kept as gpurun_out/<tag>.txt
tag=$1; counters=$2; shift 2
export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rm -rf gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/bench_shape.py 64 2000 12 0 > /dev/null 2> gpurun_out/pmc_$tag.err
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $f | grep -i "pair_mask\|pair_fill\|verify_window" > gpurun_out/$tag.txt
rm -rf gpurun_out/pmc_$tag
cat gpurun_out/$tag.txt
