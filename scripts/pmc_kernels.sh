#!/bin/bash
# scripts/pmc_kernels.sh <tag> "<kernel name regex>" V S N "<counters pass 1>" ["<counters pass 2>" ...] -- rocprofv3 --pmc passes over scripts/bench_shape.py V S N
# (one pass of the chain), per-kernel averages of the kernels that match, printed and kept as gpurun_out/<tag>.txt (round 6: where the products' kernels wait)
tag=$1; rx=$2; V=$3; S=$4; N=$5; shift 5
export TMPDIR=/tmp
: > gpurun_out/$tag.txt
p=0
for counters in "$@"; do
  rm -rf gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/bench_shape.py $V $S $N 0 > /dev/null 2> gpurun_out/pmc_$tag.err
  f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 scripts/pmc_summary.py $f | grep -E "$rx" >> gpurun_out/$tag.txt
  rm -rf gpurun_out/pmc_$tag
  p=$((p+1))
done
cat gpurun_out/$tag.txt
