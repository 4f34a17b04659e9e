#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the run-table kernels' variants on one shape: scripts/sweep_rt_variants.sh <out dir> V S N
out=$1; V=${2:-40}; S=${3:-4000}; N=${4:-24}
mkdir -p $out; export TMPDIR=/tmp
for var in "-1 1 -1" "4 1 4" "16 512 16" "64 1 64" "0 1 0" "1 1 1"; do
  set -- $var
  tag="pg$1_rg$2_cg$3"
  L3D_PROD_PAIR_G=$1 L3D_PROD_ROW_GROUP=$2 L3D_RT_G=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -- python3 scripts/bench_shape.py $V $S $N 1 > $out/$tag.json 2> $out/$tag.err
  f=$(find $out/trace_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag $(python3 -c "import json;d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_pass'])")"
  grep -E "k_prodv|k_prod_early|k_exist_count_rt|k_place|k_kept_write_chain" $f | awk -F'","' '{printf "   %-28s calls %s avg_us %.1f\n", substr($1,2,38), $2, $4/1000}'
  rm -rf $out/trace_$tag
done
