#!/bin/bash
# One measurement set of the bench command on the GPU box (run through gpurun from the repo root):
#   L3D_COMMIT=$(git rev-parse --short HEAD) gpurun -- 'L3D_COMMIT=<sha> scripts/measure_round.sh r3_v1'
# Another shape than bench.py's default (64 x 2000 x 12): L3D_SHAPE="40,4000,24" L3D_BENCH_ARGS="--views-per-gpu 40 --segments 4000 --neighbors 24
# --cpu-sample-segments 60" -- the arguments go to every bench.py command below, the shape is stamped into the summaries (`_shape`), and
# bench.py prices a run only against the summaries of its own (segments, neighbours).
# writes gpurun_out/<tag>/: bench.json (plain run), bench_under_rocprof.json + kernel_stats.csv (rocprofv3 --kernel-trace --stats),
# fetch/ write/ valu/ (separate --pmc passes).  Copy the summaries into profiles/ afterwards (scripts/make_traffic.py, make_valu.py).
set -u
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
xa=${L3D_BENCH_ARGS:-}
python3 bench.py $xa > $out/bench.json 2> $out/bench.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $xa --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $xa --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extras --no-cold > /dev/null 2> $out/fetch.err
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $xa --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extras --no-cold > /dev/null 2> $out/write.err
timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/valu -- python3 bench.py $xa --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extras --no-cold > /dev/null 2> $out/valu.err
f=$(find $out/fetch -name "*counter_collection.csv" | head -1); w=$(find $out/write -name "*counter_collection.csv" | head -1); v=$(find $out/valu -name "*counter_collection.csv" | head -1)
python3 scripts/make_traffic.py $f $w > $out/traffic.json
python3 scripts/make_valu.py $v > $out/valu.json
# the raw counter dumps are large: keep the summaries only
rm -rf $out/fetch $out/write $out/valu $out/trace
ls -la $out
