#!/bin/bash
# Where the cycles of the three big kernels go: SQ counter passes over ONE pass of the bench scene (run through gpurun from the repo root):
#   gpurun -- 'L3D_COMMIT=<sha> scripts/measure_stalls.sh r4'
# rocprofv3 serialises dispatches while it collects counters, so every figure is the kernel ALONE on the chip (no second stream beside it).
# Counter names are checked against `rocprofv3 -L` first (a pass with one unknown name collects nothing); 8 SQ counters per pass.
# Writes gpurun_out/<tag>_stalls/stalls.json (scripts/make_stalls.py); copy it to profiles/<round>_stalls.json.
set -u
tag=${1:-r4}
out=gpurun_out/${tag}_stalls
mkdir -p $out
export TMPDIR=/tmp
xa=${L3D_BENCH_ARGS:-}      # (another shape: see scripts/measure_round.sh)
rocprofv3 -L > $out/avail.txt 2>&1
want=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT SQ_INSTS_VALU_F64"
 "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT64"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
p=0
files=""
for set in "${want[@]}"; do
  have=""
  for c in $set; do
    if grep -qw "$c" $out/avail.txt; then have="$have $c"; else echo "counter $c not available on this box" >> $out/missing.txt; fi
  done
  if [ -n "$have" ]; then
    timeout 420 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $out/pass$p -- python3 bench.py $xa --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extras --no-cold > /dev/null 2> $out/pass$p.err
    f=$(find $out/pass$p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && files="$files $f"
  fi
  p=$((p+1))
done
python3 scripts/make_stalls.py $files > $out/stalls.json
for d in $out/pass*; do [ -d "$d" ] && rm -rf $d; done
rm -f $out/avail.txt
ls -la $out
