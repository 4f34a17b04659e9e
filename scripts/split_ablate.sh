#!/bin/bash
# timing-only ablations of k_chain_verify (results are wrong in these modes): kernels one at a time on one stream
for d in 0 1 2 3 4; do
  L3D_SPLIT_DEBUG=$d L3D_CHAIN_SERIAL=1 python bench.py --steps 4 --warmup 2 --no-extras 2>/dev/null | python scripts/show_bench.py debug$d
done
