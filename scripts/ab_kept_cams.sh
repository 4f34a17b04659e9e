#!/bin/bash
for t in 0 1 0 1; do
  echo "== cfg5 40x4000x24 kept_cams $t"; L3D_KEPT_CAMS=$t python3 scripts/bench_shape.py 40 4000 24 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_pass'], d['kept'], {k:d['kernels_ms'][k] for k in ('verify_window','cand_move','exist','kept_write','pair_mask','pair_fill')})"
done
for t in 0 1 0 1; do
  echo "== cfg2 64x2000x12 kept_cams $t"; L3D_KEPT_CAMS=$t python3 scripts/bench_shape.py 64 2000 12 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_pass'], d['median_ms'], d['kept'], {k:d['kernels_ms'][k] for k in ('verify_window','cand_move','exist','kept_write','pair_mask','pair_fill')})"
done
