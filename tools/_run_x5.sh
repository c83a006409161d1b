#!/bin/bash
# A/B of blend development variants: prints blend us per pair for each
cd $GRAFT_REPO_ROOT
O=gpurun_out/x5; mkdir -p $O
for cfg in C2 C3; do for b in "$@"; do
  timeout 120 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows 2 --blend $b --pairs 8 --iters 2 > $O/q_${cfg}_b$b.log 2>&1
  echo "$cfg blend=$b $(grep -o "'blend': [0-9.]*" $O/q_${cfg}_b$b.log) $(grep -o '"ms_per_pair": [0-9.]*' $O/q_${cfg}_b$b.log | tail -1) $(grep -o 'mean [0-9.e-]*' $O/q_${cfg}_b$b.log)"
done; done
