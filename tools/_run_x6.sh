#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x6; mkdir -p $O
for cfg in C2 C3; do for m in 0 4 8 16 32 128; do
  timeout 120 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows 2 --blend 4 --pairs 8 --iters 2 --morton $m > $O/q_${cfg}_m$m.log 2>&1
  echo "$cfg morton=$m $(grep "^{'project'" $O/q_${cfg}_m$m.log) $(grep -o '"ms_per_pair": [0-9.]*' $O/q_${cfg}_m$m.log | tail -1)"
done; done
