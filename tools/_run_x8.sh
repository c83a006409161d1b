#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x8; mkdir -p $O
for cfg in C2 C3; do for rows in 1 2; do
  timeout 120 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows $rows --blend 4 --pairs 8 --iters 2 --pack 2 > $O/q_${cfg}_r$rows.log 2>&1
  echo "$cfg rows=$rows $(grep "^{'project'" $O/q_${cfg}_r$rows.log) $(grep -o '"ms_per_pair": [0-9.]*' $O/q_${cfg}_r$rows.log | tail -1)"
done; done
