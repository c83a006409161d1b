#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x2; mkdir -p $O
for cfg in C2 C3; do
  timeout 300 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows 2 --blend 4 --pairs 8 --morton 1 > $O/q_${cfg}_morton.log 2>&1; tail -2 $O/q_${cfg}_morton.log
done
