#!/bin/bash
# one gpurun call: GPU test suite + the round profile (bench, rocprofv3 kernel stats, PMC passes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log | cut -c1-300
bash tools/profile_round.sh r2c 2>&1 | tail -3
