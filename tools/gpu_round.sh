#!/bin/bash
# One gpurun call per round: the GPU test suite, then bench + rocprofv3 kernel stats + PMC passes (tools/profile_round.sh).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r2d'   ->  gpurun_out/<tag>/...;  then  python tools/pmc_summary.py <tag> C2
TAG=${1:-r2}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log | cut -c1-300
bash tools/profile_round.sh $TAG 2>&1 | tail -3
