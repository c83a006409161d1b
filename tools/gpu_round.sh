#!/bin/bash
# One gpurun call: the GPU test suite, then the driver's bench command.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r4a'   ->  gpurun_out/<tag>/...
TAG=${1:-r4}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests/ -m gpu -q --durations=12 > $O/gpu_tests.log 2>&1; tail -25 $O/gpu_tests.log | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; head -c 700 $O/bench.json
