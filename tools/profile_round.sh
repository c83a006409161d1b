#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + HBM-traffic PMC passes.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
# kernel stats of the serial order (--inflight 1: kernels in isolation = what bench.py's `stages` time with hipEvents)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline --inflight 1 > $OUT/stats_bench.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 16 > /dev/null 2> $OUT/pmc_$c.err
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_SQ -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 16 > /dev/null 2> $OUT/pmc_SQ.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_SQ2 -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 16 > /dev/null 2> $OUT/pmc_SQ2.err
find $OUT -name "*.csv" | head -20
tail -c 400 $OUT/bench.json
