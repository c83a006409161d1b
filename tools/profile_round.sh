#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes for one config.
#   tools/profile_round.sh <tag> [config]      -> gpurun_out/<tag>_<config>/...      then  python tools/pmc_summary.py <tag> <config>
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r1}
CFG=${2:-C2}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${TAG}_${CFG}
mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "C2" ]; then
  # the driver's exact command
  timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
else
  timeout 600 python $R/bench.py --config $CFG --steps 12 --warmup 3 --no-c3 --cpu-seconds 4 > $OUT/bench.json 2> $OUT/bench.err
fi
# serial on one stream (--inflight 1) with the DEFAULT launch shapes (2 stereo pairs per launch); warm-up = steps so that every
# TSDF sweep of the run covers the same 12 frames (per-launch averages are then per 12-frame sweep)
ARGS="--config $CFG --no-cpu-baseline --no-parity --no-c3 --no-trained-like --no-steady-state --inflight 1 --steps 12 --warmup 12 --min-repeats 2 --min-seconds 0.05"
# kernel stats of the serial order (--inflight 1: kernels in isolation = what bench.py's `stages` time with hipEvents)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_$c.err
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_SQ -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_SQ.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_SQ2 -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_SQ2.err
if [ "$CFG" = "C2" ]; then
  # kernel trace of the PIPELINED default (what runs when): tools/trace_summary.py -> trace_summary.txt + a condensed trace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/bench.py --no-cpu-baseline --no-parity --no-c3 --no-trained-like --no-steady-state --min-repeats 3 --min-seconds 0.02 > /dev/null 2> $OUT/trace.err
  T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python $R/tools/trace_summary.py $T 20 > $OUT/trace_summary.txt 2>&1
  find $OUT/trace -name "*kernel_trace.csv" -delete    # the raw trace is large; the condensed window stays
fi
find $OUT -name "*.csv" | wc -l
tail -c 300 $OUT/bench.json
