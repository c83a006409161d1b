#!/bin/bash
# End-of-round evidence on the GPU box (via gpurun): the profile rounds (tools/profile_round.sh <tag> <config>: the driver's
# bench command, rocprofv3 kernel stats + PMC passes of the serial order; summarise with tools/pmc_summary.py <tag> <config>)
# and the bench lines of the other BASELINE configs with their parity objects.
#   tools/validate_round.sh <tag> [configs...]        default configs: C2 C3
cd $GRAFT_REPO_ROOT
TAG=${1:-r4}; shift
CFGS=${@:-C2 C3}
for c in $CFGS; do bash tools/profile_round.sh $TAG $c 2>&1 | tail -1 | cut -c1-200; done
for c in C4 C5; do
  O=gpurun_out/${TAG}_$c; mkdir -p $O
  timeout 600 python bench.py --config $c --steps 12 --warmup 4 --no-c3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json
done
