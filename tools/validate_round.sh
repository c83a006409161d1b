#!/bin/bash
# End-of-round validation on the GPU box (via gpurun): full GPU test suite, smoke(), then the profile rounds
# (tools/profile_round.sh <tag> <config>; summarise with tools/pmc_summary.py <tag> <config>).
#   tools/validate_round.sh <tag> [configs...]        default configs: C2 C3
cd $GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
CFGS=${@:-C2 C3}
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in $CFGS; do bash tools/profile_round.sh $TAG $c 2>&1 | tail -1 | cut -c1-200; done
