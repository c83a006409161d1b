#!/bin/bash
# End-of-round validation on the GPU box (via gpurun): full GPU test suite, smoke(), then the profile rounds of C2 and C3
# (tools/profile_round.sh; summarise with tools/pmc_summary.py r<N> C2 / C3).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r3 C2 2>&1 | tail -1 | cut -c1-200
bash tools/profile_round.sh r3 C3 2>&1 | tail -1 | cut -c1-200
