cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log | cut -c1-300
