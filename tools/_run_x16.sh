#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x16; mkdir -p $O
timeout 300 python -m pytest tests/test_tsdf_parity.py tests/test_pipeline_overlap.py -m gpu -x -q > $O/t.log 2>&1; tail -2 $O/t.log
timeout 200 python bench.py --no-cpu-baseline --no-parity --no-c3 --min-seconds 0.7 > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.load(open("$O/b.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], d["timing"]["ms_per_step_min"], "tsdf", d["tsdf"]["mvoxel_updates_per_s_kernels"], {k: v["avg_us"] for k, v in d["stages"].items() if "tsdf" in k})
PY
