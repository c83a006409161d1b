#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x14; mkdir -p $O
for fb in 8 16 25 32 49; do
  timeout 200 python bench.py --no-cpu-baseline --no-c3 --no-parity --fuse-batch $fb --min-seconds 0.5 > $O/b_fb$fb.json 2> $O/b_fb$fb.err
  python - <<PY
import json
d=json.load(open("$O/b_fb$fb.json"))
print("fuse_batch=$fb value", d["value"], "ms/step", d["ms_per_step"], "tsdf kernels", d["tsdf"]["mvoxel_updates_per_s_kernels"], "integrate us", d["stages"]["tsdf_integrate"]["avg_us"], "touch", d["stages"]["tsdf_touch"]["avg_us"])
PY
done
