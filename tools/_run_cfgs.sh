#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c_cfgs; mkdir -p $O
for cfg in C3 C4 C5; do
  timeout 400 python bench.py --config $cfg --steps 16 --no-cpu-baseline --no-c3 --no-parity --min-seconds 0.5 > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_${cfg}.json"))
    print("$cfg value", d["value"], "ms/step", d["ms_per_step"], "instr", d.get("instrumented_ms_per_step"), "raster", d["raster_roofline"]["frac_of_8TBps"], "tsdf", d["tsdf"]["mvoxel_updates_per_s_kernels"])
except Exception as e:
    print("$cfg failed", e); print(open("$O/bench_${cfg}.err").read()[-600:])
PY
done
