#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x11; mkdir -p $O
for inf in 4 6 8 12 16; do
  timeout 200 python bench.py --no-cpu-baseline --no-c3 --no-parity --inflight $inf --min-seconds 0.5 > $O/b_i$inf.json 2> $O/b_i$inf.err
  python - <<PY
import json
d=json.load(open("$O/b_i$inf.json"))
print("inflight=$inf value", d["value"], "ms/step", d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"])
PY
done
