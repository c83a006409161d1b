#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x15; mkdir -p $O
timeout 300 python -m pytest tests/test_raster_parity.py -m gpu -x -q > $O/t.log 2>&1; tail -2 $O/t.log
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-parity --min-seconds 0.7 > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
d=json.load(open("$O/b$i.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], d["timing"]["ms_per_step_min"], "spatial", d["config"]["spatial_order"], {k: v["avg_us"] for k, v in d["stages"].items()})
print("C3", d["c3"]["ms_per_pair_wall"], {k: v["avg_us"] for k, v in d["c3"]["stages"].items()})
PY
done
