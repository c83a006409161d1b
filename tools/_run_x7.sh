#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x7; mkdir -p $O
timeout 300 python -m pytest tests/test_raster_parity.py -m gpu -x -q -k "packed or golden or tile_rows" > $O/t.log 2>&1; tail -3 $O/t.log
for cfg in C2 C3; do for pk in 1 2; do
  timeout 120 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows 2 --blend 4 --pairs 8 --iters 2 --pack $pk > $O/q_${cfg}_p$pk.log 2>&1
  echo "$cfg pack=$pk $(grep "^{'project'" $O/q_${cfg}_p$pk.log) $(grep -o '"ms_per_pair": [0-9.]*' $O/q_${cfg}_p$pk.log | tail -1) $(grep -o 'mean [0-9.e-]*' $O/q_${cfg}_p$pk.log)"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w -- python $GRAFT_REPO_ROOT/tools/quick_raster_bench.py --config C3 --cull 1 --rows 2 --pairs 4 --iters 1 --pack 2 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/$O/pmc_w/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "scatter" in k or "count" in k or "project" in k: print(k, round(sum(v)/len(v)), "KiB WRITE_SIZE avg (C3, packed)")
PY
