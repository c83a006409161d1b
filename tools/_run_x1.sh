#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/x1; mkdir -p $O
for cfg in C2 C3; do for b in 4 7; do
  timeout 200 python tools/quick_raster_bench.py --config $cfg --cull 1 --rows 2 --blend $b --pairs 8 > $O/q_${cfg}_b$b.log 2>&1; tail -2 $O/q_${cfg}_b$b.log
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w -- python $GRAFT_REPO_ROOT/tools/quick_raster_bench.py --config C3 --cull 1 --rows 2 --pairs 4 --iters 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/$O/pmc_w/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, round(sum(v)/len(v)), "KiB WRITE_SIZE avg")
PY
