#!/bin/bash
# Development PMC passes over tools/quick_raster_bench.py (run on the GPU box):  tools/pmc_dev.sh <tag> <bench args...>
# Each counter set is its own rocprofv3 run (--kernel-trace + --pmc only).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32" \
           "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -- python $R/tools/quick_raster_bench.py --pairs 4 --iters 1 "$@" > $OUT/pmc_$i.log 2>&1
done
python - <<PY
import collections, csv, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
s = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("k_")}
json.dump(s, open("$OUT/pmc.json", "w"), indent=1, sort_keys=True)
for k, cs in s.items():
    print(k, {c: round(v) for c, v in sorted(cs.items())})
PY
