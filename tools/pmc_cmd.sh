#!/bin/bash
# Development PMC passes over an arbitrary python command (run on the GPU box):  tools/pmc_cmd.sh <tag> <script + args...>
# Each counter set is its own rocprofv3 run (--kernel-trace + --pmc only); per-kernel averages -> gpurun_out/<tag>/pmc.json
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32" \
           "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -- python "$@" > $OUT/pmc_$i.log 2>&1 )
done
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python "$@" > $OUT/stats.log 2>&1 )
python - <<PY
import collections, csv, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
s = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("k_")}
for f in glob.glob("$OUT/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k in s:
            s[k]["avg_us"] = float(r["AverageNs"]) / 1e3
            s[k]["calls"] = int(r["Calls"])
json.dump(s, open("$OUT/pmc.json", "w"), indent=1, sort_keys=True)
for k, x in sorted(s.items()):
    w = x.get("SQ_WAVES", 0) or 1
    life = x.get("SQ_WAVE_CYCLES", 0) * 4 / w
    print(k, "us", round(x.get("avg_us", 0), 1), "waves", int(w), "life", int(life), "VALU/w", int(x.get("SQ_INSTS_VALU", 0) / w), "SALU/w", int(x.get("SQ_INSTS_SALU", 0) / w),
          "LDS/w", int(x.get("SQ_INSTS_LDS", 0) / w), "VMEM/w", int(x.get("SQ_INSTS_VMEM", 0) / w),
          "valu_frac %.2f" % (x.get("SQ_ACTIVE_INST_VALU", 0) * 4 / w / max(life, 1)), "lds_frac %.2f" % (x.get("SQ_ACTIVE_INST_LDS", 0) * 4 / w / max(life, 1)),
          "any_frac %.2f" % (x.get("SQ_ACTIVE_INST_ANY", 0) * 4 / w / max(life, 1)), "wait_any %.2f" % (x.get("SQ_WAIT_ANY", 0) * 4 / w / max(life, 1)),
          "wait_inst %.2f" % (x.get("SQ_WAIT_INST_ANY", 0) * 4 / w / max(life, 1)), "wait_lds %.2f" % (x.get("SQ_WAIT_INST_LDS", 0) * 4 / w / max(life, 1)),
          "bankconf", int(x.get("SQ_LDS_BANK_CONFLICT", 0)), "lds_idx", int(x.get("SQ_LDS_IDX_ACTIVE", 0)),
          "fetchMB %.1f" % (x.get("FETCH_SIZE", 0) / 1024), "writeMB %.1f" % (x.get("WRITE_SIZE", 0) / 1024))
PY
