"""Summarise a tools/profile_round.sh output directory into profiles/:
   <tag>_kernel_stats.csv (rocprofv3 --stats), <tag>_pmc.json (per-kernel averages of every counter) and
   profiles/pmc_traffic.json (HBM bytes per launch, read by bench.py for roofline.traffic).
HBM bytes follow the guide (MI355X_MICROARCH.md, HBM) and this repo's own calibration of the counter for gathers
(tools/ubench/gather_fetch.hip -> profiles/r5_gather_fetch_calibration.json, round 5): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE = read requests x 64 B.  A coalesced STREAM (4, 8 or 16 B per lane) travels as 128-B requests: the counter shows
exactly HALF the bytes -> x2.  A GATHER (4- to 64-byte pieces at scattered places, or a wave's 8x8-pixel patch of an image)
travels as 64-B requests, one per line touched: the counter IS the bytes moved (64 B per line, however little of it is used)
-> x1.  Per stage: READ_FACTOR below; a stage that mixes both is bracketed (x1 = lower bound, x2 = upper bound) and the
midpoint is quoted.  bytes = (factor * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"{tag}_{cfg}")
if not os.path.isdir(src):
    src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: a directory can hold the files of several runs; the last one counts."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_bench_{cfg}_kernel_stats.csv"))
for name in ("bench.json", "stats_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{cfg}_{name}"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "pmc_*")):
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("k_")}
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_{cfg}.json"), "w"), indent=1, sort_keys=True)
prefix_of = (("k_project", "project"), ("k_count_tiles", "count_tiles"), ("k_hist_colscan", "hist_colscan"), ("k_tile_scan", "tile_scan"),
             ("k_scatter", "scatter"), ("k_sort_tiles", "sort_tiles"), ("k_tsdf_touch", "tsdf_touch"),
             ("k_tsdf_integrate", "tsdf_integrate"), ("k_blend", "blend"))
# what the reads of a stage are (see the module docstring): streams are doubled, gathers are not
READ_FACTOR = {"project": 2.0, "count_tiles": 2.0, "hist_colscan": 2.0, "tile_scan": 2.0, "scatter": 2.0, "sort_tiles": 2.0,
               "blend": 1.0,            # record DMA = 16-byte pieces gathered by id (the 8-byte key stream is <= 1/6 of the reads)
               "tsdf_touch": 1.0,       # every 4th pixel of every 4th row
               "tsdf_integrate": 1.5}   # state planes stream (x2), depth / colour are patch gathers (x1): bracketed [1, 2]
stage_of = {}
for k in summary:
    for pre, st in prefix_of:
        if k.startswith(pre):
            stage_of[k] = st
            break
bench = json.load(open(os.path.join(src, "bench.json"))) if os.path.exists(os.path.join(src, "bench.json")) else {}
traffic = {}
for k, cs in sorted(summary.items()):
    stage = stage_of.get(k)
    if stage and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        if stage in traffic:      # a stage made of several kernels (the per-tile sort: one kernel per size class): sums
            t = traffic[stage]
            t["kernel"] += " + " + k
            t["hbm_bytes_per_launch"] += int((READ_FACTOR[stage] * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)
            t["fetch_kib"] += cs["FETCH_SIZE"]
            t["write_kib"] += cs["WRITE_SIZE"]
            for name, key in (("valu_insts_per_launch", "SQ_INSTS_VALU"), ("salu_insts_per_launch", "SQ_INSTS_SALU"),
                              ("branch_insts_per_launch", "SQ_INSTS_BRANCH"), ("lds_bank_conflict_cycles", "SQ_LDS_BANK_CONFLICT")):
                if key in cs and t.get(name) is not None:
                    t[name] += int(cs[key])
            continue
        traffic[stage] = dict(kernel=k, hbm_bytes_per_launch=int((READ_FACTOR[stage] * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024),
                              read_factor=READ_FACTOR[stage],
                              hbm_bytes_bracket=[int((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024), int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)],
                              fetch_kib=cs["FETCH_SIZE"], write_kib=cs["WRITE_SIZE"],
                              valu_insts_per_launch=int(cs["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in cs else None,
                              valu_trans_per_launch=int(cs["SQ_INSTS_VALU_TRANS_F32"]) if "SQ_INSTS_VALU_TRANS_F32" in cs else None,
                              date=f"round {tag[1]}" if tag[:1] == "r" and tag[1:2].isdigit() else tag,
                              salu_insts_per_launch=int(cs["SQ_INSTS_SALU"]) if "SQ_INSTS_SALU" in cs else None,
                              branch_insts_per_launch=int(cs["SQ_INSTS_BRANCH"]) if "SQ_INSTS_BRANCH" in cs else None,
                              lds_bank_conflict_cycles=int(cs["SQ_LDS_BANK_CONFLICT"]) if "SQ_LDS_BANK_CONFLICT" in cs else None,
                              cull=bench.get("config", {}).get("exact_tile_cull", 1), source=f"profiles/{tag}_pmc_{cfg}.json")
# the TSDF launches of the profiled command cover a whole sweep: record its frames so that bench.py can quote per-frame bytes
sb = os.path.join(src, "stats_bench.json")
try:
    sweep = int(json.load(open(sb)).get("config", {}).get("tsdf_fuse_batch", 1))
except Exception:
    sweep = 1
try:
    ppl = int(json.load(open(sb)).get("config", {}).get("pairs_per_launch", 1))
except Exception:
    ppl = 1
for st, t in traffic.items():
    if st in ("tsdf_touch", "tsdf_integrate"):
        t["frames_per_launch"] = sweep       # a launch = one sweep (profile_round.sh: warm-up = steps, so every sweep has this size)
    else:
        t["pairs_per_launch"] = ppl          # a raster launch covers this many stereo pairs (GS2M_OPT_PAIR_BATCH)
tp = os.path.join(dst, "pmc_traffic.json")
allt = json.load(open(tp)) if os.path.exists(tp) else {}
allt[cfg] = traffic
json.dump(allt, open(tp, "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
