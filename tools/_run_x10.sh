#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/x10; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-c3 --no-parity --min-seconds 0.2 --min-repeats 3 > $O/bench.json 2> $O/err.log
python - <<PY
import csv, glob, collections
f = glob.glob("$O/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"]) for r in rows]
ev.sort()
# take a window in the middle of the timed region: the last 40 % of the blend launches
bl = [e for e in ev if e[2].startswith("k_blend")]
t0 = bl[len(bl) // 2][0]; t1 = bl[-20][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
# sweep: time with blend running, time with anything running, concurrency histogram
pts = []
for s, e, n, q in win:
    pts.append((s, 1, n.startswith("k_blend"))); pts.append((e, -1, n.startswith("k_blend")))
pts.sort()
nb = na = 0; last = t0; t_blend = t_any = t_blend_only = t_idle = t_other_only = 0
for t, d, isb in pts:
    dt = t - last
    if na > 0: t_any += dt
    else: t_idle += dt
    if nb > 0: t_blend += dt
    if nb > 0 and na == nb: t_blend_only += dt
    if nb == 0 and na > 0: t_other_only += dt
    last = t
    na += d
    if isb: nb += d
tot = t1 - t0
nblend = sum(1 for e in win if e[2].startswith("k_blend"))
print(f"window {tot/1e3:.0f} us, {nblend} blend launches -> {tot/1e3/nblend:.1f} us per pair")
print(f"blend running {100*t_blend/tot:.1f} %, blend alone {100*t_blend_only/tot:.1f} %, only other kernels {100*t_other_only/tot:.1f} %, idle {100*t_idle/tot:.1f} %")
dur = collections.defaultdict(list)
for s, e, n, q in win: dur[n].append((e - s) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:60]:60s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us  sum/pair {sum(v)/nblend:7.1f}")
PY
