"""Development aid: what runs when in a rocprofv3 --kernel-trace of the pipelined bench.
   python tools/trace_summary.py <kernel_trace.csv> [window_ms]
Takes the last `window_ms` (default 40) of the trace that contain compositing kernels and reports, per kernel family, the
mean duration under the pipelined load, and for the window: time covered by >= 1 compositing kernel, by >= 2, by any kernel,
by none; also writes a condensed copy (queue, kernel, start_us, end_us) next to the input."""
import csv, sys, collections

path = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0")))
rows.sort()
blend = [r for r in rows if r[2].startswith("k_blend")]
t_end = blend[-1][1]
t_beg = t_end - int(win_ms * 1e6)
sel = [r for r in rows if r[1] > t_beg and r[0] < t_end]
fam = lambda n: n.split("<")[0]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        s, e = max(s, t_beg), min(e, t_end)
        if e <= s:
            continue
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def depth_time(iv, k):
    ev = []
    for s, e in iv:
        s, e = max(s, t_beg), min(e, t_end)
        if e > s:
            ev += [(s, 1), (e, -1)]
    ev.sort()
    d, last, tot = 0, t_beg, 0
    for t, x in ev:
        if d >= k:
            tot += t - last
        d += x
        last = t
    return tot


span = t_end - t_beg
bl = [(s, e) for s, e, n, q in sel if n.startswith("k_blend")]
allk = [(s, e) for s, e, n, q in sel]
nb = len([1 for s, e, n, q in sel if n.startswith("k_blend") and s >= t_beg])
print(f"window {span/1e6:.1f} ms, {nb} compositing launches -> {span/1e3/max(nb,1):.1f} us per pair")
print(f"  >=1 compositing kernel running: {100*union(bl)/span:.1f} %   >=2: {100*depth_time(bl,2)/span:.1f} %")
print(f"  any kernel running: {100*union(allk)/span:.1f} %   >=2 kernels: {100*depth_time(allk,2)/span:.1f} %   >=3: {100*depth_time(allk,3)/span:.1f} %")
other = [(s, e) for s, e, n, q in sel if not n.startswith("k_blend")]
print(f"  >=1 non-compositing kernel running: {100*union(other)/span:.1f} %   while NO compositing kernel runs: "
      f"{100*(union(allk)-union(bl))/span:.1f} %   idle: {100*(span-union(allk))/span:.1f} %")
acc = collections.defaultdict(list)
for s, e, n, q in sel:
    acc[fam(n)].append((e - s) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:28s} n={len(v):4d} mean {sum(v)/len(v):7.1f} us  sum/pair {sum(v)/max(nb,1):7.1f} us")
with open(path.replace(".csv", "_condensed.csv"), "w") as f:
    f.write("queue,kernel,start_us,end_us\n")
    for s, e, n, q in sel:
        f.write(f"{q},{n.split('(')[0][:40]},{(s-t_beg)/1e3:.1f},{(e-t_beg)/1e3:.1f}\n")
