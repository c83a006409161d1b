"""Rows of BASELINE.md section 4 from the committed bench lines of a round:  python tools/baseline_table.py r6f"""
import json, os, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for c in ("C2", "C3", "C4", "C5"):
    f = os.path.join(R, "profiles", f"{tag}_{c}_bench.json")
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    st = d["stages"]
    ras = sum(st[k]["avg_us"] for k in st if not k.startswith("tsdf"))
    ts = sum(st[k]["avg_us"] for k in st if k.startswith("tsdf"))
    ss = d.get("steady_state") or {}
    p = d.get("parity") or {}
    rr = d["raster_roofline"]
    print(f"| {c} (`{tag}_{c}_bench.json`) | 1 | **{d['value']:.0f}** | {d['ms_per_step']:.4f} | {ss.get('steady_state_ms_per_step')} / {ss.get('fill_drain_ms')} | "
          f"{ras:.1f} + {ts:.1f} | {rr['frac_of_8TBps']:.3f} ({rr['with_reference_instance_count']['frac_of_8TBps']:.3f}) | "
          f"{d['tsdf'].get('mvoxel_updates_per_s_job', 0):.0f} / {d['tsdf'].get('mvoxel_updates_per_s_kernels', 0):.0f} | "
          f"max {p.get('max_abs', 0):.1e}, clean max {p.get('max_abs_clean', 0):.1e}, {p.get('unexplained_pixels')} unexplained, PSNR {p.get('psnr_db', 0):.1f} dB, "
          f"{p.get('u8_flipped_pixels')} u8 pixels, {p.get('radii_mismatches')} radii |")
    print("   stages:", {k: v["avg_us"] for k, v in st.items()})
    for sub in ("c3", "trained_like"):
        if d.get(sub) and "stages" in d[sub]:
            print("  ", sub, d[sub]["raster_roofline"], {k: v["avg_us"] for k, v in d[sub]["stages"].items()}, d[sub].get("vs_synth_v1"))
    if "roofline" in d:
        r = d["roofline"]
        print("   roofline:", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_us")}, (r.get("valu") or {}).get("frac"), (r.get("valu") or {}).get("frac_of_issue_bound"))
    if "cpu_baseline" in d and d["cpu_baseline"]:
        cb = d["cpu_baseline"]
        print("   cpu:", {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")})
