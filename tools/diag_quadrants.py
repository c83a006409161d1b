"""Development aid (CPU): compositing-loop census for two pixel layouts of the one-wave-per-tile kernel.
For every (instance, 16x16 tile) pair whose alpha >= 1/255 box reaches the tile: regions evaluated with
 (a) 8x8 quadrants (lane = 1 pixel per quadrant), (b) 16x4 row strips (lane = 1 pixel per strip).
Saturation is ignored (it cuts both the same way)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from gs2mesh_amd import synthetic

cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
W, H = cfg.width, cfg.height
g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
pose = synthetic.ring_poses(1, cfg.ring_radius)[0]
cam = synthetic.stereo_cameras(pose, W, H, cfg.focal, cfg.focal, cfg.baseline)[0]
s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
geom = oracle.preprocess(g["xyz"], s, q, o, shs, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H, cam.tanfovx, cam.tanfovy)
r = geom["radii"]; vis = r > 0
m = geom["means2D"][vis].astype(np.float64); co = geom["conic_opacity"][vis].astype(np.float64); r = r[vis].astype(np.int64)
ca, cb, cc, op = co.T
det = ca * cc - cb * cb
t2 = np.maximum(2.0 * (np.log(op * 255.0) + 1e-4), 0)
hx = np.sqrt(t2 * cc / det) * 1.001 + 0.01
hy = np.sqrt(t2 * ca / det) * 1.001 + 0.01
box = (op * 255.0 >= 0.9999) & (det > 0)
gx, gy = (W + 15) // 16, (H + 15) // 16
x0 = np.clip((m[:, 0] - r) // 16, 0, gx).astype(int); x1 = np.clip((m[:, 0] + r + 15) // 16, 0, gx).astype(int)
y0 = np.clip((m[:, 1] - r) // 16, 0, gy).astype(int); y1 = np.clip((m[:, 1] + r + 15) // 16, 0, gy).astype(int)
I = Qq = Qs = Qs8 = 0; inst = 0
# per Gaussian: for tile columns / rows, number of regions hit factorises: quadrants = (#x halves hit) * (#y halves hit) per tile
def spans(lo, hi, t0, t1, cell, per):
    """for tiles t0..t1-1 (16 px each): number of `cell`-px cells (per tile: `per`) intersecting [lo, hi]"""
    out = []
    for t in range(t0, t1):
        base = 16 * t
        c = 0
        for k in range(per):
            a = base + k * cell
            if lo <= a + cell - 1 and hi >= a: c += 1
        out.append(c)
    return np.array(out)
idx = np.nonzero(box)[0]
rng = np.random.default_rng(0)
sample = rng.choice(idx, size=min(len(idx), 60000), replace=False)
for i in sample:
    xs2 = spans(m[i, 0] - hx[i], m[i, 0] + hx[i], x0[i], x1[i], 8, 2)
    ys2 = spans(m[i, 1] - hy[i], m[i, 1] + hy[i], y0[i], y1[i], 8, 2)
    ys4 = spans(m[i, 1] - hy[i], m[i, 1] + hy[i], y0[i], y1[i], 4, 4)
    xs1 = (xs2 > 0).astype(int)
    inst += len(xs2) * len(ys2)
    I += int(np.outer(ys2 > 0, xs2 > 0).sum())
    Qq += int(np.outer(ys2, xs2).sum())
    Qs += int(np.outer(ys4, xs1).sum())
sc = len(idx) / len(sample)
print(f"{cfg.name}: visible {vis.sum()}, boxed {len(idx)}; per eye (scaled): rect instances {inst*sc/1e6:.2f} M, iterations I {I*sc/1e6:.2f} M, "
      f"quadrant evals {Qq*sc/1e6:.2f} M ({Qq/I:.2f}/iter), strip evals {Qs*sc/1e6:.2f} M ({Qs/I:.2f}/iter)")
print(f"VALU model per eye: quadrants {(I*10+Qq*12.5)*sc/1e6:.1f} M, strips {(I*6+Qs*12.5)*sc/1e6:.1f} M")
print("median hx, hy:", np.median(hx[idx]), np.median(hy[idx]), " mean radius", r.mean())
