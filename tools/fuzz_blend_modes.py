import os, sys, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer
rng = np.random.default_rng(77)
bad = 0
for it in range(40):
    W, H = int(rng.integers(40, 400)), int(rng.integers(40, 300))
    P = int(rng.integers(200, 20000))
    f = float(rng.uniform(100, 400))
    log_s = math.log(float(rng.uniform(0.004, 0.08)))
    g = synthetic.trained_like(P, 1000 + it, log_s, focal=f) if it % 2 else synthetic.synth_v1(P, 1000 + it, log_s)
    if it % 3 == 0:   # push many opacities to the cap
        g["opacity"] = np.where(rng.random(g["opacity"].shape) < 0.6, 6.0, g["opacity"]).astype(np.float32)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    pose = synthetic.ring_pose(float(rng.uniform(0, 6.28)), 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W, H, f, f, 0.245)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bg = np.asarray(rng.random(3), np.float32)
    imgs = {}
    for rows in (1, 2):
        for cull in (0, 1, 2):
            for mode in (0, 2, 3):
                r = Rasterizer(0)
                r.set_option(_lib.OPT_EXACT_TILE_CULL, cull); r.set_option(_lib.OPT_TILE_ROWS, rows); r.set_option(_lib.OPT_BLEND_MODE, mode)
                img, _ = r.forward(d(g["xyz"]), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center), d(bg), W, H,
                                   cam.tanfovx, cam.tanfovy, shs=d(shs), scales=d(s), rotations=d(q))
                imgs[(rows, cull, mode)] = img.cpu().numpy()
                r.close()
    ref = imgs[(1, 0, 0)]
    for k, v in imgs.items():
        if not np.array_equal(v, ref):
            bad += 1
            print("MISMATCH", it, W, H, P, k, float(np.abs(v - ref).max()), int((v != ref).sum()))
print("fuzz done: 40 scenes x 18 settings (tile rows x cull level x loop form), mismatches:", bad)
