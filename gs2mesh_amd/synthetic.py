"""Synthetic workloads for BASELINE.json's configs.

There are no datasets or trained splats in the build/bench environment, so every
measurement and full-size parity property runs on seeded synthetic inputs of the named
shapes: ``synth_v1`` Gaussians, ring-of-cameras stereo poses, analytic sphere depth maps.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .graphics import Camera

SH_C0 = 0.28209479177387814  # GS/utils/sh_utils.py:24


def RGB2SH(rgb):
    """GS/utils/sh_utils.py:114-115."""
    return (rgb - 0.5) / SH_C0


@dataclass
class Config:
    name: str
    P: int
    seed: int
    log_s_mu: float
    width: int
    height: int
    focal: float
    n_pairs: int
    baseline_pct: float
    tsdf_n: int          # "N^3" label
    tsdf_extent: float
    sdf_trunc: float
    sphere_radius: float
    ring_radius: float = 3.5

    @property
    def voxel_length(self):
        return self.tsdf_extent / self.tsdf_n

    @property
    def baseline(self):
        return self.ring_radius * self.baseline_pct / 100.0


# BASELINE.md section 3 / SURVEY.md 8(d)
CONFIGS = {
    "C1": Config("C1", 10_000, 1001, math.log(0.02), 400, 300, 400.0, 2, 7.0, 128, 2.0, 0.16, 0.6),
    "C2": Config("C2", 300_000, 1002, math.log(0.006), 1600, 1200, 2900.0, 49, 7.0, 512, 2.0, 0.04, 0.6),
    "C3": Config("C3", 2_000_000, 1003, math.log(0.004), 1280, 840, 1100.0, 200, 7.0, 512, 2.0, 0.04, 0.6),
    "C4": Config("C4", 2_500_000, 1004, math.log(0.004), 1920, 1080, 1150.0, 300, 7.0, 1024, 4.0, 0.04, 1.2),
    "C5": Config("C5", 500_000, 1005, math.log(0.006), 1920, 1440, 1500.0, 120, 14.0, 512, 2.0, 0.04, 0.6),
}


def synth_v1(P, seed, log_s_mu):
    """Pre-activation Gaussian parameters in GaussianModel layout (all float32):
    xyz[P,3], features_dc[P,1,3], features_rest[P,15,3], scaling[P,3] (log), rotation[P,4]
    (wxyz, unnormalised), opacity[P,1] (logit)."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, size=(P, 3)).astype(np.float32)
    scaling = rng.normal(log_s_mu, 0.4, size=(P, 3)).astype(np.float32)
    rotation = rng.normal(0.0, 1.0, size=(P, 4)).astype(np.float32)
    o = rng.uniform(0.05, 0.95, size=(P, 1))
    opacity = np.log(o / (1.0 - o)).astype(np.float32)
    f_dc = RGB2SH(rng.uniform(0.0, 1.0, size=(P, 1, 3))).astype(np.float32)
    f_rest = rng.normal(0.0, 0.05, size=(P, 15, 3)).astype(np.float32)
    return dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling, rotation=rotation,
                opacity=opacity)


def trained_like(P, seed, log_s_mu, focal=2900.0, ring_radius=3.5):
    """Gaussians with the statistics of a TRAINED splat that ``synth_v1`` lacks (same layout): strongly anisotropic
    scales (10-100 : 1, surface-aligned discs and needles), 30 % of the opacities saturated at sigmoid >= 0.985 (the
    reference's 0.99 alpha cap, forward.cu:343), a mass of near-transparent ones, 0.1 % background splats whose screen
    radius exceeds 300 px (hundreds of tiles each: the > 64-tile walk of the binning), most centres on a surface shell,
    duplicated centres (exact depth ties in every view) and SH rest coefficients large enough to clamp colours at 0."""
    rng = np.random.default_rng(seed)
    n_shell = int(0.7 * P)
    d = rng.normal(size=(n_shell, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = d * (0.6 + rng.normal(0.0, 0.01, size=(n_shell, 1)))
    xyz = np.concatenate([shell, rng.uniform(-1.0, 1.0, size=(P - n_shell, 3))]).astype(np.float32)
    n_dup = max(2, P // 50)                                   # exact depth ties: copies of other centres
    dst = rng.choice(P, n_dup, replace=False)
    xyz[dst] = xyz[rng.choice(P, n_dup)]
    scaling = rng.normal(log_s_mu, 0.5, size=(P, 3))
    thin = rng.integers(1, 3, size=P)                         # 1 thin axis = disc, 2 = needle
    ratio = np.log(rng.uniform(10.0, 100.0, size=P))
    for a in range(3):
        scaling[:, a] -= np.where((a < thin), ratio, 0.0)
    perm = np.argsort(rng.random((P, 3)), axis=1)             # which axes are the thin ones
    scaling = np.take_along_axis(scaling, perm, axis=1)
    n_big = max(1, P // 1000)                                 # background splats: radius = 3 sigma f / z > 300 px
    big = rng.choice(P, n_big, replace=False)
    sigma = rng.uniform(1.0, 2.0, size=n_big) * 300.0 * ring_radius / (3.0 * focal)
    scaling[big, 0] = np.log(sigma)
    scaling[big, 1] = np.log(sigma * rng.uniform(0.3, 1.0, size=n_big))
    scaling[big, 2] = np.log(sigma * 0.01)
    u = rng.random(P)
    o = np.where(u < 0.30, rng.uniform(0.985, 0.9997, size=P),
                 np.where(u < 0.60, rng.uniform(0.004, 0.1, size=P), rng.uniform(0.1, 0.98, size=P)))
    o[big] = rng.uniform(0.02, 0.3, size=n_big)
    opacity = np.log(o / (1.0 - o)).astype(np.float32).reshape(P, 1)
    rotation = rng.normal(0.0, 1.0, size=(P, 4)).astype(np.float32)
    f_dc = RGB2SH(rng.uniform(0.0, 1.0, size=(P, 1, 3))).astype(np.float32)
    f_rest = rng.normal(0.0, 0.15, size=(P, 15, 3)).astype(np.float32)
    return dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling.astype(np.float32), rotation=rotation,
                opacity=opacity)


def ring_pose(azimuth, radius=3.5):
    """COLMAP world->camera [R|t] of a camera on a horizontal ring looking at the origin,
    up = -y (COLMAP: +x right, +y down, +z forward)."""
    c = np.array([radius * math.sin(azimuth), 0.0, -radius * math.cos(azimuth)])
    z = -c / np.linalg.norm(c)                 # forward: towards origin
    y = np.array([0.0, 1.0, 0.0])              # image-down = world +y  (up = -y)
    x = np.cross(y, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=0)            # rows = camera axes in world coords -> world->cam
    t = -R @ c
    return R, t


def ring_poses(n, radius=3.5, first=0, total=None):
    """n world->cam poses [n,3,4]; azimuths are k*2pi/total for k = first..first+n-1."""
    total = n if total is None else total
    out = np.zeros((n, 3, 4))
    for i in range(n):
        R, t = ring_pose(2.0 * math.pi * (first + i) / total, radius)
        out[i, :, :3] = R
        out[i, :, 3] = t
    return out


def stereo_cameras(pose_w2c, width, height, fx, fy, baseline):
    """(left, right) ``Camera`` for one COLMAP pose, exactly the pair Renderer.render_image_pair
    builds: R_gs = R_w2c^T, T_gs = t (left), T_gs = t - (baseline,0,0) (right)
    (verified against the reference's Euler detour in tests/test_golden_cameras.py)."""
    R = np.asarray(pose_w2c)[:, :3]
    t = np.asarray(pose_w2c)[:, 3]
    FoVx = 2 * np.arctan2(width, 2 * fx)
    FoVy = 2 * np.arctan2(height, 2 * fy)
    left = Camera(0, R.T, t, FoVx, FoVy, width, height)
    right = Camera(0, R.T, t - np.array([baseline, 0.0, 0.0]), FoVx, FoVy, width, height)
    return left, right


def sphere_depth(pose_w2c, width, height, fx, fy, cx, cy, radius):
    """Analytic z-depth map [H,W] float32 of a sphere of `radius` at the world origin
    (0 = no hit = invalid)."""
    R = np.asarray(pose_w2c)[:, :3]
    t = np.asarray(pose_w2c)[:, 3]
    u = (np.arange(width) - cx) / fx
    v = (np.arange(height) - cy) / fy
    dx, dy = np.meshgrid(u, v)
    d = np.stack([dx, dy, np.ones_like(dx)], axis=-1)          # camera-space ray, z = 1
    oc = t                                                      # sphere centre in camera coords = R*0 + t
    a = (d * d).sum(-1)
    b = -2.0 * (d @ oc)
    c = float(oc @ oc) - radius * radius
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), 0.0)
    s = np.where(s > 0, s, 0.0)
    return s.astype(np.float32)                                 # z-depth = s * d_z = s


def color_pattern(width, height):
    """Fixed u8 colour image ((u+v)%256, u%256, v%256)."""
    u, v = np.meshgrid(np.arange(width), np.arange(height))
    return np.stack([(u + v) % 256, u % 256, v % 256], axis=-1).astype(np.uint8)


def sphere_depth_torch(pose_w2c, width, height, fx, fy, cx, cy, radius, device):
    """`sphere_depth` evaluated on the device (float64 maths, float32 result)."""
    import torch
    t = torch.as_tensor(np.asarray(pose_w2c)[:, 3], dtype=torch.float64, device=device)
    u = (torch.arange(width, dtype=torch.float64, device=device) - cx) / fx
    v = (torch.arange(height, dtype=torch.float64, device=device) - cy) / fy
    dy, dx = torch.meshgrid(v, u, indexing="ij")
    a = dx * dx + dy * dy + 1.0
    b = -2.0 * (dx * t[0] + dy * t[1] + t[2])
    c = float((t * t).sum()) - radius * radius
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = torch.where(hit, (-b - torch.sqrt(torch.clamp(disc, min=0.0))) / (2 * a), torch.zeros_like(a))
    s = torch.where(s > 0, s, torch.zeros_like(s))
    return s.to(torch.float32).contiguous()
