"""Real spherical harmonics on the host (torch), for ``render(..., pipe.convert_SHs_python=True)``
(GS/gaussian_renderer/__init__.py:73-78 calls GS/utils/sh_utils.py:eval_sh).  The kernels evaluate the same basis
(forward.cu:20-71); this module restates it as a basis-matrix product and is pinned by tests/golden/sh_rgb.npz, which the
reference's own eval_sh produced.

Basis of degree <= 3 in the 3DGS ordering and sign convention (16 functions of the unit direction (x, y, z)):
  l=0: c0;   l=1: -c1 y, c1 z, -c1 x;
  l=2: c20 xy, c21 yz, c22 (2zz - xx - yy), c23 xz, c24 (xx - yy);
  l=3: c30 y(3xx - yy), c31 xyz, c32 y(4zz - xx - yy), c33 z(2zz - 3xx - 3yy), c34 x(4zz - xx - yy), c35 z(xx - yy),
       c36 x(xx - 3yy).
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[..., (deg+1)^2] values of the basis functions at unit directions ``dirs`` [..., 3]."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, C0)]
    if deg >= 1:
        cols += [-C1 * y, C1 * z, -C1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        cols += [C2[0] * (x * y), C2[1] * (y * z), C2[2] * (2.0 * zz - xx - yy), C2[3] * (x * z), C2[4] * (xx - yy)]
    if deg >= 3:
        cols += [C3[0] * y * (3.0 * xx - yy), C3[1] * (x * y) * z, C3[2] * y * (4.0 * zz - xx - yy),
                 C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), C3[4] * x * (4.0 * zz - xx - yy), C3[5] * z * (xx - yy),
                 C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """``sh`` [..., C, >= (deg+1)^2] coefficients, ``dirs`` [..., 3] unit directions -> [..., C]
    (same call signature as the reference's eval_sh)."""
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise ValueError(f"need {n} SH coefficients, got {sh.shape[-1]}")
    return (sh[..., :n] * sh_basis(deg, dirs).unsqueeze(-2)).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
