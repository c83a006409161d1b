"""``render()`` with the reference's signature and return dict (GS/gaussian_renderer/__init__.py:18-100),
forward only, on the HIP rasteriser.  Kept for callers that hold a 3DGS ``Camera``-like object and a
``GaussianModel``; the pipeline class ``Renderer`` uses the fused stereo entry point instead."""
from __future__ import annotations

import math

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def _dev(x, device):
    return torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Background tensor (bg_color) must be on the GPU, as in the reference."""
    device = pc.get_xyz.device
    screenspace_points = torch.zeros_like(pc.get_xyz)       # gradient carrier in the reference (:26); unused
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=_dev(bg_color, device), scale_modifier=scaling_modifier,
        viewmatrix=_dev(viewpoint_camera.world_view_transform, device),
        projmatrix=_dev(viewpoint_camera.full_proj_transform, device), sh_degree=pc.active_sh_degree,
        campos=_dev(viewpoint_camera.camera_center, device), prefiltered=False,
        debug=bool(getattr(pipe, "debug", False)))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    # 3D covariance: from scale / rotation inside the rasteriser, or (pipe.compute_cov3D_python, :56-61) on the host side
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    # colour: SH evaluated by the rasteriser, or (pipe.convert_SHs_python, :67-78) in torch, or the caller's override
    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            from .sh_utils import eval_sh
            n_coef = (pc.max_sh_degree + 1) ** 2
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, n_coef)
            dir_pp = pc.get_xyz - _dev(viewpoint_camera.camera_center, device).repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    with torch.no_grad():
        rendered_image, radii = rasterizer(means3D=pc.get_xyz, means2D=screenspace_points, shs=shs,
                                           colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales,
                                           rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}
