"""Pipeline-level drop-in for ``gs2mesh_utils.renderer_utils.Renderer`` (renderer_utils.py:105-395).

Same constructor signature, attributes (``cameras``, ``left_cameras``, ``baseline``, ``poses``,
``output_dir_root``, ``args``, ``__len__``), methods (``render_folder_name``, ``save_camera_data``,
``prepare_renderer``, ``render_image_pair``) and on-disk layout (``<root>/<NNN>/{left,right}.png``,
``camera_data.json``), so ``run_single.py:76-103`` and ``Stereo.run`` (stereo_utils.py:95-103) work
unchanged.  What changes underneath:

  * both eyes are rendered in ONE fused pass (``gs2m_render_views``) straight from the PLY's
    pre-activation parameters -- no per-eye ``Camera`` upload of a random image
    (renderer_utils.py:386), no ``torch.cat`` of the SH block per call;
  * ``prepare_renderer`` reads only ``point_cloud/iteration_N/point_cloud.ply``; the reference builds a
    3DGS ``Scene`` that loads every training image just to reach ``load_ply``
    (GS/scene/__init__.py:71-81);
  * ``render_pair_device`` is the in-memory hand-off: u8 HWC images on the device,
    quantised exactly as ``cv2.imwrite`` would (round-half-even, saturate).
"""
from __future__ import annotations

import copy
import json
import os

import numpy as np
import torch
from scipy.optimize import least_squares

from .colmap_io import poses_from_file, read_cameras_text
from .gaussian_model import GaussianModel, read_gaussian_ply
from .graphics import Camera
from .poses import (RT_from_rot_pos, calculate_right_camera_pose, convert_R_T_to_GS, eul2rotm,
                    intrinsic_from_camera_params, rotm2eul)
from . import _lib
from .rasterizer import Rasterizer, camera_from


def sort_camera_coordinates(coordinates):
    """Greedy nearest-neighbour ordering starting from the lowest-z camera
    (renderer_utils.py:33-99): among the two nearest unvisited cameras take the one closer in z."""
    coordinates = np.asarray(coordinates)
    n = len(coordinates)
    visited = np.zeros(n, dtype=bool)
    order = []
    cur = int(np.argmin(coordinates[:, 2]))
    while not visited.all():
        visited[cur] = True
        order.append(cur)
        if visited.all():
            break
        d = np.linalg.norm(coordinates - coordinates[cur], axis=1)
        d[visited] = np.inf
        d[cur] = np.inf
        cand = np.argsort(d)[:2]
        if len(cand) == 0:
            break
        cur = int(cand[np.argmin(np.abs(coordinates[cand][:, 2] - coordinates[cur][2]))])
    return order


class Renderer:
    def __init__(self, base_dir, colmap_dir, output_dir_root, args, dataset='custom', splatting='custom',
                 experiment_name=None, device='cuda'):
        self.args = args
        self.render_name = args.colmap_name
        self.white_background = args.GS_white_background
        self.base_dir = base_dir
        self.colmap_dir = colmap_dir
        self.output_dir_root = output_dir_root
        self.device = device
        self.splatting_iteration = args.GS_iterations
        self.splatting_dir = os.path.join(base_dir, 'splatting_output', splatting, self.render_name)
        self.splatting_ply_file_path = os.path.join(self.splatting_dir, 'point_cloud',
                                                    f"iteration_{self.splatting_iteration}", 'point_cloud.ply')
        self.poses = torch.from_numpy(poses_from_file(os.path.join(self.colmap_dir, 'sparse', '0', 'images.txt')))
        poses_inv = [np.linalg.inv(np.vstack((p, np.array([0, 0, 0, 1])))) for p in self.poses.numpy()]
        camera_rotations = [rotm2eul(p[:3, :3]) for p in poses_inv]
        for i in range(len(camera_rotations)):           # renderer_utils.py:136-139: flip y/z, back to Euler
            rot = eul2rotm(camera_rotations[i])
            rot[:, 1:] *= -1
            camera_rotations[i] = rotm2eul(rot)
        camera_locations = [p[:3, 3].tolist() for p in poses_inv]
        cams = read_cameras_text(os.path.join(self.colmap_dir, 'sparse', '0', 'cameras.txt'))
        camera_params = []
        for k in sorted(cams):
            c = cams[k]
            sr = c.model == 'SIMPLE_RADIAL'
            camera_params.append({'width': c.width, 'height': c.height, 'fx': c.params[0],
                                  'fy': c.params[0 if sr else 1], 'cx': c.params[1 if sr else 2],
                                  'cy': c.params[2 if sr else 3]})
        if len(camera_params) != len(camera_locations):
            camera_params = [camera_params[0]] * len(camera_locations)
        if args.renderer_baseline_absolute is not None:
            self.baseline = args.renderer_baseline_absolute
        else:
            ts = np.array(camera_locations)
            if args.renderer_scene_360:
                radius = np.median(np.linalg.norm(ts - ts.mean(axis=0), axis=1))
                if args.dataset_name == "DTU":
                    radius *= 2
            else:
                x, y, z = ts[:, 0], ts[:, 1], ts[:, 2]
                res = least_squares(lambda p: np.sqrt((x - p[0]) ** 2 + (y - p[1]) ** 2 + (z - p[2]) ** 2) - p[3],
                                    np.array([x.mean(), y.mean(), z.mean(), 1.0]))
                radius = res.x[3]
            self.baseline = radius * (args.renderer_baseline_percentage / 100)
        if args.renderer_sort_cameras:
            self.sorted_camera_indices = sort_camera_coordinates(np.array(camera_locations))
            self.poses = self.poses[torch.tensor(self.sorted_camera_indices)]
        else:
            self.sorted_camera_indices = range(len(camera_locations))
        self.cameras = []
        for i in range(len(camera_locations)):
            ci = self.sorted_camera_indices[i]
            cp = camera_params[ci]
            R_right, T_right = calculate_right_camera_pose(camera_rotations[ci], camera_locations[ci], self.baseline)
            common = {'width': cp['width'], 'height': cp['height'], 'fx': float(cp['fx']), 'fy': float(cp['fy']),
                      'cx': float(cp['cx']), 'cy': float(cp['cy'])}
            ext = RT_from_rot_pos(tuple(camera_rotations[ci]), tuple(camera_locations[ci]))
            self.cameras.append({
                'left': {'rot': tuple(camera_rotations[ci].tolist()), 'pos': tuple(camera_locations[ci]), **common,
                         'intrinsic': intrinsic_from_camera_params(cp), 'extrinsic': ext, 'baseline': self.baseline},
                'right': {'rot': R_right, 'pos': T_right, **common, 'intrinsic': intrinsic_from_camera_params(cp),
                          'extrinsic': ext.copy()}})
        print(f"num views: {len(self.cameras)}")
        print(f"baseline: {self.baseline}")
        self.left_cameras = [c['left'] for c in self.cameras]
        if args.renderer_save_json:
            self.save_camera_data()
        try:        # visualisation-only point cloud (renderer_utils.py:216)
            self.GS_ply_points = read_gaussian_ply(self.splatting_ply_file_path)["xyz"]
        except OSError:
            self.GS_ply_points = np.zeros((0, 3), np.float32)
        self._raster = None
        self._views = {}

    def __len__(self):
        return len(self.cameras)

    def render_folder_name(self, render_number):
        return os.path.join(self.output_dir_root, f"{render_number:03}")

    def save_camera_data(self):
        os.makedirs(self.output_dir_root, exist_ok=True)
        cams = copy.deepcopy(self.cameras)
        for c in cams:
            for eye in ('left', 'right'):
                c[eye]['intrinsic'] = c[eye]['intrinsic'].tolist()
                c[eye]['extrinsic'] = c[eye]['extrinsic'].tolist()
        with open(os.path.join(self.output_dir_root, 'camera_data.json'), 'w') as f:
            json.dump(cams, f, indent=4)

    def prepare_renderer(self):
        """Load the trained splat (sh_degree 3, renderer_utils.py:344) onto the device."""
        dev = self.device if self.device != 'cuda' else f"cuda:{torch.cuda.current_device()}"
        self.gaussians = GaussianModel(3, device=dev)
        self.gaussians.load_ply(self.splatting_ply_file_path)
        bg = [1, 1, 1] if self.white_background else [0, 0, 0]
        self.background = torch.tensor(bg, dtype=torch.float32, device=dev)
        self._bg_host = tuple(float(b) for b in bg)
        self._raster = Rasterizer(torch.device(dev).index or 0)
        from .rasterizer import auto_blend_mode, auto_cull_level
        raw = self.gaussians.raw()
        self._raster.set_option(_lib.OPT_BLEND_MODE, auto_blend_mode(raw))   # same image; a trained splat has many opacities at the cap
        self._raster.set_option(_lib.OPT_EXACT_TILE_CULL, auto_cull_level(int(raw["xyz"].shape[0])))   # image-preserving; fewer instances to sort/blend
        self._raster.set_option(_lib.OPT_TILE_ROWS, 2)            # 16 x 32 binning tiles: same image, ~30 % fewer instances
        # one-time re-layout: Morton-ordered packed copy of the splat (a trained splat is stored in densification order,
        # i.e. spatially random) + wave-transposed SH; images / radii are those of the model as loaded
        from .rasterizer import auto_spatial_order
        if auto_spatial_order(int(raw["xyz"].shape[0])):    # same rule as RenderFusePipeline(spatial_order="auto")
            self._raster.pack_model(raw)
        else:
            self._raster.pack_sh(raw)
        self._views = {}

    def _pair(self, camera_number):
        if camera_number not in self._views:
            out = []
            for name in ('left', 'right'):
                c = self.cameras[camera_number][name]
                R, T = convert_R_T_to_GS(tuple(c['rot']), tuple(c['pos']))
                w, h = c['width'], c['height']
                FoVx = 2 * np.arctan2(w, 2 * c['fx'])      # principal point ignored, as in the reference
                FoVy = 2 * np.arctan2(h, 2 * c['fy'])
                out.append(camera_from(Camera(0, R, T, FoVx, FoVy, w, h)))
            self._views[camera_number] = out
        return self._views[camera_number]

    def render_pair_device(self, camera_number, want_color=False):
        """In-memory hand-off: dict(rgb8=[2,H,W,3] u8 device tensor (left, right), color=[2,3,H,W] f32)."""
        if self._raster is None:
            raise RuntimeError("call prepare_renderer() first")
        with torch.no_grad():
            return self._raster.render_views(self.gaussians.raw(), self._pair(camera_number), bg=self._bg_host,
                                             want_color=want_color, want_rgb8=True)

    def render_image_pair(self, camera_number, visualize=False):
        """Render the stereo pair of view `camera_number` and write left.png / right.png
        (renderer_utils.py:363-395)."""
        from PIL import Image as PILImage
        res = self.render_pair_device(camera_number)
        rgb8 = res["rgb8"].cpu().numpy()
        out_dir = self.render_folder_name(camera_number)
        os.makedirs(out_dir, exist_ok=True)
        for k, name in enumerate(('left', 'right')):
            PILImage.fromarray(rgb8[k], mode="RGB").save(os.path.join(out_dir, f'{name}.png'))
        if visualize:
            import matplotlib.pyplot as plt
            plt.imshow(rgb8[0])
            plt.imshow(rgb8[1], alpha=0.5)
            plt.show()
