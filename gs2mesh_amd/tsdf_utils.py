"""Pipeline-level drop-in for ``gs2mesh_utils.tsdf_utils.TSDF`` (tsdf_utils.py:23-142), fuse half.

``TSDF(renderer, stereo, args, out_name).run()`` walks the views exactly like the reference
(TSDF_dilate / TSDF_valid / TSDF_skip, masks, occlusion mask, min / max depth in baselines,
TSDF_scale on the extrinsic translation and the depth) and integrates them with the HIP block-sparse
volume instead of Open3D's CPU ``ScalableTSDFVolume``.  The per-pixel preprocessing
(depth *= mask, depth < min -> 0, depth/scale, >= trunc -> 0; tsdf_utils.py:68-93) is fused into the
integration kernels; only the 10x10 closing + erosion of the object mask (tsdf_utils.py:73-77) stays on
the host (scipy.ndimage; cv2 is not a dependency).  Frames can come from disk (``left.png``,
``out_<model>/depth.npy``, ...: the reference layout, so ``--skip_rendering`` style resumes work) or
from memory via ``frame_source`` (the in-memory hand-off).

After ``run()``: ``self.volume`` (gs2mesh_amd.integration.ScalableTSDFVolume) and ``self.mesh`` (marching
cubes on the GPU, ``gs2m_tsdf_extract``); ``save_mesh`` / ``clean_mesh`` write the reference's
``<out_name>_mesh.ply`` / ``<out_name>_cleaned_mesh.ply`` (tsdf_utils.py:112-142).
"""
from __future__ import annotations

import os

import numpy as np

from .integration import (Image, PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume,
                          TSDFVolumeColorType)


def _morph(m, k, erode):
    """cv2.erode / cv2.dilate with a k x k box of ones, default anchor (k//2, k//2) and default constant
    border (+inf for erode, -inf for dilate):  dst(y,x) = min|max over dy,dx in [-k//2, k-1-k//2] of
    src(y+dy, x+dx).  Written with explicit offsets so that even kernel sizes (10, the reference
    default) are unambiguous."""
    a = k // 2
    b = k - 1 - a
    pad = np.pad(m, ((a, b), (a, b)), constant_values=bool(erode))
    # separable: a box is the product of a row and a column segment
    win = np.lib.stride_tricks.sliding_window_view(pad, k, axis=1)
    r = win.all(axis=-1) if erode else win.any(axis=-1)
    win = np.lib.stride_tricks.sliding_window_view(r, k, axis=0)
    return win.all(axis=-1) if erode else win.any(axis=-1)


def preprocess_object_mask(mask, invert=False, erode=True, closing_kernel_size=10, erosion_kernel_size=10):
    """tsdf_utils.py:69-77: optional inversion, then cv2.MORPH_CLOSE (dilate, erode) and cv2.erode with
    k x k boxes of ones; returns a bool mask."""
    m = np.asarray(mask).astype(bool)
    if invert:
        m = ~m
    if erode:
        m = _morph(_morph(m, closing_kernel_size, erode=False), closing_kernel_size, erode=True)
        m = _morph(m, erosion_kernel_size, erode=True)
    return m


class TSDF:
    def __init__(self, renderer, stereo, args, out_name, frame_source=None, max_blocks=None, lib=None):
        self.model_name = stereo.model_name if stereo is not None else getattr(args, "stereo_model", "DLNR_Middlebury")
        self.renderer = renderer
        self.out_name = out_name
        self.args = args
        self.frame_source = frame_source      # callable(camera_number) -> dict(image, depth[, mask, occlusion])
        self.max_blocks = max_blocks
        self._lib = lib                       # None = the HIP library (tests inject the emulator build)
        self.volume = None
        self.mesh = None

    def _load_frame(self, camera_number):
        if self.frame_source is not None:
            return self.frame_source(camera_number)
        from PIL import Image as PILImage
        a = self.args
        d = self.renderer.render_folder_name(camera_number)
        fr = dict(image=np.array(PILImage.open(os.path.join(d, 'left.png'))).astype(np.uint8),
                  depth=np.load(os.path.join(d, f'out_{self.model_name}', 'depth.npy')))
        if a.TSDF_use_mask:
            fr["mask"] = np.load(os.path.join(d, 'left_mask.npy')).astype(bool)
        if a.TSDF_use_occlusion_mask:
            fr["occlusion"] = np.load(os.path.join(d, f'out_{self.model_name}', 'occlusion_mask.npy')).astype(bool)
        return fr

    def run(self, visualize=False):
        a = self.args
        n = len(self.renderer)
        valid = a.TSDF_valid if a.TSDF_valid is not None else list(range(n))
        skip = a.TSDF_skip if a.TSDF_skip is not None else []
        voxel_length = a.TSDF_voxel / 512
        kw = {} if self.max_blocks is None else dict(max_blocks=self.max_blocks)
        if self._lib is not None:
            kw["lib"] = self._lib
        volume = ScalableTSDFVolume(voxel_length=float(voxel_length), sdf_trunc=a.TSDF_sdf_trunc,
                                    color_type=TSDFVolumeColorType.RGB8, **kw)
        baseline = self.renderer.baseline
        for camera_number, left_camera in enumerate(self.renderer.left_cameras):
            if camera_number % a.TSDF_dilate != 0:
                continue
            if valid is not None and camera_number not in valid:
                continue
            if skip is not None and camera_number in skip:
                continue
            fr = self._load_frame(camera_number)
            mask = None
            if a.TSDF_use_mask and fr.get("mask") is not None:
                mask = preprocess_object_mask(fr["mask"], a.TSDF_invert_mask, a.TSDF_erode_mask,
                                              a.TSDF_closing_kernel_size, a.TSDF_erosion_kernel_size)
            if a.TSDF_use_occlusion_mask and fr.get("occlusion") is not None:
                occ = fr["occlusion"]
                occ = np.asarray(occ.cpu() if hasattr(occ, "cpu") else occ).astype(bool)
                mask = occ if mask is None else (np.asarray(mask).astype(bool) & occ)
            if mask is not None:
                mask = np.asarray(mask).astype(np.uint8)
            extrinsic = left_camera['extrinsic'].copy()
            extrinsic[:3, 3] /= a.TSDF_scale
            depth_trunc = baseline * a.TSDF_max_depth_baselines / a.TSDF_scale
            rgbd = RGBDImage.create_from_color_and_depth(Image(fr["image"]), Image(fr["depth"]),
                                                         depth_scale=a.TSDF_scale, depth_trunc=depth_trunc,
                                                         convert_rgb_to_intensity=False)
            intr = PinholeCameraIntrinsic(left_camera['width'], left_camera['height'], left_camera['fx'],
                                          left_camera['fy'], left_camera['cx'], left_camera['cy'])
            volume.integrate(rgbd, intr, np.linalg.inv(extrinsic), mask=mask,
                             min_depth=a.TSDF_min_depth_baselines * baseline)
        volume.status()
        self.volume = volume
        self.mesh = volume.extract_triangle_mesh()                  # tsdf_utils.py:108
        self.mesh.scale(a.TSDF_scale, (0, 0, 0))                    # :109
        self.mesh.compute_vertex_normals()                          # :110

    def save_mesh(self):
        """tsdf_utils.py:112-120."""
        from .mesh import write_triangle_mesh
        write_triangle_mesh(os.path.join(self.renderer.output_dir_root, f'{self.out_name}_mesh.ply'), self.mesh)
        print("SAVED MESH")

    def clean_mesh(self):
        """tsdf_utils.py:122-142: drop connected components with fewer than TSDF_cleaning_threshold / TSDF_scale
        triangles.  As in the reference the method rebinds ``self.clean_mesh`` to the cleaned mesh (:138)."""
        import copy
        from .mesh import write_triangle_mesh
        thres = self.args.TSDF_cleaning_threshold / self.args.TSDF_scale
        triangle_clusters, cluster_n_triangles, cluster_area = self.mesh.cluster_connected_triangles()
        triangles_to_remove = cluster_n_triangles[triangle_clusters] < thres
        self.clean_mesh = copy.deepcopy(self.mesh)
        self.clean_mesh.remove_triangles_by_mask(triangles_to_remove)
        self.clean_mesh.remove_unreferenced_vertices()
        write_triangle_mesh(os.path.join(self.renderer.output_dir_root, f'{self.out_name}_cleaned_mesh.ply'),
                            self.clean_mesh)
        print("SAVED CLEANED MESH")
