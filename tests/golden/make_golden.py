"""Generate golden vectors from the REFERENCE's own Python (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference, writes tests/golden/*.npz
    python tests/golden/make_golden.py --ply      # only the splat PLY fixture (the reference's save_ply, executed)
    python tests/golden/make_golden.py --open3d   # only where open3d==0.17.0 is installed: golden TSDF volume + mesh

The reference has no unit tests or golden vectors.  What it does have
are independent pure-Python implementations of sub-steps of the hot path that import on CPU:

  * GS/utils/sh_utils.py:57-112      eval_sh              -> pins SH -> RGB (forward.cu:20-71)
  * GS/utils/general_utils.py:64-110 build_rotation / build_scaling_rotation / strip_symmetric
    + GS/scene/gaussian_model.py:27-31 build_covariance_from_scaling_rotation
                                                            -> pins Sigma = R S^2 R^T packing
                                                               (forward.cu:118-152)
  * GS/utils/graphics_utils.py:38-71 getWorld2View2 / getProjectionMatrix, combined as in
    GS/scene/cameras.py:54-57                              -> pins the per-view uniforms
  * the rasteriser forward itself (forward.cu, rasterizer_impl.cu kernels), compiled for the CPU by
    oracle/build_ref.py                                    -> pins projection / binning / compositing
  * gs2mesh_utils/transformation_utils.py:23-63,83-141,207-224 (eul2rotm, rotm2eul,
    convert_R_T_to_GS, RT_from_rot_pos, calculate_right_camera_pose) chained exactly as
    Renderer.__init__ / render_image_pair do (renderer_utils.py:132-141,178-206,378-386)
                                                            -> pins the stereo camera poses

`general_utils.build_rotation` hard-codes device='cuda'; we run it with torch.zeros patched
to ignore the device keyword (CPU container).  cv2 is stubbed (only get_shading uses it).
The outputs are committed as small .npz fixtures; /root/reference is never read at test time.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
GS = os.path.join(REF, "third_party", "gaussian-splatting")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, GS)
    sys.path.insert(0, REF)
    from utils import sh_utils, graphics_utils, general_utils  # noqa
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_transformation_utils", os.path.join(REF, "gs2mesh_utils", "transformation_utils.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    return sh_utils, graphics_utils, general_utils, tu


def make_ply_fixture():
    """The splat PLY as the REFERENCE writes it: GaussianModel.construct_list_of_attributes + save_ply
    (GS/scene/gaussian_model.py:177-208) are lifted out of the file with ast and executed on CPU tensors; `plyfile`
    (absent here) is replaced by a 20-line stand-in that serialises the structured array the reference builds exactly as
    plyfile's binary writer does (header: `property float <name>` per field in dtype order, then the raw records).  The
    loader under test (gs2mesh_amd/gaussian_model.py) never sees our own writer."""
    import ast
    src = open(os.path.join(GS, "scene", "gaussian_model.py")).read()
    cls = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ClassDef) and n.name == "GaussianModel")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("construct_list_of_attributes", "save_ply")]
    assert len(fns) == 2
    captured = {}

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            return (name, elements)

    class PlyData:
        def __init__(self, els):
            self.els = els

        def write(self, path):
            (name, arr), = self.els
            with open(path, "wb") as f:
                f.write(b"ply\nformat binary_little_endian 1.0\n")
                f.write(f"element {name} {len(arr)}\n".encode())
                for field in arr.dtype.names:
                    assert arr.dtype[field] == np.dtype("f4")
                    f.write(f"property float {field}\n".encode())
                f.write(b"end_header\n")
                f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())
            captured["n"] = len(arr)

    ns = {"np": np, "torch": torch, "os": os, "mkdir_p": lambda p: None, "PlyElement": PlyElement, "PlyData": PlyData}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "gaussian_model.py", "exec"), ns)
    rng = np.random.default_rng(77)
    P = 7
    me = types.SimpleNamespace(
        _xyz=torch.tensor(rng.normal(0, 1, (P, 3)), dtype=torch.float32),
        _features_dc=torch.tensor(rng.normal(0, 1, (P, 1, 3)), dtype=torch.float32),
        _features_rest=torch.tensor(rng.normal(0, 0.1, (P, 15, 3)), dtype=torch.float32),
        _opacity=torch.tensor(rng.normal(0, 2, (P, 1)), dtype=torch.float32),
        _scaling=torch.tensor(rng.normal(-4, 0.5, (P, 3)), dtype=torch.float32),
        _rotation=torch.tensor(rng.normal(0, 1, (P, 4)), dtype=torch.float32))
    me.construct_list_of_attributes = lambda: ns["construct_list_of_attributes"](me)
    out = os.path.join(OUT, "reference_point_cloud.ply")
    ns["save_ply"](me, out)
    assert captured["n"] == P
    np.savez(os.path.join(OUT, "reference_point_cloud.npz"), **{k[1:]: getattr(me, k).numpy() for k in
                                                                ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")})
    print("wrote", out)


def make_open3d_golden():
    """Golden volumes from the REAL dependency (open3d==0.17.0, requirements.txt:15) -- only where it is installed
    (it is not in the build image: the tests that read tests/golden/open3d_tsdf.npz skip with that reason).  Same
    frames as tests/test_tsdf_parity.py::frames(3, 160, 120, 170.0), the reference's call sequence
    (gs2mesh_utils/tsdf_utils.py:53-56,88-93,106-108)."""
    import open3d as o3d
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    sys.path.insert(0, os.path.dirname(OUT))
    from test_tsdf_parity import frames
    frs, (W, H, fx, fy, cx, cy) = frames(3, 160, 120, 170.0)
    voxel, trunc = 2.0 / 128, 0.08
    vol = o3d.pipelines.integration.ScalableTSDFVolume(voxel_length=voxel, sdf_trunc=trunc,
                                                       color_type=o3d.pipelines.integration.TSDFVolumeColorType.RGB8)
    intr = o3d.camera.PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    for d, c, E in frs:
        rgbd = o3d.geometry.RGBDImage.create_from_color_and_depth(
            o3d.geometry.Image(np.ascontiguousarray(c)), o3d.geometry.Image(np.ascontiguousarray(d, np.float32)),
            depth_scale=1.0, depth_trunc=1e9, convert_rgb_to_intensity=False)
        vol.integrate(rgbd, intr, E)
    mesh = vol.extract_triangle_mesh()
    pc = vol.extract_voxel_point_cloud()          # voxel centres with weight > 0; colour channel = (tsdf + 1) / 2
    np.savez_compressed(os.path.join(OUT, "open3d_tsdf.npz"), version=o3d.__version__, voxel=voxel, trunc=trunc,
                        vertices=np.asarray(mesh.vertices), triangles=np.asarray(mesh.triangles),
                        vertex_colors=np.asarray(mesh.vertex_colors), voxel_points=np.asarray(pc.points),
                        voxel_tsdf01=np.asarray(pc.colors)[:, 0])
    print("wrote open3d_tsdf.npz from open3d", o3d.__version__)


def main():
    if "--open3d" in sys.argv:
        return make_open3d_golden()
    if "--ply" in sys.argv:
        return make_ply_fixture()
    sh_utils, graphics_utils, general_utils, tu = _import_reference()
    rng = np.random.default_rng(20240611)

    # ---- 1. SH -> RGB ------------------------------------------------------------------
    P = 512
    shs = rng.normal(0, 0.4, size=(P, 16, 3)).astype(np.float32)      # get_features layout [P,16,3]
    xyz = rng.uniform(-2, 2, size=(P, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, 4.0], np.float32)
    out = {}
    for deg in range(4):
        # GS/gaussian_renderer/__init__.py:73-78 (convert_SHs_python path)
        shs_view = torch.from_numpy(shs).transpose(1, 2).reshape(-1, 3, 16)
        dir_pp = torch.from_numpy(xyz) - torch.from_numpy(campos)[None].repeat(P, 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = sh_utils.eval_sh(deg, shs_view, dir_pp_normalized)
        out[f"rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()
    np.savez_compressed(os.path.join(OUT, "sh_rgb.npz"), shs=shs, xyz=xyz, campos=campos, **out)

    # ---- 2. Sigma = R S^2 R^T -----------------------------------------------------------
    scales = np.exp(rng.normal(np.log(0.05), 0.6, size=(P, 3))).astype(np.float32)
    rots_raw = rng.normal(0, 1, size=(P, 4)).astype(np.float32)
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    torch.zeros = cpu_zeros
    try:
        covs = {}
        for mod in (1.0, 0.5):
            # gaussian_model.py:27-31 ; rotation passed is the RAW _rotation (gaussian_model.py:117-118),
            # build_rotation normalises it internally (general_utils.py:79-81)
            L = general_utils.build_scaling_rotation(mod * torch.from_numpy(scales), torch.from_numpy(rots_raw))
            cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))
            covs[f"cov_mod{mod}"] = cov.numpy()
    finally:
        torch.zeros = real_zeros
    rots_n = torch.nn.functional.normalize(torch.from_numpy(rots_raw)).numpy()
    np.savez_compressed(os.path.join(OUT, "cov3d.npz"), scales=scales, rots_raw=rots_raw, rots_normalized=rots_n,
                        **covs)

    # ---- 3 + 4. stereo poses and per-view uniforms ------------------------------------
    from scipy.spatial.transform import Rotation
    n = 12
    Rw2c = Rotation.random(n, random_state=7).as_matrix()
    t = rng.uniform(-3, 3, size=(n, 3))
    poses = np.concatenate([Rw2c, t[..., None]], axis=-1)            # [n,3,4] world->cam (poses_from_file)
    W_, H_ = 1600, 1200
    fx, fy = 2892.33, 2883.18
    baseline = 0.245
    rec = dict(poses=poses, width=W_, height=H_, fx=fx, fy=fy, baseline=baseline)
    keys = ["left_rot", "left_pos", "right_rot", "right_pos", "extrinsic", "R_gs_left", "T_gs_left", "R_gs_right",
            "T_gs_right", "wvt_left", "proj", "full_left", "center_left", "wvt_right", "full_right", "center_right"]
    acc = {k: [] for k in keys}
    for i in range(n):
        # renderer_utils.py:134-141
        pose_inv = np.linalg.inv(np.vstack((poses[i], np.array([0, 0, 0, 1]))))
        cam_rot = tu.rotm2eul(pose_inv[:3, :3])
        rotation = tu.eul2rotm(cam_rot)
        rotation[:, 1:] *= -1
        cam_rot = tu.rotm2eul(rotation)
        cam_loc = pose_inv[:3, 3].tolist()
        # renderer_utils.py:181,192
        R_right, T_right = tu.calculate_right_camera_pose(cam_rot, cam_loc, baseline)
        extrinsic = tu.RT_from_rot_pos(tuple(cam_rot), tuple(cam_loc))
        acc["left_rot"].append(np.asarray(cam_rot, np.float64))
        acc["left_pos"].append(np.asarray(cam_loc, np.float64))
        acc["right_rot"].append(np.asarray(R_right, np.float64))
        acc["right_pos"].append(np.asarray(T_right, np.float64))
        acc["extrinsic"].append(extrinsic)
        FoVx = 2 * np.arctan2(W_, 2 * fx)   # renderer_utils.py:384-385
        FoVy = 2 * np.arctan2(H_, 2 * fy)
        proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        for eye, (rot, pos) in (("left", (tuple(cam_rot.tolist()), tuple(cam_loc))), ("right", (R_right, T_right))):
            R, T = tu.convert_R_T_to_GS(tuple(rot), tuple(pos))      # renderer_utils.py:381
            # GS/scene/cameras.py:54-57 on CPU
            wvt = torch.tensor(graphics_utils.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
            center = wvt.inverse()[3, :3]
            acc[f"R_gs_{eye}"].append(np.asarray(R, np.float64))
            acc[f"T_gs_{eye}"].append(np.asarray(T, np.float64))
            acc[f"wvt_{eye}"].append(wvt.numpy())
            acc[f"full_{eye}"].append(full.numpy())
            acc[f"center_{eye}"].append(center.numpy())
        acc["proj"].append(proj.numpy())
    np.savez_compressed(os.path.join(OUT, "stereo_cameras.npz"), **rec, **{k: np.stack(v) for k, v in acc.items()})
    # ---- 5. left-right consistency mask: execute the reference's OWN function body ----------------
    # Stereo.get_occlusion_mask (gs2mesh_utils/stereo_utils.py:149-179) is pure numpy but lives in a module
    # that imports DLNR / cv2; the function is lifted out of the file with ast and compiled on its own.
    import ast
    src = open(os.path.join(REF, "gs2mesh_utils", "stereo_utils.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "get_occlusion_mask")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "stereo_utils.py", "exec"), ns)
    H2, W2 = 96, 160
    xs = np.arange(W2)[None, :].repeat(H2, 0)
    true_disp = (18.0 + 10.0 * np.sin(xs / 23.0) + rng.normal(0, 0.05, (H2, W2))).astype(np.float32)
    L2R = true_disp.copy()
    R2L = (np.roll(true_disp, -18, axis=1) + rng.normal(0, 0.4, (H2, W2))).astype(np.float32)
    L2R[10:30, 40:70] += 9.0            # inconsistent patch -> occluded
    L2R[:, :12] = 30.0                  # projects left of the image -> occluded
    L2R[50:60, 100:120] = -400.0        # projects right of the image
    masks = {f"mask_thr{t}": ns["get_occlusion_mask"](None, L2R, R2L, t) for t in (1, 3)}
    np.savez_compressed(os.path.join(OUT, "occlusion_mask.npz"), L2R=L2R, R2L=R2L, **masks)

    # ---- 6. the rasteriser forward, run by the REFERENCE'S OWN KERNELS compiled for the CPU -------------
    # oracle/_ref (oracle/build_ref.py): cuda_rasterizer/forward.cu + three kernels of rasterizer_impl.cu built from
    # the sources where they lie, -ffp-contract=off (the literal IEEE sequence of the source; nvcc contracts
    # some a*b+c into FMAs, which moves results by ulps -- the HIP parity tolerances cover that).
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import math
    import oracle
    from gs2mesh_amd import synthetic
    W3, H3, f3 = 200, 136, 180.0                       # ragged: 12.5 x 8.5 tiles
    g = synthetic.synth_v1(2000, 4242, math.log(0.03))
    sc, qn, op = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs3 = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    pose = synthetic.ring_pose(0.3, 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W3, H3, f3, f3, 0.2)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    ref = oracle.ref_forward(g["xyz"], op, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W3, H3,
                             cam.tanfovx, cam.tanfovy, bg, shs=shs3, scales=sc, rotations=qn)
    np.savez_compressed(
        os.path.join(OUT, "ref_forward.npz"), W=W3, H=H3, xyz=g["xyz"], scales=sc, rotations=qn, opacity=op, shs=shs3,
        bg=bg, viewmatrix=np.asarray(cam.world_view_transform, np.float32), projmatrix=np.asarray(cam.full_proj_transform, np.float32),
        campos=np.asarray(cam.camera_center, np.float32), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        **{"out_" + k: v for k, v in ref.items() if k != "num_rendered"}, out_num_rendered=ref["num_rendered"])
    make_ply_fixture()
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
