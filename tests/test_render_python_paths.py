"""The host-side alternatives of render() (GS/gaussian_renderer/__init__.py:56-79: pipe.compute_cov3D_python,
pipe.convert_SHs_python) and the debug snapshot of the operator shim (DGR __init__.py:83-90), pinned by the golden vectors the
reference's own Python produced (tests/golden/sh_rgb.npz, cov3d.npz)."""
import os

import numpy as np
import pytest
import torch


def test_eval_sh_matches_the_reference(golden_dir):
    from gs2mesh_amd.sh_utils import eval_sh
    g = np.load(os.path.join(golden_dir, "sh_rgb.npz"))
    shs = torch.from_numpy(g["shs"])                                  # get_features layout [P,16,3]
    xyz, campos = torch.from_numpy(g["xyz"]), torch.from_numpy(g["campos"])
    d = xyz - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    for deg in range(4):
        rgb = torch.clamp_min(eval_sh(deg, shs.transpose(1, 2).reshape(-1, 3, 16), d) + 0.5, 0.0).numpy()
        np.testing.assert_allclose(rgb, g[f"rgb_deg{deg}"], atol=2e-6, rtol=0)


def test_get_covariance_matches_the_reference(golden_dir):
    from gs2mesh_amd.gaussian_model import GaussianModel
    g = np.load(os.path.join(golden_dir, "cov3d.npz"))
    gm = GaussianModel(3, device="cpu")
    P = g["scales"].shape[0]
    z = np.zeros
    gm.load_arrays(z((P, 3)), z((P, 1, 3)), z((P, 15, 3)), np.log(g["scales"]), g["rots_raw"], z((P, 1)))
    for mod in (1.0, 0.5):
        np.testing.assert_allclose(gm.get_covariance(mod).numpy(), g[f"cov_mod{mod}"], rtol=3e-6, atol=1e-9)


def test_render_python_paths_feed_precomputed_inputs(monkeypatch):
    """render() with the two pipe switches hands colors_precomp / cov3D_precomp (not shs / scales+rotations) to the
    rasteriser, with the values of the host implementations."""
    from types import SimpleNamespace
    import gs2mesh_amd.gaussian_renderer as gr
    from gs2mesh_amd.gaussian_model import GaussianModel
    from gs2mesh_amd.graphics import Camera
    from gs2mesh_amd import synthetic
    g = synthetic.synth_v1(50, 3, -3.0)
    gm = GaussianModel(3, device="cpu")
    gm.load_arrays(g["xyz"], g["features_dc"], g["features_rest"], g["scaling"], g["rotation"], g["opacity"])
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 1.0, 0.8, 64, 48)
    seen = {}

    class FakeRasterizer:
        def __init__(self, raster_settings):
            pass

        def __call__(self, **kw):
            seen.update(kw)
            return torch.zeros(3, 48, 64), torch.zeros(50, dtype=torch.int32)

    monkeypatch.setattr(gr, "GaussianRasterizer", FakeRasterizer)
    out = gr.render(cam, gm, SimpleNamespace(compute_cov3D_python=True, convert_SHs_python=True, debug=False), torch.zeros(3))
    assert seen["shs"] is None and seen["scales"] is None and seen["rotations"] is None
    assert seen["cov3D_precomp"].shape == (50, 6) and seen["colors_precomp"].shape == (50, 3)
    assert (seen["colors_precomp"] >= 0).all() and set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
    seen.clear()
    gr.render(cam, gm, SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False), torch.zeros(3))
    assert seen["cov3D_precomp"] is None and seen["colors_precomp"] is None and seen["shs"].shape == (50, 16, 3)


def test_debug_mode_leaves_a_snapshot_on_failure(monkeypatch, tmp_path):
    import gs2mesh_amd.diff_gaussian_rasterization as dgr

    class Boom:
        def forward(self, *a, **k):
            raise RuntimeError("simulated device failure")

    monkeypatch.setattr(dgr, "_handle", lambda device: Boom())
    monkeypatch.chdir(tmp_path)
    P = 5
    rs = dgr.GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3, torch.zeros(3),
                                          False, True)
    r = dgr.GaussianRasterizer(rs)
    with pytest.raises(RuntimeError, match="simulated device failure"):
        r(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.ones(P, 1), shs=torch.zeros(P, 16, 3),
          scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    snap = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert len(snap) == 19 and snap[1].shape == (P, 3) and snap[14].shape == (P, 16, 3) and snap[18] is True
    # without debug: no dump
    os.remove(tmp_path / "snapshot_fw.dump")
    r2 = dgr.GaussianRasterizer(rs._replace(debug=False))
    with pytest.raises(RuntimeError):
        r2(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.ones(P, 1), shs=torch.zeros(P, 16, 3),
           scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    assert not os.path.exists(tmp_path / "snapshot_fw.dump")
