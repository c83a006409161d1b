"""gs2mesh_amd -- MI355X-native render -> fuse hot path of GS2Mesh.

Host side (Python on PyTorch-ROCm) of the C-ABI library ``libgs2mesh_amd.so`` (hand-written
HIP kernels for gfx950, sources in ``gs2mesh_amd/csrc``).  Mirrors the reference's
interfaces for this path only:

  * ``gs2mesh_amd.diff_gaussian_rasterization``  operator API of the rasteriser
    (DGR/diff_gaussian_rasterization/__init__.py)
  * ``gs2mesh_amd.integration``                  Open3D ScalableTSDFVolume subset used by
    gs2mesh_utils/tsdf_utils.py
  * ``gs2mesh_amd.renderer_utils.Renderer`` / ``gs2mesh_amd.tsdf_utils.TSDF``
    pipeline classes called by run_single.py

There is no CPU fallback: importing the package works anywhere, but every op raises
``RuntimeError`` if the HIP library has not been built (``python -m gs2mesh_amd.build``).
"""
__version__ = "0.1.0"
