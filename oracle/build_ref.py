"""Build oracle/_ref/libref_raster.so: the REFERENCE's own rasteriser kernels compiled for the CPU.

    python oracle/build_ref.py            # needs /root/reference (the build container); a no-op elsewhere

The reference's forward path is C++/CUDA that compiles from its own few source files once the CUDA execution
model is shimmed (oracle/ref_shim/, oracle/ref_driver.cpp): cuda_rasterizer/forward.cu + auxiliary.h + config.h,
three small kernels of rasterizer_impl.cu, and the vendored header-only glm.  Nothing is copied into the
repository: the only text derived from the reference is a temporary include file under oracle/_ref/ (git-ignored)
holding forward.cu minus its two host launchers (the `<<< >>>` launch syntax is not C++) and the three kernels of
rasterizer_impl.cu (that file's host code needs CUB).  Test infrastructure only."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GS2M_REFERENCE", "/root/reference")
DGR = os.path.join(REF, "third_party", "gaussian-splatting", "submodules", "diff-gaussian-rasterization")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libref_raster.so")


def available() -> bool:
    return os.path.exists(os.path.join(DGR, "cuda_rasterizer", "forward.cu"))


def build(force: bool = False) -> str | None:
    if not available():
        return LIB if os.path.exists(LIB) else None
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(HERE, "ref_driver.cpp"), os.path.join(HERE, "build_ref.py")] + \
           [os.path.join(HERE, "ref_shim", f) for f in os.listdir(os.path.join(HERE, "ref_shim")) if f.endswith(".h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(p) for p in srcs):
        return LIB
    fwd = open(os.path.join(DGR, "cuda_rasterizer", "forward.cu")).read()
    cut = fwd.index("void FORWARD::render(")                      # host launchers start here
    impl = open(os.path.join(DGR, "cuda_rasterizer", "rasterizer_impl.cu")).read()
    a = impl.index("__global__ void checkFrustum(")
    b = impl.index("void CudaRasterizer::Rasterizer::markVisible(")
    inc = os.path.join(OUT, "ref_kernels.inc")
    with open(inc, "w") as f:
        f.write("// GENERATED at build time from the reference sources (see oracle/build_ref.py); not tracked.\n")
        f.write(fwd[:cut])
        f.write("\n")
        f.write(impl[a:b])
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared", "-w",
           "-I", os.path.join(HERE, "ref_shim"), "-I", os.path.join(DGR, "cuda_rasterizer"),
           "-I", os.path.join(DGR, "third_party", "glm"), f'-DREF_KERNELS_INC="{inc}"',
           os.path.join(HERE, "ref_driver.cpp"), "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("building oracle/_ref failed:\n" + r.stdout.decode(errors="replace")[-6000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
