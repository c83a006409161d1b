"""Build libgs2mesh_amd.so (HIP, gfx950) in-tree:  python -m gs2mesh_amd.build

Explicit hipcc invocations, one object per stage so that floating-point contraction can be set
per stage (projection / TSDF: -ffp-contract=off = literal IEEE sequence, bit-comparable with
the CPU oracle; blend: default FMA contraction).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgs2mesh_amd.so")
ARCH = "gfx950"

# (source, extra flags)
SOURCES = [
    ("raster_project.hip", ["-ffp-contract=off"]),
    ("raster_bin.hip", []),
    # packed-f32 SLP costs register copies in the compositing loop
    ("raster_blend.hip", ["-fno-slp-vectorize"]),
    ("raster_api.hip", []),
    ("tsdf_kernels.hip", ["-ffp-contract=off"]),
    ("tsdf_api.hip", []),
    ("stereo_kernels.hip", ["-ffp-contract=off"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(CSRC, "hip"), "-I", CSRC,
          "-Wall", "-Wno-unused-function"]


def _hipcc():
    h = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(h):
        raise RuntimeError("hipcc not found (need ROCm)")
    return h


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        if "_obj" in root:
            continue
        for f in files:
            if f.endswith((".h", ".hip")):
                out.append(os.path.join(root, f))
    out.append(os.path.join(HERE, "..", "include", "gs2mesh_amd.h"))
    return out


def build(force: bool = False, verbose: bool = False, extra: list[str] | None = None) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    newest = max(os.path.getmtime(p) for p in _deps())
    objs = []
    procs = []
    for src, flags in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= newest:
            continue
        cmd = [hipcc, "-c", s, "-o", o] + COMMON + flags + (extra or []) + os.environ.get("GS2M_BUILD_EXTRA", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    if force or procs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
