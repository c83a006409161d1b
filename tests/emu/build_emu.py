"""Build tests/emu/_build/libgs2mesh_emu.so: the kernel sources of gs2mesh_amd/csrc compiled
with g++ against the CPU fiber emulator (tests/emu/platform.h).  TEST INFRASTRUCTURE ONLY."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gs2mesh_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libgs2mesh_emu.so")
SRCS = ["raster_project.hip", "raster_bin.hip", "raster_blend.hip", "raster_api.hip", "tsdf_kernels.hip",
        "tsdf_api.hip", "stereo_kernels.hip"]


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(HERE, f) for f in ("platform.h", "emu_runtime.cpp")]
    for root, _, files in os.walk(CSRC):
        if "_obj" in root or root.endswith("/hip"):
            continue
        deps += [os.path.join(root, f) for f in files if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(ROOT, "include", "gs2mesh_amd.h"))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
        return LIB
    common = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
              "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-unused-variable",
              "-I", HERE, "-I", CSRC]
    objs = []
    procs = []
    for s in SRCS:
        p = os.path.join(CSRC, s)
        if not os.path.exists(p):
            continue
        o = os.path.join(OUT, s.replace(".hip", ".o"))
        objs.append(o)
        procs.append((s, subprocess.Popen(common + ["-x", "c++", "-c", p, "-o", o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT)))
    o = os.path.join(OUT, "emu_runtime.o")
    objs.append(o)
    procs.append(("emu_runtime.cpp", subprocess.Popen(common + ["-c", os.path.join(HERE, "emu_runtime.cpp"), "-o", o],
                                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ failed on {s}:\n{out.decode(errors='replace')}")
    r = subprocess.run(["g++", "-shared", "-fopenmp", "-o", LIB] + objs, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force=True))
