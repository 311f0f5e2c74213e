"""Builds libgs3d_hip.so (the C-ABI library of include/gs3d.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The library is built in-tree (wgpu-3dgs-core_amd/lib/), is
git-ignored and travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgs3d_hip.so")
SOURCES = [os.path.join(CSRC, "gs3d.hip"), os.path.join(CSRC, "gs_ply.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in
                  ("gs_kernel_lib.h", "gs_render_kernels.h", "gs_bundle_kernels.h", "gs_internal.h")] + [
    os.path.join(ROOT, "include", "gs3d.h")]

# -ffp-contract=off: the render path's results are defined operation by operation;
# fused multiply-adds appear only where __builtin_fmaf is written.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + ["-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed building libgs3d_hip.so")
    if verbose and res.stdout.strip():
        print(res.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
