"""Builds libtravgpu.so (the gfx950 HIP kernels + the C-ABI shim) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels with
the gpurun snapshot, so the GPU box uses the prebuilt file.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = ["csrc/te_kernels.hip", "csrc/te_shim.hip", "csrc/te_fast_step.hip", "csrc/te_slide_normals.hip", "csrc/te_footprint.hip"]
HEADERS = ["csrc/te_internal.h", "csrc/te_march.h", "csrc/te_cell.h", "../include/travgpu.h"]
LIB = os.path.join(_HERE, "libtravgpu.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build libtravgpu.so)")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(_HERE, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc")]
    cmd += [os.path.join(_HERE, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
