"""Builds the gfx950 HIP library in-tree (rgbd360_amd/lib/librgbd360_hip.so).

hipcc cross-compiles without a GPU; the resulting .so travels with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "rgbd360_api.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "photo_icp_kernels.h"), os.path.join(_HERE, "csrc", "gn_math.h"),
        os.path.join(_HERE, "csrc", "frame360_kernels.h"), os.path.join(_HERE, "csrc", "occlusion_kernels.h"),
        os.path.join(_HERE, "csrc", "pinhole_kernels.h"), os.path.join(_HERE, "csrc", "pbmap_register.h"), os.path.join(_HERE, "csrc", "multi_gpu.h"), os.path.join(_HERE, "csrc", "sequence_engine.h"), os.path.join(_HERE, "csrc", "rig_dense.h"), os.path.join(_HERE, "csrc", "host_wait.h"),
        os.path.join(_HERE, "..", "include", "rgbd360_hip.h"), os.path.join(_HERE, "..", "include", "rgbd360_hip_diag.h")]
LIB = os.path.join(_HERE, "lib", "librgbd360_hip.so")

# -ffp-contract=off: the warp front end must round exactly like the CPU oracle (see photo_icp_kernels.h);
# contraction is re-enabled per block where it is harmless.
# -fno-slp-vectorize: packed f32 math is not faster than scalar on gfx950 (tools/ubench/valu_rate.hip) and the
# packing costs ~50 v_mov per pixel in the fused kernel.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-kernarg-preload-count=16",     # leading scalar kernel arguments arrive in SGPRs (gfx940+): no s_load round trip before the state load
         "-Wno-unused-value"]
# RCCL (= NCCL on ROCm): the multi-GPU sequence entry all-gathers the solved poses over xGMI (csrc/multi_gpu.h)
LINK = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC] + LINK
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
