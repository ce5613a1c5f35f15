"""Builds the gfx950 HIP library in-tree (rgbd360_amd/lib/librgbd360_hip.so).

hipcc cross-compiles without a GPU; the resulting .so travels with the repository snapshot.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "rgbd360_api.hip")
import glob


def deps():
    """Every source the one translation unit can include: csrc/*.h, csrc/*.hip and the public headers (a fixed list went stale twice)."""
    here = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_HERE, "csrc", "*.hip")))
    return here + sorted(glob.glob(os.path.join(_HERE, "..", "include", "*.h")))


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
    return any(os.path.getmtime(d) > t for d in deps())


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_SELECT_E32 = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+), ([^,]+), ([^,]+), vcc(\s*(;.*)?)$")


def widen_selects(asm_in: str, asm_out: str) -> int:
    """Rewrites every `v_cndmask_b32_e32 vD, a, b, vcc` of a device-ISA dump into the VOP3 encoding of the same instruction.
    Measured on gfx950 (tools/ubench/vcc_select.hip, valu_peak2.hip): the VOP2 form costs a SIMD 9.6 ns per wave-instruction
    whoever wrote VCC (only the select directly behind the v_cmp that produced VCC is exempt), the VOP3 form - naming VCC or any
    SGPR pair - 2.1-2.4 ns, like every other simple VALU instruction.  The compiler's shrink pass prefers the 4-byte VOP2 form and
    has no switch; same instruction, same operands, same result: only the encoding (and 4 bytes of code) changes.
    In the product kernels the selects are never back to back, and the whole-library A/B (tools/ab_libs.py run default plain; Frame360
    kernel trace of both builds) moved nothing by more than 1 %: the product build stays the one-shot hipcc build, and this path is
    kept for RGBD360_BUILD_SELECTS_VOP3=1 and for A/B runs."""
    n = 0
    out = []
    with open(asm_in) as f:
        for line in f:
            m = _SELECT_E32.match(line.rstrip("\n"))
            if m:
                line = "%sv_cndmask_b32_e64%s%s, %s, %s, vcc%s\n" % (m.group(1), m.group(2), m.group(3), m.group(4).strip(), m.group(5).strip(), m.group(6) or "")
                n += 1
            out.append(line)
    with open(asm_out, "w") as f:
        f.writelines(out)
    return n


def compile_library(out: str, extra_flags=(), verbose: bool = False, selects_vop3: bool = True) -> str:
    """hipcc in two halves: device code to assembly, the select encoding widened (widen_selects), assembled + linked + bundled with the
    LLVM tools hipcc itself drives, then the host half compiled with that bundle embedded (`-fcuda-include-gpubinary`, what the driver
    does internally).  selects_vop3=False is the plain one-shot hipcc build (A/B: tools/ab_libs.py)."""
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    flags = FLAGS + list(extra_flags)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not selects_vop3:
        run([hipcc()] + flags + ["-o", out, SRC] + LINK)
        return out
    dev_flags = [f for f in flags if f not in ("-shared", "-pthread")]
    with tempfile.TemporaryDirectory(prefix="rgbd360_build_") as tmp:
        asm, asm2, obj, hsaco, fb = (os.path.join(tmp, n) for n in ("dev.s", "dev_e64.s", "dev.o", "dev.hsaco", "dev.hipfb"))
        run([hipcc()] + dev_flags + ["-S", "--cuda-device-only", "-o", asm, SRC])
        n = widen_selects(asm, asm2)
        if verbose:
            print("widen_selects: %d v_cndmask_b32_e32 -> _e64" % n)
        run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm2, "-o", obj])
        run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, obj])
        run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
        run([hipcc()] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-o", out, SRC] + LINK)
    return out


def build(force: bool = False, verbose: bool = False, debug_knobs: bool = False) -> str:
    """debug_knobs: -DRGBD360_DEBUG_KNOBS, the library then reads the A/B environment variables of the measurement tools (csrc/knobs.h)."""
    if force or needs_build():
        compile_library(LIB, extra_flags=["-DRGBD360_DEBUG_KNOBS"] if debug_knobs else (), verbose=verbose,
                        selects_vop3=os.environ.get("RGBD360_BUILD_SELECTS_VOP3", "0") == "1")
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, debug_knobs="--debug-knobs" in sys.argv))
