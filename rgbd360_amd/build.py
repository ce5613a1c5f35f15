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
_CSRC = os.path.join(_HERE, "csrc")
# Three translation units since round 6 (one 3139-line .hip until then: 45 s per build, whatever was touched): contexts + alignment,
# the Frame360 stages, and the host-only entry points.  Each lists the headers only IT includes, so an edit of the per-pixel pass does
# not rebuild the Frame360 kernels and vice versa; the units compile in parallel.
_SHARED = ["knobs.h", "host_wait.h", "device_math.h", "f360_state.h"]
UNITS = {
    "rgbd360_api.hip": ["photo_icp_kernels.h", "occlusion_kernels.h", "pinhole_kernels.h", "gn_math.h", "sequence_engine.h", "rig_dense.h", "multi_gpu.h", "libm_f32.h"] + _SHARED,
    "rgbd360_frame360.hip": ["frame360_kernels.h", "pbmap_register.h"] + _SHARED,
    "rgbd360_host.cpp": ["depth_model.h", "pbmap_register.h"],
}
SRC = os.path.join(_CSRC, "rgbd360_api.hip")
import glob


def unit_deps(unit):
    return [os.path.join(_CSRC, unit)] + [os.path.join(_CSRC, h) for h in UNITS[unit]] + sorted(glob.glob(os.path.join(_HERE, "..", "include", "*.h")))


def deps():
    """Every source of the library: the units, the headers under csrc/ and the public headers."""
    here = sorted(glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.cpp")))
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


def _obj_path(unit: str, tag: str) -> str:
    return os.path.join(_HERE, "lib", "obj", tag, os.path.splitext(unit)[0] + ".o")


def compile_unit(unit: str, obj: str, flags, verbose: bool = False, selects_vop3: bool = False) -> None:
    """One translation unit -> a relocatable object (host code + the gfx950 code object embedded).  selects_vop3: device code to assembly,
    the select encoding widened (widen_selects), assembled / linked / bundled with the LLVM tools hipcc itself drives, then the host half
    compiled with that bundle embedded (`-fcuda-include-gpubinary`, what the driver does internally)."""
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    src = os.path.join(_CSRC, unit)
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    cflags = [f for f in flags if f not in ("-shared", "-pthread")]
    if unit.endswith(".cpp"):          # host only: no offload
        run([hipcc(), "-x", "c++", "-O3", "-std=c++17", "-fPIC", "-pthread", "-c", src, "-o", obj])
        return
    if not selects_vop3:
        run([hipcc()] + cflags + ["-pthread", "-c", src, "-o", obj])
        return
    with tempfile.TemporaryDirectory(prefix="rgbd360_build_") as tmp:
        asm, asm2, dobj, hsaco, fb = (os.path.join(tmp, n) for n in ("dev.s", "dev_e64.s", "dev.o", "dev.hsaco", "dev.hipfb"))
        run([hipcc()] + cflags + ["-S", "--cuda-device-only", "-o", asm, src])
        n = widen_selects(asm, asm2)
        if verbose:
            print("widen_selects: %d v_cndmask_b32_e32 -> _e64" % n)
        run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm2, "-o", dobj])
        run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dobj])
        run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
        run([hipcc()] + cflags + ["-pthread", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", obj])


def compile_library(out: str, extra_flags=(), verbose: bool = False, selects_vop3: bool = True, force: bool = True) -> str:
    """The units whose sources changed (all of them with `force`) are compiled side by side, then linked.  Objects are kept under
    lib/obj/<flag tag>/ so that the next build only redoes what was touched."""
    from concurrent.futures import ThreadPoolExecutor
    flags = FLAGS + list(extra_flags)
    tag = ("vop3" if selects_vop3 else "plain") + ("_" + "_".join(f.lstrip("-") for f in extra_flags) if extra_flags else "")
    todo = []
    for unit in UNITS:
        obj = _obj_path(unit, tag)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in unit_deps(unit)):
            todo.append((unit, obj))
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        for f in [ex.submit(compile_unit, u, o, flags, verbose, selects_vop3) for u, o in todo]:
            f.result()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + [_obj_path(u, tag) for u in UNITS] + LINK
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False, debug_knobs: bool = False) -> str:
    """debug_knobs: -DRGBD360_DEBUG_KNOBS, the library then reads the A/B environment variables of the measurement tools (csrc/knobs.h)."""
    if force or needs_build():
        compile_library(LIB, extra_flags=["-DRGBD360_DEBUG_KNOBS"] if debug_knobs else (), verbose=verbose,
                        selects_vop3=os.environ.get("RGBD360_BUILD_SELECTS_VOP3", "0") == "1", force=force)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, debug_knobs="--debug-knobs" in sys.argv))
