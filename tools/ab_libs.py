"""Same-box A/B of two (or more) builds of the library: kernel / iteration / alignment times of the single-pair path, Infinity-Cache
resident and HBM-fed (rotating over 8 copies of the pair), plus a hash of the poses (builds that only differ in scheduling must agree).
    python tools/ab_libs.py build NAME=-DFLAG[,-DFLAG2] ...      cross-compiles rgbd360_amd/lib/librgbd360_hip_NAME.so (no GPU needed)
    python tools/ab_libs.py run [rounds] NAME[@ENV=VAL] ...        on the GPU box; NAME 'default' = the product library; @ENV=VAL sets a variable for that arm"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbd360_amd import build as B


def lib_of(name):
    return B.LIB if name == "default" else os.path.join(os.path.dirname(B.LIB), "librgbd360_hip_%s.so" % name)


if sys.argv[1] == "build":
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition("=")
        fl = [f for f in flags.split(",") if f]
        B.compile_library(lib_of(name), [f for f in fl if f != "plain"], selects_vop3="plain" not in fl)     # NAME=plain: one-shot hipcc build (VOP2 selects)
        print(lib_of(name))
    sys.exit(0)

CHILD = r'''
import sys, time, hashlib, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
regs = []
for _ in range(8):
    r = RegisterPhotoICP(); r.setNumPyr(4)
    r.setTargetFrame(rgbA, dA); r.setSourceFrame(rgbB, dB)
    regs.append(r)
reg = regs[0]
h = hashlib.sha1()
out = []
for method in (0, 2):
    reg.alignFrames360(np.eye(4), method)
    pose = reg.getOptimalPose()
    h.update(pose.tobytes()); h.update(np.asarray(reg.num_iterations, np.int32).tobytes())
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): reg.alignFrames360(np.eye(4), method)
        best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    us = min(reg.time_eval_kernel(0, pose, method, True, 100) for _ in range(5))
    fus = min(reg.time_eval_kernel(0, pose, method, 2, 100) for _ in range(5))
    rus = min(RegisterPhotoICP.time_eval_kernel_rotating(regs, 0, pose, method, True, 80) for _ in range(5))
    rfus = min(RegisterPhotoICP.time_eval_kernel_rotating(regs, 0, pose, method, 2, 80) for _ in range(5))
    reg.forced_iters(0, np.eye(4), method, 200)
    r = [reg.forced_iters(0, np.eye(4), method, 400) for _ in range(3)]
    h.update(r[0]["pose"].tobytes())
    out.append("m%%d pass %%.2f | fused %%.2f | HBM-fed pass %%.2f fused %%.2f | align %%.1f us" %% (method, us, fus, rus, rfus, best))
if len(sys.argv) > 2 and sys.argv[2] == "4k":
    (a4, d4), (b4, e4), _ = synth.make_pair(4096, 2048, seed=1234)
    r4 = RegisterPhotoICP(); r4.setNumPyr(5); r4.setTargetFrame(a4, d4); r4.setSourceFrame(b4, e4)
    r4.alignFrames360(np.eye(4), 2); p4 = r4.getOptimalPose(); h.update(p4.tobytes())
    out.append("4k pass m0 %%.1f m2 %%.1f fused m2 %%.1f" %% tuple([min(r4.time_eval_kernel(0, p4, m, True, 30) for _ in range(3)) for m in (0, 2)] + [min(r4.time_eval_kernel(0, p4, 2, 2, 30) for _ in range(3))]))
print("; ".join(out), "| poses", h.hexdigest()[:12])
''' % ROOT
args = sys.argv[2:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
extra = ["4k"] if "4k" in args else []
names = [a for a in args if a != "4k"]
for rnd in range(rounds):
    for name in names:
        lib, _, envs = name.partition("@")
        env = dict(os.environ)
        for kv in envs.split("@"):
            if "=" in kv:
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        r = subprocess.run([sys.executable, "-c", CHILD, lib_of(lib)] + extra, capture_output=True, text=True, env=env)
        print("%-10s|" % name, r.stdout.strip() or r.stderr.strip()[-600:], flush=True)
