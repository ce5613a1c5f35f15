"""Diagnostic: where a launch of the per-pixel pass spends its time, Infinity-Cache-resident (back-to-back launches over one pair) vs
HBM-fed (launches rotating over 8 copies of the pair).  Needs the RGBD360_EVAL_STAMPS build:
    python tools/eval_stamps.py build      (cross-compiles, no GPU needed)      then on the GPU box:  python tools/eval_stamps.py [fused]
Per-block stamps (100 MHz): start, pose arrived, gate passed / first warp stage issued, loop done, wave reduction done, end."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbd360_amd import _lib, build as B
LIB = os.path.join(os.path.dirname(B.LIB), "librgbd360_hip_estamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call([B.hipcc()] + B.FLAGS + ["-DRGBD360_EVAL_STAMPS", "-o", LIB, B.SRC] + B.LINK)
    print(LIB)
    sys.exit(0)
import numpy as np
_lib.LIB_PATH = LIB
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
mode = 2 if (len(sys.argv) > 1 and sys.argv[1] == "fused") else True
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
regs = []
for _ in range(8):
    r = RegisterPhotoICP(); r.setNumPyr(4)
    r.setTargetFrame(rgbA, dA); r.setSourceFrame(rgbB, dB)
    regs.append(r)
reg = regs[0]
reg.alignFrames360(np.eye(4), 2)
pose = reg.getOptimalPose()
L = reg._L
L.rgbd360_debug_eval_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.rgbd360_debug_eval_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]


def report(tag, r):
    blocks = np.zeros(2 * 256)
    nb = L.rgbd360_debug_eval_blocks(r._ctx(), 0, blocks.ctypes.data_as(C.c_void_p))
    se = blocks[:2 * nb].reshape(nb, 2) / 100.0          # us
    t0 = se[:, 0].min()
    st = np.zeros(12)
    L.rgbd360_debug_eval_stamps(r._ctx(), 0, st.ctypes.data_as(C.c_void_p))
    print("%s: blocks start %.2f..%.2f us after the first, run %.2f (min) %.2f (median) %.2f (max) us, last end %.2f us" % (
        tag, 0.0, (se[:, 0] - t0).max(), (se[:, 1] - se[:, 0]).min(), np.median(se[:, 1] - se[:, 0]), (se[:, 1] - se[:, 0]).max(), (se[:, 1] - t0).max()))
    run = se[:, 1] - se[:, 0]
    print("   block run time us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f | by position (32 blocks each): %s" % (
        run.mean(), np.percentile(run, 10), np.percentile(run, 50), np.percentile(run, 90), run.max(),
        " ".join("%.1f" % run[k:k + 32].mean() for k in range(0, nb, 32))))
    wv = np.zeros(16 * 256)
    L.rgbd360_debug_eval_waves.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    nbw = L.rgbd360_debug_eval_waves(r._ctx(), 0, wv.ctypes.data_as(C.c_void_p))
    wv = wv[:16 * nbw].reshape(nbw, 16) / 100.0
    print("   per-wave loop end, mean over blocks (us from block start), wave 0..15: " + " ".join("%.2f" % x for x in wv.mean(0)))
    print("   last wave - first wave per block: mean %.2f us; slowest wave index histogram: %s" % (
        (wv.max(1) - wv.min(1)).mean(), np.bincount(wv.argmax(1), minlength=16).tolist()))
    for name, row in (("block 0", st[:6]), ("block nb-1", st[6:])):
        print("   %s: pose %.2f | first stage %.2f | loop %.2f | wave reduce %.2f | end %.2f us" % ((name,) + tuple(row[:5] / 100.0)))


for method in (0, 2):
    us = reg.time_eval_kernel(0, pose, method, mode, 50)
    report("method %d resident (%.2f us/launch)" % (method, us), reg)
    us = RegisterPhotoICP.time_eval_kernel_rotating(regs, 0, pose, method, mode, 80)
    report("method %d HBM-fed  (%.2f us/launch)" % (method, us), regs[7])
