"""What the contract's torch.cuda.synchronize() costs behind a forced-schedule call, by the device's schedule flag
(hipSetDeviceFlags: 0 auto, 1 spin, 2 yield, 4 blocking).  python tools/sync_flags_probe.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
(rgbA, dA), (rgbB, dB), _ = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4); reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB); reg.sync()
reg.forced_iters(0, np.eye(4), 0, 50)
call = reg.forced_iters_call(0, np.eye(4), 0, 20)
def med(f, n=41):
    t = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e6
for flag in (None, 1, 2, 4, 0, 1):
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag)) if flag is not None else "-"
    f = ctypes.c_uint(99); hip.hipGetDeviceFlags(ctypes.byref(f))
    print("flag %s (rc %s, now %d): K=20 call %.1f us, call+sync %.1f us, idle sync %.1f us" % (flag, rc, f.value, med(call), med(lambda: (call(), torch.cuda.synchronize())), med(torch.cuda.synchronize)))
