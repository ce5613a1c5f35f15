"""Fixed cost of one forced-schedule call (the K steps of bench.py's `value` are one such call): wall time of the prepared C call for
several K, with and without the contract's torch.cuda.synchronize() behind it.  python tools/call_overhead.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
torch.cuda.init(); torch.zeros(1, device="cuda")
(rgbA, dA), (rgbB, dB), _ = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4); reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB); reg.sync()
reg.forced_iters(0, np.eye(4), 0, 50)
def med(f, n=41):
    t = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e6
res = {}
for K in (1, 2, 5, 10, 20, 40, 100):
    call = reg.forced_iters_call(0, np.eye(4), 0, K)
    a = med(call)
    b = med(lambda: (call(), torch.cuda.synchronize()))
    res[K] = (a, b)
    print("K=%3d  call %7.1f us  call+sync %7.1f us  per step %6.2f / %6.2f" % (K, a, b, a / K, b / K))
(a20, b20), (a40, b40) = res[20], res[40]
print("per-iteration slope %.2f us; fixed cost of a call %.1f us (call only) / %.1f us (with torch.cuda.synchronize)" % ((a40 - a20) / 20, 2 * a20 - a40, 2 * b20 - b40))
print("torch.cuda.synchronize() on an idle device: %.1f us" % med(lambda: torch.cuda.synchronize()))
