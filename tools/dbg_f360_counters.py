"""Join / pointer-chase counters of the connected-component passes (unions, find hops, longest walk), printed by a library
built with -DF360_DEBUG_COUNTERS:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -fno-slp-vectorize \
        -DF360_DEBUG_COUNTERS -o rgbd360_amd/lib/librgbd360_hip_dbg.so rgbd360_amd/csrc/rgbd360_api.hip
  python tools/dbg_f360_counters.py [width]"""
import os, sys
sys.path.insert(0, os.getcwd())
import rgbd360_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "librgbd360_hip_dbg.so")
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
(rgbA, dA), _, _ = synth.make_pair(W, W // 2, seed=5)
st = Frame360Stages(RegisterPhotoICP())
for _ in range(2): st.frame_planes(dA, convention=2, angular_threshold=0.03)
