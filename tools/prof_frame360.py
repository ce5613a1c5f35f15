import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
(rgbA, dA), _, _ = synth.make_pair(W, W // 2, seed=5)
st = Frame360Stages(RegisterPhotoICP())
for _ in range(3): st.frame_planes(dA, convention=2, angular_threshold=0.03)
