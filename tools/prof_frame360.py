import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
(rgbA, dA), _, _ = synth.make_pair(2048, 1024, seed=5)
st = Frame360Stages(RegisterPhotoICP())
for _ in range(3): st.frame_planes(dA, convention=2, angular_threshold=0.03)
