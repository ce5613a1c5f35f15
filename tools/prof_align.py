import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
frames = [synth.render(synth.trajectory_pose(k, 7), 2048, 1024, 7) for k in range(2)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(*frames[0]); reg.setSourceFrame(*frames[1])
for _ in range(8): reg.alignFrames360(np.eye(4), 2)
