import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ANG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03        # 0.03 merges the room into one region at full size; 0.03 * 1024 / W separates the walls
MIN_INLIERS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
WHICH = int(sys.argv[4]) if len(sys.argv) > 4 else 0             # 0 = frame A (identity orientation), 1 = frame B (rotated 10 degrees)
COLOUR = int(sys.argv[5]) if len(sys.argv) > 5 else 0            # 1: with the frame's colour image registered (colour descriptors + dominant colour)
pair = synth.make_pair(W, W // 2, seed=5, trans=0.3, rot_deg=10.0) if WHICH else synth.make_pair(W, W // 2, seed=5)
dA = pair[WHICH][1]
st = Frame360Stages(RegisterPhotoICP())
if COLOUR: st.set_color_image(pair[WHICH][0])
for _ in range(3): out = st.frame_planes(dA, convention=2, angular_threshold=ANG, min_inliers=MIN_INLIERS, max_curvature=0.0013, max_planes=4096)
print("planes", len(out["planes"]))
