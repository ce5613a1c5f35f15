"""Writes n synthetic sequence frames as raw files for examples/odometry_replay.cpp: dump_sequence.py <dir> <n> <W> <H>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
d, n, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.makedirs(d, exist_ok=True)
for k in range(n):
    rgb, depth = synth.render(synth.trajectory_pose(k, 7), W, H, 7)
    rgb.tofile(os.path.join(d, "frame_%03d.rgb" % k))
    depth.tofile(os.path.join(d, "frame_%03d.depth" % k))
