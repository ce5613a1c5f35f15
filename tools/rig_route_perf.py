"""Wall time of the per-sensor plane route of the 8-camera rig (examples/register_pair_planes.cpp in Python): per sensor
rgbd360_sensor_cloud (pinhole cloud, down-sampled by 2) + rgbd360_cloud_planes (bilateral filter, normal map, regions, rig frame),
then RegisterPbMap on the two frames' plane lists.  python tools/rig_route_perf.py"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rgbd360_amd import pbmap, synth
from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP

R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])
T_rig_sensor = [synth.make_pose(synth.rodrigues(np.array([1.0, 0, 0]), math.radians(45.0 * s)) @ R0, np.zeros(3)) for s in range(8)]
T_w1 = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
M = synth.default_motion(17, 0.08, 3.0)
frames = [[synth.render_pinhole(T @ T_rig_sensor[s], 320, 240, 5)[1] for s in range(8)] for T in (T_w1, T_w1 @ M)]
st = Frame360Stages(RegisterPhotoICP())


def frame_planes(depths, timing=None):
    planes = []
    for s in range(8):
        t0 = time.perf_counter()
        cloud = st.sensor_cloud(depths[s], 2, 0.3, 10.0)
        t1 = time.perf_counter()
        planes += st.cloud_planes(cloud, 120, 160, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, 0, T_rig_sensor[s])
        t2 = time.perf_counter()
        if timing is not None:
            timing[0] += t1 - t0
            timing[1] += t2 - t1
    return planes


def frame_planes_fused(depths):
    planes = []
    for s in range(8):
        planes += st.sensor_planes(depths[s], 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, T_rig_sensor[s])
    return planes


p1, p2 = frame_planes(frames[0]), frame_planes(frames[1])
assert [(p["root"], p["count"]) for p in frame_planes_fused(frames[0])] == [(p["root"], p["count"]) for p in p1]
t0 = time.perf_counter()
for _ in range(10):
    frame_planes_fused(frames[0])
print("fused rgbd360_sensor_planes, 8 sensors: %.2f ms per frame" % ((time.perf_counter() - t0) / 10 * 1e3))
n = 10
tm = [0.0, 0.0]
t0 = time.perf_counter()
for _ in range(n):
    p1 = frame_planes(frames[0], tm)
t_frame = (time.perf_counter() - t0) / n
# one context (stream + buffers) and one host thread per sensor: the per-sensor chains are launch-latency bound, the eight of a
# frame are independent (Frame360.h:489-503 runs them on 8 OpenMP threads too) -- ctypes releases the GIL during the calls
from concurrent.futures import ThreadPoolExecutor
stages = [Frame360Stages(RegisterPhotoICP()) for _ in range(8)]
pool = ThreadPoolExecutor(8)


def frame_planes_threads(depths):
    def one(s):
        return stages[s].sensor_planes(depths[s], 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, T_rig_sensor[s])
    out = []
    for ps in pool.map(one, range(8)):
        out += ps
    return out


assert [(p["root"], p["count"]) for p in frame_planes_threads(frames[0])] == [(p["root"], p["count"]) for p in p1]
t0 = time.perf_counter()
for _ in range(10):
    frame_planes_threads(frames[0])
print("8 contexts on 8 host threads (rgbd360_sensor_planes each): %.2f ms per frame" % ((time.perf_counter() - t0) / 10 * 1e3))
t0 = time.perf_counter()
m1, m2 = pbmap.merge_planes(p1), pbmap.merge_planes(p2)
t_merge = (time.perf_counter() - t0) / 2
print("Frame360::mergePlanes: %d -> %d and %d -> %d planes, %.2f ms per frame through the Python mirror" % (len(p1), len(m1), len(p2), len(m2), t_merge * 1e3))
regm = pbmap.RegisterRGBD360(odometry_config=True)
goodm = regm.RegisterPbMap(m1, m2, 25, pbmap.ODOMETRY_6DoF)
print("RegisterPbMap on the merged planes: good %s, %d matched, pose error vs the rig motion %.2e rad %.2e m" % (
    goodm, len(regm.getMatchedPlanes()), *synth.pose_error(regm.getPose(), M)))
reg = pbmap.RegisterRGBD360(odometry_config=True)
t0 = time.perf_counter()
good = reg.RegisterPbMap(p1, p2, 25, pbmap.ODOMETRY_6DoF)
t_match = time.perf_counter() - t0
print("8 sensors x 320x240 -> 160x120 clouds: %.2f ms per frame (sensor clouds %.2f ms, filter + normals + regions + plane lists %.2f ms); %d / %d planes" % (
    t_frame * 1e3, tm[0] / n * 1e3, tm[1] / n * 1e3, len(p1), len(p2)))
print("RegisterPbMap (ODOMETRY_6DoF): %.2f ms through the Python mirror, good %s, %d matched, pose error vs the rig motion %.2e rad %.2e m" % (
    t_match * 1e3, good, len(reg.getMatchedPlanes()), *synth.pose_error(reg.getPose(), M)))
