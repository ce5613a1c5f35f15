"""The C++ adapter (include/rgbd360/RegisterPhotoICP.hpp) and the app-shaped driver that replays the call sequence of
the reference's OdometryRGBD360.cpp through it."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(out_dir):
    from rgbd360_amd import build
    lib = build.build()
    exe = os.path.join(str(out_dir), "odometry_replay")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "odometry_replay.cpp"), "-L" + os.path.dirname(lib), "-lrgbd360_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    return exe


def test_adapter_compiles_against_the_c_abi(tmp_path):
    exe = build_example(tmp_path)
    # no frames: the driver must fail on its own I/O check (exit 3), i.e. it linked and started
    assert subprocess.call([exe, str(tmp_path / "missing"), "2", "8", "8"]) == 3


@pytest.mark.gpu
def test_odometry_replay_matches_python_host(tmp_path, hip_lib):
    from rgbd360_amd import synth
    from rgbd360_amd.batch import align_sequence
    from rgbd360_amd.register import RegisterPhotoICP
    exe = build_example(tmp_path)
    seq = tmp_path / "seq"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_sequence.py"), str(seq), "3", "256", "128"])
    run = subprocess.run([exe, str(seq), "3", "256", "128"], text=True, capture_output=True, check=True)
    rows = [l.split() for l in run.stdout.strip().splitlines()]
    assert len(rows) == 2 and all(r[3] == "0" for r in rows)
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    frames = {k: synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(3)}
    poses, status, _ = align_sequence(reg, lambda k: frames[k], 0, 2, 2)
    for j, r in enumerate(rows):
        rel_t = np.array([float(x) for x in r[7:10]])
        assert np.allclose(rel_t, poses[j][:3, 3], atol=2e-5)
    # RegisterPhotoICP::calcEntropy (RPI.h:4789-4797) of the last pair, C++ adapter vs the Python mirror vs the formula on the Hessian
    ent = [float(l.split()[2]) for l in run.stderr.splitlines() if l.startswith("entropy ")]
    H = reg.getHessian().astype(np.float64)
    want = 0.5 * (6 * (1 + np.log(2 * 3.14159265359)) + np.log(np.linalg.det(np.linalg.inv(H))))
    assert len(ent) == 2 and abs(ent[1] - reg.calcEntropy()) < 1e-3 and abs(reg.calcEntropy() - want) < 1e-6 * max(1.0, abs(want)), (ent, reg.calcEntropy(), want)
    # the frame loop inside the library (RegisterPhotoICP::alignSequence -> rgbd360_align360_batch): the same lines
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_sequence.py"), str(seq), "6", "256", "128"])
    pairwise = subprocess.check_output([exe, str(seq), "6", "256", "128"], text=True)
    batched = subprocess.check_output([exe, str(seq), "6", "256", "128", "--sequence"], text=True)
    assert len(pairwise.strip().splitlines()) == 5 and pairwise == batched
    # and sharded over the GPUs of the node from this one C++ process (rgbd360_multi_*; this box has one device)
    multi = subprocess.check_output([exe, str(seq), "6", "256", "128", "--multi", "1"], text=True)
    assert multi == batched
    forced = subprocess.check_output([exe, str(seq), "6", "256", "128", "--multi", "1"], text=True, env=dict(os.environ, RGBD360_FORCE_RCCL="1"))
    forced = "".join(l + "\n" for l in forced.splitlines() if l.startswith("pair "))      # RCCL prints a version banner on stdout
    assert forced == batched                        # the rows went through ncclAllGather (one rank) and came back unchanged


def test_register_rgbd360_adapter_matches_python_mirror(tmp_path):
    """include/rgbd360/RegisterRGBD360.hpp (reference RegisterRGBD360.h:47-338 surface) through a small driver: same
    matches, pose, entropy and matched area as rgbd360_amd.pbmap.RegisterRGBD360.  Host-only code: runs without a GPU."""
    import math
    from rgbd360_amd import build, pbmap, synth
    from tests.test_pbmap_register import room_planes
    lib = build.build()
    exe = os.path.join(str(tmp_path), "pbmap_register_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "pbmap_register_demo.cpp"), "-L" + os.path.dirname(lib), "-lrgbd360_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ synth.default_motion(21, 0.06, 2.0)
    rng = np.random.default_rng(3)
    frames = [room_planes(T_wA, rng, 0.003, 0.003), room_planes(T_wB, rng, 0.003, 0.003)]
    txt = tmp_path / "planes.txt"
    with open(txt, "w") as f:
        for k, planes in enumerate(frames):
            if k:
                f.write("--\n")
            for p in planes:
                vals = [*p["centroid"], *p["normal"], p["d"], p["curvature"], p["area"], p["elongation"]]
                f.write(" ".join(repr(float(np.float32(v))) for v in vals) + "\n")
    out = subprocess.check_output([exe, str(txt), "25", "2", "1"], text=True).strip().splitlines()
    reg = pbmap.RegisterRGBD360(odometry_config=True)
    assert reg.RegisterPbMap(frames[0], frames[1], 25, pbmap.ODOMETRY_6DoF)
    assert out[0] == "status 0 good 1"
    assert out[1].split()[1:] == [f"{i}:{j}" for i, j in sorted(reg.getMatchedPlanes().items())]
    pose = np.array([[float(x) for x in l.split()] for l in out[2:6]])
    assert np.abs(pose - reg.getPose()).max() < 1e-6
    assert abs(float(out[6].split()[1]) - reg.calcEntropy()) < 1e-3
    assert abs(float(out[7].split()[1]) - reg.getAreaMatched()) < 1e-3
    assert synth.pose_error(pose, np.linalg.inv(T_wA) @ T_wB)[0] < math.radians(0.5)
    assert subprocess.call([exe, str(tmp_path / "missing"), "0", "0", "0"]) == 3


@pytest.mark.gpu
def test_odometry_replay_with_pbmap_initial_guess(tmp_path, hip_lib):
    """--pbmap: planes of both frames on the device (rgbd360::segmentPlanes), RegisterPbMap in ODOMETRY_6DoF, the pose seeds
    alignFrames360 (KFsphere_SLAM.cpp:149).  The plane pose is already within a few mm of the motion, and the dense result
    agrees with the run that starts from the identity."""
    from rgbd360_amd import synth
    exe = build_example(tmp_path)
    seq = tmp_path / "seq"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_sequence.py"), str(seq), "3", "512", "256"])
    plain = subprocess.check_output([exe, str(seq), "3", "512", "256"], text=True).strip().splitlines()
    seeded = subprocess.check_output([exe, str(seq), "3", "512", "256", "--pbmap"], text=True).strip().splitlines()
    pb = [l.split() for l in seeded if l.startswith("pbmap")]
    pairs = [l.split() for l in seeded if l.startswith("pair")]
    assert len(pb) == 2 and len(pairs) == 2 and len(plain) == 2
    for j in range(2):
        assert pb[j][3] == "1" and int(pb[j][5]) >= 4, pb[j]
        T_gt = np.linalg.inv(synth.trajectory_pose(j, 7)) @ synth.trajectory_pose(j + 1, 7)
        t_pb = np.array([float(x) for x in pb[j][7:10]])
        assert np.linalg.norm(t_pb - T_gt[:3, 3]) < 0.01, (t_pb, T_gt[:3, 3])
        t_seeded = np.array([float(x) for x in pairs[j][7:10]])
        t_plain = np.array([float(x) for x in plain[j].split()[7:10]])
        assert pairs[j][3] == "0" and np.linalg.norm(t_seeded - T_gt[:3, 3]) < 5e-3
        assert np.linalg.norm(t_seeded - t_plain) < 5e-3
    # the one-call form (rgbd360::RegisterFrames): same relative translations as the explicit sequence, links valid
    linked = [l.split() for l in subprocess.check_output([exe, str(seq), "3", "512", "256", "--link"], text=True).strip().splitlines()]
    assert len(linked) == 2
    for j in range(2):
        assert linked[j][3] == "1" and int(linked[j][5]) >= 4, linked[j]
        assert np.allclose([float(x) for x in linked[j][7:10]], [float(x) for x in pairs[j][7:10]], atol=1e-5)


def _write_rig_frame(path, T_w_rig, T_rig_sensor, seed):
    """An 8-sensor frame of the synthetic room in the reference's sphere_images_%d.bin layout (Frame360::serialize)."""
    import struct
    from rgbd360_amd import synth
    with open(path, "wb") as f:
        f.write(b"\x16\x00\x00\x00\x00\x00\x00\x00serialization::archive" + b"\x00" * 15)
        for s in range(8):
            rgb, dep = synth.render_pinhole(T_w_rig @ T_rig_sensor[s], 320, 240, seed)
            f.write(struct.pack("<iiQQ", 320, 240, 3, 16) + np.ascontiguousarray(rgb).tobytes())
            f.write(struct.pack("<iiQQ", 320, 240, 2, 2) + np.ascontiguousarray(dep, np.uint16).tobytes())
        f.write(struct.pack("<iiQQ", 0, 0, 0, 0))


def build_pair_planes(out_dir):
    from rgbd360_amd import build
    lib = build.build()
    exe = os.path.join(str(out_dir), "register_pair_planes")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "register_pair_planes.cpp"), "-L" + os.path.dirname(lib), "-lrgbd360_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    return exe


def test_register_pair_planes_compiles_and_checks_its_inputs(tmp_path):
    exe = build_pair_planes(tmp_path)
    assert subprocess.call([exe, str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path)]) == 3     # no such frames
    assert subprocess.call([exe]) == 2


@pytest.mark.gpu
def test_register_pair_planes_on_a_synthetic_rig(tmp_path, hip_lib):
    """The plane side of RegisterPairRGBD360.cpp on the reference's file formats, end to end on the device: two 8-sensor frames of
    the synthetic room written as sphere_images_%d.bin + Rt_0N.txt -> per sensor cloud (pinhole, down-sampled by 2) -> bilateral
    filter -> normal map -> regions -> rig frame -> RegisterPbMap.  The rig moved by 8 cm / 3 degrees: the plane pose recovers it."""
    import math
    from rgbd360_amd import synth
    exe = build_pair_planes(tmp_path)
    R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])          # sensor axes in the rig frame: image down = -x (x is up)
    T_rig_sensor = [synth.make_pose(synth.rodrigues(np.array([1.0, 0, 0]), math.radians(45.0 * s)) @ R0, np.zeros(3)) for s in range(8)]
    T_w1 = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    M = synth.default_motion(17, 0.08, 3.0)
    _write_rig_frame(tmp_path / "f1.bin", T_w1, T_rig_sensor, 5)
    _write_rig_frame(tmp_path / "f2.bin", T_w1 @ M, T_rig_sensor, 5)
    for s in range(8):
        np.savetxt(tmp_path / ("Rt_0%d.txt" % (s + 1)), T_rig_sensor[s])
    out = subprocess.check_output([exe, str(tmp_path / "f1.bin"), str(tmp_path / "f2.bin"), str(tmp_path), "2"], text=True).strip().splitlines()
    n1, n2 = (int(x) for x in out[0].split()[1:3])
    pieces = [int(x.strip(')')) for x in out[0].split()[4:6]]
    assert 5 <= n1 <= pieces[0] and 5 <= n2 <= pieces[1] and pieces[0] >= 8, out[0]        # co-planar pieces were merged
    st = out[1].split()
    assert st[1] == "0" and st[3] == "1" and int(st[5]) >= 5, out[1]
    T = np.array([[float(x) for x in l.split()] for l in out[2:6]])
    rot, tr = synth.pose_error(T, M)
    assert rot < math.radians(0.3) and tr < 0.015, (rot, tr)


def build_frame360_pair(out_dir):
    from rgbd360_amd import build
    lib = build.build()
    exe = os.path.join(str(out_dir), "frame360_pair")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "frame360_pair.cpp"), "-L" + os.path.dirname(lib), "-lrgbd360_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    return exe


def test_frame360_adapter_compiles_and_checks_its_inputs(tmp_path):
    """include/rgbd360/Frame360.hpp (Frame360 / Calib360 with the reference's member and method names, Frame360.h:93-1148, Calib360.h:44-134)
    compiles warning-free in the call shape of RegisterPairRGBD360.cpp:60-90; missing calibration files and frames are reported."""
    exe = build_frame360_pair(tmp_path)
    assert subprocess.call([exe]) == 2
    assert subprocess.call([exe, str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path)]) == 3         # no Rt_0N.txt there
    for s in range(8):
        np.savetxt(tmp_path / ("Rt_0%d.txt" % (s + 1)), np.eye(4))
    assert subprocess.call([exe, str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path)], stderr=subprocess.DEVNULL) == 3     # no such frames


@pytest.mark.gpu
def test_frame360_adapter_on_a_synthetic_rig(tmp_path, hip_lib):
    """RegisterPairRGBD360.cpp's sequence through the adapter classes, end to end on the device: loadFrame -> stitchSphericalImage ->
    getPlanes (eight getPlanesSensor on eight contexts, groupPlanes, mergePlanes) for two frames of the synthetic room, RegisterPbMap on
    the Frame360 objects, then the dense alignment of the two stitched panoramas from the plane pose.  The rig moved by 8 cm / 3 degrees."""
    import math
    from rgbd360_amd import synth
    exe = build_frame360_pair(tmp_path)
    R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])
    T_rig_sensor = [synth.make_pose(synth.rodrigues(np.array([1.0, 0, 0]), math.radians(45.0 * s)) @ R0, np.zeros(3)) for s in range(8)]
    T_w1 = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    M = synth.default_motion(17, 0.08, 3.0)
    _write_rig_frame(tmp_path / "f1.bin", T_w1, T_rig_sensor, 5)
    _write_rig_frame(tmp_path / "f2.bin", T_w1 @ M, T_rig_sensor, 5)
    for s in range(8):
        np.savetxt(tmp_path / ("Rt_0%d.txt" % (s + 1)), T_rig_sensor[s])
    plain = subprocess.check_output([exe, str(tmp_path / "f1.bin"), str(tmp_path / "f2.bin"), str(tmp_path), "2"], text=True)
    # with an intrinsics directory the frames go through undistort() first (RegisterPairRGBD360.cpp:69): identity models (multiplier 1 in
    # every slice of every bin) leave every number of the run unchanged -- the float-metres route of clouds and planes equals the millimetre one
    import struct
    one = struct.pack("<d", 10.0) + struct.pack("<i", 5) + struct.pack("<d", 2.0)
    for vec in (np.full(5, 100.0), np.ones(5), np.ones(5), np.ones(5)):
        one += struct.pack("<iii", 4, 5, 1) + vec.astype(np.float32).tobytes()
    os.makedirs(tmp_path / "intr")
    for s in range(8):
        with open(tmp_path / "intr" / ("distortion_model%d" % (s + 1)), "wb") as f:
            f.write(b"DiscreteDepthDistortionModel v01\n" + struct.pack("<iiii", 640, 480, 8, 6) + struct.pack("<d", 2.0) + struct.pack("<ii", 80, 80) + one * 6400)
    undist = subprocess.check_output([exe, str(tmp_path / "f1.bin"), str(tmp_path / "f2.bin"), str(tmp_path), "2", str(tmp_path / "intr")], text=True)
    assert undist == plain
    out = plain.strip().splitlines()
    c = out[0].replace(",", "").split()
    assert int(c[2]) == 8 * 160 * 120 and int(c[5]) > 1000 and int(c[7]) > 0.9 * int(c[5]), out[0]      # buildSphereCloud: the room's walls, in the planes' frame
    out = out[1:]
    w = out[0].split()
    n1, n2 = int(w[1]), int(w[2])
    pieces = [int(w[4]), int(w[5].strip(")"))]
    assert 5 <= n1 < pieces[0] and 5 <= n2 < pieces[1], out[0]                 # pieces of one wall seen by several sensors were pooled
    assert float(w[8]) > 20.0 and float(w[9]) > 20.0 and 30 < int(w[12]) < 230, out[0]     # planar area (m2) of the room's walls; mean grey level
    st = out[1].split()
    assert st[1] == "0" and st[3] == "1" and int(st[5]) >= 4, out[1]
    Tp = np.array([[float(x) for x in l.split()] for l in out[2:6]])
    rot, tr = synth.pose_error(Tp, M)
    assert rot < math.radians(0.5) and tr < 0.02, (rot, tr)
    assert out[6].split()[2] == "0", out[6]
    Td = np.array([[float(x) for x in l.split()] for l in out[7:11]])
    # the stitched panorama lives in the rig frame rotated like the sphere images: the dense pose is compared through its magnitude
    ang = math.acos(max(-1.0, min(1.0, (np.trace(Td[:3, :3]) - 1) / 2)))
    assert abs(ang - math.radians(3.0)) < math.radians(0.5) and abs(np.linalg.norm(Td[:3, 3]) - 0.08) < 0.03, (ang, Td[:3, 3])


def build_stereo_planes(out_dir):
    from rgbd360_amd import build
    lib = build.build()
    exe = os.path.join(str(out_dir), "frame360_stereo_planes")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "frame360_stereo_planes.cpp"), "-L" + os.path.dirname(lib), "-lrgbd360_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    return exe


def test_frame360_stereo_adapter_compiles_and_checks_its_inputs(tmp_path):
    exe = build_stereo_planes(tmp_path)
    assert subprocess.call([exe]) == 2
    assert subprocess.call([exe, str(tmp_path / "none.raw")], stderr=subprocess.DEVNULL) == 3
    (tmp_path / "short.raw").write_bytes(b"\x10\x00\x20\x00" + b"\x00" * 100)            # 16 x 32 announced, 25 floats present
    assert subprocess.call([exe, str(tmp_path / "short.raw")], stderr=subprocess.DEVNULL) == 3


@pytest.mark.gpu
def test_frame360_stereo_adapter_finds_the_walls(tmp_path, hip_lib):
    """Frame360_stereo through its adapter (loadDepth -> buildSphereCloud -> getPlanesStereo, Frame360_stereo.h:268-311, 454-512, 847-980):
    the range panorama of the synthetic room in the stereo camera's geometry (2048 x 665, rows from -61 to +56 degrees of elevation),
    written in the sensor's raw layout (uint16 height, width, then the image column by column).  The walls come back as planes with
    axis-aligned normals at their true distances."""
    import struct
    from oracle import oracle as O
    from rgbd360_amd import synth
    exe = build_stereo_planes(tmp_path)
    H, W = 665, 2048
    rays = O.sphere_cloud(np.ones((H, W), np.float32), 1).astype(np.float64)            # unit rays of every pixel (the oracle's restatement of :470-490)
    cam = np.asarray(synth.CAM_A, float)
    lo, hi = np.asarray(synth.ROOM_LO, float), np.asarray(synth.ROOM_HI, float)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(rays > 0, (hi - cam) / rays, np.where(rays < 0, (lo - cam) / rays, np.inf))
    depth = t.min(axis=1).reshape(H, W).astype(np.float32)
    with open(tmp_path / "room.raw", "wb") as f:
        f.write(struct.pack("<HH", H, W) + np.ascontiguousarray(depth.T).tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "room.raw")], text=True).strip().splitlines()
    head = out[0].replace(",", "").split()
    assert head[1] == str(H) and head[3] == str(W) and int(head[6]) == H * W and int(head[8]) >= 5, out[0]
    walls = {}                                   # per wall direction the region with the most inliers
    for l in out[1:]:
        v = [float(x) for x in l.split()]
        n = np.array(v[1:4])
        ax = int(np.argmax(np.abs(n)))
        key = (ax, int(np.sign(n[ax])))
        if abs(n[ax]) > 0.9999 and v[0] > walls.get(key, (0, 0.0))[0]:
            walls[key] = (v[0], v[4])
    assert len(walls) >= 4 and sum(c for c, _ in walls.values()) > 0.35 * H * W, (walls, out[:12])
    for (ax, sgn), (cnt, d) in walls.items():   # normals point at the camera: a wall at hi has normal -e_ax, distance hi - cam
        want = (hi[ax] - cam[ax]) if sgn < 0 else (cam[ax] - lo[ax])
        assert abs(d - want) < 0.02, (ax, sgn, cnt, d, want)


# ---- the reference's own signatures (Eigen / cv::Mat) on the adapter: compiled against mock headers ------------------------------
MOCK = os.path.join(ROOT, "tests", "mock_headers")
REF = "/root/reference"

_HARNESS_HEAD = r'''
#include <iostream>
#include <map>
#include <utility>
#include <vector>
#include <RegisterPhotoICP.h>      // include/rgbd360/compat/RegisterPhotoICP.h: the reference's header name, class in the global namespace
#ifndef RGBD360_HAVE_EIGEN
#error "the mock Eigen headers were not picked up"
#endif
#ifndef RGBD360_HAVE_OPENCV
#error "the mock OpenCV headers were not picked up"
#endif
using namespace std;
namespace pcl { inline double getTime() { return 0.0; } }
struct Frame360 { cv::Mat sphereRGB, sphereDepth; };
struct Map360 {
    std::vector<Frame360*> vpSpheres;
    std::map<unsigned, std::map<unsigned, std::pair<Eigen::Matrix4f, Eigen::Matrix<float,6,6> > > > mmConnectionKFs;
};
struct Optimizer { void addEdge(int, int, const Eigen::Matrix<double,4,4>&, const Eigen::Matrix<double,6,6>&) {} };
'''


def _compile(tmp_path, name, source, link=False):
    from rgbd360_amd import build
    src = tmp_path / (name + ".cpp")
    src.write_text(source)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror=return-type", "-I" + MOCK, "-I" + os.path.join(ROOT, "include", "rgbd360", "compat"),
           "-I" + os.path.join(ROOT, "include"), str(src)]
    if link:
        lib = build.build()
        exe = str(tmp_path / name)
        subprocess.check_call(cmd + ["-L" + os.path.dirname(lib), "-lrgbd360_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
        return exe
    subprocess.check_call(cmd + ["-c", "-o", str(tmp_path / (name + ".o"))])
    return None


def test_adapter_has_the_reference_signatures_with_eigen_and_opencv(tmp_path):
    """With Eigen / OpenCV on the include path (here: the mock headers of tests/mock_headers) the adapter's public surface is the
    reference's (RPI.h:171-199, 273-288, 480-516, 4254, 4519): cv::Mat frames, by-value Eigen::Matrix4f guesses with the reference's
    defaults, Eigen return types that multiply and convert like the call sites need, public num_iterations / pyramids / LUT.
    The RGBD360_HAVE_EIGEN / RGBD360_HAVE_OPENCV branches of the header are compiled AND linked here (no GPU needed for that)."""
    body = _HARNESS_HEAD + r'''
#include <type_traits>
static_assert(std::is_same<decltype(std::declval<RegisterPhotoICP&>().getOptimalPose()), Eigen::Matrix4f>::value, "RPI.h:273");
static_assert(std::is_same<decltype(std::declval<RegisterPhotoICP&>().getHessian()), Eigen::Matrix<float,6,6> >::value, "RPI.h:279");
static_assert(std::is_same<decltype(std::declval<RegisterPhotoICP&>().getGradient()), Eigen::Matrix<float,6,1> >::value, "RPI.h:285");
static_assert(std::is_same<decltype(RegisterPhotoICP::num_iterations), std::vector<int> >::value, "RPI.h:177");
static_assert(std::is_same<decltype(RegisterPhotoICP::graySrcPyr), std::vector<cv::Mat> >::value, "RPI.h:198");
static_assert(std::is_same<decltype(RegisterPhotoICP::LUT_xyz_sphere), std::vector<Eigen::Vector3f> >::value, "RPI.h:171");
int main(int argc, char**) {
    if (argc < 2) return 3;                       // (never run without a GPU: this is a compile + link check)
    Frame360 a, b;
    a.sphereRGB.create(8, 16, CV_8UC3); a.sphereDepth.create(8, 16, CV_16UC1);
    b.sphereRGB.create(8, 16, CV_8UC3); b.sphereDepth.create(8, 16, CV_16UC1);
    RegisterPhotoICP align360;
    align360.setNumPyr(1);
    align360.setTargetFrame(a.sphereRGB, a.sphereDepth);
    align360.setSourceFrame(b.sphereRGB, b.sphereDepth);
    align360.alignFrames360();                                            // the reference's defaults (RPI.h:4519)
    align360.alignFrames360(Eigen::Matrix4f::Identity(), RegisterPhotoICP::PHOTO_DEPTH, 2);
    Eigen::Matrix3f K = Eigen::Matrix3f::Identity();
    align360.setCameraMatrix(K);
    align360.alignFrames(Eigen::Matrix4f::Identity(), RegisterPhotoICP::PHOTO_DEPTH);
    Eigen::Matrix4f T = align360.getOptimalPose();
    Eigen::Matrix<double,6,6> info = align360.getHessian().cast<double>();
    Eigen::Matrix<float,6,1> g = align360.getGradient();
    float entropy = align360.calcEntropy();                                // RPI.h:4789
    (void)entropy;
    align360.downloadPyramids();
    align360.downloadLUT(0);
    Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
    const bool ok = rgbd360::Register(a, b, guess);                        // bool Register(Frame360&, Frame360&, Eigen::Matrix4f&)
    std::cout << T << "\n" << info(0, 0) << " " << g(0) << " " << align360.num_iterations.size() << " " << align360.graySrcPyr.size() << " " << ok << std::endl;
    return 0;
}
'''
    exe = _compile(tmp_path, "eigen_surface", body, link=True)
    assert subprocess.call([exe]) == 3


@pytest.mark.skipif(not os.path.isdir(REF), reason="reads the reference's call sites in place (build container only)")
def test_reference_call_sites_compile_character_for_character(tmp_path):
    """The north star's "drops into OdometryRGBD360 ... unchanged": the lines of the reference's applications that use
    RegisterPhotoICP are read IN PLACE from /root/reference (nothing of them is stored in this repository), pasted unmodified into
    function bodies whose parameters stand for the surrounding variables, and compiled against the adapter with the mock Eigen /
    OpenCV headers: OdometryRGBD360.cpp:189-193, LoopClosure360.h:306-323, KFsphere_SLAM.cpp:144-153 and :399-402."""
    def lines(path, lo, hi):
        with open(os.path.join(REF, path), encoding="latin-1") as f:
            src = f.read().splitlines()
        out = "\n".join(src[lo - 1:hi])
        assert "align360" in out
        return out
    body = _HARNESS_HEAD + "\nvoid odometry_site(Frame360* frame360_1, Frame360* frame360_2, RegisterPhotoICP& align360, Eigen::Matrix4f& rigidTransf_dense, const Eigen::Matrix4f& rotOffset) {\n"
    body += lines("Registration/OdometryRGBD360.cpp", 189, 193)
    body += "\n}\nvoid loop_closure_site(Map360& Map, Frame360* newKF, unsigned compareLocalIdx, unsigned newFrameID, Eigen::Matrix4f rotOffset, Eigen::Matrix4f relativePose,\n"
    body += "                       Optimizer& optimizer, std::map<unsigned, std::map<unsigned, float> >& connectionsLC) {\n"
    body += lines("include/LoopClosure360.h", 306, 323)
    body += "\n}\nvoid keyframe_site(Map360& Map, unsigned nearestKF, Frame360* candidateKF, std::pair<Eigen::Matrix4f, Eigen::Matrix<float,6,6> > candidateKF_connection,\n"
    body += "                   RegisterPhotoICP& align360, Eigen::Matrix4f rotOffset) {\n"
    body += lines("SLAM/KFsphere_SLAM.cpp", 144, 153)
    body += "\n}\nvoid candidate_site(Frame360*& candidateKF, Frame360* frame360, std::pair<Eigen::Matrix4f, Eigen::Matrix<float,6,6> >& candidateKF_connection,\n"
    body += "                    float& candidateKF_sso, RegisterPhotoICP& align360, Eigen::Matrix4f rigidTransf_dense) {\n"
    body += lines("SLAM/KFsphere_SLAM.cpp", 399, 402)
    body += "\n}\n"
    _compile(tmp_path, "reference_call_sites", body)
