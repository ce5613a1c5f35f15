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
    out = subprocess.check_output([exe, str(seq), "3", "256", "128"], text=True)
    rows = [l.split() for l in out.strip().splitlines()]
    assert len(rows) == 2 and all(r[3] == "0" for r in rows)
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    frames = {k: synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(3)}
    poses, status, _ = align_sequence(reg, lambda k: frames[k], 0, 2, 2)
    for j, r in enumerate(rows):
        rel_t = np.array([float(x) for x in r[7:10]])
        assert np.allclose(rel_t, poses[j][:3, 3], atol=2e-5)
    # the frame loop inside the library (RegisterPhotoICP::alignSequence -> rgbd360_align360_batch): the same lines
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_sequence.py"), str(seq), "6", "256", "128"])
    pairwise = subprocess.check_output([exe, str(seq), "6", "256", "128"], text=True)
    batched = subprocess.check_output([exe, str(seq), "6", "256", "128", "--sequence"], text=True)
    assert len(pairwise.strip().splitlines()) == 5 and pairwise == batched
