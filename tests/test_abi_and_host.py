"""CPU tests of the boundary: the C-ABI library loads and exports every declared symbol; host-side logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from rgbd360_amd import build
    path = build.build()          # hipcc cross-compiles for gfx950 without a GPU
    return C.CDLL(path)


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(rgbd360_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol(built_lib):
    from rgbd360_amd import _lib
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [s for s in decl if not hasattr(built_lib, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == decl      # the ctypes binding covers the whole header and nothing else


def test_no_torch_types_in_the_abi():
    for hdr in ("rgbd360_hip.h", "rgbd360_hip_diag.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        assert 'extern "C"' in txt
        code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)          # declarations only, comments stripped
        for banned in ("torch", "at::", "std::", "Tensor", "hipStream_t", "template"):
            assert banned not in code, (hdr, banned)


def test_measurement_entry_points_live_in_the_diag_header():
    """The product ABI (rgbd360_hip.h) carries no timers / self-tests / forced schedules: those are rgbd360_hip_diag.h."""
    main = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rgbd360_hip.h")).read(), flags=re.S)
    for name in ("rgbd360_time_eval_kernel", "rgbd360_time_solve_kernel", "rgbd360_selftest_math", "rgbd360_forced_iters"):
        assert name not in main, name


def test_default_params_match_reference_defaults(built_lib):
    from rgbd360_amd import _lib
    L = _lib.load()
    p = _lib.Params()
    L.rgbd360_default_params(C.byref(p))
    # RegisterPhotoICP.h:201-221, 4593-4595
    assert (p.n_pyr, p.max_iters, p.mask_seams) == (4, 10, 1)
    assert p.min_depth == np.float32(0.3) and p.max_depth == np.float32(6.0)
    assert p.sigma_photo == np.float32(6.0 / 255) and p.sigma_depth == np.float32(0.2)
    assert p.thres_sal_photo == np.float32(0.01) and p.thres_sal_depth == np.float32(0.01)
    assert p.tol_residual == np.float32(1e-3) and p.tol_update == np.float32(1e-4)


def test_product_path_fails_loudly_without_a_gpu(built_lib):
    """No CPU fallback: on a machine without a HIP device context creation must fail, not degrade."""
    from rgbd360_amd import _lib
    from rgbd360_amd.register import RegisterPhotoICP, Rgbd360Error
    L = _lib.load()
    if L.rgbd360_device_count() > 0:
        pytest.skip("a GPU is visible here")
    reg = RegisterPhotoICP()
    rgb = np.zeros((32, 64, 3), np.uint8)
    d = np.zeros((32, 64), np.uint16)
    with pytest.raises(Rgbd360Error):
        reg.setTargetFrame(rgb, d)


def test_product_package_never_imports_the_oracle():
    """The oracle is a checker only: no import, include, link or dlopen of anything under oracle/ from the product."""
    pkg = os.path.join(ROOT, "rgbd360_amd")
    banned = [r"^\s*(from|import)\s+oracle\b", r"import_module\([^)]*oracle", r"#\s*include[^\n]*oracle", r"liboracle",
              r"oracle_[a-z0-9_]+\s*\(", r"CDLL\([^)]*oracle"]
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                for pat in banned:
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dirpath, fn), pat)
    # and the build of the product does not compile or link the oracle's sources
    code = "\n".join(l for l in open(os.path.join(pkg, "build.py")).read().splitlines() if not l.lstrip().startswith("#"))
    assert "oracle" not in code


def test_host_argument_validation():
    from rgbd360_amd.register import RegisterPhotoICP, Rgbd360Error
    reg = RegisterPhotoICP()
    with pytest.raises(Rgbd360Error):
        reg.setTargetFrame(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint16))          # not HxWx3
    with pytest.raises(Rgbd360Error):
        reg.setTargetFrame(np.zeros((8, 8, 3), np.uint8), np.zeros((8, 8), np.float64))       # bad depth dtype
    with pytest.raises(Rgbd360Error):
        reg.setTargetFrame(np.zeros((8, 8, 3), np.uint8), np.zeros((4, 8), np.uint16))        # size mismatch
    with pytest.raises(Rgbd360Error):
        reg.setVisualization(True)
    assert reg.nPyrLevels == 4
    reg.setNumPyr(5)
    assert reg.nPyrLevels == 5


def test_pose_layout_helpers():
    from rgbd360_amd.register import pose_from_cm, pose_to_cm
    T = np.arange(16, dtype=np.float32).reshape(4, 4)
    cm = pose_to_cm(T)
    assert cm[12] == T[0, 3] and cm[1] == T[1, 0]       # column-major like Eigen::Matrix4f::data()
    assert np.array_equal(pose_from_cm(cm), T)


def test_bin_loader_reads_the_boost_archive_layout(tmp_path):
    """rgbd360_load_frame_bin (host-only) on a file written in the layout of Frame360::serialize / cvmat_serialization.h."""
    import struct
    from rgbd360_amd.register import Rgbd360Error, load_frame_bin
    rng = np.random.default_rng(0)
    rows, cols = 12, 16
    rgb = rng.integers(0, 256, size=(8, rows, cols, 3), dtype=np.uint8)
    dep = rng.integers(0, 5000, size=(8, rows, cols)).astype(np.uint16)
    path = tmp_path / "sphere_images_0.bin"
    with open(path, "wb") as f:
        f.write(b"\x16\x00\x00\x00\x00\x00\x00\x00serialization::archive" + b"\x00" * 15)     # 45-byte archive header
        for s in range(8):
            f.write(struct.pack("<iiQQ", cols, rows, 3, 16) + rgb[s].tobytes())
            f.write(struct.pack("<iiQQ", cols, rows, 2, 2) + dep[s].tobytes())
        f.write(struct.pack("<iiQQ", 0, 0, 0, 0))                                                  # empty timestamp Mat
    a, b = load_frame_bin(str(path))
    assert np.array_equal(a, rgb) and np.array_equal(b, dep)
    with pytest.raises(Rgbd360Error):
        load_frame_bin(str(tmp_path / "missing.bin"))
    bad = tmp_path / "bad.bin"
    bad.write_bytes(open(path, "rb").read()[:500])
    with pytest.raises(Rgbd360Error):
        load_frame_bin(str(bad))
    # buffers sized for another image size are never written: the call is refused
    import ctypes as C
    from rgbd360_amd import _lib
    L = _lib.load()
    small_rgb = np.zeros((8, 4, 4, 3), np.uint8)
    small_dep = np.zeros((8, 4, 4), np.uint16)
    r, c = C.c_int(4), C.c_int(4)
    rc = L.rgbd360_load_frame_bin(str(path).encode(), small_rgb.ctypes.data_as(C.c_void_p), small_dep.ctypes.data_as(C.c_void_p),
                                  C.byref(r), C.byref(c))
    assert rc == -4 and not small_rgb.any() and not small_dep.any()
    # absurd record headers (negative / huge sizes, wrong element type) are rejected before anything is read
    for hdr in (struct.pack("<iiQQ", -1, rows, 3, 16), struct.pack("<iiQQ", 100000, rows, 3, 16), struct.pack("<iiQQ", cols, rows, 4, 24)):
        evil = tmp_path / "evil.bin"
        evil.write_bytes(b"\x00" * 45 + hdr + b"\x00" * 64)
        with pytest.raises(Rgbd360Error):
            load_frame_bin(str(evil))


def test_build_select_widening_only_changes_the_encoding(tmp_path):
    """rgbd360_amd/build.py can re-encode VOP2 selects as VOP3 in the device assembly (an optional build): the rewrite
    touches exactly the `v_cndmask_b32_e32 ..., vcc` lines, keeps operands and comments, and leaves everything else byte for byte."""
    from rgbd360_amd import build as B
    src = tmp_path / "in.s"
    dst = tmp_path / "out.s"
    lines = [
        "\tv_cndmask_b32_e32 v17, 0, v11, vcc\n",
        "\tv_cndmask_b32_e32 v5, v6, v7, vcc ; select\n",
        "\tv_cndmask_b32_e64 v1, v2, v3, s[4:5]\n",          # already VOP3
        "\tv_cndmask_b32_e32 v1, v2, v3, s[4:5]\n",          # not a VCC select: left alone
        "\tv_fma_f32 v0, v1, v2, v3\n",
        "\tv_cndmask_b32_sdwa v1, v2, v3, vcc dst_sel:DWORD\n",
        ".LBB0_1:\n",
    ]
    src.write_text("".join(lines))
    assert B.widen_selects(str(src), str(dst)) == 2
    out = dst.read_text().splitlines(keepends=True)
    assert out[0] == "\tv_cndmask_b32_e64 v17, 0, v11, vcc\n"
    assert out[1].startswith("\tv_cndmask_b32_e64 v5, v6, v7, vcc") and "; select" in out[1]
    assert out[2:] == lines[2:]


def test_build_dependencies_cover_every_source_of_the_library():
    """needs_build() must see an edit to ANY file the one translation unit includes (a hand-kept list missed rig_dense.h and
    host_wait.h in round 3): the dependency set is a glob of csrc/ and include/."""
    from rgbd360_amd import build
    names = {os.path.basename(d) for d in build.deps()}
    here = os.path.join(ROOT, "rgbd360_amd", "csrc")
    assert {f for f in os.listdir(here) if f.endswith((".h", ".hip"))} <= names
    assert {"rig_dense.h", "host_wait.h", "rgbd360_api.hip", "rgbd360_hip.h", "rgbd360_hip_diag.h"} <= names
