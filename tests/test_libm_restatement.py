"""csrc/libm_f32.h -- asinf / atanf / atan2f / roundf restated operation for operation from glibc's machine code, what
rgbd360_set_index_arithmetic(ctx, 1) computes the warp with -- against the C library of this host (no GPU: the host compile of the
header).  The quick form: every 61st float through the one-argument functions, 20 million drawn pairs through atan2f; the full form
(every float, 4e9 pairs, one minute on eight cores) is `tools/libm_f32_check.cpp` without arguments, recorded in
profiles/r06_reference_arithmetic.txt.  The device compile is checked by tests/test_gpu_parity.py (rgbd360_selftest_libm)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_libm_restatement_equals_the_c_library_on_the_host(tmp_path):
    exe = str(tmp_path / "libm_f32_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-pthread", "-o", exe, os.path.join(ROOT, "tools", "libm_f32_check.cpp")], check=True)
    out = subprocess.run([exe, "20", "61"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 4 and all(" 0 differ" in l for l in lines), out.stdout
