import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    # a GPU box gives a 1-GPU job 16 of its 256 hardware threads: an OpenMP team of 256 on 16 cores turns every small parallel
    # region of the oracle (pyramid rows of a 160 x 120 sensor image) into tens of milliseconds
    O.set_num_threads(min(16, O.num_threads()))
    return O


@pytest.fixture(scope="session")
def small_pair():
    """256x128 synthetic pair + ground truth (same generator call as tests/golden/make_golden.py)."""
    from rgbd360_amd import synth
    return synth.make_pair(256, 128, seed=1234)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail (not skip) when it is missing or no device is visible."""
    from rgbd360_amd import _lib
    L = _lib.load()
    assert L.rgbd360_device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return L
