import os
import sys

import pytest

# the ranked allpairs path tolerates device/host floating-point drift in production; the suite wants to SEE any (vsx_search.cpp)
os.environ.setdefault("VSX_RANK_STRICT", "1")
# r05: in production a sparse-task class needs >= 4 096 tasks to be worth its own launches (vsx_host.cpp); the fixtures are small, and
# the suite wants the sparse-task kernels exercised by every one- to four-target task it plans (the golden vectors are one-target
# tasks).  test_sparse_class_threshold covers the production threshold in a child process; VSX_SPARSE=0 (test_alternate_kernel_modes)
# the whole-wave classes.
os.environ.setdefault("VSX_SPARSE_MIN", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from vsearch_amd import _lib
        return _lib.load().vsx_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build(ref=True)      # builds liboracle.so; oracle/_ref only where /root/reference exists
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must run the HIP path: a missing extension or device is a FAILURE there, not a skip."""
    from vsearch_amd import _lib
    lib = _lib.load()
    assert lib.vsx_device_count() > 0, "no gfx950 device visible: -m gpu tests need the MI355X"
    return lib
