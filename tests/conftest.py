import os
import sys

import pytest

# torch ships its own libamdhip64; a process that loads /opt/rocm's copy first (through
# libblitzar_amd.so) cannot initialise torch.cuda afterwards ("No HIP GPUs are available").  The
# tests that hand torch-allocated device memory to the C ABI therefore need torch loaded first;
# the library itself never depends on torch.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    from blitzar_amd import api
    try:
        return api.load().bzamd_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """the reference's own CPU backend (oracle/_ref), test infrastructure only"""
    from oracle import ref_oracle
    if not ref_oracle.available():
        pytest.skip("oracle/_ref/libblitzar_ref.so not built")
    return ref_oracle


@pytest.fixture(scope="session")
def gpu_backend():
    """sxt_init(SXT_GPU_BACKEND) once per session; GPU tests fail loudly without a device"""
    from blitzar_amd import api
    assert _has_gpu(), "no HIP device visible: GPU tests must run on the MI355X box"
    api.reset_for_testing()
    assert api.init(api.SXT_GPU_BACKEND, 100) == 0
    assert api.load().bzamd_active_backend() == api.SXT_GPU_BACKEND
    yield api
    api.reset_for_testing()


@pytest.fixture()
def cpu_backend():
    """sxt_init(SXT_CPU_BACKEND): the explicit host backend (config 1 plumbing)"""
    from blitzar_amd import api
    api.reset_for_testing()
    assert api.init(api.SXT_CPU_BACKEND, 10) == 0
    yield api
    api.reset_for_testing()
