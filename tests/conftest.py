import os
import sys

import pytest

# Parity tests compare with the oracle's model of ONE reference tier: pin it (an index otherwise follows the host's CPUID, as the
# reference does -- tests/test_host_logic.py covers that rule).  Tier-specific tests set the variable themselves.
os.environ.setdefault("VECSIM_GPU_TIER", "avx512")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _has_gpu():
    try:
        from vectorsimilarity_amd import _capi
        return _capi.load().VecSimGpu_DeviceCount() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: a green GPU tier that ran nothing
    # is worse than a red one.  Plain runs (no -m) skip GPU tests when there is no device.
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def vso():
    from oracle import vso as m
    m.build()
    m.lib()
    return m
