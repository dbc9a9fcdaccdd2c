import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# every batching turn of the test session cross-checks its completion flag: what the host reads when the flag lands must be
# what it reads after a full stream synchronisation (raftq_capi.hip wait_turn; read once per process by the library)
os.environ.setdefault("RAFTQ_CYCLE_CHECK", "1")


# tests/test_hostsim.py re-runs the node / pipe suites with the host C++ under ASan / TSan and the engine calls answered by
# the CPU oracle (tests/c/libraftq_hostsim*.so).  That library is a TEST double: the package cannot be pointed at it (its
# loader only takes builds inside raftsql_amd/), so the sub-run names it in RAFTQ_TEST_ENGINE_DOUBLE and THIS file, not the
# product, swaps the path before anything is loaded.
_DOUBLE = os.environ.get("RAFTQ_TEST_ENGINE_DOUBLE")
if _DOUBLE:
    assert os.path.dirname(os.path.abspath(_DOUBLE)) == os.path.join(ROOT, "tests", "c"), "the engine double lives under tests/c"
    from raftsql_amd import _lib as _product_lib

    assert _product_lib._lib is None, "the product library is already loaded"
    _product_lib.LIB_PATH = _DOUBLE


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/raftq_oracle.h)."""
    from oracle import pyoracle

    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_engine_cls():
    """QuorumEngine bound to the native library; refuses to run without it."""
    from raftsql_amd.engine import QuorumEngine, device_count

    if _DOUBLE:  # tests/test_hostsim.py's sub-run: no GPU is needed, the oracle answers the engine calls
        return QuorumEngine
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a visible GPU")

    assert device_count() >= 1, "libraftq.so sees no HIP device"
    return QuorumEngine


@pytest.fixture(params=["device", "host"], ids=["stage-in-device-memory", "stage-in-host-memory"])
def stage_mode(request, monkeypatch):
    """Where the handles a test creates put their ack / inbound staging buffers (read by raftq_create): fine-grained
    device memory behind the large BAR (the default on the MI355X box) or pinned host memory (RAFTQ_STAGE=host: every
    machine without a host-addressable aperture).  Both forms are part of the suite the driver runs."""
    if request.param == "host":
        monkeypatch.setenv("RAFTQ_STAGE", "host")
    else:
        monkeypatch.delenv("RAFTQ_STAGE", raising=False)
    return request.param
