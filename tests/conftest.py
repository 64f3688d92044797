import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # skip reasons and warnings belong in the record the driver keeps (it runs `pytest -x -q`): which size a memory-gated test ran
    # at, why a multi-GPU test did not run on a 1-GPU box
    if "s" not in (config.option.reportchars or ""):
        config.option.reportchars = (config.option.reportchars or "") + "s"
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the real reference binary oracle/_ref/buildG_ref")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first(request):
    """tests that hand device pointers between torch and the library (sharded flow, processing order) need torch's HIP runtime
    initialised BEFORE libdisco_hip.so opens the device — afterwards torch reports "No HIP GPUs are available". Done once per
    session whenever GPU tests are selected; a no-op on the CPU suite."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import torch

        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    yield
