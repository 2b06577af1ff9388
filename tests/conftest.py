import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path; a missing GPU under -m gpu is a failure, not a skip."""
    assert _gpu_available(), "no HIP device visible: -m gpu tests need an MI355X"
    from flame_ros_amd import lib
    lib.load()  # raises if libflame_hip.so is missing: no fallback
    return 0


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import build_oracle
    return build_oracle()


@pytest.fixture(scope="module", autouse=True)
def _free_lease_at_module_start():
    """On a GPU box every test module starts from a device lease that is free: a resident launch that gave up in an EARLIER
    module (beside that module's foreign kernels) leaves a back-off of 16+ solves behind, during which plans are sized for
    launches -- not what a module's own assertions about resident tiles expect (tests/util.py settle_lease)."""
    if _gpu_available():
        try:
            from tests.util import settle_lease
            settle_lease()
        except Exception:  # noqa: BLE001 -- never the reason a module fails
            pass
    yield
