import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _reset_datapath():
    """Every test starts on the exact-fp32 datapath with an empty packed-weight registry (tests that exercise the bf16
    paths, or the entrypoint which selects bf16x3 itself, must not leak that choice into the next test)."""
    try:
        from ddpo_amd import lib as L
    except Exception:
        yield
        return
    L.DATAPATH = "fp32"
    L.PACKED.clear()
    yield
    L.DATAPATH = "fp32"
    L.PACKED.clear()
