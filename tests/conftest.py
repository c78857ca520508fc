import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# tests and tools run on deterministic random-init weights (no checkpoints exist offline); a plain run of the entrypoint
# refuses to (ddpo_amd/utils/serialization.py) — tests/test_pretrained_layouts.py checks that refusal explicitly
os.environ.setdefault("DDPO_ALLOW_SYNTHETIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch) is what the slow tests wait for.  torch's default thread count on the GPU box is the host's 256 hardware threads,
    # which OVERSUBSCRIBES its convolutions: one CFG step of the SD-1.5 oracle takes 12.1 s at the default 128 threads, 7.5 s on 64, 4.65 s on 32 and 4.25 s on 16; forward + backward at 32x32: 16.6 / 8.2 / 4.3 / 3.6 s (round 5:
    # tools/oracle_threads.py, profiles/r05_oracle_threads.log).  DDPO_ORACLE_THREADS overrides.
    try:
        import torch
        n = int(os.environ.get("DDPO_ORACLE_THREADS", "0")) or min(os.cpu_count() or 1, 16)
        torch.set_num_threads(n)
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a HIP device (or without the built library) skips the gpu-marked tests instead of
    failing 300 of them; on the GPU box nothing is skipped here — a missing library there must fail loudly."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible): run `pytest -m gpu` on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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


def parity_record(line):
    """Append one measured-parity line (error margins of a passing test are invisible under `pytest -q`) to $DDPO_PARITY_LOG, if set
    (tools/r04_final.sh sets it and copies the file into profiles/)."""
    path = os.environ.get("DDPO_PARITY_LOG")
    print(line)
    if path:
        with open(path, "a") as f:
            f.write(line.strip() + "\n")
