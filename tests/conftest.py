import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


def _gpu_count():
    try:
        from gumbi_amd import engine

        return engine.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """Skip-free guard: `-m gpu` tests are only selected on the GPU box; fail loudly otherwise."""
    n = _gpu_count()
    if n < 1:
        pytest.fail("this test needs an MI355X and libgumbi_hip.so; none is visible (no CPU fallback exists)")
    return n
