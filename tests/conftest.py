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


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests`` on a machine without an MI355X skips the gpu-marked tests; an explicit
    ``-m gpu`` (the GPU box) or GUMBI_REQUIRE_GPU=1 keeps them, and the ``gpu`` fixture then FAILS them when
    no device is visible -- the product has no CPU fallback and the tests must not pretend otherwise."""
    import os

    asked = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if asked or os.environ.get("GUMBI_REQUIRE_GPU") == "1" or _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (select with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def gpu():
    """No CPU fallback exists: selected gpu tests fail loudly when no device / library is visible."""
    n = _gpu_count()
    if n < 1:
        pytest.fail("this test needs an MI355X and libgumbi_hip.so; none is visible (no CPU fallback exists)")
    return n


def pytest_runtest_teardown(item, nextitem):
    """GUMBI_TEST_CEILING=1 (diagnostic): the register-only MFMA rate a FRESH process sees after every test -- a test that
    leaves work running on the GPU (a worker that never drained a stream) shows up as a halved rate."""
    import os

    if os.environ.get("GUMBI_TEST_CEILING") != "1":
        return
    import subprocess

    out = subprocess.run([sys.executable, str(ROOT / "tools" / "gpu_ceiling_now.py")], capture_output=True, text=True)
    last = (out.stdout.strip().splitlines() or ["?"])[-1]
    print(f"\n[ceiling after {item.name}] {last[:60]}", flush=True)
