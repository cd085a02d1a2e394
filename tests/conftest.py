import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # Build the checkers before collection: the oracle always (gcc), the unmodified reference when
    # /root/reference is mounted (build container) -- the "needs reference" skips are decided at import time.
    try:
        from oracle import pyoracle
        pyoracle.build()
    except Exception as e:                       # a missing toolchain must not hide the real test errors
        print(f"conftest: building oracle/ failed: {e}", file=sys.stderr)


def _gpu_present() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return Path("/dev/kfd").exists()


@pytest.fixture(scope="session")
def gpu_available():
    return _gpu_present()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: leave gpu tests alone.
    pass
