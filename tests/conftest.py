import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from sketchy_amd import _lib
        return _lib.load().skx_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must FAIL (not skip) when the HIP library is missing; they skip only when
    there is genuinely no device (e.g. someone runs -m gpu in the CPU container)."""
    from sketchy_amd import _lib
    _lib.load()  # raises ImportError loudly if the extension was not built
    if not _has_gpu():
        pytest.skip("no HIP device visible")
    return 0
