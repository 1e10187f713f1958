import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# libamc.so's fault-injection hooks (AMC_FAIL_NEXT_MEMSET / _MEMCPY, csrc/match_common.hip) exist only in a process that
# had this set when the library made its first checked memset: the test processes do, production does not
os.environ.setdefault("AMC_TEST_HOOKS", "1")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from pycolmap_amd import _capi
        return _capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def amc_ctx():
    """A live libamc context on device 0.  GPU tests must FAIL (not skip) if the native library
    is missing: there is no fallback path to hide behind."""
    from pycolmap_amd import _capi
    ctx = _capi.Context(0)
    yield ctx
    ctx.close()
