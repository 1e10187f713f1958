"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/*.h declares,
and fails loudly (no fallback) when there is no GPU.  No compute calls here."""
import ctypes
import re
from pathlib import Path

import pytest

from pycolmap_amd import _capi

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(amc_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    decl = declared_symbols()
    assert "amc_match_pairs" in decl and len(decl) >= 10
    for name in decl:
        assert hasattr(lib, name), f"libamc.so does not export {name}"
    assert set(_capi.EXPORTED_SYMBOLS) <= set(decl)
    assert lib.amc_abi_version() == 2


def test_no_gpu_means_loud_failure_not_fallback():
    n = _capi.device_count()
    if n > 0:
        pytest.skip("a GPU is visible; this test covers the CPU-only container")
    with pytest.raises(_capi.AmcError):
        _capi.Context(0)


def test_struct_layout_matches_header():
    # amc_match_opts: 2 doubles + 2 int32 = 24 bytes; amc_match_result ends with a pointer
    assert ctypes.sizeof(_capi.MatchOpts) == 24
    assert ctypes.sizeof(_capi.MatchResult) == 8 * 3 + 8 * 4 + 8 * 3 + 8 + 8
