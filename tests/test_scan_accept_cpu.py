"""The match scan's accept bit (pycolmap_amd/csrc/scan_accept.h) must be a SUPERSET of COLMAP's one-way acceptance
tests evaluated with the host-libm acos table (one_way_accepts, amc_internal.h; FindBestMatchesOneWayBruteForce,
SURVEY.md A.2): resolve_index re-tests every kept row exactly, a dropped row is lost.  build_scan_accept proves this
for the options at hand; here the same property is checked from outside, by brute force against the table, for
COLMAP's default options and a spread of others, and the kept set is shown to be tight (the point of the bit)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "shim" / "_build" / "libscanacceptshim.so"
LUT_SIZE = 262145


@pytest.fixture(scope="module")
def shim():
    SHIM.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "shim" / "scan_accept_shim.cc"
    hdr = ROOT / "pycolmap_amd" / "csrc" / "scan_accept.h"
    if not SHIM.exists() or SHIM.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared", "-fPIC", str(src),
                        "-o", str(SHIM)], check=True)
    lib = C.CDLL(str(SHIM))
    lib.sa_sizeof.restype = C.c_uint
    return lib


@pytest.fixture(scope="module")
def lut():
    # the table amc_ctx_create builds: acosf(min(d * 2^-18, 1)) with the host libm, in float32
    libm = C.CDLL("libm.so.6")
    libm.acosf.restype = C.c_float
    libm.acosf.argtypes = [C.c_float]
    d = np.arange(LUT_SIZE, dtype=np.float32) * np.float32(2.0 ** -18)
    d = np.minimum(d, np.float32(1.0))
    # vectorised acos differs from libm's acosf in the last bit here and there: take libm's, value by value, once
    out = np.empty(LUT_SIZE, dtype=np.float32)
    for i in range(LUT_SIZE):
        out[i] = libm.acosf(float(d[i]))
    return out


def build(lib, lut, ratio, dist):
    buf = (C.c_char * lib.sa_sizeof())()
    lib.sa_build(lut.ctypes.data_as(C.c_void_p), C.c_uint(LUT_SIZE), C.c_float(ratio), C.c_float(dist), buf)
    raw = np.frombuffer(buf, dtype=np.uint32).copy()
    return buf, {"coef": raw[:9].view(np.float32), "margin": raw[9:10].view(np.float32)[0], "min_best": int(raw[10]),
                 "trivial": int(raw[11])}


def exact_accepts(lut, best, second, ratio, dist):
    b = np.minimum(best, LUT_SIZE - 1)
    s = np.minimum(second, LUT_SIZE - 1)
    ab, as_ = lut[b], lut[s]
    rej = (best == 0) | (ab > np.float32(dist)) | (ab >= np.float32(ratio) * as_)
    return ~rej


def keep(lib, buf, best, second):
    out = np.zeros(len(best), dtype=np.uint8)
    lib.sa_eval(buf, best.ctypes.data_as(C.c_void_p), second.ctypes.data_as(C.c_void_p), C.c_uint(len(best)),
                out.ctypes.data_as(C.c_void_p))
    return out.astype(bool)


@pytest.mark.parametrize("ratio,dist", [(0.8, 0.7), (0.6, 0.7), (0.95, 1.2), (1.0, 0.7), (0.3, 0.5), (0.8, 3.2), (0.8, 0.0),
                                        (1.5, 0.7), (0.0, 0.7)])
def test_scan_accept_is_a_tight_superset_of_the_exact_tests(shim, lut, ratio, dist):
    buf, a = build(shim, lut, ratio, dist)
    rng = np.random.default_rng(int(ratio * 1000) + int(dist * 10))
    n = 400_000
    # seconds anywhere below the best, bests over the whole table, values past 2^18 (saturated descriptors) included
    best = rng.integers(0, 300_000, n).astype(np.uint32)
    second = (best * rng.random(n)).astype(np.uint32)
    # and a band around the ratio threshold, where the decision is made: second near the critical value of each best
    bb = rng.integers(1000, 262144, n).astype(np.uint32)
    th = np.arccos(np.minimum(bb / 262144.0, 1.0))
    with np.errstate(invalid="ignore", divide="ignore"):
        crit = 262144.0 * np.cos(np.minimum(th / max(ratio, 1e-6), np.pi / 2))
    ss = np.clip(crit + rng.integers(-40, 41, n), 0, 2 ** 20).astype(np.uint32)
    best = np.concatenate([best, bb, np.array([0, 1, 262143, 262144, 262145, 8323200], dtype=np.uint32)])
    second = np.concatenate([second, ss, np.array([0, 0, 262143, 262144, 262144, 8323200], dtype=np.uint32)])
    ex = exact_accepts(lut, best, second, ratio, dist)
    kp = keep(shim, buf, best, second)
    assert not np.any(ex & ~kp), "the scan would drop a row COLMAP accepts"
    if not a["trivial"]:
        # tight: what is kept beyond the exact set sits within the margin (16 units of 2^-18) of the ratio threshold -
        # 24 more on the best value and the exact test accepts too
        # (values past 2^18 saturate the table on both sides: exact rejects equal saturated values, the scan keeps them)
        extra = kp & ~ex & (best <= 262144 - 24)
        assert np.all(exact_accepts(lut, best[extra] + np.uint32(24), second[extra], ratio, dist))


def test_default_options_are_not_trivial(shim, lut):
    _, a = build(shim, lut, 0.8, 0.7)
    assert a["trivial"] == 0
    # test 1 exactly: the first table entry at or under max_distance
    assert lut[a["min_best"]] <= np.float32(0.7) < lut[a["min_best"] - 1]


def test_exhaustive_over_second_at_the_critical_best(shim, lut):
    """The proof inside build_scan_accept, redone here for the default options: for EVERY second value, the smallest
    best the exact ratio test accepts is kept."""
    ratio, dist = 0.8, 0.7
    buf, a = build(shim, lut, ratio, dist)
    s = np.arange(LUT_SIZE, dtype=np.uint32)
    rhs = np.float32(ratio) * lut
    # lut is non-increasing: the first index with lut[b] < rhs[s]
    bcrit1 = np.searchsorted(-lut, -rhs, side="right").astype(np.uint32)   # count of entries with lut >= rhs
    ok = bcrit1 < LUT_SIZE
    b = np.maximum(bcrit1[ok], 1).astype(np.uint32)
    big = np.full(len(b), 0, dtype=np.uint32)
    assert np.all(exact_accepts(lut, b, s[ok], ratio, 10.0) | (b == 1))
    probe_buf, _ = build(shim, lut, ratio, 10.0)   # test 1 out of the way (acos <= pi/2 < 10 always)
    assert np.all(keep(shim, probe_buf, b, s[ok]))
    del big
