"""The property guided matching's candidate-generation kernel (match_guided.hip) rests on, checked on the CPU: for
every point, the keypoints the kernel lists (pycolmap_amd/csrc/guided_region.h, compiled for the host by
tests/shim/guided_shim.cc: the same grid, the same per-row regions) contain every keypoint the float32 filter of
FeatureMatcher::MatchGuided accepts (oracle_guided_filter) - for real epipolar geometries, wild models, odd keypoint
layouts and thresholds from 0 to 1e4.  Pairs the setup turns down (they keep the dense kernel) are counted, not
checked."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as o

ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "shim" / "_build" / "libguidedshim.so"


@pytest.fixture(scope="module")
def shim():
    SHIM.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "shim" / "guided_shim.cc"
    hdr = ROOT / "pycolmap_amd" / "csrc" / "guided_region.h"
    if not SHIM.exists() or SHIM.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared", "-fPIC", str(src),
                        "-o", str(SHIM)], check=True)
    return C.CDLL(str(SHIM))


def accepted_matrix(kind, model32, max_residual, kp1, kp2):
    """[n1, n2] bool: the float32 filter does NOT reject (x1, x2) - NaN counts as accepted, as in MatchGuided."""
    lib = o.load()
    lib.oracle_guided_filter.restype = C.c_int
    lib.oracle_guided_filter.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    mp = model32.ctypes.data_as(C.c_void_p)
    out = np.zeros((len(kp1), len(kp2)), dtype=bool)
    for i, (x1, y1) in enumerate(kp1):
        for j, (x2, y2) in enumerate(kp2):
            out[i, j] = not lib.oracle_guided_filter(kind, mp, C.c_float(max_residual), C.c_float(x1), C.c_float(y1),
                                                     C.c_float(x2), C.c_float(y2))
    return out


def layouts(rng, n):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        xy = rng.uniform([0, 0], [1600, 1200], size=(n, 2))
    elif kind == 1:
        xy = rng.normal([800, 600], [rng.uniform(1, 300), rng.uniform(1, 300)], size=(n, 2))
    elif kind == 2:
        t = rng.uniform(0, 1, size=n)
        xy = np.stack([100 + 1400 * t, 300 + rng.uniform(-500, 500) * t], axis=1)
    elif kind == 3:
        xy = np.repeat(rng.uniform(0, 1000, size=(max(n // 4, 1), 2)), 4, axis=0)[:n]
    elif kind == 4:
        xy = np.round(rng.uniform([0, 0], [1600, 1200], size=(n, 2)))      # integer coordinates: exact cell boundaries
    else:
        xy = rng.uniform(-1e4, 1e4, size=(1, 2)) + rng.uniform(0, 10.0 ** rng.uniform(-2, 3), size=(n, 2))
    return np.ascontiguousarray(xy, dtype=np.float32)


def models(rng, style):
    e = np.array([rng.uniform(-5000, 6000), rng.uniform(-5000, 6000), 1.0])
    ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
    A = np.eye(3) + rng.normal(size=(3, 3)) * [[0.05, 0.05, 30], [0.05, 0.05, 30], [1e-5, 1e-5, 0.05]]
    F = [ex @ A, ex, np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), np.array([[0, 0, 1.0], [0, 0, 0], [-1.0, 0, 0]]),
         rng.normal(size=(3, 3)) * [1e-6, 1e-6, 1e-3], rng.normal(size=(3, 3))][style]
    H = [A, np.diag([rng.uniform(0.1, 10), rng.uniform(0.1, 10), 1.0]), np.eye(3) + rng.normal(size=(3, 3)) * 1e-4,
         np.array([[1, 0, 0], [0, 1, 0], [rng.uniform(-3e-4, 3e-4), rng.uniform(-3e-4, 3e-4), 1.0]]), np.linalg.inv(A),
         rng.normal(size=(3, 3))][style]
    s = 10.0 ** rng.uniform(-4, 4)
    return (F * s).astype(np.float32).reshape(9), (H * s).astype(np.float32).reshape(9)


def test_listed_keypoints_contain_every_accepted_one(shim):
    rng = np.random.default_rng(31)
    shim.shim_guided_candidates.restype = C.c_int
    checked = dense = 0
    listed = accepted = 0
    for trial in range(300):
        n1, n2 = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        kp1, kp2 = layouts(rng, n1), layouts(rng, n2)
        n1, n2 = len(kp1), len(kp2)
        Fm, Hm = models(rng, trial % 6)
        max_error = float(rng.choice([0.0, 0.3, 1.0, 4.0, 4.0, 16.0, 100.0, 1e4]))
        mr = np.float32(max_error * max_error)
        for kind, model in ((1, Fm), (2, Hm)):
            acc = None
            for direction in (0, 1):
                nr, nc = (n1, n2) if direction == 0 else (n2, n1)
                cand = np.zeros((nr, nc), dtype=np.uint8)
                rc = shim.shim_guided_candidates(kind, model.ctypes.data_as(C.c_void_p), C.c_float(mr),
                                                 kp1.ctypes.data_as(C.c_void_p), n1, kp2.ctypes.data_as(C.c_void_p), n2,
                                                 direction, cand.ctypes.data_as(C.c_void_p))
                assert rc in (0, 1)
                if rc == 0:
                    dense += 1
                    continue
                if acc is None:
                    acc = accepted_matrix(kind, model, float(mr), kp1, kp2)
                a = acc if direction == 0 else acc.T
                missing = a & (cand == 0)
                assert not missing.any(), (f"trial {trial} kind {kind} dir {direction} max_error {max_error}: "
                                           f"{int(missing.sum())} accepted pairings not listed, first {np.argwhere(missing)[0]}")
                checked += 1
                listed += int(cand.sum())
                accepted += int(a.sum())
    assert checked > 300 and dense > 5           # both routings occur
    assert listed >= accepted


def test_real_geometry_lists_a_small_fraction(shim):
    """A true epipolar geometry with a 4-pixel threshold on 1600 x 1200 images: the kernel's point is that it lists a
    few per cent of the matrix, and of course still everything the filter accepts."""
    rng = np.random.default_rng(32)
    shim.shim_guided_candidates.restype = C.c_int
    n = 1500
    kp1 = rng.uniform([0, 0], [1600, 1200], size=(n, 2)).astype(np.float32)
    kp2 = rng.uniform([0, 0], [1600, 1200], size=(n, 2)).astype(np.float32)
    K = np.array([[1200.0, 0, 800], [0, 1200, 600], [0, 0, 1]])
    t = np.array([1.0, 0.1, 0.05])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    ang = 0.1
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    model = (F / np.abs(F).max()).astype(np.float32).reshape(9)
    sub = slice(0, 200)
    acc = accepted_matrix(1, model, 16.0, kp1[sub], kp2)
    cand = np.zeros((n, n), dtype=np.uint8)
    rc = shim.shim_guided_candidates(1, model.ctypes.data_as(C.c_void_p), C.c_float(16.0), kp1.ctypes.data_as(C.c_void_p), n,
                                     kp2.ctypes.data_as(C.c_void_p), n, 0, cand.ctypes.data_as(C.c_void_p))
    assert rc == 1
    assert not (acc & (cand[sub] == 0)).any()
    frac_listed, frac_acc = cand.mean(), acc.mean()
    assert frac_acc < 0.02 and frac_listed < 0.08, (frac_listed, frac_acc)
