"""Loads the deviation-budget builds of the verification oracle (oracle/Makefile `variants`: D3, D1, D2 switched
off one at a time and all together; oracle/tvg_oracle.cc header) and runs EstimateTwoViewGeometry through them.
Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess
from pathlib import Path

import numpy as np

import oracle_lib as o

ROOT = Path(__file__).resolve().parents[2]
BUILD = ROOT / "oracle" / "_build"
NAMES = ("seqsum", "lapacksvd", "companion", "upstreamlike")
WHAT = {"seqsum": "D3 off (sequential sums)", "lapacksvd": "D1 off (LAPACK SVD)", "companion": "D2 off (companion-matrix roots)",
        "upstreamlike": "D1 + D2 + D3 off"}
_libs: dict = {}


def lapack_library() -> str | None:
    """The OpenBLAS (with LAPACKE) that scipy bundles, or None."""
    try:
        import scipy
    except ImportError:
        return None
    base = Path(scipy.__file__).resolve().parent.parent
    hits = sorted(glob.glob(str(base / "scipy.libs" / "libscipy_openblas*.so")))
    return hits[0] if hits else None


def load(name: str) -> C.CDLL:
    if name not in _libs:
        lib = BUILD / f"liboracle_{name}.so"
        if not lib.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "variants"], check=True, capture_output=True)
        if name != "seqsum":
            path = lapack_library()
            if path is None:
                raise RuntimeError("no LAPACK library found (scipy's bundled OpenBLAS)")
            os.environ["ORACLE_LAPACK_LIB"] = path
        _libs[name] = C.CDLL(str(lib))
    return _libs[name]


def estimate_two_view_geometry(name, cam1, pts1, cam2, pts2, matches, opts=None, seed=0):
    """oracle_lib.estimate_two_view_geometry through variant `name` (None: the default oracle)."""
    if name is None:
        return o.estimate_two_view_geometry(cam1, pts1, cam2, pts2, matches, opts, seed)
    lib = load(name)
    opts = opts or o.tvg_default_options()
    p1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
    res = o.TvgResult()
    mask = np.zeros(max(1, len(m)), dtype=np.uint8)
    lib.oracle_estimate_two_view_geometry.restype = C.c_int
    rc = lib.oracle_estimate_two_view_geometry(
        C.byref(cam1), o._p(p1), C.c_size_t(len(p1)), C.byref(cam2), o._p(p2), C.c_size_t(len(p2)),
        o._p(m), C.c_size_t(len(m)), C.byref(opts), C.c_uint32(seed), C.byref(res), o._p(mask))
    assert rc == 0
    return dict(config=int(res.config), config_name=o.CONFIG_NAMES[res.config], num_inliers=int(res.num_inliers),
                E=np.array(res.E).reshape(3, 3), F=np.array(res.F).reshape(3, 3), H=np.array(res.H).reshape(3, 3),
                trials=list(res.trials), inl=list(res.inl), inlier_mask=mask[:len(m)].astype(bool))
