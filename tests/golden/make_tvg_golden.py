"""Generates tests/golden/tvg_golden_v4.npz: seeded two-view scenes with the oracle's
EstimateTwoViewGeometry (+ EstimateTwoViewGeometryPose) results.

As for the match fixture, the reference holds no golden vectors for this path (SURVEY.md section
8c) and COLMAP 3.9.1 cannot be run here, so the fixture is produced by our own oracle
(oracle/tvg_oracle.cc) and pins IT - and through it the HIP path - against regressions; it does
not pin the oracle against COLMAP ("parity unpinned").  Floating-point results are stored as raw
bit patterns (uint64).  The oracle's libm calls (log in the dynamic trial count, acos in the
triangulation angle) are glibc's; the fixture was generated with glibc 2.35 on x86-64.

Run from the repo root:  python tests/golden/make_tvg_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib as o  # noqa: E402
from pycolmap_amd import synth  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))
from scenes import CAMS, CASES, GOLDEN_SEED  # noqa: E402  (shared with make_reference_golden.py)

FIELDS = ("E", "F", "H", "qvec", "tvec", "R")


def build(verbose=True):
    """Every array of the fixture, from the oracle as it is now (tests/test_oracle_frozen_cpu.py compares them with the file)."""
    rng = np.random.default_rng(GOLDEN_SEED)
    out = {"num_cases": np.int64(len(CASES)), "oracle_version": np.array(o.tvg_version())}
    for k, (kw, prior, c1, c2, okw) in enumerate(CASES):
        sc = synth.two_view_scene(rng, **kw)
        if c1 >= 2 or c2 >= 2:
            sc = synth.recamera_scene(sc, CAMS[c1][0], CAMS[c1][1], CAMS[c2][0], CAMS[c2][1])
        out[f"pts1_{k}"] = sc["pts1"]
        out[f"pts2_{k}"] = sc["pts2"]
        out[f"matches_{k}"] = sc["matches"]
        out[f"cams_{k}"] = np.array([c1, c2, int(prior)], dtype=np.int64)
        keys = sorted(okw)
        out[f"opt_keys_{k}"] = np.array(keys, dtype="U32")
        out[f"opt_vals_{k}"] = np.array([float(okw[x]) for x in keys], dtype=np.float64)
        for pose in (0, 1):
            opts = o.tvg_default_options(compute_relative_pose=pose, **okw)
            cam1 = o.make_camera(CAMS[c1][0], 1600, 1200, CAMS[c1][1], prior=prior)
            cam2 = o.make_camera(CAMS[c2][0], 1600, 1200, CAMS[c2][1], prior=prior)
            r = o.estimate_two_view_geometry(cam1, sc["pts1"], cam2, sc["pts2"], sc["matches"], opts, seed=0)
            tag = f"{k}_p{pose}"
            out[f"config_{tag}"] = np.int64(r["config"])
            out[f"mask_{tag}"] = r["inlier_mask"]
            out[f"trials_{tag}"] = np.array(r["trials"], dtype=np.int64)
            out[f"inl_{tag}"] = np.array(r["inl"], dtype=np.int64)
            out[f"points3D_{tag}"] = np.int64(r["num_points3D"])
            out[f"tri_angle_{tag}"] = np.array([r["tri_angle"]]).view(np.uint64)
            for f in FIELDS:
                out[f"{f}_{tag}"] = np.ascontiguousarray(r[f], dtype=np.float64).reshape(-1).view(np.uint64)
            if verbose:
                print(k, pose, r["config_name"], r["num_inliers"], r["trials"], round(r["tri_angle"], 5))
    return out


def main():
    np.savez_compressed(Path(__file__).with_name("tvg_golden_v4.npz"), **build())


if __name__ == "__main__":
    main()
