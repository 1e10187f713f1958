"""Seed-only scene definitions shared by the fixture generators (pure numpy + pycolmap_amd/synth.py loaded BY PATH, so
that the scripts also run on a machine that has the real pycolmap and nothing of this repo built):

  golden_scenes()  the 20 cases of tests/golden/tvg_golden_v4.npz (make_tvg_golden.py), same generator, same order
  bench_scenes()   the 64 calibrated scenes of bench.py's verify leg (default_rng(7))
"""
import importlib.util
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
_spec = importlib.util.spec_from_file_location("amc_synth_by_path", ROOT / "pycolmap_amd" / "synth.py")
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

# indices 0 / 1: the two pinhole cameras of the v1 fixture; 2..: one camera per distortion model
# (synth.EXAMPLE_CAMERAS); a case using one of those sees its scene through that camera
# (synth.recamera_scene) so that the calibrated path has real geometry to find
CAMS = [("PINHOLE", (1200.0, 1200.0, 800.0, 600.0)), ("SIMPLE_PINHOLE", (1150.0, 805.0, 598.0))] + \
       [(m, synth.EXAMPLE_CAMERAS[m]) for m in ("SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV",
                                                "FOV", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE",
                                                "THIN_PRISM_FISHEYE")]
# (scene kwargs, prior focal length, camera index image 1 / image 2, option overrides)
CASES = [
    (dict(num_inliers=300, num_outliers=100), False, 0, 0, {}),
    (dict(num_inliers=300, num_outliers=100), True, 0, 0, {}),
    (dict(num_inliers=200, num_outliers=150, planar=True), False, 0, 0, {}),
    (dict(num_inliers=200, num_outliers=150, planar=True), True, 0, 1, {}),
    (dict(num_inliers=250, num_outliers=50, pure_rotation=True, noise=0.05), True, 0, 0, {}),
    (dict(num_inliers=60, num_outliers=40, noise=1.0), True, 1, 0, {}),
    (dict(num_inliers=0, num_outliers=40), False, 0, 0, {}),
    (dict(num_inliers=12, num_outliers=0), True, 0, 0, {}),                      # fewer than min_num_inliers
    (dict(num_inliers=150, num_outliers=60), True, 0, 0, dict(force_H_use=1)),
    (dict(num_inliers=150, num_outliers=60), True, 1, 1, dict(max_error=2.0, confidence=0.99, min_num_trials=50,
                                                             max_num_trials=2000, min_inlier_ratio=0.1)),
    (dict(num_inliers=180, num_outliers=90, planar=True), False, 0, 0, dict(detect_watermark=0, max_H_inlier_ratio=0.5)),
    (dict(num_inliers=400, num_outliers=300, noise=0.3), True, 0, 0, dict(min_E_F_inlier_ratio=0.8)),
    # cameras with distortion parameters (Camera::CamFromImg = IterativeUndistortion / closed forms)
    (dict(num_inliers=260, num_outliers=90), True, 2, 2, {}),                    # SIMPLE_RADIAL: extract_features' default
    (dict(num_inliers=220, num_outliers=120), True, 3, 4, {}),                   # RADIAL x OPENCV
    (dict(num_inliers=200, num_outliers=80, planar=True), True, 4, 0, {}),       # OPENCV x PINHOLE, planar
    (dict(num_inliers=240, num_outliers=100), True, 5, 6, {}),                   # OPENCV_FISHEYE x FULL_OPENCV
    (dict(num_inliers=240, num_outliers=100), True, 7, 8, {}),                   # FOV x SIMPLE_RADIAL_FISHEYE
    (dict(num_inliers=240, num_outliers=100), True, 9, 10, {}),                  # RADIAL_FISHEYE x THIN_PRISM_FISHEYE
    (dict(num_inliers=150, num_outliers=60), False, 2, 4, {}),                   # no prior focal length: F + H only
    (dict(num_inliers=250, num_outliers=50, pure_rotation=True, noise=0.05), True, 2, 3, {}),
]
GOLDEN_SEED = 20260924
RANSAC_OPTION_KEYS = ("max_error", "min_inlier_ratio", "confidence", "dyn_num_trials_multiplier", "min_num_trials",
                      "max_num_trials")


def golden_scenes():
    """Yields dict(name, pts1, pts2, matches, cam1 = (model, params), cam2, prior, opts) - the scenes of
    tvg_golden_v4.npz, bit for bit (same generator state sequence as make_tvg_golden.py)."""
    rng = np.random.default_rng(GOLDEN_SEED)
    for k, (kw, prior, c1, c2, okw) in enumerate(CASES):
        sc = synth.two_view_scene(rng, **kw)
        if c1 >= 2 or c2 >= 2:
            sc = synth.recamera_scene(sc, CAMS[c1][0], CAMS[c1][1], CAMS[c2][0], CAMS[c2][1])
        yield dict(name=f"golden{k}", pts1=sc["pts1"], pts2=sc["pts2"], matches=sc["matches"], cam1=CAMS[c1], cam2=CAMS[c2],
                   cam_index=(c1, c2), prior=bool(prior), opts=dict(okw), width=1600, height=1200)


def bench_scenes(distinct=64):
    """The verify leg of bench.py: `distinct` seeded calibrated scenes (PINHOLE, prior focal length), a quarter planar."""
    rng = np.random.default_rng(7)
    for k in range(distinct):
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(150, 450)), num_outliers=int(rng.integers(50, 200)),
                                  planar=(k % 4 == 3))
        cam = ("PINHOLE", (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0))
        yield dict(name=f"bench{k}", pts1=sc["pts1"], pts2=sc["pts2"], matches=sc["matches"], cam1=cam, cam2=cam,
                   cam_index=(-1, -1), prior=True, opts={}, width=sc["width"], height=sc["height"])


def all_scenes():
    yield from golden_scenes()
    yield from bench_scenes()
