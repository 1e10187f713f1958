"""Loader of tests/golden/tvg_golden_v4.npz (made by tests/golden/make_tvg_golden.py)."""
from pathlib import Path

import numpy as np

PATH = Path(__file__).parent / "golden" / "tvg_golden_v4.npz"
import sys
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pycolmap_amd import synth  # noqa: E402  (EXAMPLE_CAMERAS only: pure numpy)

CAMS = [("PINHOLE", (1200.0, 1200.0, 800.0, 600.0)), ("SIMPLE_PINHOLE", (1150.0, 805.0, 598.0))] + \
       [(m, synth.EXAMPLE_CAMERAS[m]) for m in ("SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV",
                                                "FOV", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE",
                                                "THIN_PRISM_FISHEYE")]
FIELDS = ("E", "F", "H", "qvec", "tvec", "R")
INT_OPTS = {"force_H_use", "detect_watermark", "min_num_trials", "max_num_trials", "min_num_inliers",
            "multiple_models", "compute_relative_pose"}


def cases():
    """Yields dicts: pts1, pts2, matches, cam1/cam2 = (model, params), prior, opts (dict), and
    want[pose] = dict(config, mask, trials, inl, points3D, tri_angle bits, E/F/H/qvec/tvec/R bits)."""
    z = np.load(PATH)
    import oracle_lib
    # the fixture pins ONE arithmetic: a fixture from another oracle version is not evidence of anything
    assert str(z["oracle_version"]) == oracle_lib.tvg_version(), (str(z["oracle_version"]), oracle_lib.tvg_version())
    for k in range(int(z["num_cases"])):
        c1, c2, prior = (int(x) for x in z[f"cams_{k}"])
        opts = {}
        for key, val in zip(z[f"opt_keys_{k}"], z[f"opt_vals_{k}"]):
            opts[str(key)] = int(val) if str(key) in INT_OPTS else float(val)
        want = {}
        for pose in (0, 1):
            tag = f"{k}_p{pose}"
            w = dict(config=int(z[f"config_{tag}"]), mask=z[f"mask_{tag}"], trials=z[f"trials_{tag}"].tolist(),
                     inl=z[f"inl_{tag}"].tolist(), points3D=int(z[f"points3D_{tag}"]),
                     tri_angle=z[f"tri_angle_{tag}"])
            for f in FIELDS:
                w[f] = z[f"{f}_{tag}"]
            want[pose] = w
        yield dict(index=k, pts1=z[f"pts1_{k}"], pts2=z[f"pts2_{k}"], matches=z[f"matches_{k}"], cam1=CAMS[c1],
                   cam2=CAMS[c2], prior=bool(prior), opts=opts, want=want)


def bits(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)
