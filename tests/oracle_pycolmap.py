"""The slice of the pycolmap API that tests/golden/make_reference_golden.py calls, served by the CPU oracle
(tests/oracle_lib.py): lets the reference pin kit record "what the oracle returns" through the very same calls it makes
against the real pycolmap, so that tests/test_reference_golden.py compares like with like.  Test infrastructure only."""
from types import SimpleNamespace

import numpy as np

import oracle_lib as o

__version__ = "oracle"
COLMAP_version = "restatement of 3.9.1 (oracle/tvg_oracle.cc)"
has_hip = False   # (not the real pycolmap: make_reference_golden.py refuses to write reference_v1.npz from this)


class Camera:
    def __init__(self, model="PINHOLE", width=1600, height=1200, params=()):
        self.model, self.width, self.height, self.params = model, int(width), int(height), tuple(float(x) for x in params)
        self.has_prior_focal_length = False

    def oc(self, prior=None):
        return o.make_camera(self.model, self.width, self.height, self.params,
                             prior=self.has_prior_focal_length if prior is None else prior)


class RANSACOptions:
    def __init__(self):
        self.max_error, self.min_inlier_ratio, self.confidence = 4.0, 0.01, 0.9999
        self.dyn_num_trials_multiplier, self.min_num_trials, self.max_num_trials = 3.0, 1000, 100000

    def kw(self):
        return dict(max_error=self.max_error, min_inlier_ratio=self.min_inlier_ratio, confidence=self.confidence,
                    dyn_num_trials_multiplier=self.dyn_num_trials_multiplier, min_num_trials=int(self.min_num_trials),
                    max_num_trials=int(self.max_num_trials))


class TwoViewGeometryOptions:
    def __init__(self):
        self.min_num_inliers, self.min_E_F_inlier_ratio, self.max_H_inlier_ratio = 15, 0.95, 0.8
        self.watermark_min_inlier_ratio, self.watermark_border_size = 0.7, 0.1
        self.detect_watermark, self.multiple_ignore_watermark, self.force_H_use = True, True, False
        self.compute_relative_pose, self.multiple_models = False, False
        self.ransac = RANSACOptions()
        self.ransac.min_inlier_ratio, self.ransac.confidence = 0.25, 0.999
        self.ransac.min_num_trials, self.ransac.max_num_trials = 100, 10000

    def oo(self):
        kw = {k: (int(v) if isinstance(v, (bool, np.bool_)) else v) for k, v in vars(self).items() if k != "ransac"}
        kw.update(self.ransac.kw())
        return o.tvg_default_options(**kw)


def _ransac(kind, key, p1, p2, opts):
    p1, p2 = np.asarray(p1, np.float64), np.asarray(p2, np.float64)
    r = o.ransac_estimate(kind, p1, p2, o.ransac_options(**opts.kw()), seed=0)
    if not r["success"]:
        return None, r
    return {key: r["model"], "num_inliers": r["num_inliers"], "inliers": r["inliers"]}, r


def fundamental_matrix_estimation(points2D1, points2D2, estimation_options=None):
    return _ransac("F", "F", points2D1, points2D2, estimation_options or RANSACOptions())[0]


def homography_matrix_estimation(points2D1, points2D2, estimation_options=None):
    return _ransac("H", "H", points2D1, points2D2, estimation_options or RANSACOptions())[0]


def _rigid(qvec_wxyz, tvec):
    q = np.asarray(qvec_wxyz, np.float64)
    return SimpleNamespace(rotation=SimpleNamespace(quat=q[[1, 2, 3, 0]]), translation=np.asarray(tvec, np.float64))


def essential_matrix_estimation(points2D1, points2D2, camera1, camera2, estimation_options=None):
    ro = estimation_options or RANSACOptions()
    p1, p2 = np.asarray(points2D1, np.float64), np.asarray(points2D2, np.float64)
    c1, c2 = camera1.oc(), camera2.oc()
    n1, n2 = o.cam_from_img(c1, p1), o.cam_from_img(c2, p2)
    kw = ro.kw()
    # essential_matrix.h:41-46: 0.5 * (e / f1 + e / f2) = (CamFromImgThreshold(e) of both cameras) / 2
    kw["max_error"] = 0.5 * (o.cam_from_img_threshold(c1, ro.max_error) + o.cam_from_img_threshold(c2, ro.max_error))
    r = o.ransac_estimate("E", n1, n2, o.ransac_options(**kw), seed=0)
    if not r["success"]:
        return None
    inl = np.flatnonzero(r["inliers"]).astype(np.uint32)
    wp = o.estimate_two_view_geometry_pose(c1, p1, c2, p2, np.c_[inl, inl], 2, E=r["model"])
    return {"E": r["model"], "cam2_from_cam1": _rigid(wp["qvec"], wp["tvec"]), "num_inliers": r["num_inliers"],
            "inliers": r["inliers"]}


def _tvg(camera1, points1, camera2, points2, matches, options, prior):
    options = options or TwoViewGeometryOptions()
    p1, p2 = np.asarray(points1, np.float64), np.asarray(points2, np.float64)
    m = np.asarray(matches, np.uint32).reshape(-1, 2) if matches is not None else \
        np.stack([np.arange(len(p1))] * 2, axis=1).astype(np.uint32)
    w = o.estimate_two_view_geometry(camera1.oc(prior), p1, camera2.oc(prior), p2, m, options.oo(), seed=0)
    lab = w["inlier_label"]
    inl = np.concatenate([m[lab == k] for k in range(1, int(lab.max()) + 1)]) if lab.max() > 0 else np.zeros((0, 2), np.uint32)
    return SimpleNamespace(config=w["config"], E=w["E"], F=w["F"], H=w["H"], inlier_matches=inl, tri_angle=w["tri_angle"],
                           cam2_from_cam1=_rigid(w["qvec"], w["tvec"]))


def estimate_two_view_geometry(camera1, points1, camera2, points2, matches=None, options=None):
    return _tvg(camera1, points1, camera2, points2, matches, options, None)


def estimate_calibrated_two_view_geometry(camera1, points1, camera2, points2, matches=None, options=None):
    # EstimateCalibratedTwoViewGeometry is entered directly: the dispatch of EstimateTwoViewGeometry (multiple_models,
    # force_H_use, the prior-focal-length test) is not in its way
    import copy
    opts = copy.deepcopy(options or TwoViewGeometryOptions())
    opts.force_H_use, opts.multiple_models = False, False
    return _tvg(camera1, points1, camera2, points2, matches, opts, True)


def squared_sampson_error(points2D1, points2D2, E):
    return list(o.sampson_error(np.asarray(points2D1, np.float64), np.asarray(points2D2, np.float64), np.asarray(E, np.float64)))


def homography_decomposition(H, K1, K2, points1, points2):
    r = o.pose_from_homography(H, K1, K2, points1, points2)
    return dict(R=r["R"], t=r["t"], n=r["n"], points3D=[x for x in r["points3D"]])
