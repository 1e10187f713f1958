"""GPU parity tests of the single-pair estimator entry points (amc_ransac_pairs,
amc_squared_sampson_error, amc_verify_pairs on double-precision points) against the CPU oracle:
what pycolmap's fundamental/homography/essential_matrix_estimation, squared_sampson_error and
estimate_two_view_geometry bind (/root/reference/pycolmap/estimators/*.h).  Bit-exact."""
import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import _capi, synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = _capi.Context(0)
    yield c
    c.close()


def correspondences(sc, jitter_rng=None):
    """Matched points of a synthetic scene as two aligned float64 arrays.  A sub-float32 jitter makes
    sure the double-precision upload path is what is being exercised."""
    p1 = sc["pts1"][sc["matches"][:, 0]].astype(np.float64)
    p2 = sc["pts2"][sc["matches"][:, 1]].astype(np.float64)
    if jitter_rng is not None:
        p1 = p1 + jitter_rng.normal(scale=1e-6, size=p1.shape)
        p2 = p2 + jitter_rng.normal(scale=1e-6, size=p2.shape)
    return p1, p2


def upload_pairs(ctx, pairs, cams=None):
    ctx.reserve_slots(2 * len(pairs))
    for i, (p1, p2) in enumerate(pairs):
        ctx.upload_points_f64(2 * i, p1)
        ctx.upload_points_f64(2 * i + 1, p2)
        if cams is not None:
            for s, cam in ((2 * i, cams[i][0]), (2 * i + 1, cams[i][1])):
                ctx.upload_camera(s, cam["model"], cam["width"], cam["height"], cam["params"], cam.get("prior", False))
    s1 = np.arange(0, 2 * len(pairs), 2, dtype=np.uint32)
    off = np.zeros(len(pairs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p1) for p1, _ in pairs])
    matches = np.concatenate([np.stack([np.arange(len(p1))] * 2, axis=1) for p1, _ in pairs]).astype(np.uint32)
    return s1, s1 + 1, off, matches


RANSAC_KW = dict(max_error=4.0, min_inlier_ratio=0.01, confidence=0.9999, min_num_trials=1000,
                 max_num_trials=100000)  # pycolmap's Python-side RANSACOptions defaults


@pytest.mark.parametrize("kind", ["F", "H"])
def test_ransac_pairs_match_oracle(ctx, kind):
    rng = np.random.default_rng(11 if kind == "F" else 12)
    pairs = []
    for i in range(12):
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(20, 400)), num_outliers=int(rng.integers(0, 200)),
                                  planar=(kind == "H") or bool(i % 3 == 0))
        pairs.append(correspondences(sc, rng))
    pairs.append((np.zeros((3, 2)), np.ones((3, 2))))       # fewer points than the minimal sample
    pairs.append((rng.uniform(0, 1000, (60, 2)), rng.uniform(0, 1000, (60, 2))))  # pure noise
    s1, s2, off, matches = upload_pairs(ctx, pairs)
    rep, mask = ctx.ransac_pairs(kind, s1, s2, off, matches, ransac=RANSAC_KW, seed=0)
    for p, (p1, p2) in enumerate(pairs):
        w = o.ransac_estimate(kind, p1, p2, o.ransac_options(**RANSAC_KW), seed=0)
        assert bool(rep[p]["success"]) == w["success"], p
        assert rep[p]["num_trials"] == w["num_trials"], p
        assert rep[p]["num_inliers"] == w["num_inliers"], p
        np.testing.assert_array_equal(mask[int(off[p]):int(off[p + 1])], w["inliers"], err_msg=str(p))
        np.testing.assert_array_equal(bits(rep[p]["model"]), bits(w["model"]), err_msg=str(p))


def test_essential_ransac_matches_oracle(ctx):
    rng = np.random.default_rng(13)
    pairs, cams, norm = [], [], []
    for i in range(8):
        f1, f2 = float(rng.uniform(700, 1500)), float(rng.uniform(700, 1500))
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(30, 300)), num_outliers=int(rng.integers(0, 120)),
                                  f=f1)
        p1, p2 = correspondences(sc, rng)
        if i % 2 == 0:
            c1 = dict(model="PINHOLE", width=1600, height=1200, params=(f1, f1 * 1.01, 800.0, 600.0))
            c2 = dict(model="SIMPLE_PINHOLE", width=1600, height=1200, params=(f2, 790.0, 610.0))
        else:
            c1 = dict(model="SIMPLE_PINHOLE", width=1600, height=1200, params=(f1, 800.0, 600.0))
            c2 = dict(model="PINHOLE", width=1600, height=1200, params=(f2, f2, 805.0, 595.0))
        pairs.append((p1, p2))
        cams.append((c1, c2))
    s1, s2, off, matches = upload_pairs(ctx, pairs, cams)
    rep, mask = ctx.ransac_pairs("E", s1, s2, off, matches, ransac=RANSAC_KW, seed=0)

    def cam_from_img(c, p):   # Camera::CamFromImg for the two pinhole models
        pr = c["params"]
        if c["model"] == "SIMPLE_PINHOLE":
            return np.stack([(p[:, 0] - pr[1]) / pr[0], (p[:, 1] - pr[2]) / pr[0]], axis=1), pr[0]
        return np.stack([(p[:, 0] - pr[2]) / pr[0], (p[:, 1] - pr[3]) / pr[1]], axis=1), (pr[0] + pr[1]) / 2.0

    for p, ((p1, p2), (c1, c2)) in enumerate(zip(pairs, cams)):
        n1, f1 = cam_from_img(c1, p1)
        n2, f2 = cam_from_img(c2, p2)
        kw = dict(RANSAC_KW)
        kw["max_error"] = (RANSAC_KW["max_error"] / f1 + RANSAC_KW["max_error"] / f2) / 2  # essential_matrix.h:41-46
        w = o.ransac_estimate("E", n1, n2, o.ransac_options(**kw), seed=0)
        assert bool(rep[p]["success"]) == w["success"], p
        assert rep[p]["num_trials"] == w["num_trials"], p
        assert rep[p]["num_inliers"] == w["num_inliers"], p
        np.testing.assert_array_equal(mask[int(off[p]):int(off[p + 1])], w["inliers"], err_msg=str(p))
        np.testing.assert_array_equal(bits(rep[p]["model"]), bits(w["model"]), err_msg=str(p))


def test_squared_sampson_error_bit_exact(ctx):
    rng = np.random.default_rng(14)
    for n in (0, 1, 63, 64, 65, 1000, 100003):
        p1 = rng.uniform(-1, 1, (n, 2))
        p2 = rng.uniform(-1, 1, (n, 2))
        E = rng.normal(size=(3, 3))
        got = ctx.squared_sampson_error(p1, p2, E)
        want = o.sampson_error(p1, p2, E)
        np.testing.assert_array_equal(bits(got), bits(want))
    # known answer: points on the epipolar line have zero error
    E = np.array([[0, -1, 0.2], [1, 0, -0.3], [-0.2, 0.3, 0]], float)   # [t]x
    x1 = np.array([[0.1, 0.2]])
    l2 = E @ np.array([0.1, 0.2, 1.0])
    x2 = np.array([[0.5, -(l2[0] * 0.5 + l2[2]) / l2[1]]])
    assert ctx.squared_sampson_error(x1, x2, E)[0] < 1e-28


def test_two_view_geometry_on_f64_points(ctx):
    """amc_verify_pairs over double-precision points = pycolmap.estimate_two_view_geometry."""
    rng = np.random.default_rng(15)
    scenes, priors = [], []
    for i in range(10):
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(30, 300)), num_outliers=int(rng.integers(0, 100)),
                                  planar=bool(i % 4 == 1), pure_rotation=bool(i % 4 == 2))
        sc["pts1"] = sc["pts1"] + rng.normal(scale=1e-6, size=sc["pts1"].shape)   # not float32-representable
        sc["pts2"] = sc["pts2"] + rng.normal(scale=1e-6, size=sc["pts2"].shape)
        scenes.append(sc)
        priors.append(bool(i % 2))
    ctx.reserve_slots(2 * len(scenes))
    for i, (sc, prior) in enumerate(zip(scenes, priors)):
        for s, pts in ((2 * i, sc["pts1"]), (2 * i + 1, sc["pts2"])):
            ctx.upload_points_f64(s, pts)
            ctx.upload_camera(s, "PINHOLE", sc["width"], sc["height"],
                              (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior)
    s1 = np.arange(0, 2 * len(scenes), 2, dtype=np.uint32)
    off = np.zeros(len(scenes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(sc["matches"]) for sc in scenes])
    matches = np.concatenate([sc["matches"] for sc in scenes])
    tvg, mask, _ = ctx.verify_pairs(s1, s1 + 1, off, matches, _capi.tvg_options(), seed=0)
    for p, (sc, prior) in enumerate(zip(scenes, priors)):
        cam = o.make_camera("PINHOLE", sc["width"], sc["height"],
                            (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=prior)
        w = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"], o.tvg_default_options(), seed=0)
        assert _capi.CONFIG_NAMES[tvg[p]["config"]] == w["config_name"], p
        assert tvg[p]["num_trials"].tolist() == w["trials"], p
        assert tvg[p]["num_inliers"] == w["num_inliers"], p
        np.testing.assert_array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"], err_msg=str(p))
        for k in "EFH":
            np.testing.assert_array_equal(bits(tvg[p][k]), bits(w[k]), err_msg=f"{p} {k}")


def test_python_estimator_api_matches_oracle():
    """The pycolmap-named functions end to end (pybind11 module -> C ABI -> kernel)."""
    import pycolmap_amd as pc
    rng = np.random.default_rng(16)
    sc = synth.two_view_scene(rng, num_inliers=200, num_outliers=80)
    p1, p2 = correspondences(sc, rng)
    ro = pc.RANSACOptions()
    oo = o.ransac_options(max_error=ro.max_error, min_inlier_ratio=ro.min_inlier_ratio, confidence=ro.confidence,
                          min_num_trials=ro.min_num_trials, max_num_trials=ro.max_num_trials)
    for fn, kind, key in ((pc.fundamental_matrix_estimation, "F", "F"), (pc.homography_matrix_estimation, "H", "H")):
        got = fn(p1, p2)
        want = o.ransac_estimate(kind, p1, p2, oo, seed=0)
        assert want["success"] and got is not None
        np.testing.assert_array_equal(bits(got[key]), bits(want["model"]))
        assert got["num_inliers"] == want["num_inliers"]
        np.testing.assert_array_equal(np.array(got["inliers"]), want["inliers"])
    # a failed RANSAC returns None (fewer correspondences than the minimal sample)
    assert pc.fundamental_matrix_estimation(p1[:5], p2[:5]) is None
    assert pc.homography_matrix_estimation(p1[:3], p2[:3]) is None
    # options as a dict, like the reference's implicit dict -> options conversion
    got = pc.homography_matrix_estimation(p1, p2, dict(max_error=2.0))
    want = o.ransac_estimate("H", p1, p2, o.ransac_options(max_error=2.0), seed=0)
    assert (got is None) == (not want["success"])
    if got is not None:
        np.testing.assert_array_equal(bits(got["H"]), bits(want["model"]))

    cam = pc.Camera(model="PINHOLE", width=sc["width"], height=sc["height"],
                    params=[sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0])
    gotE = pc.essential_matrix_estimation(p1, p2, cam, cam)
    n1 = (p1 - [sc["width"] / 2.0, sc["height"] / 2.0]) / sc["f"]
    n2 = (p2 - [sc["width"] / 2.0, sc["height"] / 2.0]) / sc["f"]
    e = (ro.max_error / sc["f"] + ro.max_error / sc["f"]) / 2
    wantE = o.ransac_estimate("E", n1, n2, o.ransac_options(max_error=e, min_inlier_ratio=ro.min_inlier_ratio,
                                                           confidence=ro.confidence, min_num_trials=ro.min_num_trials,
                                                           max_num_trials=ro.max_num_trials), seed=0)
    assert gotE is not None and wantE["success"]
    np.testing.assert_array_equal(bits(gotE["E"]), bits(wantE["model"]))
    np.testing.assert_array_equal(np.array(gotE["inliers"]), wantE["inliers"])
    # cam2_from_cam1: PoseFromEssentialMatrix on the inlier correspondences
    ocamE = o.make_camera("PINHOLE", sc["width"], sc["height"], (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0))
    inl = np.flatnonzero(wantE["inliers"]).astype(np.uint32)
    wp = o.estimate_two_view_geometry_pose(ocamE, p1, ocamE, p2, np.c_[inl, inl], 2, E=wantE["model"])
    rig = gotE["cam2_from_cam1"]
    np.testing.assert_array_equal(bits(rig.rotation.quat), bits(wp["qvec"][[1, 2, 3, 0]]))   # Eigen order x, y, z, w
    np.testing.assert_array_equal(bits(rig.translation), bits(wp["tvec"]))
    assert np.allclose(rig.rotation.matrix(), wp["R"], atol=1e-12) and np.allclose(rig.matrix()[:, 3], wp["tvec"])

    res = pc.squared_sampson_error(n1, n2, gotE["E"])
    np.testing.assert_array_equal(bits(np.array(res)), bits(o.sampson_error(n1, n2, gotE["E"])))

    # estimate_two_view_geometry: explicit matches and the identity default
    for prior in (False, True):
        cam.has_prior_focal_length = prior
        ocam = o.make_camera("PINHOLE", sc["width"], sc["height"],
                             (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=prior)
        # compute_relative_pose, and the same pose through estimate_two_view_geometry_pose afterwards
        gp = pc.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                           pc.TwoViewGeometryOptions(compute_relative_pose=True))
        wpp = o.estimate_two_view_geometry(ocam, sc["pts1"], ocam, sc["pts2"], sc["matches"],
                                           o.tvg_default_options(compute_relative_pose=1))
        assert gp.config.name == wpp["config_name"] and wpp["pose_ok"]
        np.testing.assert_array_equal(bits(gp.cam2_from_cam1.rotation.quat), bits(wpp["qvec"][[1, 2, 3, 0]]))
        np.testing.assert_array_equal(bits(gp.cam2_from_cam1.translation), bits(wpp["tvec"]))
        assert bits(gp.tri_angle) == bits(wpp["tri_angle"]) and gp.tri_angle > 0
        g = pc.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"])
        assert g.tri_angle == 0.0 and np.array_equal(g.cam2_from_cam1.rotation.quat, [0, 0, 0, 1])
        assert pc.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], g) is True
        np.testing.assert_array_equal(bits(g.cam2_from_cam1.rotation.quat), bits(wpp["qvec"][[1, 2, 3, 0]]))
        np.testing.assert_array_equal(bits(g.cam2_from_cam1.translation), bits(wpp["tvec"]))
        assert bits(g.tri_angle) == bits(wpp["tri_angle"]) and g.config.name == wpp["config_name"]
        assert pc.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], pc.TwoViewGeometry()) is False
        # invert(): E^T, F^T, H^-1, swapped match columns, inverse pose; twice = back (up to rounding in H / pose)
        import copy
        gi = copy.deepcopy(g)
        gi.invert()
        np.testing.assert_array_equal(gi.E, g.E.T)
        np.testing.assert_array_equal(gi.F, g.F.T)
        np.testing.assert_array_equal(gi.inlier_matches, g.inlier_matches[:, ::-1])
        assert np.allclose(gi.H @ g.H / (gi.H @ g.H)[2, 2], np.eye(3), atol=1e-9)
        T, Ti = np.vstack([g.cam2_from_cam1.matrix(), [0, 0, 0, 1]]), np.vstack([gi.cam2_from_cam1.matrix(), [0, 0, 0, 1]])
        assert np.allclose(T @ Ti, np.eye(4), atol=1e-12)
        gi.invert()
        np.testing.assert_array_equal(gi.inlier_matches, g.inlier_matches)
        assert np.allclose(gi.cam2_from_cam1.matrix(), g.cam2_from_cam1.matrix(), atol=1e-12)
        g = pc.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"])
        w = o.estimate_two_view_geometry(ocam, sc["pts1"], ocam, sc["pts2"], sc["matches"], o.tvg_default_options())
        assert g.config.name == w["config_name"]
        np.testing.assert_array_equal(g.inlier_matches, sc["matches"][w["inlier_mask"]])
        for k in "EFH":
            np.testing.assert_array_equal(bits(getattr(g, k)), bits(w[k]))
    ident = np.stack([np.arange(len(p1))] * 2, axis=1).astype(np.uint32)
    g = pc.estimate_two_view_geometry(cam, p1, cam, p2)
    w = o.estimate_two_view_geometry(ocam, p1, ocam, p2, ident, o.tvg_default_options())
    assert g.config.name == w["config_name"]
    np.testing.assert_array_equal(g.inlier_matches, ident[w["inlier_mask"]])
    # the calibrated entry point runs E + F + H whatever has_prior_focal_length says
    cam.has_prior_focal_length = False
    gc = pc.estimate_calibrated_two_view_geometry(cam, p1, cam, p2)
    ocam_p = o.make_camera("PINHOLE", sc["width"], sc["height"],
                           (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=True)
    wc = o.estimate_two_view_geometry(ocam_p, p1, ocam_p, p2, ident, o.tvg_default_options())
    assert gc.config.name == wc["config_name"]
    np.testing.assert_array_equal(bits(gc.E), bits(wc["E"]))


def test_multiple_models_through_the_python_api():
    """options.multiple_models: config MULTIPLE, inlier matches = the geometries' lists one after the other."""
    import pycolmap_amd as pc
    rng = np.random.default_rng(17)
    a = synth.two_view_scene(rng, num_inliers=160, num_outliers=30, extra_keypoints=5)
    b = synth.two_view_scene(rng, num_inliers=120, num_outliers=0, extra_keypoints=5)
    pts1 = np.concatenate([a["pts1"], b["pts1"]])
    pts2 = np.concatenate([a["pts2"], b["pts2"]])
    matches = np.concatenate([a["matches"].astype(np.int64),
                              b["matches"].astype(np.int64) + [len(a["pts1"]), len(a["pts2"])]]).astype(np.uint32)
    cam = pc.Camera(model="PINHOLE", width=a["width"], height=a["height"],
                    params=[a["f"], a["f"], a["width"] / 2.0, a["height"] / 2.0])
    g = pc.estimate_two_view_geometry(cam, pts1, cam, pts2, matches, dict(multiple_models=True))
    ocam = o.make_camera("PINHOLE", a["width"], a["height"], (a["f"], a["f"], a["width"] / 2.0, a["height"] / 2.0))
    w = o.estimate_two_view_geometry(ocam, pts1, ocam, pts2, matches, o.tvg_default_options(multiple_models=1))
    assert g.config.name == w["config_name"] == "MULTIPLE"
    want = np.concatenate([matches[w["inlier_label"] == k] for k in range(1, int(w["inlier_label"].max()) + 1)])
    np.testing.assert_array_equal(g.inlier_matches, want)
    assert len(want) >= 250
