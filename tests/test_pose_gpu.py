"""GPU parity tests of the relative pose (pose.hip: amc_pose_pairs and amc_verify_pairs with
compute_relative_pose) against the oracle's EstimateTwoViewGeometryPose on identical inputs.
Bit-exact: R, t, quaternion, tri_angle, the number of points in front of both cameras, config."""
import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import _capi, synth

pytestmark = pytest.mark.gpu

CAM_A = dict(model="PINHOLE", width=1600, height=1200, params=(1200.0, 1190.0, 800.0, 600.0))
CAM_B = dict(model="SIMPLE_PINHOLE", width=1600, height=1200, params=(1210.0, 801.0, 599.0))


def bits(a):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def ocam(c, prior=True):
    return o.make_camera(c["model"], c["width"], c["height"], c["params"], prior=prior)


def assert_pose_equal(got, want, tag=""):
    assert bool(got["ok"]) == want["pose_ok"], tag
    assert int(got["config"]) == want["config"], f"{tag}: {got['config']} vs {want['config_name']}"
    assert int(got["num_points3D"]) == want["num_points3D"], tag
    np.testing.assert_array_equal(bits(got["R"]), bits(want["R"]), err_msg=f"{tag} R")
    np.testing.assert_array_equal(bits(got["tvec"]), bits(want["tvec"]), err_msg=f"{tag} t")
    np.testing.assert_array_equal(bits(got["qvec"]), bits(want["qvec"]), err_msg=f"{tag} q")
    assert bits(got["tri_angle"]) == bits(want["tri_angle"]), f"{tag}: {got['tri_angle']} vs {want['tri_angle']}"


def test_pose_pairs_all_configs_bit_exact(amc_ctx):
    """A batch of general / planar / panoramic scenes with mixed cameras, every config value and small
    correspondence counts (empty, one, two, three) through amc_pose_pairs."""
    rng = np.random.default_rng(2024)
    kinds = [dict(), dict(planar=True), dict(pure_rotation=True, noise=0.05)]
    scenes, cams, geoms = [], [], []
    for i in range(9):
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(30, 400)), num_outliers=int(rng.integers(0, 80)),
                                  **kinds[i % 3])
        c1, c2 = CAM_A, (CAM_B if i % 2 else CAM_A)
        r = o.estimate_two_view_geometry(ocam(c1), sc["pts1"], ocam(c2), sc["pts2"], sc["matches"])
        scenes.append(sc)
        cams.append((c1, c2))
        geoms.append(r)
    amc_ctx.reserve_slots(2 * len(scenes))
    for i, (sc, (c1, c2)) in enumerate(zip(scenes, cams)):
        amc_ctx.upload_points_f64(2 * i, sc["pts1"])
        amc_ctx.upload_points_f64(2 * i + 1, sc["pts2"])
        amc_ctx.upload_camera(2 * i, c1["model"], c1["width"], c1["height"], c1["params"], True)
        amc_ctx.upload_camera(2 * i + 1, c2["model"], c2["width"], c2["height"], c2["params"], True)
    jobs = []  # (scene, config, inlier matches)
    for i, (sc, r) in enumerate(zip(scenes, geoms)):
        m = sc["matches"][r["inlier_mask"]]
        for cfg in range(9):
            jobs.append((i, cfg, m))
        for n in (0, 1, 2, 3, 64, 65):
            jobs.append((i, 2, m[:n]))
            jobs.append((i, 6, m[:n]))
    s1 = np.array([2 * j[0] for j in jobs], dtype=np.uint32)
    off = np.zeros(len(jobs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(j[2]) for j in jobs])
    mm = np.concatenate([j[2] for j in jobs])
    cfgs = np.array([j[1] for j in jobs], dtype=np.int32)
    E = np.stack([geoms[j[0]]["E"] for j in jobs])
    H = np.stack([geoms[j[0]]["H"] for j in jobs])
    got = amc_ctx.pose_pairs(s1, s1 + 1, off, mm, cfgs, E, H)
    seen = set()
    for k, (i, cfg, m) in enumerate(jobs):
        c1, c2 = cams[i]
        want = o.estimate_two_view_geometry_pose(ocam(c1), scenes[i]["pts1"], ocam(c2), scenes[i]["pts2"], m, cfg,
                                                 E=geoms[i]["E"], H=geoms[i]["H"])
        assert_pose_equal(got[k], want, f"job {k} scene {i} cfg {cfg} n {len(m)}")
        seen.add(want["config_name"])
    assert {"CALIBRATED", "UNCALIBRATED", "PLANAR", "PANORAMIC"} <= seen


def test_verify_with_relative_pose_bit_exact(amc_ctx):
    """TwoViewGeometryOptions.compute_relative_pose through amc_verify_pairs: the estimation as before,
    then the pose; PLANAR_OR_PANORAMIC resolves to PLANAR / PANORAMIC in tvg.config as well."""
    rng = np.random.default_rng(31)
    scenes, priors = [], []
    for i in range(12):
        kind = [dict(), dict(planar=True), dict(pure_rotation=True, noise=0.05), dict(num_inliers=0, num_outliers=30)][i % 4]
        kw = dict(num_inliers=int(rng.integers(40, 350)), num_outliers=int(rng.integers(0, 120)))
        kw.update(kind)
        scenes.append(synth.two_view_scene(rng, **kw))
        priors.append(i % 3 != 2)
    amc_ctx.reserve_slots(2 * len(scenes))
    for i, sc in enumerate(scenes):
        for s, pts in ((2 * i, sc["pts1"]), (2 * i + 1, sc["pts2"])):
            amc_ctx.upload_keypoints(s, pts.astype(np.float32))
            amc_ctx.upload_camera(s, CAM_A["model"], CAM_A["width"], CAM_A["height"], CAM_A["params"], priors[i])
    s1 = np.arange(0, 2 * len(scenes), 2, dtype=np.uint32)
    off = np.zeros(len(scenes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(sc["matches"]) for sc in scenes])
    mm = np.concatenate([sc["matches"] for sc in scenes])
    for multiple in (0, 1):
        tvg, mask, st = amc_ctx.verify_pairs(s1, s1 + 1, off, mm,
                                             _capi.tvg_options(compute_relative_pose=1, multiple_models=multiple), seed=0)
        assert "pose" in st and len(st["pose"]) == len(scenes)
        names = set()
        for p, sc in enumerate(scenes):
            cam = ocam(CAM_A, priors[p])
            w = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                             o.tvg_default_options(compute_relative_pose=1, multiple_models=multiple), seed=0)
            assert _capi.CONFIG_NAMES[tvg[p]["config"]] == w["config_name"], f"pair {p}"
            np.testing.assert_array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"], err_msg=f"pair {p}")
            if w["config_name"] != "MULTIPLE":
                for k in "EFH":
                    np.testing.assert_array_equal(bits(tvg[p][k]), bits(w[k]), err_msg=f"pair {p} {k}")
            assert_pose_equal(st["pose"][p], w, f"pair {p} multiple={multiple}")
            names.add(w["config_name"])
        if not multiple:
            assert {"CALIBRATED", "UNCALIBRATED", "PLANAR", "DEGENERATE"} <= names, names
            assert "PLANAR_OR_PANORAMIC" not in names
    # without the option the result carries no poses and PLANAR_OR_PANORAMIC stays
    tvg, mask, st = amc_ctx.verify_pairs(s1, s1 + 1, off, mm, _capi.tvg_options(), seed=0)
    assert "pose" not in st and "PLANAR_OR_PANORAMIC" in {_capi.CONFIG_NAMES[c] for c in tvg["config"]}


def test_pose_many_inliers_and_planted_motion(amc_ctx):
    """Thousands of inliers (the median selection runs over many 64-wide chunks) and the result is the
    planted motion."""
    rng = np.random.default_rng(8)
    sc = synth.two_view_scene(rng, num_inliers=6000, num_outliers=500, noise=0.3)
    amc_ctx.reserve_slots(2)
    amc_ctx.upload_keypoints(0, sc["pts1"].astype(np.float32))
    amc_ctx.upload_keypoints(1, sc["pts2"].astype(np.float32))
    cam = ("PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0))
    amc_ctx.upload_camera(0, *cam, True)
    amc_ctx.upload_camera(1, *cam, True)
    off = np.array([0, len(sc["matches"])], dtype=np.uint64)
    tvg, mask, st = amc_ctx.verify_pairs([0], [1], off, sc["matches"], _capi.tvg_options(compute_relative_pose=1), seed=0)
    oc = o.make_camera("PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0), prior=True)
    w = o.estimate_two_view_geometry(oc, sc["pts1"], oc, sc["pts2"], sc["matches"],
                                     o.tvg_default_options(compute_relative_pose=1), seed=0)
    assert w["config_name"] == "CALIBRATED" and w["num_points3D"] > 5000
    assert_pose_equal(st["pose"][0], w, "large")
    R = st["pose"][0]["R"]
    assert np.degrees(np.arccos(np.clip((np.trace(R.T @ sc["R"]) - 1) / 2, -1, 1))) < 0.3
    assert float(st["pose"][0]["tvec"] @ (sc["t"] / np.linalg.norm(sc["t"]))) > 0.999


def test_pose_rejects_unsupported_input(amc_ctx):
    rng = np.random.default_rng(1)
    sc = synth.two_view_scene(rng, num_inliers=50, num_outliers=0)
    amc_ctx.reserve_slots(2)
    amc_ctx.upload_points_f64(0, sc["pts1"])
    amc_ctx.upload_points_f64(1, sc["pts2"])
    off = np.array([0, len(sc["matches"])], dtype=np.uint64)
    with pytest.raises(_capi.AmcError):   # no cameras uploaded
        amc_ctx.pose_pairs([0], [1], off, sc["matches"], [2], [sc["E_true"]])
    amc_ctx.upload_camera(0, *("PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0)), True)
    amc_ctx.upload_camera(1, *("PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0)), True)
    bad = sc["matches"].copy()
    bad[3, 1] = 10 ** 6
    with pytest.raises(_capi.AmcError):   # match index past the keypoints
        amc_ctx.pose_pairs([0], [1], off, bad, [2], [sc["E_true"]])
    got = amc_ctx.pose_pairs(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint64),
                             np.zeros((0, 2), np.uint32), [], None)
    assert len(got) == 0
