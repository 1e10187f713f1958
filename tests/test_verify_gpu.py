"""GPU parity tests of two-view verification: libamc.so's amc_verify_pairs vs the CPU oracle on
identical seeded inputs.  Bit-exact: configs, inlier masks, trial counts AND the bit patterns of
the E/F/H models (FP64, same operation order, no contraction)."""
import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import _capi, synth

pytestmark = pytest.mark.gpu


def bits(a):
    """Bit patterns, with every NaN mapped to one pattern: x86 and gfx950 produce default NaNs of opposite
    sign (0/0 in a degenerate solve), which carries no information."""
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def build_batch(scenes, priors):
    """One image slot pair per scene."""
    slots, cams = [], []
    for sc, prior in zip(scenes, priors):
        cam = dict(model="PINHOLE", width=sc["width"], height=sc["height"],
                   params=(sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=prior)
        slots += [sc["pts1"], sc["pts2"]]
        cams += [cam, cam]
    return slots, cams


def run_both(ctx, scenes, priors, opts_kw=None, seed=0):
    opts_kw = opts_kw or {}
    slots, cams = build_batch(scenes, priors)
    ctx.reserve_slots(len(slots))
    for i, (kp, cam) in enumerate(zip(slots, cams)):
        ctx.upload_keypoints(i, kp.astype(np.float32))
        ctx.upload_camera(i, cam["model"], cam["width"], cam["height"], cam["params"], cam["prior"])
    s1 = np.arange(0, len(slots), 2, dtype=np.uint32)
    s2 = s1 + 1
    off = np.zeros(len(scenes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(sc["matches"]) for sc in scenes])
    matches = np.concatenate([sc["matches"] for sc in scenes]) if scenes else np.zeros((0, 2), np.uint32)
    ransac_kw = opts_kw.get("ransac", {})
    tvg, mask, st = ctx.verify_pairs(s1, s2, off, matches, _capi.tvg_options(**opts_kw), seed=seed)
    okw = {k: v for k, v in opts_kw.items() if k != "ransac"}
    okw.update(ransac_kw)
    want = []
    for sc, prior in zip(scenes, priors):
        cam = o.make_camera("PINHOLE", sc["width"], sc["height"],
                            (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=prior)
        want.append(o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                                 o.tvg_default_options(**okw), seed=seed))
    return tvg, mask, off, want


def assert_pair_equal(p, tvg, mask, off, want):
    g, w = tvg[p], want[p]
    name = _capi.CONFIG_NAMES[g["config"]]
    assert name == w["config_name"], f"pair {p}: {name} != {w['config_name']}"
    assert g["num_trials"].tolist() == w["trials"], f"pair {p} trials {g['num_trials']} vs {w['trials']}"
    assert g["model_inliers"].tolist() == w["inl"], f"pair {p}"
    assert g["num_inliers"] == w["num_inliers"], f"pair {p}"
    np.testing.assert_array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"], err_msg=f"pair {p}")
    for k in "EFH":
        np.testing.assert_array_equal(bits(g[k]), bits(w[k]), err_msg=f"pair {p} model {k}")


def test_uncalibrated_general_and_planar(amc_ctx):
    rng = np.random.default_rng(0)
    scenes = [synth.two_view_scene(rng, num_inliers=300, num_outliers=100),
              synth.two_view_scene(rng, num_inliers=200, num_outliers=150, planar=True),
              synth.two_view_scene(rng, num_inliers=60, num_outliers=40, noise=1.0),
              synth.two_view_scene(rng, num_inliers=0, num_outliers=30)]
    tvg, mask, off, want = run_both(amc_ctx, scenes, [False] * 4)
    for p in range(4):
        assert_pair_equal(p, tvg, mask, off, want)
    names = [_capi.CONFIG_NAMES[c] for c in tvg["config"]]
    assert names[0] == "UNCALIBRATED" and names[1] == "PLANAR_OR_PANORAMIC" and names[3] == "DEGENERATE"


def test_calibrated_with_essential_matrix(amc_ctx):
    rng = np.random.default_rng(1)
    scenes = [synth.two_view_scene(rng, num_inliers=250, num_outliers=90),
              synth.two_view_scene(rng, num_inliers=150, num_outliers=60, planar=True),
              synth.two_view_scene(rng, num_inliers=80, num_outliers=120)]
    tvg, mask, off, want = run_both(amc_ctx, scenes, [True] * 3)
    for p in range(3):
        assert_pair_equal(p, tvg, mask, off, want)
    assert _capi.CONFIG_NAMES[tvg["config"][0]] == "CALIBRATED" and tvg["num_trials"][0][0] > 0


def test_mixed_batch_sizes_and_edge_cases(amc_ctx):
    rng = np.random.default_rng(2)
    scenes = [synth.two_view_scene(rng, num_inliers=ni, num_outliers=no, planar=pl)
              for ni, no, pl in [(20, 5, False), (14, 0, False), (15, 0, False), (700, 300, False),
                                 (33, 31, True), (64, 0, False), (65, 63, False), (128, 128, True)]]
    priors = [False, False, True, False, True, False, True, False]
    tvg, mask, off, want = run_both(amc_ctx, scenes, priors)
    for p in range(len(scenes)):
        assert_pair_equal(p, tvg, mask, off, want)
    assert _capi.CONFIG_NAMES[tvg["config"][1]] == "DEGENERATE"      # 14 matches < min_num_inliers


def test_tiny_match_counts(amc_ctx):
    """0 .. 9 correspondences with min_num_inliers = 0: every RANSAC sees fewer points than, exactly
    as many as, or barely more than its minimal sample."""
    rng = np.random.default_rng(31)
    base = synth.two_view_scene(rng, num_inliers=40, num_outliers=0)
    scenes = []
    for m in (0, 1, 3, 4, 5, 6, 7, 8, 9):
        sc = dict(base)
        sc["matches"] = base["matches"][:m].copy()
        scenes.append(sc)
    for prior in (False, True):
        tvg, mask, off, want = run_both(amc_ctx, scenes, [prior] * len(scenes), dict(min_num_inliers=0))
        for p in range(len(scenes)):
            assert_pair_equal(p, tvg, mask, off, want)


@pytest.mark.parametrize("serial", [False, True])
def test_large_match_counts(amc_ctx, monkeypatch, serial):
    """Pairs whose correspondences do not fit a wave's LDS share (points stay in HBM), and pairs
    whose index arrays need a whole workgroup's LDS (one wave per workgroup), mixed with small ones: three size
    classes in one call - beside each other on two streams (the default), or one after the other."""
    if serial:
        monkeypatch.setenv("AMC_TVG_SERIAL_CLASSES", "1")
    rng = np.random.default_rng(21)
    scenes = [synth.two_view_scene(rng, num_inliers=ni, num_outliers=no, planar=pl, extra_keypoints=10)
              for ni, no, pl in [(3000, 2000, False), (100, 40, False), (6000, 3000, True), (900, 500, False),
                                 (5000, 4500, False)]]
    priors = [True, False, False, True, False]
    tvg, mask, off, want = run_both(amc_ctx, scenes, priors)
    for p in range(len(scenes)):
        assert_pair_equal(p, tvg, mask, off, want)


def test_more_matches_than_the_index_arrays_fit_in_lds(amc_ctx):
    """Round 3's limit: a pair's two 16-bit index arrays (sampler permutation, inlier list: 4 bytes per match) had to
    fit one workgroup's LDS, ~38,000 matches.  Beyond it the "big" builds of the two kernels (tvg_e_big.hip /
    tvg_fh_big.hip) keep the arrays in the wave's global workspace: 50,000 and 65,535 matches (the most 16-bit indices
    name), calibrated and not, general and planar, next to small pairs of the same call - bit for bit the oracle."""
    rng = np.random.default_rng(50)
    scenes = [synth.two_view_scene(rng, num_inliers=46000, num_outliers=4000),
              synth.two_view_scene(rng, num_inliers=200, num_outliers=80),
              synth.two_view_scene(rng, num_inliers=60000, num_outliers=5535, planar=True),
              synth.two_view_scene(rng, num_inliers=39000, num_outliers=3000)]
    assert [len(sc["matches"]) for sc in scenes] == [50000, 280, 65535, 42000]
    for priors in ([True, True, False, False], [False, False, True, True]):
        tvg, mask, off, want = run_both(amc_ctx, scenes, priors)
        for p in range(len(scenes)):
            assert_pair_equal(p, tvg, mask, off, want)
    one_more = synth.two_view_scene(rng, num_inliers=65000, num_outliers=536)
    with pytest.raises(Exception, match="65535"):
        run_both(amc_ctx, [one_more], [False])


def test_watermark_and_option_variants(amc_ctx):
    rng = np.random.default_rng(9)
    w, h, n = 1600, 1200, 80
    x = np.r_[rng.uniform(5, 150, n // 2), rng.uniform(w - 150, w - 5, n // 2)]
    y = rng.uniform(5, 150, n)
    p1 = np.c_[x, y].astype(np.float32).astype(np.float64)
    p2 = (p1 + np.array([3.0, -2.0]) + rng.normal(0, 0.05, size=(n, 2))).astype(np.float32).astype(np.float64)
    wm = dict(pts1=p1, pts2=p2, matches=np.c_[np.arange(n), np.arange(n)].astype(np.uint32),
              width=w, height=h, f=1200.0)
    gen = synth.two_view_scene(rng, num_inliers=200, num_outliers=100)
    tvg, mask, off, want = run_both(amc_ctx, [wm, gen], [False, False])
    assert_pair_equal(0, tvg, mask, off, want)
    assert_pair_equal(1, tvg, mask, off, want)
    assert _capi.CONFIG_NAMES[tvg["config"][0]] == "WATERMARK"
    for kw in [dict(detect_watermark=0), dict(force_H_use=1), dict(min_num_inliers=30, max_H_inlier_ratio=0.5),
               dict(ransac=dict(max_error=2.0, confidence=0.99, min_num_trials=50, max_num_trials=2000))]:
        tvg, mask, off, want = run_both(amc_ctx, [wm, gen], [True, True], kw)
        assert_pair_equal(0, tvg, mask, off, want)
        assert_pair_equal(1, tvg, mask, off, want)


def test_small_max_num_trials_never_overrun_by_the_essential_chunk(amc_ctx):
    """the essential-matrix RANSAC cuts its 64-trial chunks short once the adaptive limit is passed, with a floor
    of 8 trials - which must not carry a chunk past a small user max_num_trials.  Trial caps just above a chunk boundary
    (65 .. 71), min_num_trials at the cap or below it, on scenes where many 5-point samples give no model (mostly
    collinear / repeated points) and on ordinary ones: report.num_trials <= max_num_trials and every field equal to the
    oracle's sequential loop."""
    rng = np.random.default_rng(77)
    scenes = [synth.two_view_scene(rng, num_inliers=120, num_outliers=60),
              synth.two_view_scene(rng, num_inliers=30, num_outliers=200)]
    # degenerate layouts: points on a line / four distinct points repeated (5-point samples mostly rank deficient)
    n = 90
    t = rng.uniform(100, 1400, n)
    line = np.c_[t, 0.5 * t + 50.0]
    rep = np.array([[100.0, 100.0], [900.0, 200.0], [400.0, 800.0], [1200.0, 1000.0]])[rng.integers(0, 4, n)]
    for p1 in (line, rep):
        p2 = p1 + np.array([4.0, -3.0])
        scenes.append(dict(pts1=p1.astype(np.float32).astype(np.float64), pts2=p2.astype(np.float32).astype(np.float64),
                           matches=np.c_[np.arange(n), np.arange(n)].astype(np.uint32), width=1600, height=1200, f=1200.0))
    for mx in (65, 66, 67, 70, 71, 130):
        for mn in (0, 5, mx):
            kw = dict(ransac=dict(min_num_trials=mn, max_num_trials=mx))
            tvg, mask, off, want = run_both(amc_ctx, scenes, [True] * len(scenes), kw)
            for p in range(len(scenes)):
                assert_pair_equal(p, tvg, mask, off, want)
                assert int(tvg["num_trials"][p][0]) <= mx, (mx, mn, p, tvg["num_trials"][p])


def test_watermark_ransac_below_its_trial_cap(amc_ctx):
    """min_num_trials below the watermark RANSAC's trial cap (18 at the default ratios; the C++ RANSACOptions
    default is 0): its adaptive trial count decides when it stops.  Watermark-like scenes with a varying share
    of translated points, so that the count lands on different sides of the cut-offs."""
    rng = np.random.default_rng(17)
    w, h = 1600, 1200
    scenes = []
    for n, frac in ((80, 1.0), (120, 0.9), (200, 0.75), (64, 0.72), (150, 0.6), (90, 0.95)):
        x = np.r_[rng.uniform(5, 150, n // 2), rng.uniform(w - 150, w - 5, n - n // 2)]
        y = rng.uniform(5, 150, n)
        p1 = np.c_[x, y]
        shift = np.where((rng.uniform(size=n) < frac)[:, None], np.array([3.0, -2.0]), rng.uniform(-60, 60, size=(n, 2)))
        p2 = p1 + shift + rng.normal(0, 0.05, size=(n, 2))
        scenes.append(dict(pts1=p1.astype(np.float32).astype(np.float64), pts2=p2.astype(np.float32).astype(np.float64),
                           matches=np.c_[np.arange(n), np.arange(n)].astype(np.uint32), width=w, height=h, f=1200.0))
    seen = set()
    for mt in (0, 1, 2, 5, 17):
        for kw in (dict(ransac=dict(min_num_trials=mt)),
                   dict(ransac=dict(min_num_trials=mt, confidence=0.9, dyn_num_trials_multiplier=1.0)),
                   dict(ransac=dict(min_num_trials=mt), watermark_min_inlier_ratio=0.5)):
            tvg, mask, off, want = run_both(amc_ctx, scenes, [False] * len(scenes), kw)
            for p in range(len(scenes)):
                assert_pair_equal(p, tvg, mask, off, want)
                seen.add((_capi.CONFIG_NAMES[tvg["config"][p]], int(tvg["num_trials"][p][3])))
    assert any(c == "WATERMARK" for c, _ in seen) and len({t for _, t in seen}) >= 3


def test_seed_changes_the_stream_and_is_reproducible(amc_ctx):
    rng = np.random.default_rng(3)
    sc = [synth.two_view_scene(rng, num_inliers=120, num_outliers=120)]
    a = run_both(amc_ctx, sc, [False], seed=0)
    b = run_both(amc_ctx, sc, [False], seed=0)
    c = run_both(amc_ctx, sc, [False], seed=7)
    assert_pair_equal(0, *a)
    assert_pair_equal(0, *c)
    np.testing.assert_array_equal(bits(a[0]["F"][0]), bits(b[0]["F"][0]))
    assert a[0]["num_trials"][0].tolist() != c[0]["num_trials"][0].tolist() or \
        not np.array_equal(bits(a[0]["F"][0]), bits(c[0]["F"][0]))


def test_invalid_inputs_raise(amc_ctx):
    rng = np.random.default_rng(4)
    sc = synth.two_view_scene(rng, num_inliers=30, num_outliers=0)
    amc_ctx.reserve_slots(2)
    amc_ctx.upload_keypoints(0, sc["pts1"].astype(np.float32))
    with pytest.raises(_capi.AmcError) as e:   # slot 1 has no keypoints/camera
        amc_ctx.verify_pairs([0], [1], [0, len(sc["matches"])], sc["matches"])
    assert e.value.code == _capi.AMC_E_STATE
    amc_ctx.upload_keypoints(1, sc["pts2"].astype(np.float32))
    for s in (0, 1):
        amc_ctx.upload_camera(s, "PINHOLE", 1600, 1200, (1200, 1200, 800, 600), False)
    bad = sc["matches"].copy()
    bad[0, 0] = 10 ** 6
    with pytest.raises(_capi.AmcError) as e:
        amc_ctx.verify_pairs([0], [1], [0, len(bad)], bad)
    assert e.value.code == _capi.AMC_E_INVALID
    tvg, mask, _ = amc_ctx.verify_pairs([], [], [0], np.zeros((0, 2), np.uint32))
    assert len(tvg) == 0 and len(mask) == 0


def two_motion_scene(rng, n_a, n_b, n_out):
    """Matches that follow two different rigid motions (plus outliers): keypoints of two scenes glued."""
    a = synth.two_view_scene(rng, num_inliers=n_a, num_outliers=n_out, extra_keypoints=5)
    b = synth.two_view_scene(rng, num_inliers=n_b, num_outliers=0, extra_keypoints=5)
    off1, off2 = len(a["pts1"]), len(a["pts2"])
    sc = dict(a)
    sc["pts1"] = np.concatenate([a["pts1"], b["pts1"]])
    sc["pts2"] = np.concatenate([a["pts2"], b["pts2"]])
    mb = b["matches"].astype(np.int64) + [off1, off2]
    m = np.concatenate([a["matches"].astype(np.int64), mb])
    sc["matches"] = m[rng.permutation(len(m))].astype(np.uint32)
    return sc


def test_multiple_models(amc_ctx):
    """TwoViewGeometryOptions.multiple_models = EstimateMultipleTwoViewGeometries: rounds on the matches
    the previous rounds left over.  The mask byte is 1 + the index of the geometry a match belongs to."""
    rng = np.random.default_rng(61)
    scenes = [two_motion_scene(rng, 200, 120, 40), two_motion_scene(rng, 150, 150, 0),
              synth.two_view_scene(rng, num_inliers=180, num_outliers=60),            # one model only
              synth.two_view_scene(rng, num_inliers=10, num_outliers=100),            # nothing: DEGENERATE
              two_motion_scene(rng, 90, 60, 200)]
    for priors, kw in (([False] * 5, dict(multiple_models=1)),
                       ([True, False, True, False, True], dict(multiple_models=1, multiple_ignore_watermark=0)),
                       ([False] * 5, dict(multiple_models=1, min_num_inliers=40))):
        slots, cams = build_batch(scenes, priors)
        amc_ctx.reserve_slots(len(slots))
        for i, (kp, cam) in enumerate(zip(slots, cams)):
            amc_ctx.upload_keypoints(i, kp.astype(np.float32))
            amc_ctx.upload_camera(i, cam["model"], cam["width"], cam["height"], cam["params"], cam["prior"])
        s1 = np.arange(0, len(slots), 2, dtype=np.uint32)
        off = np.zeros(len(scenes) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(sc["matches"]) for sc in scenes])
        matches = np.concatenate([sc["matches"] for sc in scenes])
        tvg, mask, st = amc_ctx.verify_pairs(s1, s1 + 1, off, matches, _capi.tvg_options(**kw), seed=0)
        names = []
        for p, (sc, prior) in enumerate(zip(scenes, priors)):
            cam = o.make_camera("PINHOLE", sc["width"], sc["height"],
                                (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=prior)
            w = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                             o.tvg_default_options(**kw), seed=0)
            g = tvg[p]
            names.append(_capi.CONFIG_NAMES[g["config"]])
            assert names[-1] == w["config_name"], p
            assert g["num_inliers"] == w["num_inliers"], p
            np.testing.assert_array_equal(st["inlier_labels"][int(off[p]):int(off[p + 1])], w["inlier_label"], err_msg=str(p))
            for k in "EFH":
                np.testing.assert_array_equal(bits(g[k]), bits(w[k]), err_msg=f"{p} {k}")
            if names[-1] != "MULTIPLE":
                assert g["num_trials"].tolist() == w["trials"], p
        assert "MULTIPLE" in names and "DEGENERATE" in names
        assert any(n not in ("MULTIPLE", "DEGENERATE") for n in names)


def test_non_finite_and_degenerate_inputs(amc_ctx):
    """NaN / inf / huge / tiny / coincident / collinear keypoints: every loop in the kernel is bounded, and
    the results still equal the oracle's (NaNs compare false on both sides)."""
    rng = np.random.default_rng(71)
    base = synth.two_view_scene(rng, num_inliers=120, num_outliers=40)

    def variant(f):
        sc = dict(base)
        sc["pts1"], sc["pts2"] = base["pts1"].copy(), base["pts2"].copy()
        f(sc)
        with np.errstate(all="ignore"):   # what the float32 keypoint blob would hold
            sc["pts1"] = sc["pts1"].astype(np.float32).astype(np.float64)
            sc["pts2"] = sc["pts2"].astype(np.float32).astype(np.float64)
        return sc

    def nan_point(sc): sc["pts1"][3, 0] = np.nan
    def inf_point(sc): sc["pts2"][5, 1] = np.inf
    def huge(sc): sc["pts1"] *= 1e30
    def tiny(sc): sc["pts1"] *= 1e-30; sc["pts2"] *= 1e-30
    def all_same(sc): sc["pts1"][:] = 100.0; sc["pts2"][:] = 200.0
    def collinear(sc): sc["pts1"][:, 1] = 50.0; sc["pts2"][:, 1] = 60.0
    def all_nan(sc): sc["pts1"][:] = np.nan

    scenes = [variant(f) for f in (nan_point, inf_point, huge, tiny, all_same, collinear, all_nan)]
    for prior in (False, True):
        tvg, mask, off, want = run_both(amc_ctx, scenes, [prior] * len(scenes))
        for p in range(len(scenes)):
            assert_pair_equal(p, tvg, mask, off, want)


def test_golden_fixture(amc_ctx):
    """The HIP path against the committed fixture (tests/golden/tvg_golden_v3.npz): every case, with and
    without compute_relative_pose, option overrides included.  No oracle call in this test."""
    import tvg_golden
    cases = list(tvg_golden.cases())
    amc_ctx.reserve_slots(2 * len(cases))
    for i, c in enumerate(cases):
        amc_ctx.upload_points_f64(2 * i, c["pts1"])
        amc_ctx.upload_points_f64(2 * i + 1, c["pts2"])
        amc_ctx.upload_camera(2 * i, c["cam1"][0], 1600, 1200, c["cam1"][1], c["prior"])
        amc_ctx.upload_camera(2 * i + 1, c["cam2"][0], 1600, 1200, c["cam2"][1], c["prior"])
    ransac_keys = {"max_error", "min_inlier_ratio", "confidence", "dyn_num_trials_multiplier", "min_num_trials",
                   "max_num_trials"}
    for i, c in enumerate(cases):
        for pose in (0, 1):
            kw = {k: v for k, v in c["opts"].items() if k not in ransac_keys}
            kw["ransac"] = {k: v for k, v in c["opts"].items() if k in ransac_keys}
            kw["compute_relative_pose"] = pose
            off = np.array([0, len(c["matches"])], dtype=np.uint64)
            tvg, mask, st = amc_ctx.verify_pairs([2 * i], [2 * i + 1], off, c["matches"], _capi.tvg_options(**kw), seed=0)
            w = c["want"][pose]
            tag = f"case {i} pose {pose}"
            g = tvg[0]
            assert int(g["config"]) == w["config"], tag
            assert g["num_trials"].tolist() == w["trials"] and g["model_inliers"].tolist() == w["inl"], tag
            np.testing.assert_array_equal(mask, w["mask"], err_msg=tag)
            for f in "EFH":
                np.testing.assert_array_equal(tvg_golden.bits(g[f]), w[f], err_msg=f"{tag} {f}")
            if pose:
                q = st["pose"][0]
                assert int(q["num_points3D"]) == w["points3D"], tag
                assert tvg_golden.bits(q["tri_angle"])[0] == w["tri_angle"][0], tag
                for f in ("qvec", "tvec", "R"):
                    np.testing.assert_array_equal(tvg_golden.bits(q[f]), w[f], err_msg=f"{tag} {f}")


@pytest.mark.parametrize("pose", [0, 1])
def test_fused_match_verify_equals_the_two_calls(amc_ctx, monkeypatch, pose):
    """amc_match_verify_pairs (the verification kernel reads the matches where the matcher left them in HBM) against
    amc_match_pairs followed by amc_verify_pairs, on a scene whose pairs range from no overlap to dense overlap;
    also with the matcher forced into many small batches (the resident table is appended batch after batch)."""
    rng = np.random.default_rng(77)
    cam = ("SIMPLE_RADIAL", synth.EXAMPLE_CAMERAS["SIMPLE_RADIAL"])
    # two unrelated scenes: pairs across them share nothing (a handful of chance matches: DEGENERATE)
    images = synth.multiview_scene(rng, num_images=5, n_feats=640, camera=cam if pose else None) + \
        synth.multiview_scene(rng, num_images=4, n_feats=640, camera=cam if pose else None)
    amc_ctx.reserve_slots(len(images))
    for k, im in enumerate(images):
        amc_ctx.upload_descriptors(k, im["descriptors"])
        amc_ctx.upload_keypoints(k, im["keypoints"])
        amc_ctx.upload_camera(k, im["model"], im["width"], im["height"], im["params"], True)
    s1, s2 = synth.exhaustive_pairs(len(images))
    s1, s2 = np.r_[s1, s2[:5]], np.r_[s2, s1[:5]]          # a few pairs in swapped order as well
    opts = _capi.tvg_options(compute_relative_pose=pose)
    off, m, _ = amc_ctx.match_pairs(s1, s2)
    tvg, mask, st = amc_ctx.verify_pairs(s1, s2, off, m, opts, seed=0)
    assert (tvg["config"] == 1).any() and (tvg["config"] >= 2).sum() >= 6     # DEGENERATE (too few matches) and real ones
    # the shipped order (host sides interleaved, one slice behind the last batch), then with many small batches; the
    # stages fully behind each other (AMC_PIPELINE_SERIAL); the device-interleaved variant kept for the A/B - slices
    # beside the next batch's scan, more batches than slice slots (the rest joins the last slice), the scan leaving 16
    # CUs to them
    # ... and a call that fits one batch cut in two (what calls of >= 4096 pairs get: AMC_HOOK_SPLIT=2 takes this small one there)
    for batch_entries, variant in ((None, {}), (None, {"AMC_HOOK_SPLIT": "2"}), (None, {"AMC_HOOK_SPLIT": "0"}),
                                   ("4096", {}), ("4096", {"AMC_PIPELINE_SERIAL": "1"}),
                                   ("4096", {"AMC_PIPELINE_INTERLEAVE": "1"}),
                                   ("20000", {"AMC_PIPELINE_INTERLEAVE": "1", "AMC_VERIFY_CUS": "16"})):
        if batch_entries:
            monkeypatch.setenv("AMC_MATCH_BATCH_ENTRIES", batch_entries)
        for k in ("AMC_PIPELINE_SERIAL", "AMC_PIPELINE_INTERLEAVE", "AMC_VERIFY_CUS", "AMC_HOOK_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in variant.items():
            monkeypatch.setenv(k, v)
        foff, fm, fmst, ftvg, fmask, fst = amc_ctx.match_verify_pairs(s1, s2, opts, seed=0)
        if variant.get("AMC_HOOK_SPLIT") == "2":
            split_launches = fmst["match_kernel_launches"]
        elif variant.get("AMC_HOOK_SPLIT") == "0":
            assert split_launches > fmst["match_kernel_launches"]      # the cut call really ran as two batches
        np.testing.assert_array_equal(foff, off)
        np.testing.assert_array_equal(fm, m)
        np.testing.assert_array_equal(fmask, mask)
        for f in ("config", "num_inliers", "num_trials", "model_inliers"):
            np.testing.assert_array_equal(ftvg[f], tvg[f], err_msg=f)
        for f in "EFH":
            np.testing.assert_array_equal(bits(ftvg[f]), bits(tvg[f]), err_msg=f)
        assert fst["work"] == st["work"] and sum(st["work"]) > 0
        if pose:
            for f in ("ok", "config", "num_points3D"):
                np.testing.assert_array_equal(fst["pose"][f], st["pose"][f])
            for f in ("qvec", "tvec", "R", "tri_angle"):
                np.testing.assert_array_equal(bits(fst["pose"][f]), bits(st["pose"][f]), err_msg=f)
    # empty list
    z = np.zeros(0, np.uint32)
    foff, fm, _, ftvg, fmask, _ = amc_ctx.match_verify_pairs(z, z, opts)
    assert len(foff) == 1 and len(fm) == 0 and len(ftvg) == 0


def test_result_views_equal_copies_and_outlive_the_call(amc_ctx):
    """verify_pairs(copy=False): the arrays are views of the library's result, released with the last of them."""
    import gc
    rng = np.random.default_rng(77)
    sc = synth.two_view_scene(rng, num_inliers=120, num_outliers=50)
    amc_ctx.reserve_slots(2)
    for j, pts in enumerate((sc["pts1"], sc["pts2"])):
        amc_ctx.upload_keypoints(j, pts.astype(np.float32))
        amc_ctx.upload_camera(j, "PINHOLE", sc["width"], sc["height"], (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), True)
    off = np.array([0, len(sc["matches"])], np.uint64)
    a = amc_ctx.verify_pairs([0], [1], off, sc["matches"], _capi.tvg_options())
    b = amc_ctx.verify_pairs([0], [1], off, sc["matches"], _capi.tvg_options(), copy=False)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and a[1].dtype == b[1].dtype == np.bool_
    part = b[1][3:40]
    want = a[1][3:40].copy()
    del b
    gc.collect()
    np.testing.assert_array_equal(part, want)      # the slice keeps the result alive


def test_verification_on_uploaded_resident_matches(amc_ctx):
    """amc_upload_matches + amc_verify_pairs(matches = NULL): the match rows stay in device memory like descriptors and
    keypoints do, a verification call (and a second one with other options, and one with the relative pose) reads them
    there.  Same results as the call that takes the rows over PCIe - and as the oracle; the table's size is checked."""
    rng = np.random.default_rng(83)
    scenes = [synth.two_view_scene(rng, num_inliers=int(n), num_outliers=int(k), planar=pl, extra_keypoints=5)
              for n, k, pl in ((200, 80, False), (60, 10, True), (9, 30, False), (300, 0, False), (3, 2, False), (150, 150, False))]
    priors = [True, False, True, True, False, False]
    tvg, mask, off, want = run_both(amc_ctx, scenes, priors)          # rows over PCIe (uploads slots, checks nothing yet)
    for p in range(len(scenes)):
        assert_pair_equal(p, tvg, mask, off, want)
    s1 = np.arange(0, 2 * len(scenes), 2, dtype=np.uint32)
    matches = np.concatenate([sc["matches"] for sc in scenes])
    assert amc_ctx.upload_matches(matches) == len(matches)
    ptr, n = amc_ctx.resident_matches()
    assert ptr != 0 and n == len(matches)
    for kw in ({}, dict(compute_relative_pose=1), dict(ransac=dict(max_error=2.0, confidence=0.99))):
        a = amc_ctx.verify_pairs(s1, s1 + 1, off, matches, _capi.tvg_options(**kw))
        assert amc_ctx.upload_matches(matches) == len(matches)     # (a host-rows call may reuse the table's memory)
        b = amc_ctx.verify_pairs(s1, s1 + 1, off, None, _capi.tvg_options(**kw))
        c = amc_ctx.verify_pairs(s1, s1 + 1, off, None, _capi.tvg_options(**kw), copy=False)   # the table is still there
        assert a[0].tobytes() == b[0].tobytes() == c[0].tobytes()
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[1], c[1])
        if kw.get("compute_relative_pose"):
            assert a[2]["pose"].tobytes() == b[2]["pose"].tobytes()
        del c
    # a sub-list of the pairs does not match the table's size; multiple_models needs the rows on the host
    with pytest.raises(_capi.AmcError):
        amc_ctx.verify_pairs(s1[:2], s1[:2] + 1, off[:3], None, _capi.tvg_options())
    with pytest.raises(_capi.AmcError):
        amc_ctx.verify_pairs(s1, s1 + 1, off, None, _capi.tvg_options(multiple_models=1))
    # an empty upload empties the table
    assert amc_ctx.upload_matches(np.zeros((0, 2), np.uint32)) == 0
    assert amc_ctx.resident_matches() == (0, 0)
    with pytest.raises(_capi.AmcError):
        amc_ctx.verify_pairs(s1, s1 + 1, off, None, _capi.tvg_options())


@pytest.mark.parametrize("hook", ["AMC_TVG_NO_S32", "AMC_TVG_EXACT_COUNT", "AMC_TVG_SLOW_SAMPLER"])
def test_alternative_counting_and_sampling_paths_agree_with_the_oracle(amc_ctx, monkeypatch, hook):
    """The counting loops have three implementations per estimator - packed-FP32 filter, division-free FP64 test,
    reference residual - and the sampler two (chunk-parallel, draw by draw); which one runs depends on the pair
    (coordinate / max_error ratio, a rejected draw).  The hooks force the slower ones: all must reproduce the oracle
    bit for bit, like the default path (checked by every other test of this file)."""
    rng = np.random.default_rng(31)
    scenes = [synth.two_view_scene(rng, num_inliers=220, num_outliers=120),
              synth.two_view_scene(rng, num_inliers=150, num_outliers=200, planar=True),
              synth.two_view_scene(rng, num_inliers=90, num_outliers=35, noise=1.2),
              synth.two_view_scene(rng, num_inliers=300, num_outliers=80, pure_rotation=True),
              synth.two_view_scene(rng, num_inliers=33, num_outliers=31)]
    monkeypatch.setenv(hook, "1")
    for priors in ([True] * 5, [False] * 5):
        tvg, mask, off, want = run_both(amc_ctx, scenes, priors, seed=3)
        for p in range(len(scenes)):
            assert_pair_equal(p, tvg, mask, off, want)


def test_large_coordinates_take_the_fp64_counting_loops(amc_ctx):
    """Coordinates of 30,000 px with a 0.5 px threshold: (largest coordinate / max_error)^2 > 1e8, so the F / E loops
    leave the packed-FP32 Sampson filter for the FP64 test - same results as the oracle."""
    rng = np.random.default_rng(32)
    scenes = []
    for _ in range(3):
        sc = synth.two_view_scene(rng, num_inliers=160, num_outliers=90, noise=0.2)
        k = 30000.0 / sc["width"]
        for key in ("pts1", "pts2"):
            sc[key] = (sc[key] * k).astype(np.float32).astype(np.float64)   # (keypoints are float32 blobs)
        sc["width"], sc["height"], sc["f"] = int(sc["width"] * k), int(sc["height"] * k), sc["f"] * k
        scenes.append(sc)
    for priors in ([True] * 3, [False] * 3):
        tvg, mask, off, want = run_both(amc_ctx, scenes, priors, opts_kw={"ransac": {"max_error": 0.5}}, seed=1)
        for p in range(3):
            assert_pair_equal(p, tvg, mask, off, want)


@pytest.mark.parametrize("slices", [2, 3, 7])
def test_sliced_verification_equals_one_slice(amc_ctx, monkeypatch, slices):
    """A verification call cut into slices (round 6: E launches on one stream, F/H launches on another, slice k's F/H
    beside slice k + 1's E; per-slice tables, masks and workspaces; pairs no kernel looks at packed as DEGENERATE) returns
    what one slice returns, bit for bit - a pair's result does not depend on the slice it ran in - and what the oracle
    says.  Mixed batch: trivial pairs (< 15 matches) between real ones, a pair large enough for the class that runs on
    the aux stream, uncalibrated pairs that skip the E kernel."""
    rng = np.random.default_rng(31)
    scenes, priors = [], []
    for k in range(23):
        if k % 5 == 2:
            scenes.append(synth.two_view_scene(rng, num_inliers=int(rng.integers(0, 8)), num_outliers=int(rng.integers(0, 6))))
        elif k == 11:
            scenes.append(synth.two_view_scene(rng, num_inliers=2100, num_outliers=700))
        else:
            scenes.append(synth.two_view_scene(rng, num_inliers=int(rng.integers(20, 260)), num_outliers=int(rng.integers(10, 120)),
                                               planar=bool(k % 4 == 0)))
        priors.append(k % 3 != 0)
    monkeypatch.setenv("AMC_TVG_SLICES", "1")
    tvg1, mask1, off, want = run_both(amc_ctx, scenes, priors)
    for p in range(len(scenes)):
        assert_pair_equal(p, tvg1, mask1, off, want)
    assert (tvg1["config"] == 1).sum() >= 4 and (tvg1["config"] >= 2).sum() >= 12
    monkeypatch.setenv("AMC_TVG_SLICES", str(slices))
    monkeypatch.setenv("AMC_TVG_MIN_PER_SLICE", "1")
    slots, cams = build_batch(scenes, priors)
    s1 = np.arange(0, len(slots), 2, dtype=np.uint32)
    matches = np.concatenate([sc["matches"] for sc in scenes])
    for opts in (_capi.tvg_options(), _capi.tvg_options(compute_relative_pose=1)):
        if opts.compute_relative_pose:
            monkeypatch.setenv("AMC_TVG_SLICES", "1")
            want_tvg, want_mask, want_st = amc_ctx.verify_pairs(s1, s1 + 1, off, matches, opts, seed=0)
            monkeypatch.setenv("AMC_TVG_SLICES", str(slices))
        else:
            want_tvg, want_mask, want_st = tvg1, mask1, None
        tvg, mask, st = amc_ctx.verify_pairs(s1, s1 + 1, off, matches, opts, seed=0)
        assert tvg.tobytes() == want_tvg.tobytes() or all(
            np.array_equal(bits(tvg[f]), bits(want_tvg[f])) if f in "EFH" else np.array_equal(tvg[f], want_tvg[f]) for f in tvg.dtype.names)
        np.testing.assert_array_equal(mask, want_mask)
        if want_st is not None:
            assert st["work"] == want_st["work"]
            for f in ("qvec", "tvec", "tri_angle"):
                np.testing.assert_array_equal(bits(st["pose"][f]), bits(want_st["pose"][f]), err_msg=f)


def test_a_short_sample_stream_is_relaunched_on_a_longer_one(monkeypatch):
    """A RANSAC that runs off the sample-stream table flags it and the call lays out a table twice as long and runs every
    slice again (in production only a streak of Lemire rejections gets there).  AMC_TVG_STREAM_SHORT builds the first
    table a quarter as long as the trial caps need: pairs with few inliers run to the cap, overrun, and the relaunch -
    two slices here - must return exactly what a full-length table returns."""
    rng = np.random.default_rng(41)
    scenes = [synth.two_view_scene(rng, num_inliers=int(rng.integers(16, 40)), num_outliers=int(rng.integers(150, 260))) for _ in range(9)]
    scenes += [synth.two_view_scene(rng, num_inliers=150, num_outliers=30) for _ in range(3)]
    priors = [True] * len(scenes)
    with _capi.Context(0) as ctx:
        tvg, mask, off, want = run_both(ctx, scenes, priors, seed=5)
    assert max(int(t["num_trials"][1]) for t in tvg) > 4000          # some F RANSAC ran (nearly) to its cap
    monkeypatch.setenv("AMC_TVG_STREAM_SHORT", "1")
    monkeypatch.setenv("AMC_TVG_SLICES", "2")
    monkeypatch.setenv("AMC_TVG_MIN_PER_SLICE", "1")
    with _capi.Context(0) as ctx:                                      # (a fresh context: no table of the full length around)
        tvg2, mask2, off2, _ = run_both(ctx, scenes, priors, seed=5)
    for p in range(len(scenes)):
        assert_pair_equal(p, tvg2, mask2, off2, want)
    np.testing.assert_array_equal(mask2, mask)
