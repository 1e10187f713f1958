"""CPU tests of the two-view-geometry oracle (oracle/tvg_oracle.cc): independent restatements
(numpy / LAPACK / a pure-Python PRNG stream), known answers, and planted-model scenes whose
inlier sets do not depend on solver numerics."""
import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import synth


# ---------------------------------------------------------------------------------- V0: PRNG ----
def mt19937_raw(seed, n):
    """std::mt19937(seed) raw outputs via numpy's legacy init_genrand seeding."""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    return [int(x) for x in bg.random_raw(n)]


class PyPrng:
    """libstdc++ >= 11 uniform_int_distribution<uint32_t> over mt19937 (Lemire's nearly
    divisionless method, /usr/include/c++/11/bits/uniform_int_dist.h:241-270)."""

    def __init__(self, seed, budget=200000):
        self.raw = mt19937_raw(seed, budget)
        self.pos = 0

    def g(self):
        v = self.raw[self.pos]
        self.pos += 1
        return v

    def uniform(self, lo, hi):
        urange = hi - lo
        if urange == 0xFFFFFFFF:
            return self.g() + lo
        r = urange + 1
        product = self.g() * r
        low = product & 0xFFFFFFFF
        if low < r:
            threshold = ((1 << 32) - r) % r
            while low < threshold:
                product = self.g() * r
                low = product & 0xFFFFFFFF
        return (product >> 32) + lo


def test_uniform_int_stream_matches_python_restatement():
    rng = np.random.default_rng(1)
    lo = rng.integers(0, 50, size=400).astype(np.uint32)
    hi = (lo + rng.integers(0, 3000, size=400)).astype(np.uint32)
    hi[::37] = lo[::37]           # degenerate range still consumes a draw
    hi[5], lo[5] = 0xFFFFFFFF, 0  # full range
    got = o.uniform_draws(0, lo, hi)
    p = PyPrng(0)
    want = np.array([p.uniform(int(a), int(b)) for a, b in zip(lo, hi)], dtype=np.uint32)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("total,k", [(7, 7), (8, 7), (100, 7), (400, 4), (33, 5), (50, 1)])
def test_sample_stream_partial_fisher_yates(total, k):
    got = o.sample_stream(0, total, k, 64)
    p = PyPrng(0)
    idx = list(range(total))
    want = []
    for _ in range(64):
        for i in range(k):
            j = p.uniform(i, total - 1)
            idx[i], idx[j] = idx[j], idx[i]
        want.append(idx[:k])
    np.testing.assert_array_equal(got, np.array(want, dtype=np.uint32))
    assert all(len(set(r)) == k for r in got.tolist())


def test_compute_num_trials_known_values():
    # RANSAC ctor clamp at min_inlier_ratio 0.25, confidence 0.999, multiplier 3
    assert o.compute_num_trials(25000, 100000, 0.999, 3.0, 4) == 5295          # H
    assert o.compute_num_trials(25000, 100000, 0.999, 3.0, 7) > 10000          # F: capped by 10000
    assert o.compute_num_trials(70000, 100000, 0.999, 3.0, 1) == 18            # watermark translation
    assert o.compute_num_trials(100, 100, 0.999, 3.0, 7) == 1                  # denom <= 0
    assert o.compute_num_trials(0, 100, 0.999, 3.0, 7) == np.iinfo(np.int64).max  # denom == 1
    import math
    for ninl, n, k in [(150, 400, 7), (302, 400, 4), (33, 90, 5)]:
        want = math.ceil(math.log(1 - 0.999) / math.log(1 - (ninl / n) ** k) * 3.0)
        assert o.compute_num_trials(ninl, n, 0.999, 3.0, k) == want


# ------------------------------------------------------------------------- small linear algebra ----
def test_det_sum64_is_a_fixed_tree():
    rng = np.random.default_rng(2)
    for n in [0, 1, 63, 64, 65, 1000]:
        x = rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
        p = np.zeros(64)
        for k in range(n):
            p[k & 63] += x[k]
        for m in (32, 16, 8, 4, 2, 1):
            p = p + p[np.arange(64) ^ m]
        assert o.det_sum64(x) == p[0]


@pytest.mark.parametrize("n", [3, 9])
def test_jacobi_eigen_vs_lapack(n):
    rng = np.random.default_rng(3)
    for _ in range(5):
        a = rng.normal(size=(n + 4, n))
        s = a.T @ a
        w, v = o.jacobi_eigen(s)
        wl = np.linalg.eigvalsh(s)
        np.testing.assert_allclose(np.sort(w), wl, rtol=1e-10, atol=1e-10 * wl.max())
        np.testing.assert_allclose(v @ np.diag(w) @ v.T, s, atol=1e-10 * np.abs(s).max())
        np.testing.assert_allclose(v.T @ v, np.eye(n), atol=1e-12)


def test_real_roots_vs_numpy():
    rng = np.random.default_rng(4)
    for deg in [1, 2, 3, 4, 7, 10]:
        for _ in range(20):
            roots_true = rng.uniform(-3, 3, size=rng.integers(0, deg + 1))
            poly = np.poly(roots_true) if len(roots_true) else np.array([1.0])
            rest = deg - len(roots_true)
            for _ in range(rest // 2):   # pad with complex pairs
                a, b = rng.uniform(-1, 1), rng.uniform(0.5, 2)
                poly = np.polymul(poly, [1, -2 * a, a * a + b * b])
            if (rest % 2) == 1:
                r = rng.uniform(-3, 3)
                poly = np.polymul(poly, [1, -r])
                roots_true = np.append(roots_true, r)
            got = o.real_roots(poly[::-1])
            assert np.all(np.diff(got) >= 0)
            want = np.sort(roots_true)
            if len(np.unique(np.round(want, 3))) == len(want):   # skip near-multiple roots
                assert len(got) == len(want)
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(o.real_roots([-6.0, 11.0, -6.0, 1.0]), [1, 2, 3], rtol=1e-12)
    assert len(o.real_roots([1.0, 0.0, 1.0])) == 0            # x^2 + 1
    assert len(o.real_roots([5.0])) == 0                        # constant
    np.testing.assert_allclose(o.real_roots([1.0, 2.0, 0.0, 0.0]), [-0.5])   # leading zeros


# --------------------------------------------------------------------- V6 / V4 residual KATs ----
def test_sampson_and_transfer_residuals_vs_numpy():
    rng = np.random.default_rng(5)
    sc = synth.two_view_scene(rng, planar=True)
    p1 = sc["pts1"][sc["matches"][:, 0]]
    p2 = sc["pts2"][sc["matches"][:, 1]]
    F = sc["F_true"]
    x1 = np.c_[p1, np.ones(len(p1))]
    x2 = np.c_[p2, np.ones(len(p2))]
    Fx1 = x1 @ F.T
    Ftx2 = x2 @ F
    want = np.sum(x2 * Fx1, axis=1) ** 2 / (Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2)
    np.testing.assert_allclose(o.sampson_error(p1, p2, F), want, rtol=1e-9)
    inl = sc["inlier"]
    assert np.median(o.sampson_error(p1, p2, F)[inl]) < 1.0 < np.median(o.sampson_error(p1, p2, F)[~inl])
    H = sc["H_true"]
    proj = x1 @ H.T
    want = np.sum((p2 - proj[:, :2] / proj[:, 2:]) ** 2, axis=1)
    np.testing.assert_allclose(o.h_residuals(p1, p2, H), want, rtol=1e-9)
    # hand-computable: identity homography -> squared displacement
    np.testing.assert_array_equal(o.h_residuals([[1, 2]], [[4, 6]], np.eye(3)), [25.0])


# ----------------------------------------------------------------------- V2, V3, V4, V5 solvers ----
def _scale_align(M, T):
    M, T = M / np.linalg.norm(M), T / np.linalg.norm(T)
    return min(np.abs(M - T).max(), np.abs(M + T).max())


def test_minimal_and_lsq_solvers_recover_planted_models():
    rng = np.random.default_rng(6)
    sc = synth.two_view_scene(rng, num_inliers=60, num_outliers=0, noise=0.0)
    m = sc["matches"]
    p1 = sc["pts1"][m[:, 0]]
    p2 = sc["pts2"][m[:, 1]]
    # float32 rounding of keypoints leaves ~1e-5 px errors: models are recovered to ~1e-6
    F7 = o.estimate_models("F7", p1[:7], p2[:7])
    assert 1 <= len(F7) <= 3
    assert min(_scale_align(F, sc["F_true"]) for F in F7) < 1e-4
    for F in F7:
        assert np.abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-10
        assert F[2, 2] == 1.0
    F8 = o.estimate_models("F8", p1, p2)
    assert len(F8) == 1 and _scale_align(F8[0], sc["F_true"]) < 1e-5
    assert np.abs(np.linalg.det(F8[0] / np.linalg.norm(F8[0]))) < 1e-12
    # essential matrix on camera coordinates
    Kinv = np.linalg.inv(sc["K"])
    n1 = (np.c_[p1, np.ones(len(p1))] @ Kinv.T)[:, :2]
    n2 = (np.c_[p2, np.ones(len(p2))] @ Kinv.T)[:, :2]
    E5 = o.estimate_models("E5", n1[:5], n2[:5])
    assert 1 <= len(E5) <= 10
    assert min(_scale_align(E, sc["E_true"]) for E in E5) < 1e-4
    for E in E5:
        En = E / np.linalg.norm(E)
        assert np.abs(np.linalg.det(En)) < 1e-8
        assert np.abs(2 * En @ En.T @ En - np.trace(En @ En.T) * En).max() < 1e-7
    Els = o.estimate_models("E5", n1, n2)     # 5-point as its own local (least-squares) estimator
    assert min(_scale_align(E, sc["E_true"]) for E in Els) < 1e-5
    # homography: planar scene
    sp = synth.two_view_scene(rng, num_inliers=40, num_outliers=0, noise=0.0, planar=True)
    q1 = sp["pts1"][sp["matches"][:, 0]]
    q2 = sp["pts2"][sp["matches"][:, 1]]
    H4 = o.estimate_models("H", q1[:4], q2[:4])
    Hn = o.estimate_models("H", q1, q2)
    assert _scale_align(H4[0], sp["H_true"]) < 1e-4 and _scale_align(Hn[0], sp["H_true"]) < 1e-6
    T = o.estimate_models("T", [[1, 1], [3, 5]], [[2, 4], [6, 6]])
    np.testing.assert_array_equal(T[0].ravel()[:2], [2.0, 2.0])


# ---------------------------------------------------------------------------- V1: LO-RANSAC ----
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_loransac_recovers_exact_planted_inlier_sets(seed):
    """Noise << threshold << outlier error: the mask must equal the planted set exactly, whatever
    the solver's low-order bits are (implementation-independent check)."""
    rng = np.random.default_rng(100 + seed)
    sc = synth.two_view_scene(rng, num_inliers=200, num_outliers=80, noise=0.1)
    m = sc["matches"]
    p1, p2 = sc["pts1"][m[:, 0]], sc["pts2"][m[:, 1]]
    r = o.ransac_estimate("F", p1, p2)
    assert r["success"]
    # the epipolar constraint is 1-D: a few of the 80 random outliers legitimately fall within
    # 4 px of a (slightly bent) epipolar geometry; every planted inlier must be found
    assert np.all(r["inliers"][sc["inlier"]])
    assert (r["inliers"] & ~sc["inlier"]).sum() <= 5
    assert r["num_inliers"] == r["inliers"].sum() >= 200
    assert np.all(o.sampson_error(p1, p2, r["model"])[r["inliers"]] <= 16.0)
    assert 1000 <= r["num_trials"] <= 100000       # Python-side min_num_trials = 1000
    # same call twice -> identical (seed 0 each time, as pycolmap's binding does)
    r2 = o.ransac_estimate("F", p1, p2)
    np.testing.assert_array_equal(r["model"], r2["model"])
    sp = synth.two_view_scene(rng, num_inliers=150, num_outliers=60, noise=0.1, planar=True)
    q1, q2 = sp["pts1"][sp["matches"][:, 0]], sp["pts2"][sp["matches"][:, 1]]
    rh = o.ransac_estimate("H", q1, q2)
    farh = o.h_residuals(q1, q2, sp["H_true"]) > 60.0
    keep = sp["inlier"] | farh
    np.testing.assert_array_equal(rh["inliers"][keep], sp["inlier"][keep])


def test_loransac_failure_and_small_inputs():
    rng = np.random.default_rng(7)
    p1 = rng.uniform(0, 1000, size=(6, 2))
    p2 = rng.uniform(0, 1000, size=(6, 2))
    r = o.ransac_estimate("F", p1, p2)             # fewer than 7 samples
    assert not r["success"] and r["num_trials"] == 0 and r["num_inliers"] == 0
    r = o.ransac_estimate("H", p1[:4], p2[:4])     # exactly minimal: model fits its own 4 points
    assert r["success"] and r["num_inliers"] == 4


# ------------------------------------------------------------------ V8, V9: model selection ----
def test_two_view_geometry_configurations():
    rng = np.random.default_rng(8)
    cam = o.make_camera(prior=False)
    camp = o.make_camera(prior=True)
    gen = synth.two_view_scene(rng, num_inliers=300, num_outliers=100)
    r = o.estimate_two_view_geometry(cam, gen["pts1"], cam, gen["pts2"], gen["matches"])
    assert r["config_name"] == "UNCALIBRATED" and r["num_inliers"] >= 295 and r["trials"][0] == 0
    assert (r["inlier_mask"] == gen["inlier"]).mean() > 0.98
    r = o.estimate_two_view_geometry(camp, gen["pts1"], camp, gen["pts2"], gen["matches"])
    assert r["config_name"] == "CALIBRATED" and r["trials"][0] > 0 and r["inl"][0] >= 290
    pl = synth.two_view_scene(rng, num_inliers=300, num_outliers=100, planar=True)
    r = o.estimate_two_view_geometry(cam, pl["pts1"], cam, pl["pts2"], pl["matches"])
    assert r["config_name"] == "PLANAR_OR_PANORAMIC"
    r = o.estimate_two_view_geometry(camp, pl["pts1"], camp, pl["pts2"], pl["matches"])
    assert r["config_name"] == "PLANAR_OR_PANORAMIC"
    # too few matches -> DEGENERATE without touching the PRNG
    r = o.estimate_two_view_geometry(cam, gen["pts1"], cam, gen["pts2"], gen["matches"][:14])
    assert r["config_name"] == "DEGENERATE" and r["num_inliers"] == 0 and r["trials"] == [0, 0, 0, 0]
    # pure noise -> DEGENERATE (no model reaches 15 inliers)
    nz = synth.two_view_scene(rng, num_inliers=0, num_outliers=30)
    r = o.estimate_two_view_geometry(cam, nz["pts1"], cam, nz["pts2"], nz["matches"])
    assert r["config_name"] == "DEGENERATE"
    # force_H_use
    r = o.estimate_two_view_geometry(cam, pl["pts1"], cam, pl["pts2"], pl["matches"],
                                     o.tvg_default_options(force_H_use=1))
    assert r["config_name"] == "PLANAR_OR_PANORAMIC" and r["trials"][1] == 0


def test_watermark_detection():
    """Matches confined to the image border that follow a pure translation -> WATERMARK."""
    rng = np.random.default_rng(9)
    w, h = 1600, 1200
    n = 80
    x = np.r_[rng.uniform(5, 150, n // 2), rng.uniform(w - 150, w - 5, n // 2)]
    y = rng.uniform(5, 150, n)
    p1 = np.c_[x, y]
    p2 = p1 + np.array([3.0, -2.0]) + rng.normal(0, 0.05, size=(n, 2))
    matches = np.c_[np.arange(n), np.arange(n)].astype(np.uint32)
    cam = o.make_camera()
    r = o.estimate_two_view_geometry(cam, p1, cam, p2, matches)
    assert r["config_name"] == "WATERMARK" and 1 <= r["trials"][3] <= 18
    r = o.estimate_two_view_geometry(cam, p1, cam, p2, matches, o.tvg_default_options(detect_watermark=0))
    assert r["config_name"] != "WATERMARK" and r["trials"][3] == 0


def test_batch_entry_point_equals_single_calls():
    """oracle_estimate_two_view_geometry_batch (OpenMP, used by bench.py's CPU baseline) returns exactly
    what the per-pair entry point returns, whatever the thread count."""
    rng = np.random.default_rng(42)
    scenes = [synth.two_view_scene(rng, num_inliers=int(rng.integers(20, 200)), num_outliers=int(rng.integers(0, 80)),
                                   planar=bool(i % 3 == 0)) for i in range(9)]
    cams = [o.make_camera("PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0), prior=bool(i % 2)) for i in range(9)]
    for threads in (1, 4):
        got = o.estimate_two_view_geometry_batch(cams, [s["pts1"] for s in scenes], cams, [s["pts2"] for s in scenes],
                                                 [s["matches"] for s in scenes], threads=threads)
        for cam, sc, g in zip(cams, scenes, got):
            w = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"])
            assert g["config"] == w["config"] and g["trials"] == w["trials"] and g["inl"] == w["inl"]
            np.testing.assert_array_equal(g["inlier_mask"], w["inlier_mask"])
            for k in "EFH":
                np.testing.assert_array_equal(g[k], w[k])


# ------------------------------------------------------- relative pose ----
def _rot_angle_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


def _quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _tri_angles_numpy(R, t, K, p1, p2):
    """Independent float64 restatement: midpoint-free angle between the two viewing rays of each
    correspondence (the triangulation angle of a noise-free point equals the angle between its rays)."""
    Kinv = np.linalg.inv(K)
    r1 = (Kinv @ np.c_[p1, np.ones(len(p1))].T).T
    r2 = (Kinv @ np.c_[p2, np.ones(len(p2))].T).T
    r2w = (R.T @ r2.T).T                       # second ray in the first camera's frame
    c = np.sum(r1 * r2w, axis=1) / np.linalg.norm(r1, axis=1) / np.linalg.norm(r2w, axis=1)
    a = np.arccos(np.clip(c, -1, 1))
    return np.minimum(a, np.pi - a)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_relative_pose_recovers_planted_motion(seed):
    """compute_relative_pose on a calibrated general scene: rotation, translation direction and
    the median triangulation angle of the planted motion; quaternion consistent with R."""
    rng = np.random.default_rng(100 + seed)
    sc = synth.two_view_scene(rng, num_inliers=250, num_outliers=80, noise=0.3)
    cam = o.make_camera(prior=True)
    r = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                     o.tvg_default_options(compute_relative_pose=1))
    assert r["config_name"] == "CALIBRATED" and r["pose_ok"]
    assert abs(np.linalg.det(r["R"]) - 1.0) < 1e-9 and np.allclose(r["R"] @ r["R"].T, np.eye(3), atol=1e-9)
    assert _rot_angle_deg(r["R"], sc["R"]) < 0.5
    tdir = sc["t"] / np.linalg.norm(sc["t"])
    assert abs(np.linalg.norm(r["tvec"]) - 1.0) < 1e-9 and float(tdir @ r["tvec"]) > 0.999
    assert np.allclose(_quat_to_rot(r["qvec"]), r["R"], atol=1e-9) and abs(np.linalg.norm(r["qvec"]) - 1) < 1e-9
    assert r["num_points3D"] >= 0.97 * r["num_inliers"]          # nearly all inliers in front of both cameras
    m = sc["matches"][r["inlier_mask"]]
    ang = _tri_angles_numpy(sc["R"], sc["t"], sc["K"], sc["pts1"][m[:, 0]], sc["pts2"][m[:, 1]])
    assert abs(r["tri_angle"] - np.median(ang)) < 0.05 * np.median(ang)
    # the same geometry through the stand-alone entry point (estimate_two_view_geometry_pose)
    p = o.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], m, r["config"], E=r["E"], H=r["H"])
    assert p["pose_ok"] and np.array_equal(p["R"], r["R"]) and np.array_equal(p["tvec"], r["tvec"])
    assert p["tri_angle"] == r["tri_angle"] and p["num_points3D"] == r["num_points3D"]


def test_relative_pose_planar_and_panoramic():
    rng = np.random.default_rng(77)
    cam = o.make_camera(prior=True)
    opts = o.tvg_default_options(compute_relative_pose=1)
    pl = synth.two_view_scene(rng, num_inliers=300, num_outliers=60, noise=0.2, planar=True)
    r = o.estimate_two_view_geometry(cam, pl["pts1"], cam, pl["pts2"], pl["matches"], opts)
    assert r["config_name"] == "PLANAR" and r["pose_ok"]     # PLANAR_OR_PANORAMIC resolved by the pose
    assert _rot_angle_deg(r["R"], pl["R"]) < 1.0
    # homography translation is in units of the plane distance: t_true / d
    tt = pl["t"] / synth.PLANE_D
    assert np.linalg.norm(r["tvec"] - tt) < 0.05 * np.linalg.norm(tt)
    assert r["tri_angle"] > 0 and r["num_points3D"] >= 0.95 * r["num_inliers"]
    pr = synth.two_view_scene(rng, num_inliers=300, num_outliers=60, noise=0.05, pure_rotation=True)
    r = o.estimate_two_view_geometry(cam, pr["pts1"], cam, pr["pts2"], pr["matches"], opts)
    assert r["pose_ok"] and r["config_name"] in ("PANORAMIC", "PLANAR")
    assert _rot_angle_deg(r["R"], pr["R"]) < 0.5
    if r["config_name"] == "PANORAMIC":
        assert r["tri_angle"] == 0.0 and np.all(r["tvec"] == 0)
    # an exact rotation homography is PANORAMIC by construction
    K = pr["K"]
    m = pr["matches"][pr["inlier"]]
    p = o.estimate_two_view_geometry_pose(cam, pr["pts1"], cam, pr["pts2"], m, 6, H=K @ pr["R"] @ np.linalg.inv(K))
    assert p["pose_ok"] and p["config_name"] == "PANORAMIC" and p["tri_angle"] == 0.0
    assert _rot_angle_deg(p["R"], pr["R"]) < 1e-6


def test_relative_pose_skips_configs_without_geometry():
    rng = np.random.default_rng(5)
    cam = o.make_camera(prior=True)
    sc = synth.two_view_scene(rng, num_inliers=100, num_outliers=10)
    m = sc["matches"][sc["inlier"]]
    for cfg in (0, 1, 7, 8):   # UNDEFINED, DEGENERATE, WATERMARK, MULTIPLE
        p = o.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], m, cfg, E=sc["E_true"])
        assert not p["pose_ok"] and p["config"] == cfg and p["tri_angle"] == 0.0
        assert np.array_equal(p["qvec"], [1, 0, 0, 0]) and np.array_equal(p["tvec"], [0, 0, 0])
    # UNCALIBRATED with the default (all-zero) E of the uncalibrated path: defined, finite output
    p = o.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], m, 3)
    assert p["pose_ok"] and np.all(np.isfinite(p["R"])) and np.all(np.isfinite(p["tvec"]))
    # no inlier matches at all
    p = o.estimate_two_view_geometry_pose(cam, sc["pts1"], cam, sc["pts2"], m[:0], 2, E=sc["E_true"])
    assert p["pose_ok"] and p["num_points3D"] == 0 and p["tri_angle"] == 0.0


# ------------------------------------------------------------------------- golden fixture ----
def test_golden_fixture_pins_the_oracle():
    """tests/golden/tvg_golden_v3.npz: the oracle still produces what it produced when the fixture was
    committed (configs, masks, trial counts, model / pose bit patterns), with and without the pose."""
    import tvg_golden
    n = 0
    for c in tvg_golden.cases():
        for pose in (0, 1):
            cam1 = o.make_camera(c["cam1"][0], 1600, 1200, c["cam1"][1], prior=c["prior"])
            cam2 = o.make_camera(c["cam2"][0], 1600, 1200, c["cam2"][1], prior=c["prior"])
            r = o.estimate_two_view_geometry(cam1, c["pts1"], cam2, c["pts2"], c["matches"],
                                             o.tvg_default_options(compute_relative_pose=pose, **c["opts"]), seed=0)
            w = c["want"][pose]
            tag = f"case {c['index']} pose {pose}"
            assert r["config"] == w["config"] and r["trials"] == w["trials"] and r["inl"] == w["inl"], tag
            np.testing.assert_array_equal(r["inlier_mask"], w["mask"], err_msg=tag)
            assert r["num_points3D"] == w["points3D"], tag
            assert tvg_golden.bits(r["tri_angle"])[0] == w["tri_angle"][0], tag
            for f in tvg_golden.FIELDS:
                np.testing.assert_array_equal(tvg_golden.bits(r[f]), w[f], err_msg=f"{tag} {f}")
            n += 1
    assert n == 40
