"""CPU bit-parity of the product's lane-local verification numerics (pycolmap_amd/csrc/tvg_math.h,
compiled for the host by a test-only shim) against the oracle.  Bit-exact: same algorithm, same
operation order, no contraction — the same property the GPU tests then check on device."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import synth

ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "shim" / "_build" / "libtvgshim.so"


@pytest.fixture(scope="module")
def shim():
    SHIM.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "shim" / "tvg_shim.cc"
    hdrs = [ROOT / "pycolmap_amd" / "csrc" / "tvg_math.h", ROOT / "pycolmap_amd" / "csrc" / "pose_math.h"]
    if not SHIM.exists() or SHIM.stat().st_mtime < max(f.stat().st_mtime for f in [src] + hdrs):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared",
                        "-fPIC", str(src), "-o", str(SHIM)], check=True)
    return C.CDLL(str(SHIM))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def scene_points(seed, **kw):
    sc = synth.two_view_scene(np.random.default_rng(seed), **kw)
    m = sc["matches"]
    return sc, np.ascontiguousarray(sc["pts1"][m[:, 0]]), np.ascontiguousarray(sc["pts2"][m[:, 1]])


@pytest.mark.parametrize("seed", range(6))
def test_minimal_solvers_bit_exact(shim, seed):
    sc, p1, p2 = scene_points(seed, num_inliers=40, num_outliers=10, noise=0.3)
    rng = np.random.default_rng(seed)
    for _ in range(8):
        idx = rng.choice(len(p1), 7, replace=False)
        a, b = np.ascontiguousarray(p1[idx]), np.ascontiguousarray(p2[idx])
        out = np.zeros((3, 9))
        n = shim.shim_estimate_f7(_p(a), _p(b), _p(out))
        want = o.estimate_models("F7", a, b)
        assert n == len(want)
        np.testing.assert_array_equal(bits(out[:n]), bits(want.reshape(n, 9)))
        out = np.zeros((1, 9))
        shim.shim_estimate_h4(_p(a[:4]), _p(b[:4]), _p(out))
        np.testing.assert_array_equal(bits(out), bits(o.estimate_models("H", a[:4], b[:4]).reshape(1, 9)))
        Kinv = np.linalg.inv(sc["K"])
        n1 = np.ascontiguousarray((np.c_[a[:5], np.ones(5)] @ Kinv.T)[:, :2])
        n2 = np.ascontiguousarray((np.c_[b[:5], np.ones(5)] @ Kinv.T)[:, :2])
        out = np.zeros((10, 9))
        n = shim.shim_estimate_e5(_p(n1), _p(n2), _p(out))
        want = o.estimate_models("E5", n1, n2)
        assert n == len(want)
        np.testing.assert_array_equal(bits(out[:n]), bits(want.reshape(n, 9)))


def test_roots_jacobi_residuals_temper_bit_exact(shim):
    rng = np.random.default_rng(11)
    for deg in (1, 2, 3, 5, 10):
        for _ in range(10):
            c = np.ascontiguousarray(rng.normal(size=deg + 1))
            out = np.zeros(12)
            n = shim.shim_real_roots(_p(c), deg, _p(out))
            want = o.real_roots(c)
            assert n == len(want)
            np.testing.assert_array_equal(bits(out[:n]), bits(want))
    for n in (3, 9):
        a = rng.normal(size=(n + 3, n))
        s = np.ascontiguousarray(a.T @ a)
        w, v = o.jacobi_eigen(s)
        s2 = s.copy()
        v2 = np.zeros((n, n))
        shim.shim_jacobi(n, _p(s2), _p(v2))
        np.testing.assert_array_equal(bits(np.diag(s2).copy()), bits(w))
        np.testing.assert_array_equal(bits(v2), bits(v))
    sc, p1, p2 = scene_points(3, planar=True)
    for kind, M, ref in ((0, sc["F_true"], o.sampson_error), (1, sc["H_true"], o.h_residuals)):
        out = np.zeros(len(p1))
        Mc = np.ascontiguousarray(M)
        shim.shim_residuals(kind, _p(Mc), _p(p1), _p(p2), len(p1), _p(out))
        np.testing.assert_array_equal(bits(out), bits(ref(p1, p2, M)))
    # tempering against numpy's MT19937 raw outputs is covered indirectly in test_oracle_tvg;
    # here: the known first output of mt19937(5489) state word tempering
    shim.shim_temper.restype = C.c_uint
    assert shim.shim_temper(0) == 0


# ------------------------------------------------------------------ relative pose (pose_math.h) ----
def _shim_pose(shim, cam1, cam2, p1, p2, config, E, H):
    iout = np.zeros(3, dtype=np.int32)
    dout = np.zeros(17)
    prm1 = np.array(list(cam1.params)[:4], dtype=np.float64)
    prm2 = np.array(list(cam2.params)[:4], dtype=np.float64)
    a, b = np.ascontiguousarray(p1, dtype=np.float64), np.ascontiguousarray(p2, dtype=np.float64)
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(9)
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    shim.shim_pose(C.c_int(cam1.model_id), _p(prm1), C.c_int(cam2.model_id), _p(prm2), _p(a), _p(b),
                   C.c_int(len(a)), C.c_int(int(config)), _p(E), _p(H), _p(iout), _p(dout))
    return dict(pose_ok=bool(iout[0]), config=int(iout[1]), num_points3D=int(iout[2]), R=dout[:9].reshape(3, 3),
                tvec=dout[9:12].copy(), qvec=dout[12:16].copy(), tri_angle=float(dout[16]))


def _same_pose(got, want):
    assert got["pose_ok"] == want["pose_ok"] and got["config"] == want["config"]
    assert got["num_points3D"] == want["num_points3D"]
    for k in ("R", "tvec", "qvec"):
        np.testing.assert_array_equal(bits(got[k]), bits(want[k]), err_msg=k)
    assert bits(got["tri_angle"]) == bits(want["tri_angle"])


@pytest.mark.parametrize("seed", range(8))
def test_relative_pose_bit_exact(shim, seed):
    """The kernel's per-pair pose algorithm (lane-local functions, selection-based median) equals the
    oracle's EstimateTwoViewGeometryPose bit for bit: calibrated, uncalibrated (E of the calibrated
    path), planar, panoramic, and a mixed SIMPLE_PINHOLE / PINHOLE camera pair."""
    rng = np.random.default_rng(500 + seed)
    kinds = [dict(), dict(planar=True), dict(pure_rotation=True, noise=0.05)]
    sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(20, 300)), num_outliers=int(rng.integers(0, 60)),
                              **kinds[seed % 3])
    cam1 = o.make_camera("PINHOLE", 1600, 1200, (1200.0, 1190.0, 800.0, 600.0), prior=True)
    cam2 = o.make_camera("SIMPLE_PINHOLE", 1600, 1200, (1210.0, 801.0, 599.0), prior=True) if seed % 2 else cam1
    r = o.estimate_two_view_geometry(cam1, sc["pts1"], cam2, sc["pts2"], sc["matches"])
    assert r["config"] in (2, 3, 6)
    m = sc["matches"][r["inlier_mask"]]
    p1, p2 = sc["pts1"][m[:, 0]], sc["pts2"][m[:, 1]]
    ident = np.c_[np.arange(len(m)), np.arange(len(m))].astype(np.uint32)
    for cfg in {r["config"], 2, 3, 4, 5, 6}:
        want = o.estimate_two_view_geometry_pose(cam1, p1, cam2, p2, ident, cfg, E=r["E"], H=r["H"])
        got = _shim_pose(shim, cam1, cam2, p1, p2, cfg, r["E"], r["H"])
        _same_pose(got, want)
    # subsets: odd / even / tiny / empty correspondence lists (median of one, of two, of none)
    for n in (0, 1, 2, 3, 8):
        want = o.estimate_two_view_geometry_pose(cam1, p1[:n], cam2, p2[:n], ident[:n], 2, E=r["E"], H=r["H"])
        _same_pose(_shim_pose(shim, cam1, cam2, p1[:n], p2[:n], 2, r["E"], r["H"]), want)


def test_relative_pose_bit_exact_odd_inputs(shim):
    """Configurations without a pose, the all-zero E of the uncalibrated path, an exact rotation
    homography and correspondences behind the cameras."""
    rng = np.random.default_rng(9)
    sc = synth.two_view_scene(rng, num_inliers=60, num_outliers=0, noise=0.2)
    cam = o.make_camera(prior=True)
    m = sc["matches"]
    p1, p2 = sc["pts1"][m[:, 0]], sc["pts2"][m[:, 1]]
    ident = np.c_[np.arange(len(m)), np.arange(len(m))].astype(np.uint32)
    Z = np.zeros((3, 3))
    K = sc["K"]
    Hrot = K @ sc["R"] @ np.linalg.inv(K)
    cases = [(0, Z, Z), (1, Z, Z), (7, sc["E_true"], Z), (8, Z, Z), (3, Z, Z), (2, -sc["E_true"], Z),
             (6, Z, Hrot), (5, Z, Hrot), (4, Z, -Hrot), (2, sc["E_true"].T, Z)]
    for cfg, E, H in cases:
        want = o.estimate_two_view_geometry_pose(cam, p1, cam, p2, ident, cfg, E=E, H=H)
        _same_pose(_shim_pose(shim, cam, cam, p1, p2, cfg, E, H), want)
    # swapped images: most triangulated points fall behind a camera for the planted E
    want = o.estimate_two_view_geometry_pose(cam, p2, cam, p1, ident, 2, E=sc["E_true"], H=Z)
    _same_pose(_shim_pose(shim, cam, cam, p2, p1, 2, sc["E_true"], Z), want)


def test_fp32_homography_prefilter_never_contradicts_the_reference(shim):
    """tvg_math.h h32_point: the counting loop decides a homography inlier in FP32 only when |t32| exceeds its own
    error bound.  Whatever it decides must be the reference decision h_residual <= max_error^2 - for fitted and for
    wild homographies, points far from and within 1e-6 px of the threshold circle, thresholds 0.25 .. 400 px^2 -
    and it must decide nearly everything that is not on the circle."""
    rng = np.random.default_rng(17)
    decided = undecided = 0
    bands_ratio = []
    n_sure = n_out = 0
    for trial in range(400):
        n = 256
        style = trial % 5
        p1 = rng.uniform([0, 0], [1600, 1200], size=(n, 2))
        if style == 0:      # a real homography, points scattered around their images
            H = np.eye(3) + rng.normal(size=(3, 3)) * [[0.1, 0.1, 60], [0.1, 0.1, 60], [1e-4, 1e-4, 0.05]]
        elif style == 1:    # from four random correspondences: the minimal models of a non-planar scene
            src = rng.uniform([0, 0], [1600, 1200], size=(4, 2))
            dst = rng.uniform([0, 0], [1600, 1200], size=(4, 2))
            out = np.zeros(9)
            shim.shim_estimate_h4(_p(np.ascontiguousarray(src)), _p(np.ascontiguousarray(dst)), _p(out))
            H = out.reshape(3, 3)
        elif style == 2:    # strongly projective
            H = np.array([[1, 0, 0], [0, 1, 0], [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 1.0]])
        elif style == 3:
            H = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-6, 6)
        else:
            H = np.diag([rng.uniform(0.2, 5), rng.uniform(0.2, 5), 1.0]) * 10.0 ** rng.uniform(-3, 3)
        if not np.all(np.isfinite(H)):
            continue
        max_res = float(rng.choice([0.25, 1.0, 16.0, 16.0, 100.0, 400.0]))
        q = np.concatenate([p1, np.ones((n, 1))], axis=1) @ H.T
        with np.errstate(all="ignore"):
            proj = q[:, :2] / q[:, 2:3]
        # image-2 points: a third random, a third on the threshold circle +- up to 1e-6 px, a third well inside
        ang = rng.uniform(0, 2 * np.pi, size=n)
        rad = np.where(np.arange(n) % 3 == 1, np.sqrt(max_res) * (1 + rng.uniform(-1, 1, size=n) * 10.0 ** rng.uniform(-9, -3, size=n)),
                       np.sqrt(max_res) * rng.uniform(0, 0.9, size=n))
        p2 = proj + np.stack([np.cos(ang), np.sin(ang)], axis=1) * rad[:, None]
        far = np.arange(n) % 3 == 0
        p2[far] = rng.uniform([0, 0], [1600, 1200], size=(int(far.sum()), 2))
        p2 = np.where(np.isfinite(p2), p2, 0.0)
        p1c, p2c = np.ascontiguousarray(p1.astype(np.float32).astype(np.float64)), np.ascontiguousarray(p2.astype(np.float32).astype(np.float64))
        Hc = np.ascontiguousarray(H.reshape(9))
        res = np.zeros(n)
        shim.shim_residuals(1, _p(Hc), _p(p1c), _p(p2c), n, _p(res))
        want = res <= max_res
        got = np.zeros(n, dtype=np.int8)
        shim.shim_h32_decisions(_p(Hc), C.c_double(max_res), _p(p1c), _p(p2c), n, _p(got))
        band, band_ref = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
        shim.shim_h32_bands(_p(Hc), C.c_double(max_res), _p(p1c), _p(p2c), n, _p(band), _p(band_ref))
        fin = np.isfinite(band_ref)
        assert np.all(band[fin] >= band_ref[fin]), "the loop's bound must cover the reference bound of the error analysis"
        if style in (0, 2):                          # tame models: the cheaper bound stays within a few times the reference
            bands_ratio.append(float(np.median(band[fin] / band_ref[fin])))
        qv = np.zeros(n, dtype=np.float32)
        shim.shim_h32_q(_p(Hc), C.c_double(max_res), _p(p1c), _p(p2c), n, _p(qv))
        sure_out = qv > 0                            # the counting loop's "outlier beyond doubt"
        assert not (sure_out & want).any() and not (sure_out & (got != 0)).any()
        dec = got >= 0
        wrong = dec & ((got == 1) != want)
        assert not wrong.any(), f"trial {trial} style {style}: FP32 decided {got[wrong][:4]} against residuals {res[wrong][:4]} (max {max_res})"
        off_circle = np.abs(np.sqrt(np.maximum(res, 0)) - np.sqrt(max_res)) > 0.05 * np.sqrt(max_res)
        if style in (0, 4) and abs(np.log10(np.abs(H).max())) < 2:
            # tame models: everything 5 % away from the circle is decided
            assert dec[off_circle & np.isfinite(res)].mean() > 0.98
            clear = off_circle & np.isfinite(res) & ~want
            n_sure += int(sure_out[clear].sum()); n_out += int(clear.sum())
        decided += int(dec.sum())
        undecided += int((~dec).sum())
    assert decided > undecided               # (a third of the points sit on the circle by construction)
    assert np.median(bands_ratio) < 4.0
    assert n_sure > 0.98 * n_out > 0             # and it recognises nearly every outlier


def test_fp32_sampson_prefilter_never_calls_an_inlier_an_outlier(shim):
    """tvg_math.h s32_outlier_q: the F / E counting loops drop a correspondence from a model's count only when q > 0.
    Whatever it drops must be an outlier by the reference residual sampson(...) <= max_error^2 - for the true model,
    minimal models of contaminated samples and wild matrices, in pixel coordinates (F) and normalised ones (E), with
    points scattered, on the inlier side and within 1e-9 .. 1e-3 of the threshold - and on tame models it must drop
    nearly every clear outlier."""
    rng = np.random.default_rng(23)
    n_sure = n_clear = 0
    for trial in range(300):
        sc, p1, p2 = scene_points(1000 + trial, num_inliers=120, num_outliers=136, noise=0.5)
        n = len(p1)
        normalised = trial % 2 == 1
        if normalised:
            Kinv = np.linalg.inv(sc["K"])
            p1 = (np.c_[p1, np.ones(n)] @ Kinv.T)[:, :2]
            p2 = (np.c_[p2, np.ones(n)] @ Kinv.T)[:, :2]
            max_res = float(rng.choice([1.0, 4.0, 16.0])) / sc["K"][0, 0] ** 2
        else:
            max_res = float(rng.choice([0.25, 1.0, 16.0, 16.0, 100.0]))
        style = (trial // 2) % 4
        tame = style == 0
        if style == 0:
            Mx = sc["E_true"] if normalised else sc["F_true"]
            Mx = Mx / np.linalg.norm(Mx) if normalised else Mx / Mx[2, 2]
        elif style == 1:      # a minimal model of a random (mostly contaminated) sample
            if normalised:
                idx = rng.choice(n, 5, replace=False)
                out = np.zeros((10, 9))
                k = shim.shim_estimate_e5(_p(np.ascontiguousarray(p1[idx])), _p(np.ascontiguousarray(p2[idx])), _p(out))
            else:
                idx = rng.choice(n, 7, replace=False)
                out = np.zeros((3, 9))
                k = shim.shim_estimate_f7(_p(np.ascontiguousarray(p1[idx])), _p(np.ascontiguousarray(p2[idx])), _p(out))
            if k == 0:
                continue
            Mx = out[int(rng.integers(k))].reshape(3, 3)
        elif style == 2:
            Mx = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-6, 6, size=(3, 3))
        else:
            Mx = (sc["E_true"] if normalised else sc["F_true"]) * 10.0 ** rng.uniform(-30, 30)
        # a third of the image-2 points moved onto the threshold (along the epipolar line's normal), +- a hair
        p1 = np.ascontiguousarray(p1.astype(np.float32).astype(np.float64))
        p2 = p2.copy()
        l = np.c_[p1, np.ones(n)] @ Mx.T
        nrm = np.hypot(l[:, 0], l[:, 1])
        with np.errstate(all="ignore"):
            dist = (np.einsum("ij,ij->i", np.c_[p2, np.ones(n)], l)) / nrm
            near = np.arange(n) % 3 == 1
            target = np.sqrt(max_res) * 1.41 * (1 + rng.uniform(-1, 1, size=n) * 10.0 ** rng.uniform(-9, -3, size=n))
            shift = (target - dist)[:, None] * (l[:, :2] / nrm[:, None])
        ok = near & np.isfinite(shift).all(axis=1)
        p2[ok] += shift[ok]
        p2 = np.ascontiguousarray(p2.astype(np.float32).astype(np.float64))
        Mc = np.ascontiguousarray(Mx.reshape(9))
        res = np.zeros(n)
        shim.shim_residuals(0, _p(Mc), _p(p1), _p(p2), n, _p(res))
        with np.errstate(invalid="ignore"):
            inlier = res <= max_res
        qv = np.zeros(n, dtype=np.float32)
        shim.shim_s32_q(_p(Mc), C.c_double(max_res), _p(p1), _p(p2), n, _p(qv))
        sure_out = qv > 0
        bad = sure_out & inlier
        assert not bad.any(), f"trial {trial} style {style}: dropped inliers, residuals {res[bad][:4]} (max {max_res})"
        if tame:
            with np.errstate(invalid="ignore"):
                clear = np.isfinite(res) & (res > 1.05 * max_res)
            n_sure += int(sure_out[clear].sum())
            n_clear += int(clear.sum())
    assert n_sure > 0.98 * n_clear > 0


def test_closed_form_h4_on_degenerate_samples_against_the_svd_dlt(shim):
    """The minimal 4-point homography is a closed form (change of projective basis) where upstream solves the 8 x 9 DLT by
    SVD (tests/ref2/tvg_ref2.py estimate_h restates that).  On a proper sample the two give the same H up to scale; on a
    degenerate one - three collinear points - the SVD still returns a (singular) null vector while the closed form
    divides by zero.  What has to hold for the RANSAC built on it: the degenerate model must never out-score anything,
    i.e. it must not count inliers beyond what the SVD model counts, and near-degenerate samples must give the same
    inlier DECISIONS as the SVD model on points away from the threshold."""
    from ref2 import tvg_ref2 as r2
    rng = np.random.default_rng(5)
    sc, p1, p2 = scene_points(3, num_inliers=200, num_outliers=60, planar=True, noise=0.3)
    max_res = 16.0

    def decisions(H):
        out = np.zeros(len(p1))
        shim.shim_residuals(1, _p(np.ascontiguousarray(H.reshape(9))), _p(p1), _p(p2), len(p1), _p(out))
        return out

    checked_near = 0
    for trial in range(60):
        a = rng.uniform(100, 1500, (4, 2))
        b = rng.uniform(100, 1100, (4, 2))
        eps = 0.0 if trial < 20 else 10.0 ** rng.uniform(-9, -2)        # exactly / nearly collinear in image 1
        t = rng.uniform(0.2, 0.8)
        a[2] = a[0] + t * (a[1] - a[0]) + eps * np.array([-(a[1] - a[0])[1], (a[1] - a[0])[0]])
        if trial % 3 == 0:
            a, b = b.copy(), a.copy()                                   # ... or in image 2
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        got = np.zeros((1, 9))
        shim.shim_estimate_h4(_p(a), _p(b), _p(got))
        # bit-exact with the oracle's closed form whatever the sample (NaNs compared as NaNs)
        want = o.estimate_models("H", a, b).reshape(1, 9)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_array_equal(bits(np.nan_to_num(got)), bits(np.nan_to_num(want)))
        r_cf = decisions(got)
        H_svd = r2.estimate_h(a, b)[0]
        r_svd = decisions(H_svd)
        in_cf, in_svd = r_cf <= max_res, r_svd <= max_res                # NaN / inf residuals compare false: outliers
        if eps == 0.0:
            # exactly degenerate: the closed form's model is non-finite or wild; it must not collect support the SVD
            # model does not have (so it can never become a best model the reference would not also have found)
            assert in_cf.sum() <= max(in_svd.sum(), 4), (trial, int(in_cf.sum()), int(in_svd.sum()))
        else:
            # nearly degenerate: four points in general position determine H exactly; both solvers see the same model up
            # to conditioning, so the decisions agree wherever the SVD residual is not within 5 % of the threshold
            clear = np.abs(r_svd - max_res) > 0.05 * max_res
            if np.all(np.isfinite(got)) and np.all(np.isfinite(H_svd)) and eps > 1e-6:
                np.testing.assert_array_equal(in_cf[clear], in_svd[clear], err_msg=f"trial {trial} eps {eps}")
                checked_near += 1
    assert checked_near >= 10
