"""GPU parity tests of guided matching (amc_match_guided_pairs = FeatureMatcher::MatchGuided) against
the CPU oracle: match -> verify -> guided re-match with the float32 epipolar / homography filter.  Two kernels
produce it: candidate generation on a keypoint grid (match_guided.hip) for sane models, the dense filtered scan
(match_dot4.hip) for the rest; AMC_GUIDED_DENSE=1 forces the dense one, which the tests use to compare the two."""
import os

import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import _capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _capi.Context(0)
    yield c
    c.close()


def upload_scene(ctx, imgs, prior):
    ctx.reserve_slots(len(imgs))
    for k, im in enumerate(imgs):
        ctx.upload_descriptors(k, im["descriptors"])
        ctx.upload_keypoints(k, im["keypoints"])
        ctx.upload_camera(k, "PINHOLE", im["width"], im["height"], im["params"], prior)


def check_guided(ctx, imgs, s1, s2, tvg, max_error, **kw):
    off, m, st = ctx.match_guided_pairs(s1, s2, tvg, max_error, **kw)
    assert st["pairs_dot4"] + st["pairs_guided_grid"] == len(s1)
    total = 0
    for p, (a, b) in enumerate(zip(s1, s2)):
        want = o.match_guided(imgs[a]["descriptors"], imgs[a]["keypoints"], imgs[b]["descriptors"],
                              imgs[b]["keypoints"], tvg[p]["config"], tvg[p]["F"], tvg[p]["H"], max_error, **kw)
        assert want is not None
        np.testing.assert_array_equal(m[int(off[p]):int(off[p + 1])], want, err_msg=f"pair {p} ({a},{b})")
        total += len(want)
    return total


@pytest.mark.parametrize("prior", [False, True])
def test_match_verify_guided_chain(ctx, prior):
    rng = np.random.default_rng(5 + int(prior))
    imgs = synth.multiview_scene(rng, num_images=5, n_feats=700, num_landmarks=900)
    upload_scene(ctx, imgs, prior)
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    off, m, _ = ctx.match_pairs(s1, s2)
    tvg, mask, _ = ctx.verify_pairs(s1, s2, off, m, _capi.tvg_options())
    ok = np.isin(tvg["config"], [2, 3, 4, 5, 6])
    assert ok.sum() >= 6
    g1, g2, gt = s1[ok], s2[ok], tvg[ok]
    n = check_guided(ctx, imgs, g1, g2, gt, 4.0)
    assert n > 10 * ok.sum()             # guided matching recovers plenty of matches on these scenes
    check_guided(ctx, imgs, g1, g2, gt, 1.0, max_ratio=0.9, max_distance=1.0)
    check_guided(ctx, imgs, g1, g2, gt, 8.0, cross_check=False)
    check_guided(ctx, imgs, g2, g1, gt, 4.0)   # swapped images with the same models: just different data
    # homography filter: same pairs, configuration forced to the H kinds
    for cfg in (4, 5, 6):
        gh = gt.copy()
        gh["config"] = cfg
        check_guided(ctx, imgs, g1, g2, gh, 4.0)


def test_guided_ragged_sizes_and_degenerate_models(ctx):
    rng = np.random.default_rng(9)
    base = synth.multiview_scene(rng, num_images=4, n_feats=333, num_landmarks=500)
    imgs = [dict(im) for im in base]
    for im, n in zip(imgs, (333, 65, 1, 200)):
        im["descriptors"] = im["descriptors"][:n]
        im["keypoints"] = im["keypoints"][:n]
    upload_scene(ctx, imgs, False)
    s1 = np.array([0, 0, 1, 3, 2, 0], np.uint32)
    s2 = np.array([1, 3, 3, 0, 0, 2], np.uint32)
    tvg = np.zeros(len(s1), dtype=_capi.TVG_DTYPE)
    tvg["config"] = [3, 6, 2, 4, 3, 5]
    for p in range(len(s1)):
        tvg["F"][p] = rng.normal(size=(3, 3)) * [1e-6, 1e-6, 1e-3]
        tvg["H"][p] = np.eye(3) + rng.normal(size=(3, 3)) * 1e-4
    tvg["F"][4] = 0.0            # all-zero F: 0/0 in the filter = NaN, NaN > t is false: nothing rejected
    tvg["H"][5] = 0.0            # all-zero H: same through the homogeneous division
    check_guided(ctx, imgs, s1, s2, tvg, 4.0)
    check_guided(ctx, imgs, s1, s2, tvg, 0.0)


def test_guided_argument_errors(ctx):
    rng = np.random.default_rng(10)
    imgs = synth.multiview_scene(rng, num_images=2, n_feats=64, num_landmarks=100)
    upload_scene(ctx, imgs, False)
    tvg = np.zeros(1, dtype=_capi.TVG_DTYPE)
    tvg["config"] = 1                                        # DEGENERATE: COLMAP does not guide those
    with pytest.raises(_capi.AmcError, match="no guided"):
        ctx.match_guided_pairs([0], [1], tvg, 4.0)
    tvg["config"] = 3
    with pytest.raises(_capi.AmcError):
        ctx.match_guided_pairs([0], [1], tvg, -1.0)
    ctx.upload_points_f64(1, imgs[1]["keypoints"][:, :2])    # replaces the float32 keypoints
    with pytest.raises(_capi.AmcError, match="keypoints"):
        ctx.match_guided_pairs([0], [1], tvg, 4.0)
    off, m, _ = ctx.match_guided_pairs([], [], np.zeros(0, dtype=_capi.TVG_DTYPE), 4.0)
    assert off.tolist() == [0] and m.shape == (0, 2)


def test_guided_resolves_repeated_structures(ctx):
    """Every feature of image 1 has its true partner AND a decoy with the identical descriptor in image
    2: plain matching rejects all of them on the ratio test (best == second), the homography filter
    removes the decoys and guided matching recovers every correspondence."""
    rng = np.random.default_rng(12)
    n = 100
    desc = synth.quantize_descriptors(rng.gamma(0.7, 1.0, size=(n, 128)))
    kp1 = rng.uniform(50, 900, size=(n, 2)).astype(np.float32)
    shift = np.array([5.0, -3.0], np.float32)
    kp2 = np.concatenate([kp1 + shift, kp1 + np.array([400.0, 350.0], np.float32)]).astype(np.float32)
    imgs = [dict(descriptors=desc, keypoints=kp1, width=1600, height=1200, params=(1200.0, 1200.0, 800.0, 600.0)),
            dict(descriptors=np.concatenate([desc, desc]), keypoints=kp2, width=1600, height=1200,
                 params=(1200.0, 1200.0, 800.0, 600.0))]
    upload_scene(ctx, imgs, False)
    off, m, _ = ctx.match_pairs([0], [1])
    assert len(m) == 0 and len(o.match(imgs[0]["descriptors"], imgs[1]["descriptors"])) == 0
    tvg = np.zeros(1, dtype=_capi.TVG_DTYPE)
    tvg["config"] = 4
    tvg["H"][0] = [[1, 0, 5], [0, 1, -3], [0, 0, 1]]
    assert check_guided(ctx, imgs, [0], [1], tvg, 4.0) == n
    off, m, _ = ctx.match_guided_pairs([0], [1], tvg, 4.0)
    np.testing.assert_array_equal(m, np.stack([np.arange(n)] * 2, axis=1))


def both_kernels(ctx, s1, s2, tvg, max_error, **kw):
    """(offsets, matches) of the default routing and of the dense kernel alone, and how many pairs took the grid."""
    off_g, m_g, st_g = ctx.match_guided_pairs(s1, s2, tvg, max_error, **kw)
    os.environ["AMC_GUIDED_DENSE"] = "1"
    try:
        off_d, m_d, st_d = ctx.match_guided_pairs(s1, s2, tvg, max_error, **kw)
    finally:
        del os.environ["AMC_GUIDED_DENSE"]
    assert st_d["pairs_guided_grid"] == 0 and st_d["pairs_dot4"] == len(s1)
    np.testing.assert_array_equal(off_g, off_d)
    np.testing.assert_array_equal(m_g, m_d)
    return off_g, m_g, st_g["pairs_guided_grid"]


def test_guided_grid_kernel_runs_and_equals_the_dense_kernel(ctx):
    """Scene geometry (F and H models from verification) at 2000 features per image: the candidate-generation kernel
    takes every pair and returns the dense kernel's matches, for several thresholds, both directions, with and
    without cross check; and the oracle agrees on a subset."""
    rng = np.random.default_rng(21)
    imgs = synth.multiview_scene(rng, num_images=6, n_feats=2000, num_landmarks=2600)
    upload_scene(ctx, imgs, True)
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    off, m, _ = ctx.match_pairs(s1, s2)
    tvg, _, _ = ctx.verify_pairs(s1, s2, off, m, _capi.tvg_options())
    ok = np.isin(tvg["config"], [2, 3, 4, 5, 6])
    assert ok.sum() >= 10
    g1, g2, gt = s1[ok], s2[ok], tvg[ok]
    for max_error, kw in ((4.0, {}), (0.5, {}), (12.0, dict(max_ratio=0.95, max_distance=1.2)), (4.0, dict(cross_check=False))):
        _, mg, ngrid = both_kernels(ctx, g1, g2, gt, max_error, **kw)
        assert ngrid == len(g1) and len(mg) > 0
    gh = gt.copy()
    gh["config"] = 4
    for p in range(len(gh)):     # a homography that really maps image 1 near image 2: the estimated H where there is one
        if not np.any(gh["H"][p]):
            gh["H"][p] = np.eye(3) + rng.normal(size=(3, 3)) * [[1e-3, 1e-3, 5.0], [1e-3, 1e-3, 5.0], [1e-7, 1e-7, 1e-3]]
    _, _, ngrid = both_kernels(ctx, g1, g2, gh, 6.0)
    assert ngrid == len(g1)
    both_kernels(ctx, g2, g1, gh, 6.0)
    check_guided(ctx, imgs, g1[:3], g2[:3], gt[:3], 4.0)
    check_guided(ctx, imgs, g1[:3], g2[:3], gh[:3], 6.0)


def test_guided_grid_random_models_and_layouts(ctx):
    """Randomised: keypoint layouts (uniform, clustered, on a line, repeated points, a tiny box, far from the origin),
    models from tame to wild (axis-parallel epipolar lines, epipole inside the image, near-singular and
    horizon-crossing homographies, huge and tiny scales), thresholds 0 .. 1e6.  Whatever the routing decides, the
    result equals the dense kernel's."""
    rng = np.random.default_rng(22)
    n_img = 8
    layouts = []
    for k in range(n_img):
        n = int(rng.integers(1, 900))
        kind = k % 6
        if kind == 0:
            xy = rng.uniform([0, 0], [1600, 1200], size=(n, 2))
        elif kind == 1:
            xy = rng.normal([800, 600], [60, 40], size=(n, 2))
        elif kind == 2:
            t = rng.uniform(0, 1, size=n)
            xy = np.stack([100 + 1400 * t, 300 + 500 * t], axis=1)
        elif kind == 3:
            xy = np.repeat(rng.uniform(0, 1000, size=(max(n // 8, 1), 2)), 8, axis=0)[:n]
        elif kind == 4:
            xy = 500 + rng.uniform(0, 1e-3, size=(n, 2))
        else:
            xy = rng.uniform([1e5, -3e4], [1e5 + 3000, -3e4 + 2000], size=(n, 2))
        n = len(xy)
        desc = synth.quantize_descriptors(rng.gamma(0.7, 1.0, size=(n, 128)))
        layouts.append(dict(descriptors=desc, keypoints=xy.astype(np.float32), width=1600, height=1200,
                            params=(1200.0, 1200.0, 800.0, 600.0)))
    upload_scene(ctx, layouts, False)
    s1, s2 = synth.exhaustive_pairs(n_img)
    s1, s2 = np.concatenate([s1, s2]), np.concatenate([s2, s1])
    total_grid = total_matches = 0
    for rep in range(6):
        tvg = np.zeros(len(s1), dtype=_capi.TVG_DTYPE)
        for p in range(len(s1)):
            tvg["config"][p] = rng.choice([2, 3, 4, 5, 6])
            style = rng.integers(0, 7)
            if style == 0:      # a real epipolar geometry: F = [e]_x A
                e = np.array([rng.uniform(-3000, 4000), rng.uniform(-3000, 4000), 1.0])
                ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
                F = ex @ (np.eye(3) + rng.normal(size=(3, 3)) * [[0.05, 0.05, 30], [0.05, 0.05, 30], [1e-5, 1e-5, 0.05]])
            elif style == 1:    # pure translation along x / y: axis-parallel lines
                t = [1.0, 0.0, 0.0] if rng.integers(2) else [0.0, 1.0, 0.0]
                F = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            elif style == 2:    # epipole inside the image
                e = np.array([rng.uniform(0, 1600), rng.uniform(0, 1200), 1.0])
                ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
                F = ex @ np.eye(3)
            elif style == 3:
                F = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-25, 25)
            else:
                F = rng.normal(size=(3, 3)) * [1e-6, 1e-6, 1e-3]
            tvg["F"][p] = F * 10.0 ** rng.uniform(-3, 3)
            if style in (0, 4):
                H = np.eye(3) + rng.normal(size=(3, 3)) * [[0.1, 0.1, 80], [0.1, 0.1, 80], [1e-5, 1e-5, 0.05]]
            elif style == 1:
                H = np.array([[1, 0, rng.uniform(-50, 50)], [0, 1, rng.uniform(-50, 50)], [0, 0, 1.0]])
            elif style == 2:    # horizon through the image
                H = np.eye(3)
                H[2] = [1.0 / 800, 0, -1.0 + rng.uniform(-0.3, 0.3)]
            elif style == 3:
                H = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-20, 20)
            elif style == 5:    # rank 1
                H = np.outer(rng.normal(size=3), rng.normal(size=3))
            else:
                H = np.diag([rng.uniform(0.2, 5), rng.uniform(0.2, 5), 1.0])
            tvg["H"][p] = H * 10.0 ** rng.uniform(-3, 3)
        max_error = [0.0, 0.7, 4.0, 25.0, 400.0, 1e6][rep]
        # random descriptors never pass the default ratio test: accept every row that has a candidate at all, so that
        # the matches depend on every row's best index (and on the cross check)
        _, mg, ngrid = both_kernels(ctx, s1, s2, tvg, max_error, cross_check=bool(rep % 2 == 0), max_ratio=1.0,
                                    max_distance=2.0)
        total_grid += ngrid
        total_matches += len(mg)
    assert total_grid > len(s1)          # the routing does send plenty of these to the grid kernel
    assert total_matches > 1000
    # and against the oracle, one round
    check_guided(ctx, layouts, s1[:20], s2[:20], tvg[:20], 4.0)


def test_guided_grid_refuses_what_it_cannot_bound(ctx):
    """Non-finite keypoints or models, an all-zero model, a homography whose horizon crosses image 1: the dense kernel."""
    rng = np.random.default_rng(23)
    imgs = synth.multiview_scene(rng, num_images=3, n_feats=300, num_landmarks=400)
    imgs[2] = dict(imgs[2])
    kp = imgs[2]["keypoints"].copy()
    kp[7, 0] = np.nan
    imgs[2]["keypoints"] = kp
    upload_scene(ctx, imgs, False)
    tvg = np.zeros(1, dtype=_capi.TVG_DTYPE)
    tvg["config"] = 3
    tvg["F"][0] = [[0, 0, 0], [0, 0, -1], [0, 1, 0]]
    _, _, ngrid = both_kernels(ctx, [0], [1], tvg, 4.0)
    assert ngrid == 1
    for s in ([0], [2]), ([2], [0]):
        _, _, ngrid = both_kernels(ctx, s[0], s[1], tvg, 4.0)
        assert ngrid == 0                                      # NaN keypoint in image 2
    bad = tvg.copy()
    bad["F"][0][1, 1] = np.inf
    assert both_kernels(ctx, [0], [1], bad, 4.0)[2] == 0
    bad["F"][0] = 0.0
    assert both_kernels(ctx, [0], [1], bad, 4.0)[2] == 0
    h = np.zeros(1, dtype=_capi.TVG_DTYPE)
    h["config"] = 4
    h["H"][0] = [[1, 0, 0], [0, 1, 0], [1.0 / 800, 0, -1]]   # w = x / 800 - 1 changes sign inside image 1
    assert both_kernels(ctx, [0], [1], h, 4.0)[2] == 0
    h["H"][0] = np.eye(3)
    assert both_kernels(ctx, [0], [1], h, 4.0)[2] == 1
