"""GPU parity tests of guided matching (amc_match_guided_pairs = FeatureMatcher::MatchGuided) against
the CPU oracle: match -> verify -> guided re-match with the float32 epipolar / homography filter."""
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
    assert st["pairs_dot4"] == len(s1)
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
