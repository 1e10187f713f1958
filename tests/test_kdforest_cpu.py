"""COLMAP's default CPU matcher (FLANN k-d forest, approximate: SURVEY.md section 8 row M4) as restated in
oracle/match_kdforest.cc - a CPU baseline timed by bench.py, not a parity target.  The checks follow the reference's
test strategy for it (self-match is the identity; agreement with brute force >= ~95 %) and pin the
index itself: with as many checks as points the forest's 2-NN is the exact 2-NN by squared L2."""
import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import synth


def pair(seed, n=1500, sigma=20.0):
    """Two views of one synthetic scene; the second image's descriptors with extra noise."""
    rng = np.random.default_rng(seed)
    sc = synth.multiview_scene(rng, num_images=2, n_feats=n)
    d2 = np.clip(sc[1]["descriptors"].astype(np.float64) + rng.normal(0, sigma, (n, 128)), 0, 255).astype(np.uint8)
    return sc[0]["descriptors"], d2


def l2_2nn(index, q):
    a = index.astype(np.int32)
    b = q.astype(np.int32)
    d = (b * b).sum(1)[:, None] + (a * a).sum(1)[None, :] - 2 * (b @ a.T)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    return order, np.take_along_axis(d, order, 1)


def test_exhaustive_checks_give_the_exact_neighbours():
    d1, d2 = pair(0, 700)
    idx, dist = o.kdforest_knn(d2, d1, checks=len(d2))
    ref_idx, ref_dist = l2_2nn(d2, d1)
    np.testing.assert_array_equal(dist, ref_dist.astype(np.float32))     # distances exact; indices up to ties
    same = (idx == ref_idx).all(1) | (ref_dist[:, 0] == ref_dist[:, 1])
    tie2 = ref_dist[:, 1] == np.sort(((d1.astype(np.int32)[:, None, :] - d2.astype(np.int32)[None, :, :]) ** 2).sum(2), 1)[:, 2]
    assert (same | tie2).all()


def test_default_checks_find_most_neighbours():
    d1, d2 = pair(1)
    idx, dist = o.kdforest_knn(d2, d1)
    ref_idx, ref_dist = l2_2nn(d2, d1)
    assert (dist[:, 0] >= ref_dist[:, 0]).all() and (dist[:, 1] >= ref_dist[:, 1]).all()      # never better than exact
    close = ref_dist[:, 0] < 0.5 * ref_dist[:, 1]                        # queries with a distinctive neighbour
    assert close.sum() > 100
    assert (idx[close, 0] == ref_idx[close, 0]).mean() > 0.97


def test_self_match_is_the_identity():
    d1, _ = pair(2, 1200)
    d = np.unique(d1, axis=0)
    m = o.match_kdforest(d, d)
    assert (m[:, 0] == m[:, 1]).all() and len(m) > 0.95 * len(d)


@pytest.mark.parametrize("cross_check", [True, False])
def test_agreement_with_brute_force(cross_check):
    agree, found = [], []
    for seed in range(3, 7):
        d1, d2 = pair(seed)
        bf = {tuple(r) for r in o.match(d1, d2, 0.8, 0.7, cross_check)}
        kd = {tuple(r) for r in o.match_kdforest(d1, d2, 0.8, 0.7, cross_check, seed=seed)}
        assert len(bf) > 100
        found.append(len(bf & kd) / len(bf))              # brute-force matches the forest finds
        agree.append(len(bf & kd) / max(1, len(kd)))      # forest matches that are brute-force matches
    assert np.mean(found) >= 0.95 and np.mean(agree) >= 0.95, (found, agree)


def test_batched_form_and_edge_cases():
    d1, d2 = pair(8, 300)
    empty = np.zeros((0, 128), np.uint8)
    one = d2[:1]
    imgs = [d1, d2, empty, one]
    s1 = np.array([0, 0, 2, 0, 1], np.uint32)
    s2 = np.array([1, 2, 1, 3, 0], np.uint32)
    off, m = o.match_pairs(imgs, s1, s2, variant="kdforest", threads=2)
    assert off[2] == off[1] and off[3] == off[2]                         # empty images: no matches
    first = m[:int(off[1])]
    assert len(first) > 20 and (first[:, 0] < len(d1)).all() and (first[:, 1] < len(d2)).all()
    single = m[int(off[3]):int(off[4])]
    assert (single[:, 1] == 0).all() and len(single) <= 1                # one candidate, cross-checked
    # (1, 0) is (0, 1) transposed up to the approximation
    a = {tuple(r) for r in first}
    b = {(int(y), int(x)) for x, y in m[int(off[4]):int(off[5])]}
    assert len(a & b) >= 0.9 * min(len(a), len(b))
    # deterministic
    off2, m2 = o.match_pairs(imgs, s1, s2, variant="kdforest", threads=1)
    np.testing.assert_array_equal(off, off2)
    np.testing.assert_array_equal(m, m2)
