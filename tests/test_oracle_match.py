"""CPU tests: the match oracle against an independent numpy restatement, hand-computable
known answers, and the committed golden fixtures (SURVEY.md section 8c lists which to create,
since the reference has none)."""
import numpy as np
import pytest

import oracle_lib
from pycolmap_amd import synth

GOLDEN = __import__("pathlib").Path(__file__).parent / "golden" / "match_golden_v1.npz"


def numpy_match(d1, d2, max_ratio=0.8, max_distance=0.7, cross_check=True):
    """Independent restatement of SURVEY.md A.2 with numpy primitives (argmax = first maximum,
    second = partition), float64 acos. Returns (matches, ambiguous_rows) where ambiguous marks
    rows whose accept/reject margin is below float32 resolution."""
    D = d1.astype(np.int32) @ d2.astype(np.int32).T

    def one_way(M):
        R, Cn = M.shape
        best_idx = M.argmax(axis=1)
        best = M[np.arange(R), best_idx]
        if Cn > 1:
            second = np.partition(M, Cn - 2, axis=1)[:, Cn - 2]
        else:
            second = np.zeros(R, M.dtype)
        second = np.maximum(second, 0)
        a_b = np.arccos(np.minimum(best / 262144.0, 1.0))
        a_s = np.arccos(np.minimum(second / 262144.0, 1.0))
        ok = (best > 0) & ~(a_b > max_distance) & ~(a_b >= max_ratio * a_s)
        margin = np.minimum(np.abs(a_b - max_distance), np.abs(a_b - max_ratio * a_s))
        amb = (best > 0) & (margin < 1e-5)
        return np.where(ok, best_idx, -1), amb

    m12, amb1 = one_way(D)
    m21, amb2 = one_way(D.T)
    out = []
    for i in range(len(d1)):
        j = m12[i]
        if j < 0:
            continue
        if cross_check and m21[j] != i:
            continue
        out.append((i, j))
    return np.array(out, dtype=np.uint32).reshape(-1, 2), amb1.any() or amb2.any()


def test_distance_matrix_equals_numpy_gram():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(37, 128), dtype=np.uint8)
    b = rng.integers(0, 256, size=(53, 128), dtype=np.uint8)
    D = oracle_lib.distance_matrix(a, b)
    assert D.dtype == np.int32
    np.testing.assert_array_equal(D, a.astype(np.int32) @ b.astype(np.int32).T)
    # the documented maximum: 128 * 255^2
    sat = np.full((1, 128), 255, np.uint8)
    assert oracle_lib.distance_matrix(sat, sat)[0, 0] == 128 * 255 * 255 == 8323200


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("opts", [(0.8, 0.7, True), (0.8, 0.7, False), (0.95, 1.3, True)])
def test_oracle_vs_numpy_restatement(seed, opts):
    rng = np.random.default_rng(seed)
    imgs = synth.scene_images(rng, 2, 150 + 7 * seed, num_landmarks=260, visible_frac=0.5)
    got = oracle_lib.match(imgs[0], imgs[1], *opts)
    want, ambiguous = numpy_match(imgs[0], imgs[1], *opts)
    if ambiguous:
        pytest.skip("a row sits within float32 resolution of a threshold")
    np.testing.assert_array_equal(got, want)
    assert len(got) > 10  # the synthetic scene really produces matches


def test_oracle_vs_numpy_at_full_size_4096():
    """BASELINE's image size: 4096 x 4096, scene-like descriptors (hundreds of true matches, ties from
    quantisation).  numpy side: exact Gram matrix through float64 BLAS (every partial sum < 2^53),
    argmax = first maximum, second = second-largest with multiplicity, acos / ratio in float32 exactly as
    A.2 writes them (no ambiguity margin: the comparison is bit for bit)."""
    rng = np.random.default_rng(4096)
    d1, d2 = synth.scene_images(rng, 2, 4096, num_landmarks=6000, visible_frac=0.4)
    D = (d1.astype(np.float64) @ d2.astype(np.float64).T).astype(np.int32)

    def one_way(M):
        R, Cn = M.shape
        best_idx = M.argmax(axis=1)
        best = M[np.arange(R), best_idx]
        second = np.maximum(np.partition(M, Cn - 2, axis=1)[:, Cn - 2], 0)
        k = np.float32(1.0) / (np.float32(512.0) * np.float32(512.0))
        a_b = np.arccos(np.minimum(k * best.astype(np.float32), np.float32(1.0)))
        a_s = np.arccos(np.minimum(k * second.astype(np.float32), np.float32(1.0)))
        assert a_b.dtype == np.float32
        ok = (best > 0) & ~(a_b > np.float32(0.7)) & ~(a_b >= np.float32(0.8) * a_s)
        return np.where(ok, best_idx, -1)

    m12, m21 = one_way(D), one_way(D.T)
    want = np.array([(i, j) for i, j in enumerate(m12) if j >= 0 and m21[j] == i], dtype=np.uint32).reshape(-1, 2)
    got = oracle_lib.match(d1, d2)
    assert len(want) > 300
    # numpy's float32 arccos is not glibc's acosf bit for bit: rows whose test flips within one float32 ulp of a
    # threshold may differ.  Count them instead of hiding them: there must be none here (seeded inputs).
    np.testing.assert_array_equal(got, want)


def _unit(k):
    v = np.zeros(128, np.uint8)
    v[k] = 255
    return v


def test_kat_identity_and_order():
    # 4 orthogonal "one-hot" descriptors (255 at one position): dist(i,i)=65025, else 0.
    d = np.stack([_unit(k) for k in range(4)])
    # best=65025 -> acos(65025/262144)=1.32 > 0.7: rejected by max_distance
    assert len(oracle_lib.match(d, d)) == 0
    # allow the distance: second=0 -> acos(0)=pi/2, 1.32 < 0.8*1.5708=1.2566? no -> rejected by ratio
    assert len(oracle_lib.match(d, d, max_ratio=0.8, max_distance=2.0)) == 0
    m = oracle_lib.match(d, d, max_ratio=0.9, max_distance=2.0)
    np.testing.assert_array_equal(m, np.array([[0, 0], [1, 1], [2, 2], [3, 3]], np.uint32))
    # permuted second image: matches follow the permutation, ascending in idx1
    perm = np.array([2, 0, 3, 1])
    m = oracle_lib.match(d, d[perm], max_ratio=0.9, max_distance=2.0)
    np.testing.assert_array_equal(m[:, 0], np.arange(4))
    np.testing.assert_array_equal(perm[m[:, 1]], np.arange(4))


def test_kat_ties_zeros_and_saturation():
    rng = np.random.default_rng(3)
    base = synth.random_descriptors(rng, 8)
    # duplicate column in image 2: best == second for the matching row -> ratio test rejects
    d2 = base.copy()
    d2[5] = d2[4]
    m = oracle_lib.match(base[:5], d2, max_ratio=0.99, max_distance=3.0, cross_check=False)
    assert 4 not in m[:, 0]
    # all-zero query row: no distance > 0 -> never matched; all-zero column never chosen
    z1 = base.copy(); z1[2] = 0
    m = oracle_lib.match(z1, base, max_ratio=0.99, max_distance=3.0, cross_check=False)
    assert 2 not in m[:, 0]
    # saturated descriptors: dist >= 512^2 clamps acos to 0 for best and second -> 0 >= r*0 rejects
    s = np.full((3, 128), 255, np.uint8)
    assert len(oracle_lib.match(s, s, max_ratio=0.99, max_distance=3.0, cross_check=False)) == 0
    # lowest index wins a tie for best (strict '>'), but the tie also makes second==best, so the
    # row is rejected whenever max_ratio <= 1; with max_ratio > 1 the lowest index must be reported
    q = base[:1]
    dup = np.concatenate([base[3:4], base[0:1], base[0:1]])
    m = oracle_lib.match(q, dup, max_ratio=1.5, max_distance=3.0, cross_check=False)
    np.testing.assert_array_equal(m, np.array([[0, 1]], np.uint32))


def test_kat_empty_and_single():
    rng = np.random.default_rng(4)
    d = synth.random_descriptors(rng, 5)
    e = np.zeros((0, 128), np.uint8)
    assert oracle_lib.match(e, d).shape == (0, 2)
    assert oracle_lib.match(d, e).shape == (0, 2)
    # single column: second = 0 -> acos(0)=pi/2
    m = oracle_lib.match(d[:1], d[:1], max_ratio=0.8, max_distance=0.7)
    np.testing.assert_array_equal(m, np.array([[0, 0]], np.uint32))


def test_without_cross_check_rows_may_share_a_column():
    """No cross check: up to n1 matches even when n2 < n1 (several rows -> one column)."""
    rng = np.random.default_rng(12)
    b = synth.random_descriptors(rng, 3)
    a = np.concatenate([b, b, b])[rng.permutation(9)]
    m = oracle_lib.match(a, b, max_ratio=0.99, max_distance=3.0, cross_check=False)
    assert len(m) == 9 and len(set(m[:, 1].tolist())) == 3
    assert len(oracle_lib.match(a, b, max_ratio=0.99, max_distance=3.0, cross_check=True)) == 0


def test_cross_check_is_transpose_symmetric():
    rng = np.random.default_rng(5)
    imgs = synth.scene_images(rng, 2, 200, num_landmarks=350, visible_frac=0.5)
    ab = oracle_lib.match(imgs[0], imgs[1])
    ba = oracle_lib.match(imgs[1], imgs[0])
    a = {(int(i), int(j)) for i, j in ab}
    b = {(int(j), int(i)) for i, j in ba}
    assert a == b and len(a) > 0


def test_acos_lut_semantics():
    lut = oracle_lib.acos_lut()
    assert lut[0] == np.float32(np.pi / 2) and lut[262144] == 0.0
    assert np.all(np.diff(lut.astype(np.float64)) <= 0)


def test_golden_fixtures():
    g = np.load(GOLDEN)
    imgs = [g[f"desc_{k}"] for k in range(int(g["num_images"]))]
    for name in ("default", "nocross", "loose", "tight"):
        r, d, cc = g[f"{name}_opts"]
        off, m = oracle_lib.match_pairs(imgs, g["slot1"], g["slot2"], float(r), float(d), bool(cc), threads=2)
        np.testing.assert_array_equal(off, g[f"{name}_offsets"])
        np.testing.assert_array_equal(m, g[f"{name}_matches"])


# ---- guided matching (SiftCPUFeatureMatcher::MatchGuided restatement) -------------------------------
def _guided_filter_np(kind, M, max_error, p1, p2):
    """Independent float32 numpy restatement of the filter: [n1, n2] bool, True = rejected."""
    f = np.float32
    m = np.asarray(M, dtype=np.float64).astype(f).reshape(9)
    x1, y1 = p1[:, 0].astype(f)[:, None], p1[:, 1].astype(f)[:, None]
    x2, y2 = p2[:, 0].astype(f)[None, :], p2[:, 1].astype(f)[None, :]
    mr = f(np.float64(max_error) * np.float64(max_error))
    with np.errstate(all="ignore"):
        if kind == "F":
            a0 = (m[0] * x1 + m[1] * y1) + m[2]
            a1 = (m[3] * x1 + m[4] * y1) + m[5]
            a2 = (m[6] * x1 + m[7] * y1) + m[8]
            b0 = (m[0] * x2 + m[3] * y2) + m[6]
            b1 = (m[1] * x2 + m[4] * y2) + m[7]
            num = (x2 * a0 + y2 * a1) + a2
            den = ((a0 * a0 + a1 * a1) + b0 * b0) + b1 * b1
            return (num * num) / den > mr
        h0 = (m[0] * x1 + m[1] * y1) + m[2]
        h1 = (m[3] * x1 + m[4] * y1) + m[5]
        h2 = (m[6] * x1 + m[7] * y1) + m[8]
        e0 = h0 / h2 - x2
        e1 = h1 / h2 - y2
        return e0 * e0 + e1 * e1 > mr


def match_from_dists_numpy(D, max_ratio, max_distance, cross_check):
    """FindBestMatchesBruteForce on a given int distance matrix with numpy primitives; the float32
    acos values come from the host-libm table (exact: (float)d * 2^-18 has no rounding)."""
    lut = oracle_lib.acos_lut()
    r, t = np.float32(max_ratio), np.float32(max_distance)

    def one_way(M):
        R, Cn = M.shape
        best_idx = M.argmax(axis=1)                       # first maximum = lowest index among ties
        best = M[np.arange(R), best_idx]
        second = np.partition(M, Cn - 2, axis=1)[:, Cn - 2] if Cn > 1 else np.zeros(R, M.dtype)
        second = np.maximum(second, 0)
        a_b = lut[np.minimum(best, 262144)]
        a_s = lut[np.minimum(second, 262144)]
        ok = (best > 0) & ~(a_b > t) & ~(a_b >= r * a_s)
        return np.where(ok, best_idx, -1)

    if D.shape[0] == 0 or D.shape[1] == 0:
        return np.zeros((0, 2), np.uint32)
    m12 = one_way(D)
    m21 = one_way(D.T)
    out = [(i, j) for i, j in enumerate(m12) if j >= 0 and (not cross_check or m21[j] == i)]
    return np.array(out, dtype=np.uint32).reshape(-1, 2)


def _guided_np(d1, kp1, d2, kp2, kind, M, max_error, max_ratio=0.8, max_distance=0.7, cross_check=True):
    dists = d1.astype(np.int64) @ d2.astype(np.int64).T
    dists[_guided_filter_np(kind, M, max_error, kp1, kp2)] = 0
    return match_from_dists_numpy(dists, max_ratio, max_distance, cross_check)


def test_guided_matching_oracle_against_numpy_restatement():
    from pycolmap_amd import synth
    rng = np.random.default_rng(77)
    imgs = synth.multiview_scene(rng, num_images=3, n_feats=300, num_landmarks=400)
    for a, b in ((0, 1), (0, 2), (2, 1)):
        sc_a, sc_b = imgs[a], imgs[b]
        kp1, kp2 = sc_a["keypoints"][:, :2], sc_b["keypoints"][:, :2]
        F = rng.normal(size=(3, 3)) * 1e-3
        H = np.eye(3) + rng.normal(size=(3, 3)) * 1e-3
        for cfg, kind, M in ((2, "F", F), (3, "F", F), (4, "H", H), (5, "H", H), (6, "H", H)):
            for err in (0.5, 4.0, 50.0):
                got = oracle_lib.match_guided(sc_a["descriptors"], kp1, sc_b["descriptors"], kp2, cfg, F, H, err)
                want = _guided_np(sc_a["descriptors"], kp1, sc_b["descriptors"], kp2, kind, M, err)
                np.testing.assert_array_equal(got, want)
        for cfg in (0, 1, 7, 8):   # UNDEFINED, DEGENERATE, WATERMARK, MULTIPLE: no guided matching
            assert oracle_lib.match_guided(sc_a["descriptors"], kp1, sc_b["descriptors"], kp2, cfg, F, H, 4.0) is None


def test_guided_matching_limits():
    """A filter that accepts everything reproduces plain matching; one that rejects everything leaves
    an all-zero distance matrix, i.e. no matches; points exactly on the epipolar line pass."""
    from pycolmap_amd import synth
    rng = np.random.default_rng(78)
    imgs = synth.multiview_scene(rng, num_images=2, n_feats=256, num_landmarks=300)
    d1, d2 = imgs[0]["descriptors"], imgs[1]["descriptors"]
    kp1, kp2 = imgs[0]["keypoints"][:, :2], imgs[1]["keypoints"][:, :2]
    F = np.array([[0, -1e-3, 0.2], [1e-3, 0, -0.3], [-0.2, 0.3, 0.0]])
    H = np.eye(3)
    plain = oracle_lib.match(d1, d2)
    np.testing.assert_array_equal(oracle_lib.match_guided(d1, kp1, d2, kp2, 3, F, H, 1e9), plain)
    np.testing.assert_array_equal(oracle_lib.match_guided(d1, kp1, d2, kp2, 4, F, H, 1e9), plain)
    assert len(oracle_lib.match_guided(d1, kp1, d2, kp2, 4, F, np.eye(3) * 0 + [[0, 0, 1e6]] * 3, 1e-3)) == 0
    # H = identity, max_error 1: only pairings closer than 1 px survive the filter
    kp2_near = (kp1 + 0.25).astype(np.float32)
    got = oracle_lib.match_guided(d1, kp1, d1, kp2_near, 6, F, np.eye(3), 1.0, max_ratio=1.0, max_distance=10.0)
    assert len(got) > 0 and np.all(got[:, 0] == got[:, 1])


# ---- the AVX-512 VNNI variant (oracle/match_vnni.c): bench.py's "optimised" CPU baseline ---------------------
def _vnni_or_skip():
    if not oracle_lib.vnni_available():
        pytest.skip("host without AVX-512 VNNI")


def test_vnni_variant_equals_the_literal_oracle():
    """Same matches as match_oracle.c on scene-like, ragged, tie-heavy, all-zero, saturated and duplicated inputs, with
    and without the cross check, at thresholds other than the defaults tooracle_lib."""
    _vnni_or_skip()
    rng = np.random.default_rng(42)
    imgs = synth.scene_images(rng, 5, 600, num_landmarks=800, visible_frac=0.5)
    imgs += [np.zeros((37, 128), np.uint8), rng.integers(0, 3, (129, 128)).astype(np.uint8), np.full((5, 128), 255, np.uint8),
             np.repeat(imgs[0][:40], 3, axis=0), imgs[1][:1], imgs[2][:17], rng.integers(0, 256, (33, 128)).astype(np.uint8)]
    for a in range(len(imgs)):
        for b in range(len(imgs)):
            for kw in (dict(), dict(cross_check=False), dict(max_ratio=0.95, max_distance=1.2)):
                np.testing.assert_array_equal(oracle_lib.match_vnni(imgs[a], imgs[b], **kw), oracle_lib.match(imgs[a], imgs[b], **kw),
                                              err_msg=f"{a} {b} {kw}")
    assert len(oracle_lib.match_vnni(np.zeros((0, 128), np.uint8), imgs[0])) == 0


def test_vnni_batched_entry_point_equals_the_literal_one():
    _vnni_or_skip()
    rng = np.random.default_rng(43)
    imgs = synth.scene_images(rng, 6, 300, num_landmarks=500, visible_frac=0.5) + [np.zeros((0, 128), np.uint8)]
    s1, s2 = np.array([0, 1, 2, 3, 6, 5, 0], np.uint32), np.array([1, 2, 3, 4, 0, 6, 5], np.uint32)
    off, m = oracle_lib.match_pairs(imgs, s1, s2, threads=3)
    voff, vm = oracle_lib.match_pairs(imgs, s1, s2, threads=3, variant="vnni")
    np.testing.assert_array_equal(off, voff)
    np.testing.assert_array_equal(m, vm)
