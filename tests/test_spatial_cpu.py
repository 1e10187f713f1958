"""match_spatial's pair generation (host logic, no GPU): `_spatial_blocks` / `_ell_to_xyz` of the C++ host layer against
an independent numpy restatement of COLMAP 3.9.1's SpatialFeatureMatcher::Run (feature/matching.cc: location matrix
in float32, exhaustive k-NN by squared L2 accumulated left to right, neighbours in ascending distance and, at equal
distance, row order; stop at the first neighbour beyond max_distance^2; skip the query itself)."""
import numpy as np
import pytest

import pycolmap_amd as pycolmap

_p = pycolmap._pycolmap


def ell_to_xyz(lat, lon, alt):
    a, b = 6378137.0, 6356752.314245
    e2 = (a * a - b * b) / (a * a)
    la, lo = np.deg2rad(lat), np.deg2rad(lon)
    n = a / np.sqrt(1 - e2 * np.sin(la) ** 2)
    return np.array([(n + alt) * np.cos(la) * np.cos(lo), (n + alt) * np.cos(la) * np.sin(lo), (n * (1 - e2) + alt) * np.sin(la)])


def spatial_blocks(ids, priors, is_gps, ignore_z, knn_max, max_distance):
    keep, loc = [], []
    for i, t in enumerate(priors):
        t = np.asarray(t, np.float64)
        if np.isnan(t[0]) or np.isnan(t[1]) or (not ignore_z and np.isnan(t[2])):
            continue
        if (t[0] == 0 and t[1] == 0 and ignore_z) or (not ignore_z and not t.any()):
            continue
        x = np.array([t[0], t[1], 0.0 if ignore_z else t[2]])
        if is_gps:
            x = ell_to_xyz(*x)
        keep.append(i)
        loc.append(x.astype(np.float32))
    if not keep:
        return []
    loc = np.stack(loc)
    n = len(keep)
    knn = min(knn_max, n)
    lim = np.float32(max_distance * max_distance)
    out = []
    for i in range(n):
        diff = loc[i][None, :] - loc                                # float32
        sq = diff * diff
        d = (sq[:, 0] + sq[:, 1]) + sq[:, 2]                        # left to right, float32
        order = np.argsort(d, kind="stable")[:knn]
        pairs = []
        for j in order:
            if j == i:
                continue
            if d[j] > lim:
                break
            pairs.append((ids[keep[i]], ids[keep[j]]))
        out.append(pairs)
    return out


def opts(**kw):
    o = pycolmap.SpatialMatchingOptions()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_defaults_are_colmaps():
    o = pycolmap.SpatialMatchingOptions()
    assert (o.is_gps, o.ignore_z, o.max_num_neighbors, o.max_distance) == (True, True, 50, 100.0)
    assert pycolmap.SpatialMatchingOptions(dict(is_gps=False)).is_gps is False


def test_ell_to_xyz_wgs84():
    a, b = 6378137.0, 6356752.314245
    np.testing.assert_allclose(_p._ell_to_xyz([0.0, 0.0, 0.0]), [a, 0, 0], atol=1e-9)
    np.testing.assert_allclose(_p._ell_to_xyz([0.0, 90.0, 10.0]), [0, a + 10, 0], atol=1e-9)
    np.testing.assert_allclose(_p._ell_to_xyz([90.0, 0.0, 0.0]), [0, 0, b], atol=1e-8)
    rng = np.random.default_rng(0)
    for _ in range(200):
        lat, lon, alt = rng.uniform(-90, 90), rng.uniform(-180, 180), rng.uniform(-100, 9000)
        np.testing.assert_allclose(_p._ell_to_xyz([lat, lon, alt]), ell_to_xyz(lat, lon, alt), rtol=1e-15, atol=1e-8)
    # one degree of latitude is ~111 km on the ground
    d = np.linalg.norm(np.array(_p._ell_to_xyz([48.0, 11.0, 0.0])) - np.array(_p._ell_to_xyz([49.0, 11.0, 0.0])))
    assert 111.0e3 < d < 111.4e3


@pytest.mark.parametrize("seed", range(12))
def test_blocks_match_the_restatement(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 90))
    is_gps, ignore_z = bool(seed & 1), bool(seed & 2)
    if is_gps:
        pri = np.stack([48 + rng.uniform(0, 0.004, n), 11 + rng.uniform(0, 0.004, n), rng.uniform(400, 460, n)], 1)
    else:
        pri = rng.uniform(-150, 150, (n, 3))
    if seed % 3 == 0:                                               # ties: a lattice
        pri[:, :2] = np.round(pri[:, :2] * (2000 if is_gps else 0.05)) / (2000 if is_gps else 0.05)
    pri[rng.random(n) < 0.1] = 0.0                                  # "unset"
    pri[rng.random(n) < 0.1] = np.nan                               # NULL columns
    if n > 3:
        pri[3, :2] = 0.0                                            # unset only under ignore_z
        pri[3, 2] = 5.0
    ids = list(rng.permutation(np.arange(1, 3 * n))[:n].astype(int))
    knn, dist = int(rng.integers(1, 60)), float(rng.choice([30.0, 100.0, 1e6]))
    got = _p._spatial_blocks(ids, pri.tolist(), opts(is_gps=is_gps, ignore_z=ignore_z, max_num_neighbors=knn, max_distance=dist))
    exp = spatial_blocks(ids, pri, is_gps, ignore_z, knn, dist)
    assert [list(map(tuple, b)) for b in got] == exp
    assert all(len(b) <= knn for b in got)


def test_rejects_bad_options():
    with pytest.raises(ValueError):
        _p._spatial_blocks([1], [[1.0, 1.0, 0.0]], opts(max_num_neighbors=0))
    with pytest.raises(ValueError):
        _p._spatial_blocks([1], [[1.0, 1.0, 0.0]], opts(max_distance=0.0))
    with pytest.raises(ValueError):
        _p._spatial_blocks([1, 2], [[1.0, 1.0, 0.0]], opts())
    assert _p._spatial_blocks([], [], opts()) == []
    assert _p._spatial_blocks([7], [[0.0, 0.0, 3.0]], opts()) == []          # nothing with a location
