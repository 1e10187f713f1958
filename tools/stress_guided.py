"""Randomised A/B of the two guided-matching kernels (candidate generation vs the dense filtered scan, forced with
AMC_GUIDED_DENSE=1): random keypoint layouts, models from real epipolar geometries to garbage, thresholds 0..1e4.
Prints RESULT mismatches=<n>.  usage: stress_guided.py [rounds=20] [seed=1]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _capi.Context(0)
bad = total = grid_pairs = 0
for rd in range(rounds):
    n_img = 10
    imgs = []
    for k in range(n_img):
        n = int(rng.integers(1, 3000))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            xy = rng.uniform([0, 0], [1600, 1200], size=(n, 2))
        elif kind == 1:
            xy = rng.normal([800, 600], [rng.uniform(1, 300), rng.uniform(1, 300)], size=(n, 2))
        elif kind == 2:
            t = rng.uniform(0, 1, size=n)
            xy = np.stack([100 + 1400 * t, 300 + rng.uniform(-500, 500) * t], axis=1) + rng.normal(0, rng.uniform(0, 2), size=(n, 2))
        elif kind == 3:
            xy = np.repeat(rng.uniform(0, 1000, size=(max(n // 8, 1), 2)), 8, axis=0)[:n]
        elif kind == 4:
            xy = rng.uniform(-1e4, 1e4, size=(1, 2)) + rng.uniform(0, 10.0 ** rng.uniform(-3, 3), size=(n, 2))
        elif kind == 5:
            xy = np.round(rng.uniform([0, 0], [1600, 1200], size=(n, 2)))          # integer grid: exact cell boundaries
        else:
            xy = rng.uniform([0, 0], [4000, 3000], size=(n, 2)) * [1.0, 1e-3]      # very flat box
        n = len(xy)
        imgs.append(dict(descriptors=synth.quantize_descriptors(rng.gamma(0.7, 1.0, size=(n, 128))),
                         keypoints=xy.astype(np.float32)))
    ctx.reserve_slots(n_img)
    for k, im in enumerate(imgs):
        ctx.upload_descriptors(k, im["descriptors"])
        ctx.upload_keypoints(k, im["keypoints"])
    s1, s2 = synth.exhaustive_pairs(n_img)
    s1, s2 = np.concatenate([s1, s2]), np.concatenate([s2, s1])
    tvg = np.zeros(len(s1), dtype=_capi.TVG_DTYPE)
    for p in range(len(s1)):
        tvg["config"][p] = rng.choice([2, 3, 4, 5, 6])
        st = int(rng.integers(0, 6))
        e = np.array([rng.uniform(-5000, 6000), rng.uniform(-5000, 6000), 1.0])
        ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
        A = np.eye(3) + rng.normal(size=(3, 3)) * [[0.05, 0.05, 30], [0.05, 0.05, 30], [1e-5, 1e-5, 0.05]]
        F = [ex @ A, ex, np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), rng.normal(size=(3, 3)) * [1e-6, 1e-6, 1e-3],
             rng.normal(size=(3, 3)), ex @ A][st]
        H = [A, np.diag([rng.uniform(0.1, 10), rng.uniform(0.1, 10), 1.0]), np.eye(3) + rng.normal(size=(3, 3)) * 1e-4,
             np.array([[1, 0, 0], [0, 1, 0], [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 1.0]]),
             rng.normal(size=(3, 3)), np.linalg.inv(A)][st]
        tvg["F"][p] = F * 10.0 ** rng.uniform(-4, 4)
        tvg["H"][p] = H * 10.0 ** rng.uniform(-4, 4)
    max_error = float(rng.choice([0.0, 0.3, 1.0, 4.0, 4.0, 16.0, 100.0, 1e4]))
    cc = bool(rng.integers(0, 2))
    # random descriptors never pass the default ratio test: most rounds accept every row that has a candidate at all,
    # so that the matches depend on every row's best index and on the cross check
    kw = dict(cross_check=cc) if rd % 4 == 3 else dict(cross_check=cc, max_ratio=1.0, max_distance=2.0)
    off_g, m_g, st_g = ctx.match_guided_pairs(s1, s2, tvg, max_error, **kw)
    os.environ["AMC_GUIDED_DENSE"] = "1"
    off_d, m_d, _ = ctx.match_guided_pairs(s1, s2, tvg, max_error, **kw)
    del os.environ["AMC_GUIDED_DENSE"]
    for p in range(len(s1)):
        a = m_g[int(off_g[p]):int(off_g[p + 1])]
        b = m_d[int(off_d[p]):int(off_d[p + 1])]
        total += 1
        if not np.array_equal(a, b):
            bad += 1
            print(f"MISMATCH round {rd} pair {p} ({s1[p]},{s2[p]}) config {tvg['config'][p]} max_error {max_error}: {len(a)} vs {len(b)} matches", flush=True)
    grid_pairs += st_g["pairs_guided_grid"]
    print(f"round {rd}: {len(s1)} pairs, max_error {max_error}, cross_check {cc}, grid {st_g['pairs_guided_grid']}, matches {len(m_g)}; "
          f"bad so far {bad}/{total}", flush=True)
print(f"RESULT mismatches={bad} of {total} (pairs by candidate generation: {grid_pairs})")
sys.exit(1 if bad else 0)
