"""Seeded synthetic inputs for the match/verify hot path.

Descriptors follow COLMAP's convention: non-negative, L2-normalised, x512, rounded, clamped to
uint8 (/root/reference/pycolmap/feature/sift.h:76-77 documents the 512 scale).  Un-normalised
random bytes would saturate acos(min(.,1)) = 0 and match nothing, so both generators normalise.
"""
from __future__ import annotations

import numpy as np


def quantize_descriptors(x: np.ndarray) -> np.ndarray:
    """L2-normalise rows, x512, round, clamp to uint8 (COLMAP's storage convention)."""
    x = np.maximum(np.asarray(x, dtype=np.float64), 0.0)
    nrm = np.linalg.norm(x, axis=1, keepdims=True)
    nrm[nrm == 0] = 1.0
    q = np.rint(512.0 * x / nrm)
    return np.clip(q, 0, 255).astype(np.uint8)


def random_descriptors(rng: np.random.Generator, n: int) -> np.ndarray:
    """i.i.d. SIFT-like descriptors (sparse non-negative gamma entries): worst-case epilogue,
    almost no accepted matches."""
    return quantize_descriptors(rng.gamma(0.7, 1.0, size=(n, 128)))


def scene_images(rng: np.random.Generator, num_images: int, n: int, num_landmarks: int | None = None,
                 visible_frac: float = 0.3, sigma_d: float = 0.08) -> list[np.ndarray]:
    """`num_images` descriptor sets of exactly n rows that share landmarks.

    Each image sees a random `visible_frac` of the landmarks (noisy copies of the landmark's
    prototype descriptor, in random row order) and is padded with pure-noise descriptors, so
    neighbouring images have a few hundred true correspondences and the ratio test / cross check
    both fire on real structure.
    """
    L = num_landmarks or max(8, int(n / max(visible_frac, 1e-6) * 0.6))
    proto = rng.gamma(0.7, 1.0, size=(L, 128))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    out = []
    for _ in range(num_images):
        k = min(n, int(round(visible_frac * L)))
        vis = rng.choice(L, size=k, replace=False)
        d = proto[vis] + rng.normal(0.0, sigma_d, size=(k, 128)) * proto[vis].mean()
        if k < n:
            d = np.concatenate([d, rng.gamma(0.7, 1.0, size=(n - k, 128)) * 0.1], axis=0)
        perm = rng.permutation(n)
        out.append(quantize_descriptors(d[perm]))
    return out


def exhaustive_pairs(num_images: int) -> tuple[np.ndarray, np.ndarray]:
    """All unordered pairs (i < j), i-major — the set COLMAP's ExhaustiveFeatureMatcher visits
    (SURVEY.md A.4); block ordering is a host-layer concern."""
    i, j = np.triu_indices(num_images, k=1)
    return i.astype(np.uint32), j.astype(np.uint32)


# ------------------------------------------------------------------------------------------------
# two-view scenes for the verification path
# ------------------------------------------------------------------------------------------------
def _rot(rng, max_angle):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


PLANE_N = np.array([0.1, -0.15, 1.0])
PLANE_D = 6.0


def two_view_scene(rng, num_inliers=300, num_outliers=100, noise=0.5, planar=False, pure_rotation=False,
                   f=1200.0, width=1600, height=1200, extra_keypoints=50):
    """Keypoints of two PINHOLE views of one scene + a match list with planted inliers/outliers.

    Returns dict(pts1 [n1,2], pts2, matches [M,2] uint32, inlier [M] bool, K, R, t, F_true, E_true,
    H_true or None).  Keypoints are rounded through float32 like the `keypoints` blob COLMAP reads
     and returned as float64.
    """
    cx, cy = width / 2.0, height / 2.0
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    R = _rot(rng, 0.25)
    if pure_rotation:
        t = np.zeros(3)
    else:
        t = rng.normal(size=3) * np.array([1.0, 0.3, 0.2])
        t = t / np.linalg.norm(t) * 0.8
    P1, P2 = [], []
    while len(P1) < num_inliers:
        X = rng.uniform([-3, -2.2, 4], [3, 2.2, 9], size=(num_inliers * 2, 3))
        if planar:
            X[:, 2] = (PLANE_D - X[:, 0] * PLANE_N[0] - X[:, 1] * PLANE_N[1]) / PLANE_N[2]
        x1 = (K @ X.T).T
        x1 = x1[:, :2] / x1[:, 2:]
        Xc2 = (R @ X.T).T + t
        x2 = (K @ Xc2.T).T
        x2 = x2[:, :2] / x2[:, 2:]
        ok = ((x1[:, 0] > 5) & (x1[:, 0] < width - 5) & (x1[:, 1] > 5) & (x1[:, 1] < height - 5) &
              (x2[:, 0] > 5) & (x2[:, 0] < width - 5) & (x2[:, 1] > 5) & (x2[:, 1] < height - 5) &
              (Xc2[:, 2] > 0.5))
        P1 += list(x1[ok])
        P2 += list(x2[ok])
    x1 = np.array(P1[:num_inliers]).reshape(-1, 2) + rng.normal(0, noise, size=(num_inliers, 2))
    x2 = np.array(P2[:num_inliers]).reshape(-1, 2) + rng.normal(0, noise, size=(num_inliers, 2))
    o1 = rng.uniform([5, 5], [width - 5, height - 5], size=(num_outliers, 2))
    o2 = rng.uniform([5, 5], [width - 5, height - 5], size=(num_outliers, 2))
    e1 = rng.uniform([5, 5], [width - 5, height - 5], size=(extra_keypoints, 2))
    e2 = rng.uniform([5, 5], [width - 5, height - 5], size=(extra_keypoints, 2))
    pts1 = np.concatenate([x1, o1, e1]).astype(np.float32).astype(np.float64)
    pts2 = np.concatenate([x2, o2, e2]).astype(np.float32).astype(np.float64)
    M = num_inliers + num_outliers
    perm1, perm2 = rng.permutation(len(pts1)), rng.permutation(len(pts2))
    inv1, inv2 = np.argsort(perm1), np.argsort(perm2)
    pts1, pts2 = pts1[perm1], pts2[perm2]
    matches = np.stack([inv1[:M], inv2[:M]], axis=1).astype(np.uint32)
    inlier = np.arange(M) < num_inliers
    order = rng.permutation(M)
    matches, inlier = matches[order], inlier[order]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ E @ Kinv
    H = None
    if planar:
        H = K @ (R + np.outer(t, PLANE_N) / PLANE_D) @ Kinv
    if pure_rotation:
        H = K @ R @ Kinv
    return dict(pts1=pts1, pts2=pts2, matches=matches, inlier=inlier, K=K, R=R, t=t, F_true=F,
                E_true=E, H_true=H, f=f, width=width, height=height)


def multiview_scene(rng, num_images=6, n_feats=600, num_landmarks=900, f=1200.0, width=1600, height=1200,
                    sigma_px=0.5, sigma_d=0.06):
    """Images of one 3-D scene with geometrically consistent keypoints AND matching descriptors:
    the input of the whole match + verify pipeline.  Cameras sit on an arc and
    look at the landmark cloud; each image keeps up to 70 % landmark features (projection + pixel
    noise, descriptor = noisy landmark prototype) and is padded with pure-noise features.
    Returns a list of dict(name, keypoints [n,4] float32 (x, y, scale, orientation), descriptors
    [n,128] uint8, model=1 (PINHOLE), width, height, params)."""
    X = rng.uniform([-4, -2.5, -2], [4, 2.5, 2], size=(num_landmarks, 3))
    proto = rng.gamma(0.7, 1.0, size=(num_landmarks, 128))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    K = np.array([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1.0]])
    images = []
    for i in range(num_images):
        ang = (i - (num_images - 1) / 2.0) * 0.12
        C = np.array([9.0 * np.sin(ang), 0.3 * np.cos(3 * ang), -9.0 * np.cos(ang)])
        z = -C / np.linalg.norm(C)
        x = np.cross([0, 1.0, 0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        Xc = (X - C) @ R.T
        uv = (Xc @ K.T)
        uv = uv[:, :2] / uv[:, 2:]
        vis = np.where((Xc[:, 2] > 1) & (uv[:, 0] > 8) & (uv[:, 0] < width - 8) & (uv[:, 1] > 8) & (uv[:, 1] < height - 8))[0]
        vis = rng.permutation(vis)[:int(0.7 * n_feats)]
        kp = uv[vis] + rng.normal(0, sigma_px, size=(len(vis), 2))
        d = proto[vis] + rng.normal(0, sigma_d, size=(len(vis), 128)) * proto[vis].mean()
        nn = n_feats - len(vis)
        kp = np.concatenate([kp, rng.uniform([8, 8], [width - 8, height - 8], size=(nn, 2))])
        d = np.concatenate([d, rng.gamma(0.7, 1.0, size=(nn, 128)) * 0.1])
        perm = rng.permutation(n_feats)
        kp4 = np.c_[kp[perm], rng.uniform(1, 4, n_feats), rng.uniform(-3.1, 3.1, n_feats)].astype(np.float32)
        images.append(dict(name=f"img_{i:04d}.jpg", keypoints=kp4, descriptors=quantize_descriptors(d[perm]),
                           model=1, width=width, height=height, params=(f, f, width / 2.0, height / 2.0)))
    return images
