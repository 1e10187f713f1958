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
