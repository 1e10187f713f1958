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
                    sigma_px=0.5, sigma_d=0.06, camera=None):
    """Images of one 3-D scene with geometrically consistent keypoints AND matching descriptors:
    the input of the whole match + verify pipeline.  Cameras sit on an arc and
    look at the landmark cloud; each image keeps up to 70 % landmark features (projection + pixel
    noise, descriptor = noisy landmark prototype) and is padded with pure-noise features.
    Returns a list of dict(name, keypoints [n,4] float32 (x, y, scale, orientation), descriptors
    [n,128] uint8, model=1 (PINHOLE), width, height, params).  camera = (model name, params): every image
    is taken with that camera instead (keypoints projected with img_from_cam below; model = its id)."""
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
        if camera is None:
            uv = (Xc @ K.T)
            uv = uv[:, :2] / uv[:, 2:]
        else:
            zc = np.where(Xc[:, 2] > 1, Xc[:, 2], 1.0)
            uv = img_from_cam(camera[0], camera[1], Xc[:, :2] / zc[:, None])
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
                           model=1 if camera is None else CAMERA_MODEL_IDS[camera[0]], width=width, height=height,
                           params=(f, f, width / 2.0, height / 2.0) if camera is None else tuple(camera[1])))
    return images


# ------------------------------------------------------------------------------------------------
# camera models (COLMAP 3.9.1 colmap/sensor/models.h): forward projection for generating inputs
# ------------------------------------------------------------------------------------------------
CAMERA_MODEL_IDS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "SIMPLE_RADIAL": 2, "RADIAL": 3, "OPENCV": 4,
                    "OPENCV_FISHEYE": 5, "FULL_OPENCV": 6, "FOV": 7, "SIMPLE_RADIAL_FISHEYE": 8,
                    "RADIAL_FISHEYE": 9, "THIN_PRISM_FISHEYE": 10}
_NUM_FOCAL = {0: 1, 1: 2, 2: 1, 3: 1, 4: 2, 5: 2, 6: 2, 7: 2, 8: 1, 9: 1, 10: 2}

# plausible parameter vectors of a 1600 x 1200 sensor, one per model (tests, golden fixtures, benches)
EXAMPLE_CAMERAS = {
    "SIMPLE_PINHOLE": (1150.0, 805.0, 598.0),
    "PINHOLE": (1200.0, 1190.0, 800.0, 600.0),
    "SIMPLE_RADIAL": (1180.0, 802.0, 597.0, -0.06),
    "RADIAL": (1180.0, 802.0, 597.0, -0.07, 0.02),
    "OPENCV": (1210.0, 1195.0, 795.0, 605.0, -0.08, 0.03, 0.001, -0.0015),
    "OPENCV_FISHEYE": (900.0, 905.0, 800.0, 600.0, -0.03, 0.01, -0.002, 0.0005),
    "FULL_OPENCV": (1210.0, 1195.0, 795.0, 605.0, -0.08, 0.03, 0.001, -0.0015, 0.004, 0.01, -0.005, 0.001),
    "FOV": (1100.0, 1105.0, 800.0, 600.0, 0.6),
    "SIMPLE_RADIAL_FISHEYE": (900.0, 800.0, 600.0, -0.02),
    "RADIAL_FISHEYE": (900.0, 800.0, 600.0, -0.03, 0.008),
    "THIN_PRISM_FISHEYE": (900.0, 905.0, 800.0, 600.0, -0.03, 0.01, 0.0005, -0.0008, -0.002, 0.0004, 0.001, -0.0012),
}


def img_from_cam(model, params, uv: np.ndarray) -> np.ndarray:
    """Camera::ImgFromCam for an N x 2 array of normalised image-plane points (float64 numpy; used to
    synthesise keypoints of distorted cameras, not a parity path)."""
    mid = CAMERA_MODEL_IDS[model] if isinstance(model, str) else int(model)
    p = np.asarray(params, dtype=np.float64)
    nf = _NUM_FOCAL[mid]
    f1, f2, c1, c2 = p[0], p[nf - 1], p[nf], p[nf + 1]
    e = p[nf + 2:]
    u, v = np.asarray(uv, dtype=np.float64).reshape(-1, 2).T
    if mid == 10:  # equidistant projection first
        r = np.sqrt(u * u + v * v)
        th = np.arctan(r)
        s = np.where(r > 1e-15, th / np.where(r > 1e-15, r, 1.0), 1.0)
        u, v = u * s, v * s
    u2, v2, uv_ = u * u, v * v, u * v
    r2 = u2 + v2
    if mid in (0, 1):
        du = dv = 0.0
    elif mid == 2:
        du, dv = u * (e[0] * r2), v * (e[0] * r2)
    elif mid == 3:
        rad = e[0] * r2 + e[1] * r2 * r2
        du, dv = u * rad, v * rad
    elif mid == 4:
        rad = e[0] * r2 + e[1] * r2 * r2
        du = u * rad + 2 * e[2] * uv_ + e[3] * (r2 + 2 * u2)
        dv = v * rad + 2 * e[3] * uv_ + e[2] * (r2 + 2 * v2)
    elif mid == 6:
        r4, r6 = r2 * r2, r2 * r2 * r2
        rad = (1 + e[0] * r2 + e[1] * r4 + e[4] * r6) / (1 + e[5] * r2 + e[6] * r4 + e[7] * r6)
        du = u * rad + 2 * e[2] * uv_ + e[3] * (r2 + 2 * u2) - u
        dv = v * rad + 2 * e[3] * uv_ + e[2] * (r2 + 2 * v2) - v
    elif mid in (5, 8, 9):
        r = np.sqrt(r2)
        th = np.arctan(r)
        th2 = th * th
        k = list(e) + [0.0] * (4 - len(e))
        thd = th * (1 + k[0] * th2 + k[1] * th2 ** 2 + k[2] * th2 ** 3 + k[3] * th2 ** 4)
        s = np.where(r > 1e-15, thd / np.where(r > 1e-15, r, 1.0), 1.0)
        du, dv = u * s - u, v * s - v
    elif mid == 7:
        om = e[0]
        r = np.sqrt(r2)
        fac = np.arctan(r * 2 * np.tan(om / 2)) / np.where(r > 1e-9, r * om, 1.0)
        fac = np.where(r > 1e-9, fac, 2 * np.tan(om / 2) / om)
        du, dv = u * fac - u, v * fac - v
    elif mid == 10:
        r4, r6, r8 = r2 * r2, r2 ** 3, r2 ** 4
        rad = e[0] * r2 + e[1] * r4 + e[4] * r6 + e[5] * r8
        du = u * rad + 2 * e[2] * uv_ + e[3] * (r2 + 2 * u2) + e[6] * r2
        dv = v * rad + 2 * e[3] * uv_ + e[2] * (r2 + 2 * v2) + e[7] * r2
    else:
        raise ValueError(f"unknown camera model {model}")
    return np.stack([f1 * (u + du) + c1, f2 * (v + dv) + c2], axis=1)


def recamera_scene(sc: dict, model1, params1, model2, params2) -> dict:
    """The two-view scene `sc` (pinhole views, focal sc['f'], principal point at the centre) seen
    through two other cameras: every keypoint is taken back to the normalised plane and projected
    with the given model; float32-rounded like the `keypoints` blob.  Geometry (R, t, inlier set)
    is unchanged, so E stays recoverable on the calibrated path."""
    out = dict(sc)
    cx, cy, f = sc["width"] / 2.0, sc["height"] / 2.0, sc["f"]
    for key, (m, p) in (("pts1", (model1, params1)), ("pts2", (model2, params2))):
        uv = (sc[key] - np.array([cx, cy])) / f
        out[key] = img_from_cam(m, p, uv).astype(np.float32).astype(np.float64)
    return out


def tower_scene(rng, num_images=500, n_feats=4096, f=1200.0, width=1600, height=1200, sigma_px=0.5, sigma_d=0.06,
                images_per_turn=48, landmark_frac=0.65, camera=None):
    """An orbit capture with real geometry AND the sparse overlap of a real exhaustive-matching job (BASELINE.json
    configs[2]): the cameras climb a helix inside a cylindrical wall of landmarks and look outward, `images_per_turn`
    images per revolution, 0.8 vertical fields of view per revolution.  An image overlaps its ~8 neighbours on either
    side (and a few images one turn below / above); most of the N(N-1)/2 pairs see disjoint parts of the wall.
    Per image: up to landmark_frac * n_feats projected landmarks (pixel noise sigma_px, descriptor = noisy landmark
    prototype), padded with noise features to exactly n_feats.  Returns the list of image dicts multiview_scene
    returns (name, keypoints [n,4] float32, descriptors [n,128] uint8, model, width, height, params)."""
    Rc, Rw = 1.0, 6.0                                    # camera helix radius, wall radius
    dphi = 2.0 * np.pi / images_per_turn
    vis_h = 2.0 * (Rw - Rc) * (height / 2.0) / f          # wall height one image sees
    dz = 0.8 * vis_h / images_per_turn              # consecutive turns share a fifth of their height
    turns = num_images / images_per_turn
    want = int(landmark_frac * n_feats)
    # landmark density: `want` of them inside one view (window ~ 2 atan(w / 2f) of the wall, vis_h tall)
    win = 2.0 * np.arctan(width / 2.0 / f) * 0.95
    total_h = turns * 0.8 * vis_h + vis_h
    L = int(want * 1.25 * (2.0 * np.pi / win) * (total_h / vis_h))
    th = rng.uniform(0, 2 * np.pi, L)
    zz = rng.uniform(-vis_h / 2, total_h - vis_h / 2, L)
    rr = Rw + rng.uniform(-0.6, 0.6, L)                   # a rough wall: real depth variation, no plane
    X = np.stack([rr * np.cos(th), rr * np.sin(th), zz], axis=1)
    proto = rng.gamma(0.7, 1.0, size=(L, 128)).astype(np.float32)
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    order = np.argsort(zz)                                # visibility test on a z-window only
    Xs, zs = X[order], zz[order]
    K = np.array([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1.0]])
    images = []
    for i in range(num_images):
        phi, zc = i * dphi, i * dz
        C = np.array([Rc * np.cos(phi), Rc * np.sin(phi), zc])
        zax = np.array([np.cos(phi), np.sin(phi), 0.0])   # looking outward
        xax = np.array([-np.sin(phi), np.cos(phi), 0.0])
        yax = np.cross(zax, xax)
        R = np.stack([xax, yax, zax])
        lo, hi = np.searchsorted(zs, [zc - vis_h, zc + vis_h])
        Xc = (Xs[lo:hi] - C) @ R.T
        front = Xc[:, 2] > 1.0
        zc_ = np.where(front, Xc[:, 2], 1.0)
        if camera is None:
            uv = Xc[:, :2] / zc_[:, None] * f + np.array([width / 2.0, height / 2.0])
        else:
            uv = img_from_cam(camera[0], camera[1], Xc[:, :2] / zc_[:, None])
        vis = np.where(front & (uv[:, 0] > 8) & (uv[:, 0] < width - 8) & (uv[:, 1] > 8) & (uv[:, 1] < height - 8))[0]
        vis = rng.permutation(vis)[:want]
        gid = order[lo:hi][vis]
        kp = uv[vis] + rng.normal(0, sigma_px, size=(len(vis), 2))
        d = proto[gid] + rng.normal(0, sigma_d, size=(len(vis), 128)).astype(np.float32) * proto[gid].mean()
        nn = n_feats - len(vis)
        kp = np.concatenate([kp, rng.uniform([8, 8], [width - 8, height - 8], size=(nn, 2))])
        d = np.concatenate([d, rng.gamma(0.7, 1.0, size=(nn, 128)).astype(np.float32) * 0.1])
        perm = rng.permutation(n_feats)
        kp4 = np.c_[kp[perm], rng.uniform(1, 4, n_feats), rng.uniform(-3.1, 3.1, n_feats)].astype(np.float32)
        images.append(dict(name=f"img_{i:05d}.jpg", keypoints=kp4, descriptors=quantize_descriptors(d[perm]),
                           model=1 if camera is None else CAMERA_MODEL_IDS[camera[0]], width=width, height=height,
                           params=(f, f, width / 2.0, height / 2.0) if camera is None else tuple(camera[1])))
    return images
