"""tests/ref2/tvg_ref2.py — a SECOND, deliberately different CPU restatement of COLMAP 3.9.1's
EstimateTwoViewGeometry, written against SURVEY.md Appendix A.3 and not against oracle/tvg_oracle.cc.

TEST INFRASTRUCTURE ONLY (nothing under pycolmap_amd/ imports it).  Its job is to give the verification
oracle an independent pin and to measure what the oracle's documented deviations cost:

  what                         oracle/tvg_oracle.cc (and the HIP kernel)        here
  ---------------------------  -----------------------------------------------  ---------------------------------
  null spaces / least squares  Gauss-Jordan, Jacobi on A^T A            (D1)     numpy.linalg.svd (LAPACK dgesdd)
  rank-2 projection of F       remove smallest right singular direction (D1)     U diag(s0, s1, 0) V^T
  polynomial roots             bracketing + bisection, ascending        (D2)     numpy.roots (companion matrix)
  long sums                    64-way strided + butterfly               (D3)     sequential, as COLMAP's loops
  5-point solver               Nister: det of a 3x3 polynomial matrix            Stewenius: 10x10 action matrix,
                                                                                 eigenvectors (numpy.linalg.eig)
  cubic of the 7-point solver  expanded minors                                   numpy.polymul / polyadd
  PRNG                         std::mt19937 + libstdc++ uniform_int (C++)        numpy MT19937 raw words + a
                                                                                 Python restatement of Lemire's
                                                                                 method as libstdc++ >= 11 does it

The control flow (LORANSAC::Estimate, ComputeNumTrials, model selection, DetectWatermark) is restated
literally from the appendix, once more, in Python.  Inputs and outputs mirror tests/oracle_lib.py.
"""
from __future__ import annotations

import math

import numpy as np

UNDEFINED, DEGENERATE, CALIBRATED, UNCALIBRATED, PLANAR, PANORAMIC, PLANAR_OR_PANORAMIC, WATERMARK, MULTIPLE = range(9)
CONFIG_NAMES = ["UNDEFINED", "DEGENERATE", "CALIBRATED", "UNCALIBRATED", "PLANAR", "PANORAMIC", "PLANAR_OR_PANORAMIC",
                "WATERMARK", "MULTIPLE"]
SIZE_MAX = (1 << 64) - 1


# ------------------------------------------------------------------------------------------------
# PRNG: std::mt19937(seed) and std::uniform_int_distribution<uint32_t> of libstdc++ >= 11
# ------------------------------------------------------------------------------------------------
class Prng:
    def __init__(self, seed: int):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(int(seed))      # init_genrand(seed): what std::mt19937(seed) does
        self._buf = []
        self.raw_draws = 0

    def _raw(self) -> int:
        if not self._buf:
            self._buf = [int(x) for x in self._bg.random_raw(624)][::-1]
        self.raw_draws += 1
        return self._buf.pop()

    def uniform(self, lo: int, hi: int) -> int:
        """uniform_int_distribution<uint32_t>(lo, hi)(gen): the generator's range is 2^32, so for any narrower
        target range libstdc++ takes Lemire's nearly divisionless path (bits/uniform_int_dist.h, _S_nd)."""
        urange = hi - lo
        if urange == 0xFFFFFFFF:
            return self._raw()
        uerange = urange + 1
        product = self._raw() * uerange
        low = product & 0xFFFFFFFF
        if low < uerange:
            threshold = ((1 << 32) - uerange) % uerange
            while low < threshold:
                product = self._raw() * uerange
                low = product & 0xFFFFFFFF
        return (product >> 32) + lo


class RandomSampler:
    """colmap/optim/random_sampler.cc: the index permutation persists across Sample() calls."""

    def __init__(self, num_samples: int):
        self.k = num_samples
        self.idx = []

    def initialize(self, total: int):
        self.idx = list(range(total))

    def sample(self, prng: Prng):
        last = len(self.idx) - 1
        for i in range(self.k):                       # Shuffle(k, &sample_idxs_)
            j = prng.uniform(i, last)
            self.idx[i], self.idx[j] = self.idx[j], self.idx[i]
        return self.idx[:self.k]


# ------------------------------------------------------------------------------------------------
# RANSAC bookkeeping
# ------------------------------------------------------------------------------------------------
def compute_num_trials(num_inliers, num_samples, confidence, multiplier, min_num_samples):
    inlier_ratio = num_inliers / float(num_samples)
    nom = 1 - confidence
    if nom <= 0:
        return SIZE_MAX
    denom = 1 - math.pow(inlier_ratio, min_num_samples)
    if denom <= 0:
        return 1
    if denom == 1.0:
        return SIZE_MAX
    return int(math.ceil(math.log(nom) / math.log(denom) * multiplier))


def seq_sum(x) -> float:
    """sum in index order, one addition at a time (numpy's add.reduce is pairwise: not this)."""
    x = np.asarray(x, dtype=np.float64)
    return float(np.cumsum(x)[-1]) if x.size else 0.0


def better(a, b):
    return a[0] > b[0] or (a[0] == b[0] and a[1] < b[1])


def support_of(res, max_residual):
    inl = res <= max_residual
    return int(inl.sum()), seq_sum(res[inl])


# ------------------------------------------------------------------------------------------------
# residuals (vectorised, operation order of SURVEY.md A.3)
# ------------------------------------------------------------------------------------------------
def sampson(E, X, Y):
    x1_0, x1_1, x2_0, x2_1 = X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]
    Ex1_0 = E[0, 0] * x1_0 + E[0, 1] * x1_1 + E[0, 2]
    Ex1_1 = E[1, 0] * x1_0 + E[1, 1] * x1_1 + E[1, 2]
    Ex1_2 = E[2, 0] * x1_0 + E[2, 1] * x1_1 + E[2, 2]
    Etx2_0 = E[0, 0] * x2_0 + E[1, 0] * x2_1 + E[2, 0]
    Etx2_1 = E[0, 1] * x2_0 + E[1, 1] * x2_1 + E[2, 1]
    x2tEx1 = x2_0 * Ex1_0 + x2_1 * Ex1_1 + Ex1_2
    with np.errstate(all="ignore"):
        return x2tEx1 * x2tEx1 / (Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1 + Etx2_0 * Etx2_0 + Etx2_1 * Etx2_1)


def h_residuals(H, X, Y):
    s_0, s_1, d_0, d_1 = X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]
    pd_0 = H[0, 0] * s_0 + H[0, 1] * s_1 + H[0, 2]
    pd_1 = H[1, 0] * s_0 + H[1, 1] * s_1 + H[1, 2]
    pd_2 = H[2, 0] * s_0 + H[2, 1] * s_1 + H[2, 2]
    with np.errstate(all="ignore"):
        inv_pd_2 = 1.0 / pd_2
        dd_0 = d_0 - pd_0 * inv_pd_2
        dd_1 = d_1 - pd_1 * inv_pd_2
        return dd_0 * dd_0 + dd_1 * dd_1


def t_residuals(t, X, Y):
    d0 = Y[:, 0] - X[:, 0] - t[0]
    d1 = Y[:, 1] - X[:, 1] - t[1]
    return d0 * d0 + d1 * d1


# ------------------------------------------------------------------------------------------------
# estimators
# ------------------------------------------------------------------------------------------------
def center_and_normalize(P):
    n = len(P)
    cx, cy = seq_sum(P[:, 0]) / n, seq_sum(P[:, 1]) / n
    dx, dy = P[:, 0] - cx, P[:, 1] - cy
    rms = math.sqrt(seq_sum(dx * dx + dy * dy) / n)
    with np.errstate(all="ignore"):
        nf = np.float64(math.sqrt(2.0)) / np.float64(rms)
    T = np.array([[nf, 0, -nf * cx], [0, nf, -nf * cy], [0, 0, 1.0]])
    np0 = T[0, 0] * P[:, 0] + T[0, 1] * P[:, 1] + T[0, 2]
    np1 = T[1, 0] * P[:, 0] + T[1, 1] * P[:, 1] + T[1, 2]
    np2 = T[2, 0] * P[:, 0] + T[2, 1] * P[:, 1] + T[2, 2]
    inv = 1.0 / np2
    return np.stack([np0 * inv, np1 * inv], axis=1), T


def _finite(*arrays):
    return all(np.all(np.isfinite(a)) for a in arrays)


def epipolar_rows(X, Y):
    x0, y0, x1, y1 = X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]
    return np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones(len(X))], axis=1)


def estimate_f7(X, Y):
    A = epipolar_rows(X, Y)
    if not _finite(A):
        return []
    Vt = np.linalg.svd(A, full_matrices=True)[2]
    f2 = Vt[8]
    f1 = Vt[7] - f2
    # det(lambda f1 + f2) as a cubic in lambda, by polynomial arithmetic on the entries (highest power first)
    e = [np.array([f1[k], f2[k]]) for k in range(9)]

    def minor(a, b, c, d):
        return np.polysub(np.polymul(e[a], e[b]), np.polymul(e[c], e[d]))

    det = np.polyadd(np.polysub(np.polymul(e[0], minor(4, 8, 5, 7)), np.polymul(e[1], minor(3, 8, 5, 6))),
                     np.polymul(e[2], minor(3, 7, 4, 6)))
    det = np.atleast_1d(det)
    if not _finite(det) or np.all(det == 0):
        return []
    roots = np.roots(det)                                    # eigenvalues of the companion matrix
    models = []
    for r in roots:
        if abs(r.imag) > 1e-10:
            continue
        F = (r.real * f1 + f2).reshape(3, 3)
        if abs(F[2, 2]) < 1e-10:
            continue
        models.append(F / F[2, 2])
    return models


def estimate_f8(X, Y):
    n1, T1 = center_and_normalize(X)
    n2, T2 = center_and_normalize(Y)
    A = np.empty((len(X), 9))
    A[:, 0:2] = n1 * n2[:, 0:1]
    A[:, 2] = n2[:, 0]
    A[:, 3:5] = n1 * n2[:, 1:2]
    A[:, 5] = n2[:, 1]
    A[:, 6:8] = n1
    A[:, 8] = 1.0
    if not _finite(A):
        return []
    Vt = np.linalg.svd(A, full_matrices=True)[2]
    Fh = Vt[8].reshape(3, 3)
    U, S, Wt = np.linalg.svd(Fh)
    S[2] = 0.0
    F = U @ np.diag(S) @ Wt
    return [T2.T @ F @ T1]


def estimate_h(X, Y):
    N = len(X)
    n1, T1 = center_and_normalize(X)
    n2, T2 = center_and_normalize(Y)
    A = np.zeros((2 * N, 9))
    s0, s1, d0, d1 = n1[:, 0], n1[:, 1], n2[:, 0], n2[:, 1]
    A[:N, 0], A[:N, 1], A[:N, 2] = -s0, -s1, -1
    A[:N, 6], A[:N, 7], A[:N, 8] = s0 * d0, s1 * d0, d0
    A[N:, 3], A[N:, 4], A[N:, 5] = -s0, -s1, -1
    A[N:, 6], A[N:, 7], A[N:, 8] = s0 * d1, s1 * d1, d1
    if not _finite(A, T1, T2):
        return [np.full((3, 3), np.nan)]
    Vt = np.linalg.svd(A, full_matrices=True)[2]
    Hh = Vt[8].reshape(3, 3)
    try:
        return [np.linalg.inv(T2) @ Hh @ T1]
    except np.linalg.LinAlgError:
        return [np.full((3, 3), np.nan)]


def estimate_t(X, Y):
    n = len(X)
    sx, sy = seq_sum(X[:, 0]) / n, seq_sum(X[:, 1]) / n
    dx, dy = seq_sum(Y[:, 0]) / n, seq_sum(Y[:, 1]) / n
    return [np.array([dx - sx, dy - sy])]


# ---- 5-point: Stewenius / Engels / Nister "Recent developments on direct relative orientation" (the action-
#      matrix formulation), NOT the determinant formulation of the oracle ------------------------------------
def _pmul(a, b):
    """product of two polynomials in (x, y, z) stored as degree tensors c[i, j, k] (full convolution)"""
    out = np.zeros(tuple(sa + sb - 1 for sa, sb in zip(a.shape, b.shape)))
    for i, j, k in zip(*np.nonzero(a)):
        out[i:i + b.shape[0], j:j + b.shape[1], k:k + b.shape[2]] += a[i, j, k] * b
    return out


def _padd(*ps):
    shape = tuple(max(p.shape[d] for p in ps) for d in range(3))
    out = np.zeros(shape)
    for p in ps:
        out[:p.shape[0], :p.shape[1], :p.shape[2]] += p
    return out


_CUBIC = [(3, 0, 0), (2, 1, 0), (1, 2, 0), (0, 3, 0), (2, 0, 1), (1, 1, 1), (0, 2, 1), (1, 0, 2), (0, 1, 2), (0, 0, 3)]
_BASIS = [(2, 0, 0), (1, 1, 0), (0, 2, 0), (1, 0, 1), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def estimate_e5(X, Y):
    A = epipolar_rows(X, Y)
    if not _finite(A):
        return []
    Vt = np.linalg.svd(A, full_matrices=True)[2]
    basis = Vt[5:9]                                           # svd.matrixV().block<9, 4>(0, 5): x, y, z, 1
    ent = []
    for k in range(9):                                        # entry k of E as a linear polynomial tensor
        p = np.zeros((2, 2, 2))
        p[1, 0, 0], p[0, 1, 0], p[0, 0, 1], p[0, 0, 0] = basis[0, k], basis[1, k], basis[2, k], basis[3, k]
        ent.append(p)
    E = [[ent[3 * i + j] for j in range(3)] for i in range(3)]
    det = _padd(_pmul(E[0][0], _padd(_pmul(E[1][1], E[2][2]), -_pmul(E[1][2], E[2][1]))),
                -_pmul(E[0][1], _padd(_pmul(E[1][0], E[2][2]), -_pmul(E[1][2], E[2][0]))),
                _pmul(E[0][2], _padd(_pmul(E[1][0], E[2][1]), -_pmul(E[1][1], E[2][0]))))
    EEt = [[_padd(*[_pmul(E[i][k], E[j][k]) for k in range(3)]) for j in range(3)] for i in range(3)]
    tr = _padd(EEt[0][0], EEt[1][1], EEt[2][2])
    eqs = [det]
    for i in range(3):
        for j in range(3):
            eqs.append(_padd(2.0 * _padd(*[_pmul(EEt[i][k], E[k][j]) for k in range(3)]), -_pmul(tr, E[i][j])))
    Mc = np.zeros((10, 20))
    for r, p in enumerate(eqs):
        q = np.zeros((4, 4, 4))
        q[:p.shape[0], :p.shape[1], :p.shape[2]] = p
        for c, (i, j, k) in enumerate(_CUBIC + _BASIS):
            Mc[r, c] = q[i, j, k]
    if not _finite(Mc):
        return []
    try:
        R = np.linalg.solve(Mc[:, :10], Mc[:, 10:])           # cubic monomial c = - sum_b R[c, b] basis[b]
    except np.linalg.LinAlgError:
        return []
    # action matrix of multiplication by x on the quotient-ring basis _BASIS
    Ax = np.zeros((10, 10))
    for b, (i, j, k) in enumerate(_BASIS):
        m = (i + 1, j, k)
        if m in _BASIS:
            Ax[b, _BASIS.index(m)] = 1.0
        else:
            Ax[b, :] = -R[_CUBIC.index(m), :]
    if not _finite(Ax):
        return []
    w, V = np.linalg.eig(Ax)
    models = []
    for q in range(10):
        if abs(w[q].imag) > 1e-10:
            continue
        v = V[:, q].real
        if abs(v[9]) < 1e-300:
            continue
        x, y, z = v[6] / v[9], v[7] / v[9], v[8] / v[9]
        if not x * x + y * y + 1.0 < 1e20:
            continue
        Ev = x * basis[0] + y * basis[1] + z * basis[2] + basis[3]
        Ev = Ev / np.linalg.norm(Ev)
        models.append(Ev.reshape(3, 3))
    return models


EST = {
    "F7": (7, estimate_f7, sampson), "F8": (8, estimate_f8, sampson), "H": (4, estimate_h, h_residuals),
    "T": (1, estimate_t, t_residuals), "E5": (5, estimate_e5, sampson),
}


# ------------------------------------------------------------------------------------------------
# LORANSAC<Estimator, LocalEstimator>::Estimate
# ------------------------------------------------------------------------------------------------
class Report:
    def __init__(self):
        self.success = False
        self.num_trials = 0
        self.support = (0, np.finfo(np.float64).max)
        self.inlier_mask = np.zeros(0, dtype=bool)
        self.model = np.zeros((3, 3))


def lo_ransac(est, local_est, opts, prng, X, Y):
    kmin, estimate, residuals = EST[est]
    lkmin, lestimate, lresiduals = EST[local_est]
    max_trials = min(int(opts["max_num_trials"]),
                     compute_num_trials(int(opts["min_inlier_ratio"] * 100000), 100000, opts["confidence"],
                                        opts["dyn_num_trials_multiplier"], kmin))       # RANSAC constructor
    rep = Report()
    n = len(X)
    if n < kmin:
        return rep
    best, best_model, best_is_local = (0, np.finfo(np.float64).max), np.zeros((3, 3)), False
    abort = False
    max_res = opts["max_error"] * opts["max_error"]
    sampler = RandomSampler(kmin)
    sampler.initialize(n)
    dyn_max = max_trials
    t = 0
    while t < max_trials:
        if abort:
            t += 1
            break
        idx = sampler.sample(prng)
        for model in estimate(X[idx], Y[idx]):
            res = residuals(model, X, Y)
            sup = support_of(res, max_res)
            if better(sup, best):
                best, best_model, best_is_local = sup, model, False
                if sup[0] > kmin and sup[0] >= lkmin:
                    best_local_res = None
                    for _ in range(10):
                        inl = res <= max_res
                        prev = best[0]
                        for lm in lestimate(X[inl], Y[inl]):
                            lres = lresiduals(lm, X, Y)
                            lsup = support_of(lres, max_res)
                            if better(lsup, best):
                                best, best_model, best_is_local = lsup, lm, True
                                best_local_res = lres
                        if best[0] <= prev:
                            break
                        res = best_local_res
                dyn_max = compute_num_trials(best[0], n, opts["confidence"], opts["dyn_num_trials_multiplier"], kmin)
            if t >= dyn_max and t >= opts["min_num_trials"]:
                abort = True
                break
        t += 1
    rep.num_trials = t
    rep.support, rep.model = best, best_model
    if best[0] < kmin:
        return rep
    rep.success = True
    fres = (lresiduals if best_is_local else residuals)(best_model, X, Y)
    rep.inlier_mask = fres <= max_res
    return rep


# ------------------------------------------------------------------------------------------------
# EstimateTwoViewGeometry (uncalibrated / calibrated / force_H_use), DetectWatermark
# ------------------------------------------------------------------------------------------------
DEFAULTS = dict(min_num_inliers=15, min_E_F_inlier_ratio=0.95, max_H_inlier_ratio=0.8, watermark_min_inlier_ratio=0.7,
                watermark_border_size=0.1, detect_watermark=1, force_H_use=0, max_error=4.0, min_inlier_ratio=0.25,
                confidence=0.999, dyn_num_trials_multiplier=3.0, min_num_trials=100, max_num_trials=10000)


def _cam_from_img(cam, P):
    model, prm = cam["model"], cam["params"]
    if model in ("SIMPLE_PINHOLE", 0):
        return np.stack([(P[:, 0] - prm[1]) / prm[0], (P[:, 1] - prm[2]) / prm[0]], axis=1), prm[0]
    if model in ("PINHOLE", 1):
        return np.stack([(P[:, 0] - prm[2]) / prm[0], (P[:, 1] - prm[3]) / prm[1]], axis=1), (prm[0] + prm[1]) / 2.0
    raise ValueError("ref2 lifts pinhole cameras only (the distortion models are pinned in test_camera_models_cpu.py)")


def _in_bbox(P, lo_x, hi_x, lo_y, hi_y):
    return (P[:, 0] >= lo_x) & (P[:, 0] <= hi_x) & (P[:, 1] >= lo_y) & (P[:, 1] <= hi_y)


def detect_watermark(cam1, P1, cam2, P2, num_inliers, mask, o, prng):
    d1 = math.sqrt(float(cam1["width"] * cam1["width"] + cam1["height"] * cam1["height"]))
    d2 = math.sqrt(float(cam2["width"] * cam2["width"] + cam2["height"] * cam2["height"]))
    b1, b2 = o["watermark_border_size"] * d1, o["watermark_border_size"] * d2
    I1, I2 = P1[mask], P2[mask]
    border = (~_in_bbox(I1, b1, cam1["width"] - b1, b1, cam1["height"] - b1) &
              ~_in_bbox(I2, b2, cam2["width"] - b2, b2, cam2["height"] - b2))
    if int(border.sum()) / num_inliers < o["watermark_min_inlier_ratio"]:
        return False, 0
    ro = dict(o)
    ro["min_inlier_ratio"] = o["watermark_min_inlier_ratio"]
    rep = lo_ransac("T", "T", ro, prng, I1, I2)
    return rep.support[0] / num_inliers >= o["watermark_min_inlier_ratio"], rep.num_trials


def estimate_two_view_geometry(cam1, pts1, cam2, pts2, matches, opts=None, seed=0):
    """cam: dict(model, width, height, params, prior).  Returns dict(config, config_name, num_inliers, inlier_mask,
    trials [E, F, H, watermark], inl [E, F, H], E, F, H)."""
    o = dict(DEFAULTS)
    o.update(opts or {})
    prng = Prng(seed)
    M = len(matches)
    out = dict(config=UNDEFINED, num_inliers=0, inlier_mask=np.zeros(M, dtype=bool), trials=[0, 0, 0, 0], inl=[0, 0, 0],
               E=np.zeros((3, 3)), F=np.zeros((3, 3)), H=np.zeros((3, 3)))

    def done(cfg):
        out["config"] = cfg
        out["config_name"] = CONFIG_NAMES[cfg]
        return out

    min_inl = o["min_num_inliers"]
    if M < min_inl:
        return done(DEGENERATE)
    m = np.asarray(matches, dtype=np.int64)
    P1, P2 = np.asarray(pts1, dtype=np.float64)[m[:, 0]], np.asarray(pts2, dtype=np.float64)[m[:, 1]]
    calibrated = (not o["force_H_use"]) and cam1["prior"] and cam2["prior"]
    E = F = None
    if calibrated:
        N1, f1 = _cam_from_img(cam1, P1)
        N2, f2 = _cam_from_img(cam2, P2)
        eo = dict(o)
        eo["max_error"] = (o["max_error"] / f1 + o["max_error"] / f2) / 2
        E = lo_ransac("E5", "E5", eo, prng, N1, N2)
        out["E"], out["trials"][0], out["inl"][0] = E.model, E.num_trials, E.support[0]
    if not o["force_H_use"]:
        F = lo_ransac("F7", "F8", o, prng, P1, P2)
        out["F"], out["trials"][1], out["inl"][1] = F.model, F.num_trials, F.support[0]
    H = lo_ransac("H", "H", o, prng, P1, P2)
    out["H"], out["trials"][2], out["inl"][2] = H.model, H.num_trials, H.support[0]
    Hi = H.support[0]

    def ratio(a, b):
        with np.errstate(all="ignore"):
            return float(np.float64(a) / np.float64(b))

    best = None
    num_inliers = 0
    if o["force_H_use"]:
        if not H.success or Hi < min_inl:
            return done(DEGENERATE)
        cfg, best, num_inliers = PLANAR_OR_PANORAMIC, H, Hi
    elif calibrated:
        Ei, Fi = E.support[0], F.support[0]
        if (not E.success and not F.success and not H.success) or (Ei < min_inl and Fi < min_inl and Hi < min_inl):
            return done(DEGENERATE)
        if E.success and ratio(Ei, Fi) > o["min_E_F_inlier_ratio"] and Ei >= min_inl:
            best, num_inliers = (E, Ei) if Ei >= Fi else (F, Fi)
            if ratio(Hi, Ei) > o["max_H_inlier_ratio"]:
                cfg = PLANAR_OR_PANORAMIC
                if Hi > num_inliers:
                    best, num_inliers = H, Hi
            else:
                cfg = CALIBRATED
        elif F.success and Fi >= min_inl:
            best, num_inliers = F, Fi
            if ratio(Hi, Fi) > o["max_H_inlier_ratio"]:
                cfg = PLANAR_OR_PANORAMIC
                if Hi > num_inliers:
                    best, num_inliers = H, Hi
            else:
                cfg = UNCALIBRATED
        elif H.success and Hi >= min_inl:
            cfg, best, num_inliers = PLANAR_OR_PANORAMIC, H, Hi
        else:
            return done(DEGENERATE)
    else:
        Fi = F.support[0]
        if (not F.success and not H.success) or (Fi < min_inl and Hi < min_inl):
            return done(DEGENERATE)
        best, num_inliers = F, Fi
        if ratio(Hi, Fi) > o["max_H_inlier_ratio"]:
            cfg = PLANAR_OR_PANORAMIC
            if Hi >= Fi:
                best, num_inliers = H, Hi
        else:
            cfg = UNCALIBRATED
    if best.success and len(best.inlier_mask):
        out["inlier_mask"] = best.inlier_mask.copy()
        out["num_inliers"] = num_inliers
        if o["detect_watermark"]:
            is_wm, tr = detect_watermark(cam1, P1, cam2, P2, num_inliers, best.inlier_mask, o, prng)
            out["trials"][3] = tr
            if is_wm:
                cfg = WATERMARK
    return done(cfg)
