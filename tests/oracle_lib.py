"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "_build" / "liboracle.so"
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
        lib = C.CDLL(str(LIB))
        lib.oracle_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                     C.c_double, C.c_int, C.c_void_p]
        lib.oracle_match.restype = C.c_int
        lib.oracle_sift_distance_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                    C.c_void_p]
        lib.oracle_sift_distance_matrix.restype = None
        lib.oracle_match_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.oracle_match_pairs.restype = C.c_int
        lib.oracle_match_pairs_vnni.argtypes = lib.oracle_match_pairs.argtypes
        lib.oracle_match_pairs_vnni.restype = C.c_int
        lib.oracle_match_vnni.argtypes = lib.oracle_match.argtypes
        lib.oracle_match_vnni.restype = C.c_int
        lib.oracle_vnni_available.restype = C.c_int
        lib.oracle_match_pairs_kdforest.argtypes = lib.oracle_match_pairs.argtypes
        lib.oracle_match_pairs_kdforest.restype = C.c_int
        lib.oracle_match_kdforest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_double, C.c_double,
                                              C.c_int, C.c_int, C.c_uint64, C.c_void_p]
        lib.oracle_match_kdforest.restype = C.c_int64
        lib.oracle_kdforest_knn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint64,
                                            C.c_void_p, C.c_void_p]
        lib.oracle_kdforest_knn.restype = C.c_int
        lib.oracle_acos_lut.argtypes = [C.c_void_p]
        lib.oracle_acos_lut.restype = None
        _lib = lib
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def match(d1: np.ndarray, d2: np.ndarray, max_ratio=0.8, max_distance=0.7, cross_check=True):
    d1 = np.ascontiguousarray(d1, dtype=np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8).reshape(-1, 128)
    out = np.zeros((max(1, len(d1)), 2), dtype=np.uint32)
    n = load().oracle_match(_p(d1), len(d1), _p(d2), len(d2), max_ratio, max_distance,
                            int(cross_check), _p(out))
    assert n >= 0
    return out[:n].copy()


def distance_matrix(d1, d2):
    d1 = np.ascontiguousarray(d1, dtype=np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8).reshape(-1, 128)
    out = np.zeros((len(d1), len(d2)), dtype=np.int32)
    load().oracle_sift_distance_matrix(_p(d1), len(d1), _p(d2), len(d2), _p(out))
    return out


def tvg_version() -> str:
    """ORACLE_TVG_VERSION of oracle/tvg_oracle.cc: the frozen arithmetic the fixtures and the HIP kernels are held to."""
    lib = load()
    lib.oracle_tvg_version.restype = C.c_char_p
    return lib.oracle_tvg_version().decode()


def match_version() -> str:
    lib = load()
    lib.oracle_match_version.restype = C.c_char_p
    return lib.oracle_match_version().decode()


def vnni_available() -> bool:
    """The AVX-512 VNNI variant of the matcher (oracle/match_vnni.c) can run on this host."""
    return bool(load().oracle_vnni_available())


def match_vnni(d1: np.ndarray, d2: np.ndarray, max_ratio=0.8, max_distance=0.7, cross_check=True):
    d1 = np.ascontiguousarray(d1, dtype=np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8).reshape(-1, 128)
    out = np.zeros((max(1, len(d1)), 2), dtype=np.uint32)
    n = load().oracle_match_vnni(_p(d1), len(d1), _p(d2), len(d2), max_ratio, max_distance, int(cross_check), _p(out))
    assert n >= 0
    return out[:n].copy()


def match_kdforest(d1: np.ndarray, d2: np.ndarray, max_ratio=0.8, max_distance=0.7, cross_check=True, checks=0, seed=0):
    """COLMAP's default CPU matcher (FLANN k-d forest: approximate), restated in oracle/match_kdforest.cc."""
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    out = np.zeros((max(1, len(d1)), 2), dtype=np.uint32)
    n = load().oracle_match_kdforest(_p(d1), len(d1), _p(d2), len(d2), max_ratio, max_distance, int(cross_check),
                                     int(checks), int(seed), _p(out))
    assert n >= 0
    return out[:n].copy()


def kdforest_knn(index_pts: np.ndarray, queries: np.ndarray, checks=0, seed=0):
    """(idx [nq,2] int32, squared L2 [nq,2] float32) from the k-d forest alone."""
    a = np.ascontiguousarray(index_pts, np.uint8)
    q = np.ascontiguousarray(queries, np.uint8)
    idx = np.full((len(q), 2), -1, np.int32)
    dist = np.full((len(q), 2), -1, np.float32)
    assert load().oracle_kdforest_knn(_p(a), len(a), _p(q), len(q), int(checks), int(seed), _p(idx), _p(dist)) == 0
    return idx, dist


def match_pairs(images, slot1, slot2, max_ratio=0.8, max_distance=0.7, cross_check=True, threads=8, variant="literal"):
    """Batched oracle over a list of uint8 [n_i,128] images. Returns (offsets, matches).
    variant "literal": oracle/match_oracle.c (the restatement); "vnni": oracle/match_vnni.c (same results, AVX-512 VNNI);
    "kdforest": oracle/match_kdforest.cc (COLMAP's default approximate CPU matcher, restated)."""
    rows = np.array([len(im) for im in images], dtype=np.uint32)
    row_off = np.zeros(len(images), dtype=np.uint64)
    row_off[1:] = np.cumsum(rows[:-1], dtype=np.uint64)
    arena = (np.concatenate([np.asarray(im, np.uint8).reshape(-1, 128) for im in images], axis=0)
             if rows.sum() else np.zeros((1, 128), np.uint8))
    arena = np.ascontiguousarray(arena)
    s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
    s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
    cap = rows[s1].astype(np.uint64)
    out_off = np.zeros(len(s1) + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(cap)
    counts = np.zeros(len(s1), dtype=np.uint32)
    out = np.zeros((max(1, int(out_off[-1])), 2), dtype=np.uint32)
    fn = {"literal": load().oracle_match_pairs, "vnni": load().oracle_match_pairs_vnni,
          "kdforest": load().oracle_match_pairs_kdforest}[variant]
    rc = fn(_p(arena), _p(row_off), _p(rows), _p(s1), _p(s2), len(s1),
            max_ratio, max_distance, int(cross_check), _p(out_off),
            _p(counts), _p(out), threads)
    assert rc == 0
    offsets = np.zeros(len(s1) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    chunks = [out[int(out_off[p]):int(out_off[p]) + int(counts[p])] for p in range(len(s1))]
    matches = np.concatenate(chunks, axis=0) if chunks and offsets[-1] else np.zeros((0, 2), np.uint32)
    return offsets, matches


def match_guided(d1, kp1, d2, kp2, config, F, H, max_error, max_ratio=0.8, max_distance=0.7, cross_check=True):
    """oracle_match_guided for one pair. kp: [n, >=2] float32. Returns [M,2] uint32, or None when the
    configuration has no guided matching."""
    a = np.ascontiguousarray(d1, dtype=np.uint8).reshape(-1, 128)
    b = np.ascontiguousarray(d2, dtype=np.uint8).reshape(-1, 128)
    k1 = np.ascontiguousarray(np.asarray(kp1, dtype=np.float32)[:, :2]) if len(a) else np.zeros((1, 2), np.float32)
    k2 = np.ascontiguousarray(np.asarray(kp2, dtype=np.float32)[:, :2]) if len(b) else np.zeros((1, 2), np.float32)
    out = np.zeros((max(1, len(a)), 2), dtype=np.uint32)
    lib = load()
    lib.oracle_match_guided.restype = C.c_int
    n = lib.oracle_match_guided(_p(a if len(a) else np.zeros((1, 128), np.uint8)), _p(k1), C.c_int(len(a)),
                                _p(b if len(b) else np.zeros((1, 128), np.uint8)), _p(k2), C.c_int(len(b)),
                                C.c_int(int(config)), _p(_d(F).reshape(9)), _p(_d(H).reshape(9)),
                                C.c_double(max_error), C.c_double(max_ratio), C.c_double(max_distance),
                                C.c_int(int(cross_check)), _p(out))
    if n == -2:
        return None
    assert n >= 0
    return out[:n].copy()


def acos_lut():
    out = np.empty(262145, dtype=np.float32)
    load().oracle_acos_lut(_p(out))
    return out


# ------------------------------------------------------------------------------------------------
# two-view geometry oracle (oracle/tvg_oracle.cc)
# ------------------------------------------------------------------------------------------------
class TvgOptions(C.Structure):
    _fields_ = [("min_num_inliers", C.c_int32), ("detect_watermark", C.c_int32),
                ("multiple_ignore_watermark", C.c_int32), ("force_H_use", C.c_int32),
                ("compute_relative_pose", C.c_int32), ("multiple_models", C.c_int32),
                ("min_E_F_inlier_ratio", C.c_double), ("max_H_inlier_ratio", C.c_double),
                ("watermark_min_inlier_ratio", C.c_double), ("watermark_border_size", C.c_double),
                ("max_error", C.c_double), ("min_inlier_ratio", C.c_double),
                ("confidence", C.c_double), ("dyn_num_trials_multiplier", C.c_double),
                ("min_num_trials", C.c_int64), ("max_num_trials", C.c_int64)]


class OCamera(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("has_prior_focal_length", C.c_int32),
                ("width", C.c_uint64), ("height", C.c_uint64), ("params", C.c_double * 12)]


class TvgResult(C.Structure):
    _fields_ = [("config", C.c_int32), ("num_inliers", C.c_int32), ("E", C.c_double * 9),
                ("F", C.c_double * 9), ("H", C.c_double * 9), ("trials", C.c_int64 * 4),
                ("inl", C.c_int64 * 3), ("pose_ok", C.c_int32), ("num_points3D", C.c_int32),
                ("qvec", C.c_double * 4), ("tvec", C.c_double * 3), ("R", C.c_double * 9),
                ("tri_angle", C.c_double)]


def _pose_fields(r) -> dict:
    return dict(pose_ok=bool(r.pose_ok), num_points3D=int(r.num_points3D), qvec=np.array(r.qvec),
                tvec=np.array(r.tvec), R=np.array(r.R).reshape(3, 3), tri_angle=float(r.tri_angle))


CONFIG_NAMES = ["UNDEFINED", "DEGENERATE", "CALIBRATED", "UNCALIBRATED", "PLANAR", "PANORAMIC",
                "PLANAR_OR_PANORAMIC", "WATERMARK", "MULTIPLE"]


def tvg_default_options(**kw) -> TvgOptions:
    o = TvgOptions()
    load().oracle_tvg_options_default(C.byref(o))
    for k, v in kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


def ransac_options(max_error=4.0, min_inlier_ratio=0.01, confidence=0.9999,
                   dyn_num_trials_multiplier=3.0, min_num_trials=1000, max_num_trials=100000):
    """pycolmap's Python-side RANSACOptions defaults (/root/reference/pycolmap/optim/bindings.h:10-18)."""
    return tvg_default_options(max_error=max_error, min_inlier_ratio=min_inlier_ratio,
                               confidence=confidence, dyn_num_trials_multiplier=dyn_num_trials_multiplier,
                               min_num_trials=min_num_trials, max_num_trials=max_num_trials)


CAMERA_MODELS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "SIMPLE_RADIAL": 2, "RADIAL": 3, "OPENCV": 4, "OPENCV_FISHEYE": 5,
                 "FULL_OPENCV": 6, "FOV": 7, "SIMPLE_RADIAL_FISHEYE": 8, "RADIAL_FISHEYE": 9,
                 "THIN_PRISM_FISHEYE": 10}


def make_camera(model="PINHOLE", width=1600, height=1200, params=(1200.0, 1200.0, 800.0, 600.0),
                prior=False) -> OCamera:
    c = OCamera()
    c.model_id = CAMERA_MODELS[model] if isinstance(model, str) else int(model)
    c.has_prior_focal_length = int(prior)
    c.width, c.height = width, height
    for i, v in enumerate(params):
        c.params[i] = float(v)
    return c


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def cam_from_img(cam: OCamera, points) -> np.ndarray:
    """Camera::CamFromImg of an N x 2 array of image points."""
    xy = _d(points).reshape(-1, 2)
    out = np.empty_like(xy)
    lib = load()
    lib.oracle_cam_from_img.restype = C.c_int
    rc = lib.oracle_cam_from_img(C.byref(cam), _p(xy), C.c_size_t(len(xy)), _p(out))
    assert rc == 0, "oracle: unknown camera model"
    return out


def img_from_cam(cam: OCamera, points) -> np.ndarray:
    """Camera::ImgFromCam of an N x 2 array of normalised image-plane points."""
    lib = load()
    lib.oracle_img_from_cam.restype = C.c_int
    uv = np.ascontiguousarray(points, np.float64).reshape(-1, 2)
    out = np.empty_like(uv)
    assert lib.oracle_img_from_cam(C.byref(cam), _p(uv), C.c_size_t(len(uv)), _p(out)) == 0
    return out


def cam_from_img_threshold(cam: OCamera, threshold: float) -> float:
    lib = load()
    lib.oracle_cam_from_img_threshold.restype = C.c_double
    return float(lib.oracle_cam_from_img_threshold(C.byref(cam), C.c_double(threshold)))


def calibration_matrix(cam: OCamera) -> np.ndarray:
    out = np.empty(9)
    load().oracle_calibration_matrix(C.byref(cam), _p(out))
    return out.reshape(3, 3)


def estimate_two_view_geometry(cam1, pts1, cam2, pts2, matches, opts=None, seed=0):
    lib = load()
    opts = opts or tvg_default_options()
    p1, p2 = _d(pts1).reshape(-1, 2), _d(pts2).reshape(-1, 2)
    m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
    res = TvgResult()
    mask = np.zeros(max(1, len(m)), dtype=np.uint8)
    lib.oracle_estimate_two_view_geometry.restype = C.c_int
    rc = lib.oracle_estimate_two_view_geometry(
        C.byref(cam1), _p(p1), C.c_size_t(len(p1)), C.byref(cam2), _p(p2), C.c_size_t(len(p2)),
        _p(m), C.c_size_t(len(m)), C.byref(opts), C.c_uint32(seed), C.byref(res), _p(mask))
    assert rc == 0, "oracle: unsupported input"
    return dict(config=int(res.config), config_name=CONFIG_NAMES[res.config],
                num_inliers=int(res.num_inliers), E=np.array(res.E).reshape(3, 3),
                F=np.array(res.F).reshape(3, 3), H=np.array(res.H).reshape(3, 3),
                trials=list(res.trials), inl=list(res.inl), inlier_mask=mask[:len(m)].astype(bool),
                inlier_label=mask[:len(m)].copy(),  # 1 + geometry index (multiple_models), else 0/1
                **_pose_fields(res))


def estimate_two_view_geometry_pose(cam1, pts1, cam2, pts2, inlier_matches, config, E=None, H=None):
    """EstimateTwoViewGeometryPose on a given geometry (config, E, H, inlier matches)."""
    lib = load()
    p1, p2 = _d(pts1).reshape(-1, 2), _d(pts2).reshape(-1, 2)
    m = np.ascontiguousarray(inlier_matches, dtype=np.uint32).reshape(-1, 2)
    E = _d(np.zeros((3, 3)) if E is None else E).reshape(9)
    H = _d(np.zeros((3, 3)) if H is None else H).reshape(9)
    res = TvgResult()
    lib.oracle_estimate_two_view_geometry_pose.restype = C.c_int
    rc = lib.oracle_estimate_two_view_geometry_pose(
        C.byref(cam1), _p(p1), C.c_size_t(len(p1)), C.byref(cam2), _p(p2), C.c_size_t(len(p2)),
        _p(m), C.c_size_t(len(m)), C.c_int32(int(config)), _p(E), _p(H), C.byref(res))
    assert rc == 0, "oracle: unsupported input"
    return dict(config=int(res.config), config_name=CONFIG_NAMES[res.config], **_pose_fields(res))


def estimate_two_view_geometry_batch(cams1, pts1, cams2, pts2, matches, opts=None, seed=0, threads=1):
    """Batched oracle_estimate_two_view_geometry on `threads` OpenMP threads (one pair per thread at a
    time).  Lists of per-pair inputs; returns a list of result dicts like estimate_two_view_geometry."""
    lib = load()
    opts = opts or tvg_default_options()
    n = len(pts1)
    p1 = [_d(x).reshape(-1, 2) for x in pts1]
    p2 = [_d(x).reshape(-1, 2) for x in pts2]
    mm = [np.ascontiguousarray(m, dtype=np.uint32).reshape(-1, 2) for m in matches]
    C1 = (OCamera * n)(*cams1)
    C2 = (OCamera * n)(*cams2)
    P1 = (C.c_void_p * n)(*[a.ctypes.data for a in p1])
    P2 = (C.c_void_p * n)(*[a.ctypes.data for a in p2])
    MM = (C.c_void_p * n)(*[a.ctypes.data for a in mm])
    N1 = (C.c_size_t * n)(*[len(a) for a in p1])
    N2 = (C.c_size_t * n)(*[len(a) for a in p2])
    M = (C.c_size_t * n)(*[len(a) for a in mm])
    moff = np.zeros(n + 1, dtype=np.uint64)
    moff[1:] = np.cumsum([max(1, len(a)) for a in mm])
    MO = (C.c_size_t * n)(*[int(x) for x in moff[:-1]])
    res = (TvgResult * n)()
    masks = np.zeros(int(moff[-1]), dtype=np.uint8)
    lib.oracle_estimate_two_view_geometry_batch.restype = C.c_int
    bad = lib.oracle_estimate_two_view_geometry_batch(
        C.c_size_t(n), C1, P1, N1, C2, P2, N2, MM, M, MO, C.byref(opts), C.c_uint32(seed), res, _p(masks),
        C.c_int(threads))
    assert bad == 0, "oracle: unsupported input"
    out = []
    for p in range(n):
        r = res[p]
        out.append(dict(config=int(r.config), config_name=CONFIG_NAMES[r.config], num_inliers=int(r.num_inliers),
                        E=np.array(r.E).reshape(3, 3), F=np.array(r.F).reshape(3, 3), H=np.array(r.H).reshape(3, 3),
                        trials=list(r.trials), inl=list(r.inl),
                        inlier_mask=masks[int(moff[p]):int(moff[p]) + len(mm[p])].astype(bool),
                        inlier_label=masks[int(moff[p]):int(moff[p]) + len(mm[p])].copy(), **_pose_fields(r)))
    return out


def pose_from_homography(H, K1, K2, points1, points2):
    """PoseFromHomographyMatrix: dict(R, t, n, points3D [m,3])."""
    lib = load()
    lib.oracle_pose_from_homography.restype = C.c_int
    lib.oracle_pose_from_homography.argtypes = [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 4
    p1 = np.ascontiguousarray(points1, np.float64).reshape(-1, 2)
    p2 = np.ascontiguousarray(points2, np.float64).reshape(-1, 2)
    mats = [np.ascontiguousarray(a, np.float64).reshape(9) for a in (H, K1, K2)]
    R, t, n, X = np.empty(9), np.empty(3), np.empty(3), np.empty((max(1, len(p1)), 3))
    m = lib.oracle_pose_from_homography(_p(mats[0]), _p(mats[1]), _p(mats[2]), _p(p1), _p(p2), len(p1), _p(R), _p(t), _p(n), _p(X))
    return dict(R=R.reshape(3, 3), t=t, n=n, points3D=X[:m].copy())


def ransac_estimate(kind, p1, p2, opts=None, seed=0):
    """kind: 'F' | 'H' | 'E'. Mirrors pycolmap's *_matrix_estimation (seed 0 per call)."""
    lib = load()
    opts = opts or ransac_options()
    a, b = _d(p1).reshape(-1, 2), _d(p2).reshape(-1, 2)
    model = np.zeros(9)
    ninl, ntr = C.c_int64(), C.c_int64()
    mask = np.zeros(max(1, len(a)), dtype=np.uint8)
    lib.oracle_ransac_estimate.restype = C.c_int
    ok = lib.oracle_ransac_estimate({"F": 0, "H": 1, "E": 2}[kind], _p(a), _p(b), C.c_size_t(len(a)),
                                    C.byref(opts), C.c_uint32(seed), _p(model), C.byref(ninl),
                                    C.byref(ntr), _p(mask))
    return dict(model=model.reshape(3, 3), num_inliers=ninl.value, num_trials=ntr.value,
                inliers=mask[:len(a)].astype(bool), success=bool(ok))


def sampson_error(p1, p2, E):
    a, b, e = _d(p1).reshape(-1, 2), _d(p2).reshape(-1, 2), _d(E).reshape(9)
    out = np.zeros(len(a))
    load().oracle_sampson_error(_p(a), _p(b), C.c_size_t(len(a)), _p(e), _p(out))
    return out


def h_residuals(p1, p2, H):
    a, b, h = _d(p1).reshape(-1, 2), _d(p2).reshape(-1, 2), _d(H).reshape(9)
    out = np.zeros(len(a))
    load().oracle_h_residuals(_p(a), _p(b), C.c_size_t(len(a)), _p(h), _p(out))
    return out


EST_KINDS = {"F7": 0, "F8": 1, "H": 2, "T": 3, "E5": 4}


def estimate_models(kind, p1, p2):
    a, b = _d(p1).reshape(-1, 2), _d(p2).reshape(-1, 2)
    models = np.zeros((10, 9))
    lib = load()
    lib.oracle_estimate_models.restype = C.c_int
    n = lib.oracle_estimate_models(EST_KINDS[kind], _p(a), _p(b), C.c_size_t(len(a)), _p(models))
    return models[:n].reshape(n, 3, 3).copy()


def sample_stream(seed, total, k, trials):
    out = np.zeros((trials, k), dtype=np.uint32)
    load().oracle_sample_stream(C.c_uint32(seed), C.c_uint32(total), C.c_uint32(k), C.c_uint32(trials), _p(out))
    return out


def uniform_draws(seed, lo, hi):
    lo = np.ascontiguousarray(lo, dtype=np.uint32)
    hi = np.ascontiguousarray(hi, dtype=np.uint32)
    out = np.zeros(len(lo), dtype=np.uint32)
    load().oracle_uniform_draws(C.c_uint32(seed), _p(lo), _p(hi), C.c_uint32(len(lo)), _p(out))
    return out


def compute_num_trials(num_inliers, num_samples, confidence, multiplier, kmin):
    lib = load()
    lib.oracle_compute_num_trials.restype = C.c_int64
    return lib.oracle_compute_num_trials(C.c_int64(num_inliers), C.c_int64(num_samples),
                                         C.c_double(confidence), C.c_double(multiplier), C.c_int(kmin))


def real_roots(coeffs_low_to_high):
    c = _d(coeffs_low_to_high)
    out = np.zeros(12)
    lib = load()
    lib.oracle_real_roots.restype = C.c_int
    n = lib.oracle_real_roots(_p(c), C.c_int(len(c) - 1), _p(out))
    return out[:n].copy()


def jacobi_eigen(a):
    a = _d(a).copy()
    n = a.shape[0]
    v = np.zeros((n, n))
    load().oracle_jacobi_eigen(C.c_int(n), _p(a), _p(v))
    return np.diag(a).copy(), v


def det_sum64(x):
    x = _d(x)
    lib = load()
    lib.oracle_det_sum64.restype = C.c_double
    return lib.oracle_det_sum64(_p(x), C.c_size_t(len(x)))
