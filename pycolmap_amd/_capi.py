"""ctypes view of libamc.so's C ABI (include/amc.h).

Thin by design: every function here maps 1:1 to an `amc_*` entry point; no computation happens
in Python and there is no fallback — if libamc.so is missing or no gfx950 device is visible the
calls raise.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libamc.so"

AMC_OK = 0
AMC_E_INVALID, AMC_E_HIP, AMC_E_NOMEM, AMC_E_STATE = -1, -2, -3, -4
KERNEL_AUTO, KERNEL_MFMA, KERNEL_DOT4 = 0, 1, 2
KERNELS = {"auto": KERNEL_AUTO, "mfma": KERNEL_MFMA, "dot4": KERNEL_DOT4}

# every symbol include/amc.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "amc_last_error", "amc_abi_version", "amc_device_count", "amc_ctx_create", "amc_ctx_destroy",
    "amc_ctx_set_stream", "amc_ctx_reserve_slots", "amc_upload_descriptors",
    "amc_upload_descriptors_device", "amc_match_pairs", "amc_match_result_free",
    "amc_match_opts_default", "amc_get_acos_lut",
    "amc_tvg_opts_default", "amc_upload_keypoints", "amc_upload_camera", "amc_verify_pairs",
    "amc_verify_result_free", "amc_upload_points_f64", "amc_ransac_pairs", "amc_ransac_result_free",
    "amc_squared_sampson_error", "amc_match_guided_pairs", "amc_ctx_grow_slots", "amc_pose_pairs",
    "amc_cam_from_img", "amc_match_verify_pairs", "amc_ctx_trim", "amc_ctx_resident_matches",
    "amc_homography_decomposition", "amc_img_from_cam",
    "amc_comm_unique_id", "amc_comm_create", "amc_comm_destroy", "amc_allgather_match_tables", "amc_gathered_tables_free",
    "amc_allgather_pair_records", "amc_gathered_records_free", "amc_allgather_inlier_tables", "amc_ctx_last_timeline",
    "amc_upload_matches",
]
COMM_ID_BYTES = 128
RANSAC_F, RANSAC_H, RANSAC_E = 0, 1, 2
RANSAC_KINDS = {"F": RANSAC_F, "H": RANSAC_H, "E": RANSAC_E}


class _MatchLease:
    """Owns one amc_match_result; frees it when the arrays viewing it are gone."""

    def __init__(self, lib, res):
        self._lib, self._res = lib, res

    def __del__(self):
        try:
            self._lib.amc_match_result_free(C.byref(self._res))
        except Exception:  # interpreter shutdown
            pass


class _VerifyLease:
    """Owns one amc_verify_result; frees it when the arrays viewing it are gone."""

    def __init__(self, lib, res):
        self._lib, self._res = lib, res

    def __del__(self):
        try:
            self._lib.amc_verify_result_free(C.byref(self._res))
        except Exception:  # interpreter shutdown
            pass


class _LeasedArray(np.ndarray):
    """ndarray view that keeps its lease alive (numpy views of it inherit the reference through .base)."""

    @staticmethod
    def wrap(arr, lease):
        out = arr.view(_LeasedArray)
        out._lease = lease
        return out

    def __array_finalize__(self, obj):
        self._lease = getattr(obj, "_lease", None)


class AmcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"amc error {code}: {msg}")
        self.code = code


class GatheredTables(C.Structure):
    """amc_gathered_tables (include/amc.h)."""
    _fields_ = [("npairs", C.c_size_t), ("offsets", C.POINTER(C.c_uint64)), ("matches", C.POINTER(C.c_uint32)),
                ("matches_device", C.c_void_p), ("num_matches", C.c_uint64), ("rows_sent", C.c_uint64),
                ("rows_received", C.c_uint64), ("world_size", C.c_int32), ("rank", C.c_int32),
                ("sizes_ms", C.c_double), ("meta_ms", C.c_double), ("rows_ms", C.c_double), ("reorder_ms", C.c_double),
                ("download_ms", C.c_double), ("total_ms", C.c_double), ("_priv", C.c_void_p)]


class GatheredRecords(C.Structure):
    """amc_gathered_records (include/amc.h)."""
    _fields_ = [("npairs", C.c_size_t), ("record_bytes", C.c_size_t), ("records", C.c_void_p), ("records_device", C.c_void_p),
                ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("world_size", C.c_int32), ("rank", C.c_int32),
                ("total_ms", C.c_double), ("_priv", C.c_void_p)]


class MatchOpts(C.Structure):
    _fields_ = [("max_ratio", C.c_double), ("max_distance", C.c_double),
                ("cross_check", C.c_int32), ("kernel", C.c_int32)]


class MatchResult(C.Structure):
    _fields_ = [("npairs", C.c_size_t), ("offsets", C.POINTER(C.c_uint64)),
                ("matches", C.POINTER(C.c_uint32)), ("num_distances", C.c_uint64),
                ("pairs_mfma", C.c_uint64), ("pairs_dot4", C.c_uint64), ("pairs_guided_grid", C.c_uint64),
                ("device_ms", C.c_double), ("match_kernel_ms", C.c_double),
                ("cross_kernel_ms", C.c_double),
                ("match_kernel_launches", C.c_uint32), ("_priv", C.c_void_p)]


class RansacOpts(C.Structure):
    _fields_ = [("max_error", C.c_double), ("min_inlier_ratio", C.c_double),
                ("confidence", C.c_double), ("dyn_num_trials_multiplier", C.c_double),
                ("min_num_trials", C.c_int64), ("max_num_trials", C.c_int64)]


class TvgOpts(C.Structure):
    _fields_ = [("min_num_inliers", C.c_int32), ("detect_watermark", C.c_int32),
                ("multiple_ignore_watermark", C.c_int32), ("force_H_use", C.c_int32),
                ("compute_relative_pose", C.c_int32), ("multiple_models", C.c_int32),
                ("min_E_F_inlier_ratio", C.c_double), ("max_H_inlier_ratio", C.c_double),
                ("watermark_min_inlier_ratio", C.c_double), ("watermark_border_size", C.c_double),
                ("ransac", RansacOpts)]


class Tvg(C.Structure):
    _fields_ = [("config", C.c_int32), ("num_inliers", C.c_int32), ("E", C.c_double * 9),
                ("F", C.c_double * 9), ("H", C.c_double * 9), ("num_trials", C.c_int64 * 4),
                ("model_inliers", C.c_int64 * 3)]


class Pose(C.Structure):
    _fields_ = [("ok", C.c_int32), ("config", C.c_int32), ("qvec", C.c_double * 4), ("tvec", C.c_double * 3),
                ("R", C.c_double * 9), ("tri_angle", C.c_double), ("num_points3D", C.c_uint32),
                ("pad_", C.c_uint32)]


class VerifyResult(C.Structure):
    _fields_ = [("npairs", C.c_size_t), ("tvg", C.POINTER(Tvg)), ("inlier_mask", C.POINTER(C.c_uint8)),
                ("device_ms", C.c_double), ("kernel_ms", C.c_double), ("kernel_launches", C.c_uint32),
                ("pose", C.POINTER(Pose)), ("pose_kernel_ms", C.c_double), ("work", C.c_uint64 * 12),
                ("_priv", C.c_void_p)]


class RansacReport(C.Structure):
    _fields_ = [("success", C.c_int32), ("num_inliers", C.c_int32), ("num_trials", C.c_int64),
                ("model", C.c_double * 9)]


class RansacResult(C.Structure):
    _fields_ = [("npairs", C.c_size_t), ("reports", C.POINTER(RansacReport)),
                ("inlier_mask", C.POINTER(C.c_uint8)), ("device_ms", C.c_double), ("_priv", C.c_void_p)]


RANSAC_DTYPE = np.dtype([("success", np.int32), ("num_inliers", np.int32), ("num_trials", np.int64),
                         ("model", np.float64, (3, 3))])
POSE_DTYPE = np.dtype([("ok", np.int32), ("config", np.int32), ("qvec", np.float64, (4,)),
                       ("tvec", np.float64, (3,)), ("R", np.float64, (3, 3)), ("tri_angle", np.float64),
                       ("num_points3D", np.uint32), ("pad_", np.uint32)])
TVG_DTYPE = np.dtype([("config", np.int32), ("num_inliers", np.int32), ("E", np.float64, (3, 3)),
                      ("F", np.float64, (3, 3)), ("H", np.float64, (3, 3)),
                      ("num_trials", np.int64, (4,)), ("model_inliers", np.int64, (3,))])
CAMERA_MODELS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "SIMPLE_RADIAL": 2, "RADIAL": 3, "OPENCV": 4, "OPENCV_FISHEYE": 5,
                 "FULL_OPENCV": 6, "FOV": 7, "SIMPLE_RADIAL_FISHEYE": 8, "RADIAL_FISHEYE": 9,
                 "THIN_PRISM_FISHEYE": 10}
CONFIG_NAMES = ["UNDEFINED", "DEGENERATE", "CALIBRATED", "UNCALIBRATED", "PLANAR", "PANORAMIC",
                "PLANAR_OR_PANORAMIC", "WATERMARK", "MULTIPLE"]

_lib = None


def load() -> C.CDLL:
    """Load libamc.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    path = Path(os.environ.get("AMC_LIB_PATH", LIB_PATH))  # override: A/B benchmarking of kernel builds
    if not path.exists():
        raise ImportError(
            f"{path} not found — run `python -m pycolmap_amd.build` (hipcc, gfx950). "
            "pycolmap_amd has no CPU fallback.")
    lib = C.CDLL(str(path))
    lib.amc_last_error.restype = C.c_char_p
    lib.amc_abi_version.restype = C.c_int
    lib.amc_device_count.restype = C.c_int
    lib.amc_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.amc_ctx_destroy.argtypes = [C.c_void_p]
    lib.amc_ctx_destroy.restype = None
    lib.amc_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.amc_ctx_trim.argtypes = [C.c_void_p]
    lib.amc_ctx_trim.restype = C.c_int
    lib.amc_ctx_resident_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.amc_ctx_resident_matches.restype = C.c_int
    lib.amc_upload_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.amc_upload_matches.restype = C.c_int
    if hasattr(lib, "amc_ctx_last_timeline"):
        lib.amc_ctx_last_timeline.argtypes = [C.c_void_p, C.c_void_p]
    lib.amc_ctx_reserve_slots.argtypes = [C.c_void_p, C.c_uint32]
    lib.amc_ctx_grow_slots.argtypes = [C.c_void_p, C.c_uint32]
    lib.amc_upload_descriptors.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.amc_upload_descriptors_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.amc_match_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(MatchOpts), C.POINTER(MatchResult)]
    lib.amc_match_result_free.argtypes = [C.POINTER(MatchResult)]
    lib.amc_match_result_free.restype = None
    lib.amc_match_opts_default.argtypes = [C.POINTER(MatchOpts)]
    lib.amc_match_opts_default.restype = None
    lib.amc_get_acos_lut.argtypes = [C.c_void_p, C.c_void_p]
    lib.amc_tvg_opts_default.argtypes = [C.POINTER(TvgOpts)]
    lib.amc_tvg_opts_default.restype = None
    lib.amc_upload_keypoints.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.amc_upload_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_int32, C.c_int32]
    lib.amc_verify_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.POINTER(TvgOpts), C.c_uint32, C.POINTER(VerifyResult)]
    lib.amc_verify_result_free.argtypes = [C.POINTER(VerifyResult)]
    lib.amc_verify_result_free.restype = None
    lib.amc_match_guided_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_double,
                                           C.POINTER(MatchOpts), C.POINTER(MatchResult)]
    lib.amc_upload_points_f64.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.amc_ransac_pairs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.POINTER(RansacOpts), C.c_uint32, C.POINTER(RansacResult)]
    lib.amc_ransac_result_free.argtypes = [C.POINTER(RansacResult)]
    lib.amc_ransac_result_free.restype = None
    lib.amc_cam_from_img.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(lib, "amc_img_from_cam"):   # (absent from a library built from an older revision: tools/ab_prev_lib.sh)
        lib.amc_img_from_cam.argtypes = lib.amc_cam_from_img.argtypes
    lib.amc_squared_sampson_error.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                              C.c_void_p]
    lib.amc_pose_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    if hasattr(lib, "amc_homography_decomposition"):
        lib.amc_homography_decomposition.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 5
    if hasattr(lib, "amc_comm_create"):
        lib.amc_comm_unique_id.argtypes = [C.c_void_p]
        lib.amc_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.amc_comm_destroy.argtypes = [C.c_void_p]
        lib.amc_comm_destroy.restype = None
        lib.amc_allgather_match_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                   C.c_int, C.POINTER(GatheredTables)]
        lib.amc_gathered_tables_free.argtypes = [C.POINTER(GatheredTables)]
        lib.amc_gathered_tables_free.restype = None
    if hasattr(lib, "amc_allgather_pair_records"):
        lib.amc_allgather_pair_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                   C.c_int, C.POINTER(GatheredRecords)]
        lib.amc_gathered_records_free.argtypes = [C.POINTER(GatheredRecords)]
        lib.amc_gathered_records_free.restype = None
        lib.amc_allgather_inlier_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                                    C.POINTER(GatheredTables)]
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != AMC_OK:
        raise AmcError(rc, load().amc_last_error().decode(errors="replace"))


def tvg_options(**kw) -> TvgOpts:
    """TwoViewGeometryOptions with COLMAP's C++ defaults; keyword overrides; `ransac` may be a
    dict of RANSACOptions fields."""
    o = TvgOpts()
    load().amc_tvg_opts_default(C.byref(o))
    for k, v in kw.items():
        if k == "ransac":
            for rk, rv in v.items():
                assert hasattr(o.ransac, rk), rk
                setattr(o.ransac, rk, rv)
        else:
            assert hasattr(o, k), k
            setattr(o, k, v)
    return o


def device_count() -> int:
    n = load().amc_device_count()
    if n < 0:
        _check(n)
    return n


def comm_unique_id() -> bytes:
    """amc_comm_unique_id (ncclGetUniqueId): one rank calls it, every rank passes the bytes to Context.comm_create."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().amc_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One amc_comm: this context's rank in the RCCL communicator of the exchange step (include/amc.h)."""

    def __init__(self, ctx: "Context", world_size: int, rank: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        self._lib, self._ctx = ctx._lib, ctx
        self.world_size, self.rank = int(world_size), int(rank)
        h = C.c_void_p()
        _check(self._lib.amc_comm_create(ctx._h, int(world_size), int(rank), unique_id, C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._lib.amc_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _tables_out(self, res, download):
        try:
            g_off = np.ctypeslib.as_array(res.offsets, shape=(int(res.npairs) + 1,)).copy()
            total = int(res.num_matches)
            g_m = None
            if download:
                g_m = (np.ctypeslib.as_array(res.matches, shape=(total, 2)).copy() if total
                       else np.zeros((0, 2), dtype=np.uint32))
            stats = {k: float(getattr(res, k)) for k in ("sizes_ms", "meta_ms", "rows_ms", "reorder_ms", "download_ms", "total_ms")}
            stats.update(num_matches=total, rows_sent=int(res.rows_sent), rows_received=int(res.rows_received),
                         world_size=int(res.world_size), rank=int(res.rank), device_ptr=int(res.matches_device or 0),
                         gather_path="C ABI: ncclAllGather (sizes) + grouped ncclSend/ncclRecv (records, rows) from device memory")
        finally:
            self._lib.amc_gathered_tables_free(C.byref(res))
        return g_off, g_m, stats

    def allgather_match_tables(self, pair_index, offsets, matches=None, download: bool = True):
        """amc_allgather_match_tables (collective).  pair_index: global positions of this rank's pairs, or None (the
        ranks' lists are appended in rank order); offsets: this rank's CSR; matches: this rank's rows on the host, or
        None = the context's device-resident table of the last match call.  Returns (global offsets uint64, global
        matches [M, 2] uint32 or None when download is False, stats dict incl. the device pointer of the table).

        Arguments this wrapper itself finds wrong (shapes, a `matches` array that does not hold offsets[-1] rows - the C
        entry point takes no length and would read past it) do NOT raise here, before the collective: the other ranks
        would wait for this one for ever.  The rank enters the exchange with offsets the library rejects ([1, 0]:
        "offsets[0] != 0"), so that every rank raises together; the local reason is appended to this rank's message."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64).reshape(-1)
        n = off.size - 1
        idx = None if pair_index is None else np.ascontiguousarray(pair_index, dtype=np.uint64)
        m = None if matches is None else np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
        local = None
        if n < 0:
            local = "offsets must have npairs + 1 entries"
        elif idx is not None and idx.shape != (n,):
            local = "one global position per local pair"
        elif m is not None and m.shape[0] != int(off[-1]) - int(off[0]):
            local = "offsets and matches disagree (%d rows for offsets[-1] = %d)" % (m.shape[0], int(off[-1]))
        if local is not None:   # poisoned entry: a one-pair CSR the library refuses collectively
            off, n, idx, m = np.array([1, 0], dtype=np.uint64), 1, None, np.zeros((1, 2), dtype=np.uint32)
        res = GatheredTables()
        rc = self._lib.amc_allgather_match_tables(
            self._ctx._h, self._h, None if idx is None else idx.ctypes.data_as(C.c_void_p), n,
            off.ctypes.data_as(C.c_void_p), None if m is None else m.ctypes.data_as(C.c_void_p), 1 if download else 0,
            C.byref(res))
        if local is not None:
            raise AmcError(rc if rc != AMC_OK else AMC_E_INVALID, "amc_allgather_match_tables: " + local)
        _check(rc)
        return self._tables_out(res, download)

    def allgather_pair_records(self, pair_index, records=None, download: bool = True, dtype=None):
        """amc_allgather_pair_records (collective): one fixed-size record per pair of every rank, in the global pair
        order.  records: a 1-D structured / plain array with one element per local pair (itemsize a multiple of 8), or
        None = the amc_tvg records of this context's last verification call, read in device memory (then pair_index
        must have that call's number of pairs and the result has TVG_DTYPE).  Returns (records of all pairs or None when
        download is False, stats)."""
        idx = None if pair_index is None else np.ascontiguousarray(pair_index, dtype=np.uint64).reshape(-1)
        rec = None if records is None else np.ascontiguousarray(records).reshape(-1)
        rdt = TVG_DTYPE if rec is None else rec.dtype
        if dtype is not None:
            rdt = np.dtype(dtype)
        n = len(rec) if rec is not None else (len(idx) if idx is not None else 0)
        width = rdt.itemsize
        if rec is not None and idx is not None and len(idx) != n:   # (collective error, as above: a record size the library refuses)
            width, bad = 4, "one global position per local record"
        else:
            bad = None
        res = GatheredRecords()
        rc = self._lib.amc_allgather_pair_records(
            self._ctx._h, self._h, None if idx is None else idx.ctypes.data_as(C.c_void_p), n,
            None if rec is None else rec.ctypes.data_as(C.c_void_p), width, 1 if download else 0, C.byref(res))
        if bad is not None:
            raise AmcError(rc if rc != AMC_OK else AMC_E_INVALID, "amc_allgather_pair_records: " + bad)
        _check(rc)
        try:
            total = int(res.npairs)
            out = None
            if download:
                out = (np.frombuffer(C.string_at(res.records, total * width), dtype=rdt).copy() if total
                       else np.zeros(0, dtype=rdt))
            stats = dict(npairs=total, record_bytes=int(res.record_bytes), bytes_sent=int(res.bytes_sent),
                         bytes_received=int(res.bytes_received), world_size=int(res.world_size), rank=int(res.rank),
                         total_ms=float(res.total_ms), device_ptr=int(res.records_device or 0))
        finally:
            self._lib.amc_gathered_records_free(C.byref(res))
        return out, stats

    def allgather_inlier_tables(self, pair_index, npairs_local: int | None = None, download: bool = True):
        """amc_allgather_inlier_tables (collective): the inlier matches of this context's last verification call
        (compacted on the device from the match table and the masks), of every rank, in the global pair order - the
        `two_view_geometries.data` blobs.  Returns what allgather_match_tables returns."""
        idx = None if pair_index is None else np.ascontiguousarray(pair_index, dtype=np.uint64).reshape(-1)
        n = len(idx) if idx is not None else int(npairs_local or 0)
        res = GatheredTables()
        _check(self._lib.amc_allgather_inlier_tables(self._ctx._h, self._h, None if idx is None else idx.ctypes.data_as(C.c_void_p),
                                                     n, 1 if download else 0, C.byref(res)))
        return self._tables_out(res, download)


class Context:
    """One amc_ctx (one GPU). Owns device copies of every uploaded image."""

    def comm_create(self, world_size: int, rank: int, unique_id: bytes) -> Comm:
        """amc_comm_create (collective over the ranks that share `unique_id`)."""
        import weakref
        comm = Comm(self, world_size, rank, unique_id)
        self.__dict__.setdefault("_comms", []).append(weakref.ref(comm))   # closed with the context, before it
        return comm

    def __init__(self, device_id: int = 0):
        self._lib = load()
        h = C.c_void_p()
        _check(self._lib.amc_ctx_create(device_id, C.byref(h)))
        self._h = h
        self.device_id = device_id
        # Bumped by everything that may free or overwrite the resident match table (a match call of any kind, trim,
        # close).  A holder of resident_matches_tensor() - memory the library owns - records it when it takes the view
        # and checks it (resident_view_valid) before / after it hands the view to anything asynchronous.
        self.resident_generation = 0

    def close(self) -> None:
        self.resident_generation = getattr(self, "resident_generation", 0) + 1
        for ref in self.__dict__.pop("_comms", []):
            comm = ref()
            if comm is not None:
                comm.close()
        if getattr(self, "_h", None):
            self._lib.amc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: int | None) -> None:
        _check(self._lib.amc_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def last_timeline(self) -> dict:
        """amc_ctx_last_timeline: the host-side timeline of the last match_verify_pairs call (ms since its entry)."""
        buf = (C.c_double * 8)()
        _check(self._lib.amc_ctx_last_timeline(self._h, buf))
        return dict(verify_setup_done=buf[0], match_returned=buf[1], verify_launched=buf[2], verify_results_on_host=buf[3],
                    call_returned=buf[4], batch_handover_host_ms_hidden=buf[5])

    def resident_matches(self):
        """(device pointer, number of matches) of the last match call's table in device memory (amc_ctx_resident_matches):
        CSR order of that call's result, valid until the next match call / trim / close."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _check(self._lib.amc_ctx_resident_matches(self._h, C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def upload_matches(self, matches) -> int:
        """Match rows (uint32 [n, 2], the CSR order of the pair list they belong to) into the resident match table
        (amc_upload_matches): verify_pairs(..., matches=None) reads them there - once, or again with other options -
        instead of taking them over PCIe in every call.  Returns the number of rows."""
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
        self.resident_generation += 1
        _check(self._lib.amc_upload_matches(self._h, m.ctypes.data_as(C.c_void_p), C.c_uint64(m.shape[0])))
        return int(m.shape[0])

    def resident_view_valid(self, generation: int) -> bool:
        """True while a view taken at `generation` (= self.resident_generation at that time) still points at live memory."""
        return generation == self.resident_generation and bool(getattr(self, "_h", None))

    def resident_matches_tensor(self, device_index: int = 0, copy: bool = False):
        """The same table as a torch int32 [n, 2] tensor.  copy=False: it ALIASES the library's device memory (no copy) -
        what the exchange step hands to RCCL - and dies with the next match call / trim / close of this context: note
        `self.resident_generation` when taking it and check `resident_view_valid()` before relying on it (bench.py and
        tools/dist_smoke.py do, around their collectives).  copy=True: a private clone, valid for as long as it is held."""
        import torch
        ptr, n = self.resident_matches()
        if n == 0:
            return torch.zeros((0, 2), dtype=torch.int32, device=torch.device("cuda", device_index))
        if copy:
            return self.resident_matches_tensor(device_index, copy=False).clone()

        class _View:   # numpy-style CUDA array interface over the raw pointer (uint32 bit patterns viewed as int32)
            __cuda_array_interface__ = {"shape": (n, 2), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}
        return torch.as_tensor(_View(), device=torch.device("cuda", device_index))

    def trim(self) -> None:
        """Release per-call scratch, staging buffers and idle result buffers (amc_ctx_trim); uploaded images stay."""
        self.resident_generation += 1
        _check(self._lib.amc_ctx_trim(self._h))

    def reserve_slots(self, n: int) -> None:
        _check(self._lib.amc_ctx_reserve_slots(self._h, n))

    def grow_slots(self, n: int) -> None:
        """Append empty slots up to n, keeping every uploaded image."""
        _check(self._lib.amc_ctx_grow_slots(self._h, n))

    def upload_descriptors(self, slot: int, desc: np.ndarray) -> None:
        d = np.ascontiguousarray(desc, dtype=np.uint8)
        if d.size and (d.ndim != 2 or d.shape[1] != 128):
            raise ValueError(f"descriptors must be N x 128 uint8, got {d.shape}")
        rows = d.shape[0] if d.ndim == 2 else 0
        _check(self._lib.amc_upload_descriptors(self._h, slot, d.ctypes.data_as(C.c_void_p), rows))

    def upload_descriptors_device(self, slot: int, dev_ptr: int, rows: int) -> None:
        _check(self._lib.amc_upload_descriptors_device(self._h, slot, C.c_void_p(dev_ptr), rows))

    def match_pairs(self, slot1, slot2, max_ratio: float = 0.8, max_distance: float = 0.7,
                    cross_check: bool = True, kernel: str = "auto", copy: bool = True):
        """Returns (offsets uint64[npairs+1], matches uint32[M,2], stats dict).  copy=False: the arrays are views of
        the library's pinned result buffer (freed when they are collected) instead of private copies."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        if s1.shape != s2.shape or s1.ndim != 1:
            raise ValueError("slot1/slot2 must be equal-length 1-D arrays")
        opts = MatchOpts(max_ratio, max_distance, 1 if cross_check else 0, KERNELS[kernel])
        res = MatchResult()
        self.resident_generation += 1
        _check(self._lib.amc_match_pairs(self._h, s1.ctypes.data_as(C.c_void_p),
                                         s2.ctypes.data_as(C.c_void_p), s1.size, C.byref(opts),
                                         C.byref(res)))
        if not copy:
            # zero-copy views of the library's pinned result buffer (what a C++ caller reads): the arrays keep the
            # result alive and amc_match_result_free runs when the last of them is collected
            return self._unpack_match(res, _MatchLease(self._lib, res))
        try:
            offsets, matches, stats = self._unpack_match(res)
        finally:
            self._lib.amc_match_result_free(C.byref(res))
        return offsets, matches, stats

    @staticmethod
    def _unpack_match(res, lease=None):
        n = int(res.npairs)

        def view(ptr, count, dtype):   # a buffer over the address: no per-call ctypes array types, no element walk
            nbytes = count * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * nbytes).from_address(C.addressof(ptr.contents)), dtype=dtype)

        offsets = view(res.offsets, n + 1, np.uint64)
        total = int(offsets[-1])
        offsets = offsets.copy() if lease is None else _LeasedArray.wrap(offsets, lease)
        if total:
            matches = view(res.matches, 2 * total, np.uint32).reshape(total, 2)
            matches = matches.copy() if lease is None else _LeasedArray.wrap(matches, lease)
        else:
            matches = np.zeros((0, 2), dtype=np.uint32)
        stats = dict(num_distances=int(res.num_distances), pairs_mfma=int(res.pairs_mfma),
                     pairs_dot4=int(res.pairs_dot4), device_ms=float(res.device_ms),
                     match_kernel_ms=float(res.match_kernel_ms),
                     cross_kernel_ms=float(res.cross_kernel_ms),
                     match_kernel_launches=int(res.match_kernel_launches))
        return offsets, matches, stats

    @staticmethod
    def _unpack_verify(res, total, labelled=False, lease=None):
        """The result as numpy arrays: copies (one per array), or - with a lease, which then owns the C result -
        views of the library's own buffers.  `labelled`: the mask holds geometry labels (multiple_models) rather
        than 0 / 1."""
        n = int(res.npairs)
        assert C.sizeof(Tvg) == TVG_DTYPE.itemsize

        def take(ptr, count, dtype):
            nbytes = count * np.dtype(dtype).itemsize
            if not nbytes:
                return np.zeros(0, dtype=dtype)
            addr = C.cast(ptr, C.c_void_p).value
            a = np.frombuffer((C.c_char * nbytes).from_address(addr), dtype=dtype)
            return a.copy() if lease is None else _LeasedArray.wrap(a, lease)

        tvg = take(res.tvg, n, TVG_DTYPE)
        labels = take(res.inlier_mask, total, np.uint8)
        mask = labels.astype(bool) if labelled else labels.view(np.bool_)   # 0 / 1 bytes are numpy bools as they are
        # inlier_labels: 1 + index of the geometry a match belongs to (multiple_models), else 0 / 1
        stats = dict(device_ms=float(res.device_ms), kernel_ms=float(res.kernel_ms), inlier_labels=labels,
                     kernel_launches=int(res.kernel_launches), work=[int(x) for x in res.work])
        stats["pose_kernel_ms"] = float(res.pose_kernel_ms)
        if res.pose:  # compute_relative_pose: one amc_pose per pair
            assert C.sizeof(Pose) == POSE_DTYPE.itemsize
            stats["pose"] = take(res.pose, n, POSE_DTYPE)
        return tvg, mask, stats

    def match_verify_pairs(self, slot1, slot2, opts: TvgOpts | None = None, seed: int = 0, max_ratio: float = 0.8,
                           max_distance: float = 0.7, cross_check: bool = True, kernel: str = "auto", copy: bool = True):
        """amc_match_verify_pairs: match every pair, then EstimateTwoViewGeometry on its matches where the matcher
        left them in HBM.  Returns (offsets, matches, match stats, tvg, inlier_mask, verify stats).  copy=False: the
        arrays are views of the library's result buffers (as a C++ caller reads them), each result released when the last
        array viewing it is garbage collected."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        if s1.shape != s2.shape or s1.ndim != 1:
            raise ValueError("slot1/slot2 must be equal-length 1-D arrays")
        mo = MatchOpts(max_ratio, max_distance, 1 if cross_check else 0, KERNELS[kernel])
        o = opts or tvg_options()
        mres, vres = MatchResult(), VerifyResult()
        self.resident_generation += 1
        _check(self._lib.amc_match_verify_pairs(self._h, s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p),
                                                s1.size, C.byref(mo), C.byref(o), seed, C.byref(mres), C.byref(vres)))
        if not copy:
            offsets, matches, mstats = self._unpack_match(mres, _MatchLease(self._lib, mres))
            tvg, mask, vstats = self._unpack_verify(vres, matches.shape[0], bool(o.multiple_models),
                                                    _VerifyLease(self._lib, vres))
            return offsets, matches, mstats, tvg, mask, vstats
        try:
            offsets, matches, mstats = self._unpack_match(mres)
            tvg, mask, vstats = self._unpack_verify(vres, matches.shape[0], bool(o.multiple_models))
        finally:
            self._lib.amc_match_result_free(C.byref(mres))
            self._lib.amc_verify_result_free(C.byref(vres))
        return offsets, matches, mstats, tvg, mask, vstats

    def match_guided_pairs(self, slot1, slot2, tvg, max_error: float, max_ratio: float = 0.8,
                           max_distance: float = 0.7, cross_check: bool = True):
        """FeatureMatcher::MatchGuided per pair: `tvg` is a TVG_DTYPE array (config, F, H used), e.g.
        the output of verify_pairs.  Returns (offsets, matches, stats) like match_pairs."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        g = np.ascontiguousarray(tvg, dtype=TVG_DTYPE)
        if s1.shape != s2.shape or s1.ndim != 1 or g.shape != s1.shape:
            raise ValueError("slot1/slot2/tvg must be equal-length 1-D arrays")
        assert C.sizeof(Tvg) == TVG_DTYPE.itemsize
        opts = MatchOpts(max_ratio, max_distance, 1 if cross_check else 0, KERNEL_AUTO)
        res = MatchResult()
        self.resident_generation += 1
        _check(self._lib.amc_match_guided_pairs(self._h, s1.ctypes.data_as(C.c_void_p),
                                                s2.ctypes.data_as(C.c_void_p), s1.size,
                                                g.ctypes.data_as(C.c_void_p), float(max_error), C.byref(opts),
                                                C.byref(res)))
        try:
            n = int(res.npairs)
            offsets = np.ctypeslib.as_array(res.offsets, shape=(n + 1,)).copy()
            total = int(offsets[-1])
            matches = (np.ctypeslib.as_array(res.matches, shape=(total, 2)).copy() if total
                       else np.zeros((0, 2), dtype=np.uint32))
            stats = dict(num_distances=int(res.num_distances), pairs_dot4=int(res.pairs_dot4),
                         pairs_guided_grid=int(res.pairs_guided_grid),
                         device_ms=float(res.device_ms))
        finally:
            self._lib.amc_match_result_free(C.byref(res))
        return offsets, matches, stats

    def upload_keypoints(self, slot: int, kp: np.ndarray) -> None:
        k = np.ascontiguousarray(kp, dtype=np.float32)
        if k.size and (k.ndim != 2 or k.shape[1] < 2):
            raise ValueError(f"keypoints must be N x (>=2) float32, got {k.shape}")
        rows = k.shape[0] if k.ndim == 2 else 0
        stride = k.shape[1] if k.ndim == 2 and rows else 2
        _check(self._lib.amc_upload_keypoints(self._h, slot, k.ctypes.data_as(C.c_void_p), rows, stride))

    def upload_points_f64(self, slot: int, pts: np.ndarray) -> None:
        """Double-precision image points (N x 2), as pycolmap's estimator bindings take them."""
        k = np.ascontiguousarray(pts, dtype=np.float64)
        if k.size and (k.ndim != 2 or k.shape[1] != 2):
            raise ValueError(f"points must be N x 2 float64, got {k.shape}")
        rows = k.shape[0] if k.ndim == 2 else 0
        _check(self._lib.amc_upload_points_f64(self._h, slot, k.ctypes.data_as(C.c_void_p), rows))

    def ransac_pairs(self, kind, slot1, slot2, match_offsets, matches, ransac: dict | RansacOpts | None = None,
                     seed: int = 0):
        """One LO-RANSAC (kind 'F' | 'H' | 'E') per pair. Returns (reports structured array [npairs],
        inlier_mask bool [total correspondences])."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        off = np.ascontiguousarray(match_offsets, dtype=np.uint64)
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
        if off.shape != (s1.size + 1,) or s1.shape != s2.shape:
            raise ValueError("match_offsets must have npairs + 1 entries")
        if int(off[-1]) != m.shape[0]:
            raise ValueError("match_offsets[-1] must equal the number of matches")
        if isinstance(ransac, RansacOpts):
            ro = ransac
        else:
            ro = tvg_options(ransac=ransac or {}).ransac
        k = RANSAC_KINDS[kind] if isinstance(kind, str) else int(kind)
        res = RansacResult()
        _check(self._lib.amc_ransac_pairs(self._h, k, s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p),
                                          s1.size, off.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p),
                                          C.byref(ro), seed, C.byref(res)))
        try:
            n = int(res.npairs)
            assert C.sizeof(RansacReport) == RANSAC_DTYPE.itemsize
            rep = (np.frombuffer(C.string_at(res.reports, n * C.sizeof(RansacReport)), dtype=RANSAC_DTYPE).copy()
                   if n else np.zeros(0, dtype=RANSAC_DTYPE))
            total = m.shape[0]
            mask = (np.ctypeslib.as_array(res.inlier_mask, shape=(total,)).astype(bool) if total
                    else np.zeros(0, dtype=bool))
        finally:
            self._lib.amc_ransac_result_free(C.byref(res))
        return rep, mask

    def squared_sampson_error(self, points1, points2, E) -> np.ndarray:
        p1 = np.ascontiguousarray(points1, dtype=np.float64).reshape(-1, 2)
        p2 = np.ascontiguousarray(points2, dtype=np.float64).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("points1 and points2 must have the same shape")
        e = np.ascontiguousarray(E, dtype=np.float64).reshape(9)
        out = np.empty(p1.shape[0], dtype=np.float64)
        _check(self._lib.amc_squared_sampson_error(self._h, p1.ctypes.data_as(C.c_void_p),
                                                   p2.ctypes.data_as(C.c_void_p), p1.shape[0],
                                                   e.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def homography_decomposition(self, H, K1, K2, points1, points2) -> dict:
        """PoseFromHomographyMatrix: dict(R [3,3], t [3], n [3], points3D [m,3])."""
        p1 = np.ascontiguousarray(points1, dtype=np.float64).reshape(-1, 2)
        p2 = np.ascontiguousarray(points2, dtype=np.float64).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("points1 and points2 must have the same shape")
        mats = [np.ascontiguousarray(a, dtype=np.float64).reshape(9) for a in (H, K1, K2)]
        R, t, n = np.empty(9), np.empty(3), np.empty(3)
        X = np.empty((max(1, len(p1)), 3))
        m = C.c_uint64(0)
        v = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(self._lib.amc_homography_decomposition(self._h, v(mats[0]), v(mats[1]), v(mats[2]), v(p1), v(p2), len(p1),
                                                      v(R), v(t), v(n), v(X), C.cast(C.byref(m), C.c_void_p)))
        return dict(R=R.reshape(3, 3), t=t, n=n, points3D=X[:m.value].copy())

    def cam_from_img(self, model: str | int, params, points) -> np.ndarray:
        """Camera::CamFromImg of an N x 2 array of image points."""
        p = np.ascontiguousarray(params, dtype=np.float64)
        xy = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        mid = CAMERA_MODELS[model] if isinstance(model, str) else int(model)
        out = np.empty_like(xy)
        _check(self._lib.amc_cam_from_img(self._h, mid, p.ctypes.data_as(C.c_void_p), p.size,
                                          xy.ctypes.data_as(C.c_void_p), xy.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def img_from_cam(self, model: str | int, params, points) -> np.ndarray:
        """Camera::ImgFromCam of an N x 2 array of normalised image-plane points."""
        p = np.ascontiguousarray(params, dtype=np.float64)
        uv = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        mid = CAMERA_MODELS[model] if isinstance(model, str) else int(model)
        out = np.empty_like(uv)
        _check(self._lib.amc_img_from_cam(self._h, mid, p.ctypes.data_as(C.c_void_p), p.size, uv.ctypes.data_as(C.c_void_p),
                                          uv.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def upload_camera(self, slot: int, model: str | int, width: int, height: int, params,
                      has_prior_focal_length: bool = False) -> None:
        p = np.ascontiguousarray(params, dtype=np.float64)
        mid = CAMERA_MODELS[model] if isinstance(model, str) else int(model)
        _check(self._lib.amc_upload_camera(self._h, slot, mid, width, height,
                                           p.ctypes.data_as(C.c_void_p), p.size,
                                           int(has_prior_focal_length)))

    def verify_pairs(self, slot1, slot2, match_offsets, matches, opts: TvgOpts | None = None, seed: int = 0,
                     copy: bool = True):
        """EstimateTwoViewGeometry per pair. Returns (tvg structured array [npairs], inlier_mask
        bool [total matches], stats).  copy=False: the arrays are views of the library's result buffers (as a C++
        caller reads them), released when the last of them is garbage collected.  matches=None: the rows are read from
        the resident match table (upload_matches, or the last match call's) - match_offsets[-1] must be its row count."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        off = np.ascontiguousarray(match_offsets, dtype=np.uint64)
        if off.shape != (s1.size + 1,) or s1.shape != s2.shape:
            raise ValueError("match_offsets must have npairs + 1 entries")
        if matches is None:
            m, nrows = None, int(off[-1]) if off.size else 0
        else:
            m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
            nrows = m.shape[0]
            if int(off[-1]) != nrows:
                raise ValueError("match_offsets[-1] must equal the number of matches")
        o = opts if opts is not None else tvg_options()
        res = VerifyResult()
        _check(self._lib.amc_verify_pairs(self._h, s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p),
                                          s1.size, off.ctypes.data_as(C.c_void_p),
                                          m.ctypes.data_as(C.c_void_p) if m is not None else None,
                                          C.byref(o), seed, C.byref(res)))
        if not copy:
            return self._unpack_verify(res, nrows, bool(o.multiple_models), _VerifyLease(self._lib, res))
        try:
            tvg, mask, stats = self._unpack_verify(res, nrows, bool(o.multiple_models))
        finally:
            self._lib.amc_verify_result_free(C.byref(res))
        return tvg, mask, stats

    def pose_pairs(self, slot1, slot2, match_offsets, inlier_matches, config, E=None, H=None) -> np.ndarray:
        """EstimateTwoViewGeometryPose for given geometries: config [npairs], E / H [npairs, 3, 3].
        Returns a POSE_DTYPE array [npairs]."""
        s1 = np.ascontiguousarray(slot1, dtype=np.uint32)
        s2 = np.ascontiguousarray(slot2, dtype=np.uint32)
        off = np.ascontiguousarray(match_offsets, dtype=np.uint64)
        m = np.ascontiguousarray(inlier_matches, dtype=np.uint32).reshape(-1, 2)
        if off.shape != (s1.size + 1,) or s1.shape != s2.shape:
            raise ValueError("match_offsets must have npairs + 1 entries")
        if int(off[-1]) != m.shape[0]:
            raise ValueError("match_offsets[-1] must equal the number of matches")
        n = s1.size
        geoms = np.zeros(n, dtype=TVG_DTYPE)
        geoms["config"] = np.asarray(config, dtype=np.int32).reshape(n)
        if E is not None:
            geoms["E"] = np.asarray(E, dtype=np.float64).reshape(n, 3, 3)
        if H is not None:
            geoms["H"] = np.asarray(H, dtype=np.float64).reshape(n, 3, 3)
        out = np.zeros(n, dtype=POSE_DTYPE)
        assert C.sizeof(Pose) == POSE_DTYPE.itemsize and C.sizeof(Tvg) == TVG_DTYPE.itemsize
        _check(self._lib.amc_pose_pairs(self._h, s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(n), off.ctypes.data_as(C.c_void_p),
                                        m.ctypes.data_as(C.c_void_p), geoms.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def acos_lut(self) -> np.ndarray:
        out = np.empty(262145, dtype=np.float32)
        _check(self._lib.amc_get_acos_lut(self._h, out.ctypes.data_as(C.c_void_p)))
        return out
