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


def match_pairs(images, slot1, slot2, max_ratio=0.8, max_distance=0.7, cross_check=True, threads=8):
    """Batched oracle over a list of uint8 [n_i,128] images. Returns (offsets, matches)."""
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
    rc = load().oracle_match_pairs(_p(arena), _p(row_off), _p(rows), _p(s1), _p(s2), len(s1),
                                   max_ratio, max_distance, int(cross_check), _p(out_off),
                                   _p(counts), _p(out), threads)
    assert rc == 0
    offsets = np.zeros(len(s1) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    chunks = [out[int(out_off[p]):int(out_off[p]) + int(counts[p])] for p in range(len(s1))]
    matches = np.concatenate(chunks, axis=0) if chunks and offsets[-1] else np.zeros((0, 2), np.uint32)
    return offsets, matches


def acos_lut():
    out = np.empty(262145, dtype=np.float32)
    load().oracle_acos_lut(_p(out))
    return out
