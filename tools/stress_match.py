#!/usr/bin/env python3
"""Randomised parity stress of amc_match_pairs (both kernels) against the CPU oracle: ragged sizes,
tie-heavy / saturated / scene-like descriptors, random thresholds.  Needs a GPU; not in the test suite.

    python tools/stress_match.py --rounds 20
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def make_image(rng, n, kind, protos):
    if kind == 0:    # scene-like: noisy copies of shared prototypes + noise rows
        idx = rng.integers(0, len(protos), n)
        d = protos[idx] + rng.normal(0, 0.05, (n, 128)) * protos[idx].mean()
        d = np.clip(d, 0, None)
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
        return np.clip(np.round(d * 512), 0, 255).astype(np.uint8)
    if kind == 1:    # few distinct values: many exact ties
        return rng.integers(0, 3, (n, 128)).astype(np.uint8) * rng.integers(1, 60)
    if kind == 2:    # arbitrary bytes incl. saturation
        return rng.integers(0, 256, (n, 128)).astype(np.uint8)
    base = rng.integers(0, 40, (max(1, n // 7), 128)).astype(np.uint8)   # duplicated rows
    return base[rng.integers(0, len(base), n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import oracle_lib as o
    from pycolmap_amd import _capi

    ctx = _capi.Context(0)
    rng = np.random.default_rng(args.seed)
    total = bad = 0
    for rnd in range(args.rounds):
        protos = rng.gamma(0.7, 1.0, size=(600, 128))
        protos /= np.linalg.norm(protos, axis=1, keepdims=True)
        nimg = int(rng.integers(3, 9))
        kind = int(rng.integers(0, 4))
        sizes = [int(rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 255, 256, 257, 700, 1023, 1024, 1025, 1500]))
                 for _ in range(nimg)]
        imgs = [make_image(rng, n, kind, protos) for n in sizes]
        ctx.reserve_slots(nimg)
        for i, d in enumerate(imgs):
            ctx.upload_descriptors(i, d)
        s1, s2 = np.meshgrid(np.arange(nimg), np.arange(nimg), indexing="ij")
        s1, s2 = s1.ravel().astype(np.uint32), s2.ravel().astype(np.uint32)      # all ordered pairs incl. self
        max_ratio = float(rng.choice([0.6, 0.8, 0.95, 1.0, 1.2]))
        max_distance = float(rng.choice([0.3, 0.7, 1.0, 1.6]))
        cross = bool(rng.integers(0, 2))
        woff, wm = o.match_pairs(imgs, s1, s2, max_ratio=max_ratio, max_distance=max_distance, cross_check=cross, threads=64)
        want = [wm[int(woff[p]):int(woff[p + 1])] for p in range(len(s1))]
        for kernel in ("mfma", "dot4"):
            off, m, _ = ctx.match_pairs(s1, s2, max_ratio=max_ratio, max_distance=max_distance, cross_check=cross,
                                        kernel=kernel)
            for p in range(len(s1)):
                got = m[int(off[p]):int(off[p + 1])]
                total += 1
                if not np.array_equal(got, want[p]):
                    bad += 1
                    print(f"MISMATCH round {rnd} kernel {kernel} pair ({s1[p]},{s2[p]}) sizes {sizes[s1[p]]}x{sizes[s2[p]]} "
                          f"kind {kind} ratio {max_ratio} dist {max_distance} cross {cross}: {len(got)} vs {len(want[p])}")
        print(f"round {rnd}: kind {kind} sizes {sizes} ok so far {total - bad}/{total}", flush=True)
    print(f"RESULT mismatches={bad} of {total}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
