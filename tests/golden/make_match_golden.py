"""Generates tests/golden/match_golden_v1.npz.

There are NO golden vectors for this path in the reference (the
reference has no tests at all, and the arithmetic lives in the absent COLMAP 3.9.1), and the
Python reference cannot be imported here (`import pycolmap` fails: it is a pybind11 module over
libcolmap).  These fixtures are therefore produced by our own oracle (oracle/match_oracle.c) and
pin it against regressions; they do NOT pin it against COLMAP ("parity unpinned").

Run from the repo root:  python tests/golden/make_match_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib  # noqa: E402
from pycolmap_amd import synth  # noqa: E402


def build(verbose=True):
    """Every array of the fixture, from the oracle as it is now (tests/test_oracle_frozen_cpu.py compares them with the file)."""
    rng = np.random.default_rng(20260923)
    imgs = synth.scene_images(rng, 4, 160, num_landmarks=300, visible_frac=0.45)
    imgs.append(synth.random_descriptors(rng, 97))          # ragged, unrelated
    imgs.append(np.zeros((0, 128), np.uint8))               # empty image
    # adversarial rows: exact duplicates (ties), all-zero rows, saturated rows
    adv = imgs[0][:64].copy()
    adv[5] = adv[4]
    adv[10] = 0
    adv[11] = 255
    adv[12, :64] = 255
    imgs.append(adv)
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    out = {"num_images": np.int64(len(imgs)), "slot1": s1, "slot2": s2, "oracle_version": np.array(oracle_lib.match_version())}
    for k, im in enumerate(imgs):
        out[f"desc_{k}"] = im
    settings = {"default": (0.8, 0.7, True), "nocross": (0.8, 0.7, False),
                "loose": (0.95, 1.2, True), "tight": (0.6, 0.5, True)}
    for name, (r, d, cc) in settings.items():
        off, m = oracle_lib.match_pairs(imgs, s1, s2, r, d, cc, threads=4)
        out[f"{name}_opts"] = np.array([r, d, float(cc)])
        out[f"{name}_offsets"] = off
        out[f"{name}_matches"] = m
        if verbose:
            print(name, "total matches", int(off[-1]))
    return out


def main():
    np.savez_compressed(Path(__file__).with_name("match_golden_v1.npz"), **build())


if __name__ == "__main__":
    main()
