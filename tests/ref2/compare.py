"""Deviation budget of the verification oracle (results in DESIGN.md section 2).

Runs a fixed, seeded set of two-view scenes through
  * the default oracle (oracle/tvg_oracle.cc: D1 Gauss-Jordan / Jacobi, D2 bisection roots, D3 64-way sums),
  * its four variants (tests/ref2/variants.py: each deviation switched off, and all three),
  * the independent numpy restatement tests/ref2/tvg_ref2.py (LAPACK SVD, companion roots, sequential sums, a
    different 5-point algorithm, its own PRNG restatement),
and reports, against the default oracle: identical config, identical inlier mask, identical trial counts, and the
Hamming distance of the masks where they differ - split into pairs whose final models agree to rounding (the mask
difference is then a residual within rounding of max_error^2) and pairs where the RANSAC took another path.

Scenes: the 20 cases of tests/golden/tvg_golden_v4.npz, the 64 scenes of bench.py's verify leg, and `--random`
(default 500) seeded random scenes with varied inlier / outlier counts, noise, planarity and priors.

  python tests/ref2/compare.py [--random N] [--no-ref2] [--out tests/ref2/deviation_budget.json]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib as o  # noqa: E402
import tvg_golden  # noqa: E402
from pycolmap_amd import synth  # noqa: E402
from ref2 import tvg_ref2, variants  # noqa: E402

PIN = ("PINHOLE", (1200.0, 1200.0, 800.0, 600.0))


def scene_set(num_random: int):
    """[(tag, cam1 (model, params), cam2, prior, scene dict, option overrides)]"""
    out = []
    for c in tvg_golden.cases():
        out.append((f"golden{c['index']}", c["cam1"], c["cam2"], c["prior"],
                    dict(pts1=c["pts1"], pts2=c["pts2"], matches=c["matches"]), c["opts"]))
    rng = np.random.default_rng(7)                      # bench.py verify_leg's scenes
    for k in range(64):
        sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(150, 450)), num_outliers=int(rng.integers(50, 200)),
                                  planar=(k % 4 == 3))
        out.append((f"bench{k}", PIN, PIN, True, sc, {}))
    rng = np.random.default_rng(20260924)
    for k in range(num_random):
        kind = k % 5
        kw = dict(num_inliers=int(rng.integers(12, 500)), num_outliers=int(rng.integers(0, 300)),
                  noise=float(rng.choice([0.2, 0.5, 1.0, 2.0])))
        if kind == 1:
            kw["planar"] = True
        if kind == 2:
            kw.update(pure_rotation=True, noise=min(kw["noise"], 0.5))
        sc = synth.two_view_scene(rng, **kw)
        out.append((f"random{k}", PIN, PIN, bool(k % 2), sc, {}))
    return out


def model_close(a, b):
    """the two 3x3 models agree to rounding, up to scale and sign"""
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if not np.isfinite(na) or not np.isfinite(nb) or na == 0 or nb == 0:
        return bool(na == nb) or (not np.isfinite(na) and not np.isfinite(nb))
    a, b = a / na, b / nb
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-9


def compare(ref, got):
    same_cfg = ref["config"] == got["config"]
    ham = int((ref["inlier_mask"] != got["inlier_mask"]).sum())
    same_trials = list(ref["trials"]) == list(got["trials"])
    same_models = all(model_close(ref[k], got[k]) for k in "EFH")
    return dict(same_config=same_cfg, hamming=ham, same_trials=same_trials, same_models=same_models,
                same_inl=list(ref["inl"]) == list(got["inl"]))


def _rows_of(args):
    """rows of the scenes with index % workers == w, for the listed variants (a worker of evaluate())"""
    num_random, names, w, workers = args
    rows = {n: [] for n in names}
    for i, (tag, c1, c2, prior, sc, okw) in enumerate(scene_set(num_random)):
        if i % workers != w:
            continue
        cam1 = o.make_camera(c1[0], 1600, 1200, c1[1], prior=prior)
        cam2 = o.make_camera(c2[0], 1600, 1200, c2[1], prior=prior)
        opts = o.tvg_default_options(**okw)
        ref = o.estimate_two_view_geometry(cam1, sc["pts1"], cam2, sc["pts2"], sc["matches"], opts, seed=0)
        for n in names:
            if n == "ref2":
                if c1[0] not in ("PINHOLE", "SIMPLE_PINHOLE") or c2[0] not in ("PINHOLE", "SIMPLE_PINHOLE"):
                    continue
                d1 = dict(model=c1[0], width=1600, height=1200, params=c1[1], prior=prior)
                d2 = dict(model=c2[0], width=1600, height=1200, params=c2[1], prior=prior)
                got = tvg_ref2.estimate_two_view_geometry(d1, sc["pts1"], d2, sc["pts2"], sc["matches"], okw, seed=0)
            else:
                got = variants.estimate_two_view_geometry(n, cam1, sc["pts1"], cam2, sc["pts2"], sc["matches"], opts, seed=0)
            r = compare(ref, got)
            r.update(tag=tag, M=int(len(sc["matches"])), config=ref["config_name"], got_config=got["config_name"], index=i)
            rows[n].append(r)
    return rows


def evaluate(num_random: int = 500, no_ref2: bool = False, workers: int = 1, verbose: bool = True):
    """The budget as a dict (what main() writes).  workers > 1: the scenes are dealt to that many forked processes; the
    rows are merged back in scene order, so the result does not depend on the split."""
    names = list(variants.NAMES) + ([] if no_ref2 else ["ref2"])
    nscenes = 20 + 64 + num_random
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_rows_of, [(num_random, names, w, workers) for w in range(workers)])
    else:
        parts = [_rows_of((num_random, names, 0, 1))]
    rows = {n: sorted((r for part in parts for r in part[n]), key=lambda r: r["index"]) for n in names}
    for n in names:
        for r in rows[n]:
            del r["index"]
    summary = {}
    for n in names:
        rs = rows[n]
        N = len(rs)
        diff = [r for r in rs if r["hamming"] > 0]
        summary[n] = dict(
            what=variants.WHAT.get(n, "independent numpy restatement (LAPACK SVD, companion roots, sequential sums, "
                                      "Stewenius 5-point, own PRNG)"),
            pairs=N,
            identical_config=sum(r["same_config"] for r in rs),
            identical_mask=sum(r["hamming"] == 0 for r in rs),
            identical_trials=sum(r["same_trials"] for r in rs),
            identical_model_inlier_counts=sum(r["same_inl"] for r in rs),
            mask_differs=len(diff),
            mask_differs_same_models=sum(r["same_models"] for r in diff),
            mask_differs_other_ransac_path=sum(not r["same_models"] for r in diff),
            hamming_mean_where_differs=float(np.mean([r["hamming"] for r in diff])) if diff else 0.0,
            hamming_max=max([r["hamming"] for r in diff], default=0),
            hamming_total_over_matches=[int(sum(r["hamming"] for r in rs)), int(sum(r["M"] for r in rs))],
            config_changes=sorted({f"{r['config']}->{r['got_config']}" for r in rs if not r["same_config"]}),
        )
        s = summary[n]
        if verbose:
            print(f"{n:13s} pairs {N:4d}  config = {s['identical_config']:4d}  mask = {s['identical_mask']:4d}  trials = "
                  f"{s['identical_trials']:4d}  | differing masks: {len(diff)} (same models {s['mask_differs_same_models']}, "
                  f"other path {s['mask_differs_other_ransac_path']}), Hamming mean {s['hamming_mean_where_differs']:.1f} "
                  f"max {s['hamming_max']}  config changes {s['config_changes']}")
    return dict(scenes=nscenes, random=num_random, summary=summary,
                differing={n: [r for r in rows[n] if r["hamming"] or not r["same_config"]] for n in names})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--random", type=int, default=500)
    ap.add_argument("--no-ref2", action="store_true")
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--out", default=str(Path(__file__).with_name("deviation_budget.json")))
    args = ap.parse_args()
    Path(args.out).write_text(json.dumps(evaluate(args.random, args.no_ref2, args.workers), indent=1))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
