#!/usr/bin/env python3
"""Randomised parity stress of amc_verify_pairs against the CPU oracle: many scenes, option sets and
seeds per run, every field compared bit for bit.  Not part of the test suite (minutes, needs a GPU):

    python tools/stress_verify.py --rounds 20 --pairs 200
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def nbits(a):
    """bit patterns with every NaN canonical (x86 / gfx950 default NaNs differ in sign)"""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--wide", action="store_true",
                    help="also: multiple_models rounds (two-motion scenes, inlier labels compared), a few pairs of "
                         "thousands of matches per round (the larger size classes of the kernels)")
    args = ap.parse_args()
    import oracle_lib as o
    from pycolmap_amd import _capi, synth

    ctx = _capi.Context(0)
    rng = np.random.default_rng(args.seed)
    pool = ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8))
    total = bad = 0
    for rnd in range(args.rounds):
        ransac = dict(max_error=float(rng.choice([1.0, 2.0, 4.0, 6.0])),
                      min_inlier_ratio=float(rng.choice([0.05, 0.1, 0.25, 0.5])),
                      confidence=float(rng.choice([0.99, 0.999, 0.9999])),
                      min_num_trials=int(rng.choice([10, 64, 100, 130, 1000])),
                      max_num_trials=int(rng.choice([100, 500, 2000, 10000])))
        if ransac["min_num_trials"] > ransac["max_num_trials"]:
            ransac["min_num_trials"] = ransac["max_num_trials"]
        kw = dict(min_num_inliers=int(rng.choice([8, 15, 30])), detect_watermark=int(rng.integers(0, 2)),
                  force_H_use=int(rng.random() < 0.15), max_H_inlier_ratio=float(rng.choice([0.5, 0.8, 0.95])),
                  min_E_F_inlier_ratio=float(rng.choice([0.8, 0.95])), compute_relative_pose=int(rng.integers(0, 2)),
                  ransac=ransac)
        multi = bool(args.wide and rng.random() < 0.25)
        if multi:   # EstimateMultipleTwoViewGeometries: no pose, the mask bytes are geometry labels
            kw["multiple_models"] = 1
            kw["compute_relative_pose"] = 0
            kw["multiple_ignore_watermark"] = int(rng.integers(0, 2))
        seed = int(rng.integers(0, 2 ** 31))
        scenes, priors = [], []
        for q in range(args.pairs):
            kind = rng.integers(0, 4)
            big = args.wide and q < 3   # a few pairs of thousands of matches: the kernels' larger size classes
            sc = synth.two_view_scene(rng, num_inliers=int(rng.integers(1500, 6000)) if big else int(rng.integers(5, 500)),
                                      num_outliers=int(rng.integers(0, 3000)) if big else int(rng.integers(0, 300)),
                                      noise=float(rng.choice([0.2, 0.5, 1.5])), planar=kind == 1,
                                      pure_rotation=kind == 2, extra_keypoints=5)
            if multi and not big and rng.random() < 0.6:   # a second rigid motion glued on (tests/test_verify_gpu.py: two_motion_scene)
                b = synth.two_view_scene(rng, num_inliers=int(rng.integers(20, 250)), num_outliers=0, extra_keypoints=5)
                o1, o2 = len(sc["pts1"]), len(sc["pts2"])
                m = np.concatenate([sc["matches"].astype(np.int64), b["matches"].astype(np.int64) + [o1, o2]])
                sc = dict(sc)
                sc["pts1"] = np.concatenate([sc["pts1"], b["pts1"]])
                sc["pts2"] = np.concatenate([sc["pts2"], b["pts2"]])
                sc["matches"] = m[rng.permutation(len(m))].astype(np.uint32)
            scenes.append(sc)
            priors.append(bool(rng.integers(0, 2)))
        ctx.reserve_slots(2 * len(scenes))
        for i, (sc, pr) in enumerate(zip(scenes, priors)):
            cam = ("PINHOLE", sc["width"], sc["height"], (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0))
            for s, pts in ((2 * i, sc["pts1"]), (2 * i + 1, sc["pts2"])):
                ctx.upload_keypoints(s, pts.astype(np.float32))
                ctx.upload_camera(s, *cam, pr)
        s1 = np.arange(0, 2 * len(scenes), 2, dtype=np.uint32)
        off = np.zeros(len(scenes) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(sc["matches"]) for sc in scenes])
        matches = np.concatenate([sc["matches"] for sc in scenes])
        try:
            tvg, mask, vst = ctx.verify_pairs(s1, s1 + 1, off, matches, _capi.tvg_options(**kw), seed=seed)
        except _capi.AmcError as e:
            print(f"round {rnd}: rejected option set ({e}); skipped")
            continue
        okw = {k: v for k, v in kw.items() if k != "ransac"}
        okw.update(ransac)

        def ref(i):
            sc, pr = scenes[i], priors[i]
            cam = o.make_camera("PINHOLE", sc["width"], sc["height"],
                                (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=pr)
            return o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"],
                                                o.tvg_default_options(**okw), seed=seed)

        want = list(pool.map(ref, range(len(scenes))))
        for p, w in enumerate(want):
            g = tvg[p]
            if multi:
                ok = (_capi.CONFIG_NAMES[g["config"]] == w["config_name"] and g["num_inliers"] == w["num_inliers"] and
                      np.array_equal(vst["inlier_labels"][int(off[p]):int(off[p + 1])], w["inlier_label"]) and
                      all(np.array_equal(bits(g[k]), bits(w[k])) for k in "EFH") and
                      (w["config_name"] == "MULTIPLE" or g["num_trials"].tolist() == w["trials"]))
            else:
                ok = (_capi.CONFIG_NAMES[g["config"]] == w["config_name"] and g["num_trials"].tolist() == w["trials"] and
                      g["model_inliers"].tolist() == w["inl"] and g["num_inliers"] == w["num_inliers"] and
                      np.array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"]) and
                      all(np.array_equal(bits(g[k]), bits(w[k])) for k in "EFH"))
            if ok and kw["compute_relative_pose"]:
                q = vst["pose"][p]
                ok = (bool(q["ok"]) == w["pose_ok"] and int(q["num_points3D"]) == w["num_points3D"] and
                      np.array_equal(nbits(q["qvec"]), nbits(w["qvec"])) and np.array_equal(nbits(q["tvec"]), nbits(w["tvec"])) and
                      np.array_equal(nbits(q["R"]), nbits(w["R"])) and np.array_equal(nbits(q["tri_angle"]), nbits(w["tri_angle"])))
                if not ok:
                    print(f"POSE MISMATCH pair {p}: gpu q={q['qvec']} t={q['tvec']} tri={q['tri_angle']} n={q['num_points3D']} "
                          f"vs oracle q={w['qvec']} t={w['tvec']} tri={w['tri_angle']} n={w['num_points3D']}")
            total += 1
            if not ok:
                bad += 1
                print(f"MISMATCH round {rnd} pair {p}: gpu {_capi.CONFIG_NAMES[g['config']]} {g['num_trials'].tolist()} "
                      f"{g['model_inliers'].tolist()} vs oracle {w['config_name']} {w['trials']} {w['inl']}  opts={kw} seed={seed}")
        print(f"round {rnd}: {len(scenes)} pairs ok so far {total - bad}/{total}", flush=True)
    print(f"RESULT mismatches={bad} of {total}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
