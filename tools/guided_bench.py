"""Throughput of guided matching (amc_match_guided_pairs) on one synthetic scene: match + verify give each pair its
model, then the guided re-match of every verified pair is timed - by the candidate-generation kernel
(match_guided.hip) and, with AMC_GUIDED_DENSE=1, by the dense filtered scan it replaces.  Entries = 2 * n1 * n2 per
pair (both directions of the cross check), as DESIGN.md counts the 5.3e11 entries/s of round 1.
usage: guided_bench.py [images=64] [features=4096] [force_H=0]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
force_h = len(sys.argv) > 3 and sys.argv[3] == "1"
rng = np.random.default_rng(1)
imgs = synth.tower_scene(rng, num_images=n_img, n_feats=n_feat) if hasattr(synth, "tower_scene") else \
    synth.multiview_scene(rng, num_images=n_img, n_feats=n_feat, num_landmarks=int(1.3 * n_feat))
ctx = _capi.Context(0)
ctx.reserve_slots(len(imgs))
for k, im in enumerate(imgs):
    ctx.upload_descriptors(k, im["descriptors"])
    ctx.upload_keypoints(k, im["keypoints"])
    ctx.upload_camera(k, "PINHOLE", im["width"], im["height"], im["params"], True)
s1, s2 = synth.exhaustive_pairs(len(imgs))
off, m, _ = ctx.match_pairs(s1, s2)
tvg, _, _ = ctx.verify_pairs(s1, s2, off, m, _capi.tvg_options())
ok = np.isin(tvg["config"], [2, 3, 4, 5, 6])
g1, g2, gt = s1[ok], s2[ok], tvg[ok].copy()
if force_h:
    gt["config"] = 4
    for p in range(len(gt)):
        if not np.any(gt["H"][p]):
            gt["H"][p] = np.eye(3)
entries = 2.0 * sum(len(imgs[a]["descriptors"]) * len(imgs[b]["descriptors"]) for a, b in zip(g1, g2))
print(f"{len(imgs)} images x {n_feat}, {len(g1)} verified pairs of {len(s1)}, configs "
      f"{dict(zip(*np.unique(gt['config'], return_counts=True)))}", flush=True)
res = {}
for name, env in (("grid", None), ("dense", "1")):
    if env:
        os.environ["AMC_GUIDED_DENSE"] = env
    ctx.match_guided_pairs(g1, g2, gt, 4.0)
    t0 = time.perf_counter()
    o2, m2, st = ctx.match_guided_pairs(g1, g2, gt, 4.0)
    dt = time.perf_counter() - t0
    os.environ.pop("AMC_GUIDED_DENSE", None)
    res[name] = (o2, m2)
    print(f"{name:5s}: {dt * 1e3:8.1f} ms wall, device {st['device_ms']:8.1f} ms, {entries / dt:.3e} entries/s, "
          f"pairs grid/dense {st['pairs_guided_grid']}/{st['pairs_dot4']}, matches {len(m2)}", flush=True)
same = np.array_equal(res["grid"][0], res["dense"][0]) and np.array_equal(res["grid"][1], res["dense"][1])
print("grid == dense:", same)
sys.exit(0 if same else 1)
