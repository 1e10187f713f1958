"""Per-phase cycle profile of ONE LO-RANSAC per pair (amc_ransac_pairs: F, H or E alone), to size the phases of each
estimator separately.  usage: dbg_tvg_single.py <F|H|E> <inliers> <outliers> <npairs>"""
import os
import sys
import time

os.environ["AMC_TVG_PROFILE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "H"
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 210
no = int(sys.argv[3]) if len(sys.argv) > 3 else 90
npairs = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
rng = np.random.default_rng(3)
scenes = [synth.two_view_scene(rng, num_inliers=ni, num_outliers=no, noise=0.5) for _ in range(16)]
ctx = _capi.Context(0)
ctx.reserve_slots(32)
for k, sc in enumerate(scenes):
    for j, pts in enumerate((sc["pts1"], sc["pts2"])):
        ctx.upload_keypoints(2 * k + j, pts.astype(np.float32))
        ctx.upload_camera(2 * k + j, "PINHOLE", 1600, 1200, (1200.0, 1200.0, 800.0, 600.0), True)
which = np.arange(npairs) % 16
s1 = (2 * which).astype(np.uint32)
off = np.zeros(npairs + 1, dtype=np.uint64)
off[1:] = np.cumsum([len(scenes[w]["matches"]) for w in which])
mm = np.concatenate([scenes[w]["matches"] for w in which])
kw = dict(max_error=4.0, min_inlier_ratio=0.25, confidence=0.999, min_num_trials=100, max_num_trials=10000)
for _ in range(2):   # (the second call of a context still pays one-off allocations of the runtime)
    ctx.ransac_pairs(kind, s1, s1 + 1, off, mm, ransac=kw)
t0 = time.perf_counter()
rep, mask = ctx.ransac_pairs(kind, s1, s1 + 1, off, mm, ransac=kw)
dt = time.perf_counter() - t0
print(f"{kind} alone, M={ni + no}, pairs={npairs}: {npairs / dt:.0f} pairs/s, trials {rep['num_trials'].mean():.1f}, "
      f"inliers {rep['num_inliers'].mean():.1f}", flush=True)
