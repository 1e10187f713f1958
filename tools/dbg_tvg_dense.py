"""Per-phase cycle profile of the verification kernel on DENSE pairs (all matches inliers, no outliers):
the regime of tools/pipeline_bench.py's synthetic scene, where trial counts are minimal and the local
optimisation over all inliers dominates.  AMC_TVG_PROFILE=1 makes the library print the breakdown."""
import os
import sys
import time

os.environ["AMC_TVG_PROFILE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402

rng = np.random.default_rng(3)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1337
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
scenes = [synth.two_view_scene(rng, num_inliers=M, num_outliers=0, noise=0.5) for _ in range(16)]
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
ctx.verify_pairs(s1, s1 + 1, off, mm, _capi.tvg_options())
t0 = time.perf_counter()
tvg, mask, st = ctx.verify_pairs(s1, s1 + 1, off, mm, _capi.tvg_options())
dt = time.perf_counter() - t0
print(f"M={M} pairs={npairs}: {npairs / dt:.0f} pairs/s, kernel {st['kernel_ms']:.1f} ms, trials E/F/H "
      f"{tvg['num_trials'][:, :3].mean(axis=0)}, configs {np.unique(tvg['config'], return_counts=True)}")
