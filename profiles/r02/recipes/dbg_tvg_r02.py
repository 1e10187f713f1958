"""Per-phase cycle profile of the verification kernel for a fixed match count (round 2): does the counting loop
use the division-free test, and what does it buy?  usage: dbg_tvg_r02.py <inliers> <outliers> <npairs>"""
import os
import sys
import time

os.environ["AMC_TVG_PROFILE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402

rng = np.random.default_rng(3)
ni = int(sys.argv[1]) if len(sys.argv) > 1 else 210
no = int(sys.argv[2]) if len(sys.argv) > 2 else 90
npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
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
ctx.verify_pairs(s1, s1 + 1, off, mm, _capi.tvg_options())
t0 = time.perf_counter()
tvg, mask, st = ctx.verify_pairs(s1, s1 + 1, off, mm, _capi.tvg_options())
dt = time.perf_counter() - t0
print(f"M={ni + no} pairs={npairs} EXACT_COUNT={os.environ.get('AMC_TVG_EXACT_COUNT', '0')}: {npairs / dt:.0f} pairs/s, kernel "
      f"{st['kernel_ms']:.1f} ms, trials E/F/H {tvg['num_trials'][:, :3].mean(axis=0)}", flush=True)
