#!/usr/bin/env python3
"""RCCL run of the sharded match + verify exchange on GPUs, one rank per device, against the single-process result
(the N > 1 logic is covered by the gloo tests; this checks the same functions on device tensors over the nccl backend):
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 python tools/dist_smoke.py [--images N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_smoke.py
tests/test_multigpu_gpu.py runs both."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pycolmap_amd import _capi, synth  # noqa: E402
from pycolmap_amd import distributed as D  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", rank=rank, world_size=world)
dev = torch.device("cuda", torch.cuda.current_device())
rng = np.random.default_rng(0)
num_images = int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 6   # 2: one pair, an empty rank at world 2
imgs = synth.multiview_scene(rng, num_images=num_images, n_feats=600, num_landmarks=800)
ctx = _capi.Context(torch.cuda.current_device())
ctx.reserve_slots(len(imgs))
for k, im in enumerate(imgs):
    ctx.upload_descriptors(k, im["descriptors"])
    ctx.upload_keypoints(k, im["keypoints"])
    ctx.upload_camera(k, "PINHOLE", im["width"], im["height"], im["params"], True)
s1_all, s2_all = synth.exhaustive_pairs(len(imgs))
s1, s2, mine = D.shard_pairs(s1_all, s2_all, rank, world)
off, m, _ = ctx.match_pairs(s1, s2)
# ---- the exchange behind the C ABI (amc_allgather_match_tables: RCCL called by the library) ----
comm = D.make_comm(ctx)
c_off, c_m = D.all_gather_match_tables(mine, off, None, comm=comm, device_matches=True)   # rows from the resident table
abi_stats = D.last_gather_stats()
assert D.last_gather_path() == "c-abi" and abi_stats["world_size"] == world
h_off, h_m = D.all_gather_match_tables(mine, off, m, comm=comm)                            # rows from the host
assert np.array_equal(c_off, h_off) and np.array_equal(c_m, h_m)
a_off, a_m, _ = D.all_gather_appended_tables(off, m, comm=comm)                            # lists appended in rank order
assert int(a_off[-1]) == int(c_off[-1]) and len(a_off) == len(c_off)
n_off, n_m = D.all_gather_match_tables(mine, off, None, comm=comm, device_matches=True, download_rank=0)
assert (n_off is None) == (rank != 0)
try:                                                                                       # positions that are no permutation:
    D.all_gather_match_tables(np.zeros_like(mine), off, m, comm=comm)                      # every rank gets the error
    assert len(s1_all) <= 1, "duplicate positions went through"
except _capi.AmcError as e:
    assert e.code == _capi.AMC_E_INVALID, e
tvg, mask, _ = ctx.verify_pairs(s1, s2, off, m, _capi.tvg_options(compute_relative_pose=1))
# the match table straight from the device memory the kernels wrote (no host round trip before the collective)
gen = ctx.resident_generation
d_off, d_m = D.all_gather_match_tables(mine, off, m, device=dev, device_matches=ctx.resident_matches_tensor(torch.cuda.current_device()))
torch.cuda.synchronize()
assert ctx.resident_view_valid(gen)   # nothing touched the library's table while the collectives read it
g_tvg, g_off, g_m, g_ioff, g_im = D.all_gather_verification(mine, tvg, off, m, mask, len(s1_all), device=dev)
assert np.array_equal(d_off, g_off) and np.array_equal(d_m, g_m)
assert np.array_equal(c_off, g_off) and np.array_equal(c_m, g_m)   # the C ABI's exchange and torch's agree
# single-process reference on the same device
woff, wm, _ = ctx.match_pairs(s1_all, s2_all)
wtvg, wmask, _ = ctx.verify_pairs(s1_all, s2_all, woff, wm, _capi.tvg_options(compute_relative_pose=1))
assert np.array_equal(g_off, woff) and np.array_equal(g_m, wm)
assert g_tvg.tobytes() == wtvg.tobytes()
assert np.array_equal(g_im, wm[wmask]) and int(g_ioff[-1]) == int(wmask.sum())
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(f"dist smoke ok: {len(s1_all)} pairs, {len(wm)} matches, {int(wmask.sum())} inlier matches, world {world}, "
          f"gather path {D.last_gather_path()}; C ABI exchange: {abi_stats['total_ms']:.2f} ms "
          f"(sizes {abi_stats['sizes_ms']:.2f}, records {abi_stats['meta_ms']:.2f}, rows {abi_stats['rows_ms']:.2f}, reorder "
          f"{abi_stats['reorder_ms']:.2f}, download {abi_stats['download_ms']:.2f}), {abi_stats['rows_sent']} rows sent")
