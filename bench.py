#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: descriptor-pair distances/s for exhaustive
SIFT matching (all-pairs 128-D u8 dot products + top-2 + acos ratio/distance tests + cross check)
of N images x 4096 descriptors on MI355X.

A "step" is one full pass of the hot path over the workload: every image pair of the set goes
through libamc.so's amc_match_pairs (match kernel -> finalize -> match tables on the host), with
descriptors already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N=1 workload = BASELINE.json configs[1] (500 x 4096, 124,750 pairs, 2.09e12 distances).
N>1: weak scaling — the image set grows as 500*sqrt(N) so every rank matches ~124,750 pairs of
one replicated descriptor arena; ranks exchange their match tables with one RCCL all-gather at
the end of each step (the exchange step north_star names).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

INT8_DENSE_PEAK_OPS = 5.0e15  # gfx950 dense int8 MFMA: 2x the 2.5 PF bf16 dense peak
                              # (/opt/skills/guides/MI355X_MICROARCH.md: "I8 ... ~2x bf16 rate")
OPS_PER_DISTANCE = 256        # 128 int8 MACs


def make_arena_torch(num_images: int, feats: int, seed: int, device):
    """Seeded SIFT-like descriptors generated on the GPU (SURVEY.md section 8d recipe).

    Landmarks sit on a ring; image i views a window of the ring that overlaps its ~8 nearest
    neighbours on either side and nothing else — like an exhaustive match of a real capture,
    most of the N^2/2 pairs have no true overlap.  Per image: 60 % noisy copies of visible
    landmark prototypes + 40 % pure-noise features, shuffled, L2-normalised, x512, rounded,
    clamped to uint8 (COLMAP's storage convention)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sigma_d = 0.08
    k = int(0.6 * feats)            # landmark features per image
    W = 2 * k                       # window of the ring an image can see
    stride = max(1, W // 8)         # neighbours up to 8 apart share landmarks
    L = max(W, num_images * stride)
    proto = torch.randn(L, 128, generator=g, device=device).abs().pow(3.0)
    proto = proto / proto.norm(dim=1, keepdim=True)
    arena = torch.empty(num_images, feats, 128, dtype=torch.uint8, device=device)
    for i in range(num_images):
        win = (i * stride + torch.randperm(W, generator=g, device=device)[:k]) % L
        d = proto[win] + torch.randn(k, 128, generator=g, device=device) * sigma_d * proto[win].mean()
        if k < feats:
            noise = torch.randn(feats - k, 128, generator=g, device=device).abs().pow(3.0) * 0.1
            d = torch.cat([d, noise], 0)
        d = d[torch.randperm(feats, generator=g, device=device)].clamp_min(0)
        d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
        arena[i] = torch.round(512.0 * d).clamp(0, 255).to(torch.uint8)
    return arena


def host_cores() -> int:
    """CPU cores this process can actually use: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes show 256 logical CPUs but run the container with a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(arena_cpu: np.ndarray, s1: np.ndarray, s2: np.ndarray, sample_pairs: int, threads: int):
    """Time the CPU oracle ("port": oracle/match_oracle.c, the literal restatement of COLMAP's
    brute-force matcher) on a bounded seeded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    rng = np.random.default_rng(1234)
    idx = np.sort(rng.choice(len(s1), size=min(sample_pairs, len(s1)), replace=False))
    used = np.unique(np.concatenate([s1[idx], s2[idx]]))
    remap = {int(u): k for k, u in enumerate(used)}
    imgs = [arena_cpu[int(u)] for u in used]
    a = np.array([remap[int(x)] for x in s1[idx]], np.uint32)
    b = np.array([remap[int(x)] for x in s2[idx]], np.uint32)
    t0 = time.perf_counter()
    off, m = oracle_lib.match_pairs(imgs, a, b, threads=threads)
    dt = time.perf_counter() - t0
    ndist = float(sum(len(imgs[int(x)]) * len(imgs[int(y)]) for x, y in zip(a, b)))
    return ndist / dt, len(idx), dt, (idx, off, m)


def verify_leg(ctx_factory, device_index: int, npairs: int, steps: int, warmup: int, cpu_sample: int):
    """BASELINE's second metric: verified image-pairs/s.  Workload (configs[2] style): `npairs`
    synthetic calibrated two-view scenes (PINHOLE, prior focal length => E, F and H LO-RANSACs +
    model selection + watermark test all run), ~300 planted inliers + ~100 outliers each, a
    quarter of the scenes planar.  64 distinct seeded scenes are reused round-robin."""
    from pycolmap_amd import _capi, synth
    rng = np.random.default_rng(7)
    distinct = 64
    scenes = [synth.two_view_scene(rng, num_inliers=int(rng.integers(150, 450)),
                                   num_outliers=int(rng.integers(50, 200)), planar=(k % 4 == 3))
              for k in range(distinct)]
    ctx = ctx_factory()
    ctx.reserve_slots(2 * distinct)
    for k, sc in enumerate(scenes):
        for j, pts in enumerate((sc["pts1"], sc["pts2"])):
            ctx.upload_keypoints(2 * k + j, pts.astype(np.float32))
            ctx.upload_camera(2 * k + j, "PINHOLE", sc["width"], sc["height"],
                              (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), True)
    which = np.arange(npairs) % distinct
    s1 = (2 * which).astype(np.uint32)
    s2 = s1 + 1
    counts = np.array([len(scenes[w]["matches"]) for w in which], dtype=np.uint64)
    off = np.zeros(npairs + 1, dtype=np.uint64)
    off[1:] = np.cumsum(counts)
    matches = np.concatenate([scenes[w]["matches"] for w in which])
    opts = _capi.tvg_options()
    for _ in range(warmup):
        ctx.verify_pairs(s1, s2, off, matches, opts)
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(steps):
        tvg, mask, st = ctx.verify_pairs(s1, s2, off, matches, opts)
        kms += st["kernel_ms"]
    dt = time.perf_counter() - t0
    out = {
        "metric": "verified image-pairs/sec (E+F+H LO-RANSAC, model selection, watermark test)",
        "value": npairs * steps / dt, "unit": "pairs/s", "pairs": npairs, "steps": steps,
        "ms_per_step": 1e3 * dt / steps, "kernel_ms_per_step": kms / steps,
        "mean_matches_per_pair": float(counts.mean()), "dtype": "f64",
        "configs": {_capi.CONFIG_NAMES[c]: int(n) for c, n in zip(*np.unique(tvg["config"], return_counts=True))},
        "mean_trials_E_F_H": [float(x) for x in tvg["num_trials"][:, :3].mean(axis=0)],
    }
    # the same workload with TwoViewGeometryOptions.compute_relative_pose (pose.hip on the selected
    # inliers after the estimation): reported beside the metric, not as the metric
    popts = _capi.tvg_options(compute_relative_pose=1)
    ctx.verify_pairs(s1, s2, off, matches, popts)
    t0 = time.perf_counter()
    ptvg, pmask, pst = ctx.verify_pairs(s1, s2, off, matches, popts)
    pdt = time.perf_counter() - t0
    out["with_relative_pose"] = {
        "value": npairs / pdt, "unit": "pairs/s", "ms_per_step": 1e3 * pdt,
        "pose_kernel_ms_per_step": pst["pose_kernel_ms"],
        "mean_points3D_per_pair": float(pst["pose"]["num_points3D"].mean()),
        "configs": {_capi.CONFIG_NAMES[c]: int(n) for c, n in zip(*np.unique(ptvg["config"], return_counts=True))},
    }
    if cpu_sample > 0:
        # the oracle on every host core (OpenMP inside the oracle library, one pair per thread at a
        # time), `cpu_sample` pairs per core; every result is also compared with the GPU's
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as o
        cores = host_cores()
        n_cpu = min(npairs, cpu_sample * cores)
        cams, p1, p2, mm = [], [], [], []
        for p in range(n_cpu):
            sc = scenes[int(which[p])]
            cams.append(o.make_camera("PINHOLE", sc["width"], sc["height"],
                                      (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), prior=True))
            p1.append(sc["pts1"]); p2.append(sc["pts2"]); mm.append(sc["matches"])
        t0 = time.perf_counter()
        want = o.estimate_two_view_geometry_batch(cams, p1, cams, p2, mm, threads=cores)
        cdt = time.perf_counter() - t0
        mism = 0
        for p, w in enumerate(want):
            g = tvg[p]
            m = mask[int(off[p]):int(off[p + 1])]
            if (g["config"] != w["config"] or not np.array_equal(m, w["inlier_mask"]) or
                    not np.array_equal(g["F"].view(np.uint64), w["F"].view(np.uint64))):
                mism += 1
        out["cpu_baseline"] = {"value": n_cpu / cdt, "unit": "pairs/s", "cores": min(cores, n_cpu), "kind": "port",
                               "sample": f"{n_cpu} pairs of the same workload, oracle/tvg_oracle.cc (OpenMP, one pair "
                                         f"per thread at a time, {min(cores, n_cpu)} threads), {cdt:.1f} s",
                               "gpu_vs_oracle_mismatching_pairs": mism}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=500, help="images at N=1 (BASELINE configs[1])")
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--kernel", default="auto", choices=["auto", "mfma", "dot4"])
    ap.add_argument("--cpu-sample-pairs", type=int, default=0,
                    help="pairs timed on the host by the oracle (0 = eight per usable host core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-pairs", type=int, default=124750,
                    help="pairs in the verification leg (0 = skip); reported under \"verify\".  Default: every pair "
                         "of the 500-image set, BASELINE.json configs[2] (the kernel keeps 2048 waves busy from a "
                         "queue, so a 4096-pair call spends a quarter of its time in the tail: 64 k vs 92-99 k pairs/s)")
    ap.add_argument("--no-cross-check", action="store_true",
                    help="diagnostic: one-way matching only (NOT the BASELINE workload)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: run the multi-GPU exchange path (process group + all-gather) even with 1 rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pycolmap_amd import _capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    # ---- workload ------------------------------------------------------------------------
    num_images = args.images if world == 1 else int(round(args.images * math.sqrt(world)))
    arena = make_arena_torch(num_images, args.feats, seed=0, device=device)
    s1_all, s2_all = synth.exhaustive_pairs(num_images)
    # shard: order by image 2 (the kernel's L2-friendly order), deal contiguous slices to ranks
    from pycolmap_amd import distributed as D
    s1, s2, mine = D.shard_pairs(s1_all, s2_all, rank, world)

    ctx = _capi.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.reserve_slots(num_images)
    for i in range(num_images):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), args.feats)
    torch.cuda.synchronize()

    def step():
        off, m, st = ctx.match_pairs(s1, s2, kernel=args.kernel, cross_check=not args.no_cross_check)
        gathered = None
        if use_dist:
            # the exchange step: RCCL all-gather of the match tables (sizes, then padded tables);
            # afterwards every rank holds the whole match graph (rank 0 would feed the SQLite writer)
            gathered = D.all_gather_match_tables(mine, off, m, device=device, as_numpy=False)
        return off, m, st, gathered

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kern_ms, kern_launches, ndist_local, last = 0.0, 0, 0, None
    for _ in range(args.steps):
        last = step()
        kern_ms += last[2]["match_kernel_ms"]
        kern_launches += last[2]["match_kernel_launches"]
        ndist_local = last[2]["num_distances"]
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nd = torch.tensor([ndist_local], device=device, dtype=torch.float64)
        dist.all_reduce(nd, op=dist.ReduceOp.SUM)
        ndist_total = float(nd.item())
    else:
        ndist_total = float(ndist_local)

    if rank == 0:
        off, m, st, _ = last
        ms_per_step = 1e3 * elapsed / args.steps
        value = ndist_total * args.steps / elapsed
        # dominant kernel: the match kernel; live HIP-event duration per launch (rank 0)
        avg_kernel_s = (kern_ms / max(kern_launches, 1)) * 1e-3
        launches_per_step = max(kern_launches // max(args.steps, 1), 1)
        ops_per_launch = ndist_local * OPS_PER_DISTANCE / launches_per_step
        achieved = ops_per_launch / avg_kernel_s if avg_kernel_s > 0 else 0.0
        out = {
            "metric": "descriptor-pair distances/sec (exhaustive SIFT match: dot + top-2 + ratio + cross-check)",
            "value": value,
            "unit": "distances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 descriptors, int8 MFMA / int32 accumulate",
            "data": "synthetic",
            "config": {
                "workload": f"{num_images} images x {args.feats} descriptors, exhaustive match + "
                            f"ratio test + cross-check (BASELINE.json configs[1] at N=1)",
                "pairs_total": int(len(s1_all)),
                "pairs_per_rank": int(len(s1)),
                "distances_total": ndist_total,
                "kernel": args.kernel,
                "pairs_mfma": st["pairs_mfma"],
                "pairs_dot4": st["pairs_dot4"],
                "matches_rank0": int(m.shape[0]),
                "sharding": "pairs sorted by image 2, contiguous slice per rank, arena replicated"
                            + ("; RCCL all-gather of match tables per step" if world > 1 else ""),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved / 1e12,
                "peak": INT8_DENSE_PEAK_OPS / 1e12,
                "unit": "TOP/s (int8; 256 ops per descriptor-pair distance)",
                "frac": achieved / INT8_DENSE_PEAK_OPS,
                "traffic": None,  # filled below from the committed PMC pass when the launch shape is the same
                "kernel": "match_mfma_kernel" if st["pairs_mfma"] else "match_dot4_kernel",
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "launches_per_step": launches_per_step,
            },
        }
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be collected inside this process,
        # so the number comes from the committed rocprofv3 --pmc pass of this same command
        # (profiles/r01/pmc_hbm_v6.*), and only when this run launches the same shape.
        try:
            pmc = json.loads((ROOT / "profiles" / "r01" / "pmc_hbm_v6.json").read_text())
            per_launch = int(len(s1)) // launches_per_step
            if st["pairs_mfma"] and args.feats == 4096 and abs(per_launch - pmc["pairs_per_launch"]) <= 1:
                out["roofline"]["traffic"] = pmc["fetch_bytes_per_launch_corrected"]
                out["roofline"]["traffic_unit"] = "bytes read from HBM per launch (FETCH_SIZE x 1024 x 2, gfx950 correction)"
                out["roofline"]["traffic_source"] = "profiles/r01/pmc_hbm_v6.txt"
                out["roofline"]["algorithmic_bytes"] = float(per_launch) * 2 * args.feats * 128
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            arena_cpu = arena.cpu().numpy()
            sample = args.cpu_sample_pairs if args.cpu_sample_pairs > 0 else 8 * cores
            v, npairs, dt, (idx, coff, cm) = cpu_baseline(arena_cpu, s1, s2, sample, cores)
            cores = min(cores, npairs)  # the oracle runs one pair per thread
            # the sample doubles as a full-size parity spot check of the timed GPU result
            mism = 0
            for k, p in enumerate(idx):
                g = m[int(off[p]):int(off[p + 1])]
                c = cm[int(coff[k]):int(coff[k + 1])]
                if g.shape != c.shape or not np.array_equal(g, c):
                    mism += 1
            out["cpu_baseline"] = {
                "value": v, "unit": "distances/s", "cores": cores, "kind": "port",
                "sample": f"{npairs} seeded pairs of the same {num_images}x{args.feats} workload, "
                          f"oracle/match_oracle.c (-O2, OpenMP, one pair per thread), {dt:.1f} s",
                "gpu_vs_oracle_mismatching_pairs": mism,
            }
        if args.verify_pairs > 0 and world == 1:
            out["verify"] = verify_leg(lambda: _capi.Context(local_rank), local_rank, args.verify_pairs,
                                       max(1, args.steps), min(1, args.warmup),
                                       0 if args.no_cpu_baseline else 256)
        final_line = json.dumps(out)
    else:
        final_line = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if final_line is not None:
        # RCCL writes a version banner to the C stdout buffer, which would otherwise be flushed at
        # exit AFTER this line: drain it first so that the JSON line is the last thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
