#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: descriptor-pair distances/s for exhaustive
SIFT matching (all-pairs 128-D u8 dot products + top-2 + acos ratio/distance tests + cross check)
of N images x 4096 descriptors on MI355X.

A "step" is one full pass of the hot path over the workload: every image pair of the set goes
through libamc.so's amc_match_pairs (match kernel -> finalize -> match tables on the host), with
descriptors already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W            # N > 1: launches itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config 3|4 | --weak]

--config 1 (default at N=1): BASELINE.json configs[1] (500 x 4096, 124,750 pairs, 2.09e12 distances).  With N>1 it
  is only run on request (--weak): weak scaling - the image set grows as 500*sqrt(N) so every rank matches ~124,750
  pairs of one replicated descriptor arena.  At N=1 the JSON line also carries
    "verify"   BASELINE's second metric, verified image-pairs/s, with its own FP64 roofline and CPU baseline,
    "pipeline" configs[2] as stated: match + F/E/H verification of the same image set, chained on the device
               (amc_match_verify_pairs) on a scene with real geometry,
    "dense"    the same match on a set where EVERY pair overlaps (reverse scan and D2H no longer negligible).
--config 3 (default at N>1): BASELINE configs[3], 2000 x 8192, the FIXED pair set sharded over the ranks (strong
            scaling) - the workload north_star states its ">= 7x at 8 GPUs" on.
--config 4: BASELINE configs[4], 10000 x 4096, sequential (overlap 50, quadratic) + loop-closure pairs
            (feature voting on the first 512 descriptors, then the match of the retrieved pairs), sharded.
In every multi-GPU mode ranks exchange their match tables with one RCCL all-gather at the end of each
step (the exchange step north_star names), inside the timed region; the line reports it separately
(config.exchange_ms_per_step) beside every rank's kernel time (config.kernel_ms_per_step_by_rank).
--cpu-dry-run (with --backend gloo): the same entry without a GPU - sharding, exchange, reductions and the JSON
line run for real over gloo on tiny sizes, the kernels are replaced by the CPU oracle.  A plumbing check for
tests/ (value is null: nothing it prints is a measurement).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import gc
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
INT8_MEASURED_CEILING_OPS = 3.944e15  # the guide's measured I8 MFMA microbenchmark ceiling
OPS_PER_DISTANCE = 256        # 128 int8 MACs
# FP64 vector peak of MI355X: half the 157.3 TFLOP/s FP32 vector peak the guide lists, an FMA counted as 2 flop.
# The verification kernels are built with -ffp-contract=off (every multiply and add rounds separately, as COLMAP's
# x86 build does): no FMA is ever issued, which halves the reachable ceiling.
FP64_VECTOR_PEAK = 78.6e12
FP64_NO_FMA_CEILING = 39.3e12
# Algorithmic FP64 work of the verification path.  Residuals are exact counts of the
# arithmetic COLMAP's residual loops perform per correspondence; the solver figures are operation counts of the
# solvers as written in tvg_math.h (elimination, polynomial set-up, an average bisection schedule).
FLOP_SAMPSON, FLOP_HRES, FLOP_TRES = 33, 20, 7
FLOP_E5_MIN, FLOP_F7_MIN, FLOP_H4_MIN = 30000, 2900, 1300
FLOP_LO_E5, FLOP_LO_F8, FLOP_LO_H = 72000, 44000, 44000      # 9 x 9 Jacobi (+ 5-point set-up / roots)
FLOP_LO_POINT = 200                                           # normalisation + design row + 45 MACs per inlier


def quiet_gc():
    """The timed loops hold large numpy / ctypes object graphs; a generation-2 collection in the middle of a step is a
    host-side stall of tens of milliseconds (seen once in five ragged steps).  The collector runs HERE, between the
    legs, and not by itself (reference counting still frees everything that is not a cycle).  Nothing measured changes."""
    gc.collect()
    gc.disable()


def verify_flops(work):
    """FP64 flop of a verification call from amc_verify_result.work (include/amc.h):
    (residuals evaluated in FP64 by the reference expression, solvers, residuals decided by the packed-FP32 pre-filters).
    COLMAP scores every model of every trial against every match; the kernels decide nearly all of those residuals in the
    packed-FP32 pre-filters of the counting loops - that is NOT FP64 work and stays out of the FP64
    numerator.  What the kernels do evaluate in FP64 (candidate re-scores, local-optimisation scores, inlier extraction,
    final masks) is counted by the kernels themselves: work[11]."""
    scoring64 = float(work[11])
    algorithmic = FLOP_SAMPSON * work[0] + FLOP_HRES * work[1] + FLOP_TRES * work[2]
    prefiltered = max(float(algorithmic) - scoring64, 0.0)
    solvers = (FLOP_E5_MIN * work[3] + FLOP_F7_MIN * work[4] + FLOP_H4_MIN * work[5] + FLOP_LO_E5 * work[6] +
               FLOP_LO_F8 * work[7] + FLOP_LO_H * work[8] + FLOP_LO_POINT * work[9])
    return float(scoring64), float(solvers), float(prefiltered)


def executed_valu(kernel_s_per_pair):
    """Executed vector instructions of the verification kernels from the committed SQ counter pass
    (profiles/rNN/pmc_tvg_rNN.json of the latest round, tools/pmc_tvg_r05.sh), valid only while the kernels' sources hash to what they were
    when the counters were taken: wave instructions x 64 lanes per pair, as a share of the FP64 issue rate."""
    try:
        import hashlib
        pmc_path = sorted((ROOT / "profiles").glob("r*/pmc_tvg_r*.json"))[-1]   # the latest round's pass; null unless the sources still match
        pmc = json.loads(pmc_path.read_text())
        if not all(hashlib.sha256((ROOT / f).read_bytes()).hexdigest() == h for f, h in pmc["kernel_source_sha256"].items()):
            return None
        lane_ops_per_pair = pmc["valu_wave_instructions_per_pair"] * 64.0
        return {"valu_lane_ops_per_pair": lane_ops_per_pair, "source": str(pmc_path.relative_to(ROOT)),
                "lane_utilisation": pmc.get("lane_utilisation"),
                "read_bytes_per_pair": pmc.get("read_bytes_per_pair"), "write_bytes_per_pair": pmc.get("write_bytes_per_pair"),
                "valu_lane_ops_per_s": lane_ops_per_pair / kernel_s_per_pair if kernel_s_per_pair > 0 else 0.0,
                "frac_of_valu_issue_peak": (lane_ops_per_pair / kernel_s_per_pair) / FP64_NO_FMA_CEILING if kernel_s_per_pair > 0 else 0.0,
                "note": "ALL vector instructions the two kernels executed (FP64, packed FP32, integer, moves), one lane-op per "
                        "lane and instruction, against the 39.3 T lane-op/s the vector ALU issues at FP64 rate; the counters "
                        "were taken on the workload named in the file, the rate uses this run's kernel time"}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def fp64_roofline(work, kernel_s, launches, pairs=0):
    scoring, solvers, scoring_h = verify_flops(work)
    total = scoring + solvers
    ach = total / kernel_s if kernel_s > 0 else 0.0
    out = {"bound": "fp64 vector (VALU); no MFMA, bytes negligible (correspondences stream through the scalar cache)",
           "achieved": ach / 1e12, "peak": FP64_VECTOR_PEAK / 1e12, "unit": "TFLOP/s (FP64)",
           "frac": ach / FP64_VECTOR_PEAK, "frac_of_no_fma_ceiling": ach / FP64_NO_FMA_CEILING,
           "no_fma_ceiling": FP64_NO_FMA_CEILING / 1e12,
           "flop_per_launch": total / max(launches, 1), "flop_scoring_share": scoring / total if total else 0.0,
           "fp32_prefilter_flop_not_counted": scoring_h,
           "achieved_if_prefiltered_residuals_were_counted_as_fp64": (total + scoring_h) / kernel_s / 1e12 if kernel_s > 0 else 0.0,
           "kernel": "tvg_e_kernel + tvg_fh_kernel", "avg_kernel_ms": 1e3 * kernel_s / max(launches, 1), "launches": launches,
           "traffic": None,
           "note": "FP64 flop the kernels execute for the algorithm: the solvers as COLMAP's loops count them and the residuals "
                   "evaluated with the reference FP64 expression (work[11]); the residuals of the trial models - all matches x "
                   "every model of every trial - are decided by packed-FP32 pre-filters (all three estimators) and are reported "
                   "separately, not in the numerator.  Useful work per second, not issue rate"}
    if pairs:
        out["executed"] = executed_valu(kernel_s / pairs)
        ex = out["executed"] or {}
        if ex.get("read_bytes_per_pair") is not None:
            # bytes through the L2's memory side per launch (a call = one E + one F/H launch: `launches` counts both), from
            # the committed counter pass named in executed.source - workspaces and spill slots, not inputs
            per_launch = pairs / max(launches, 1)
            out["traffic"] = ex["read_bytes_per_pair"] * per_launch
            out["traffic_written"] = ex["write_bytes_per_pair"] * per_launch
            out["traffic_rate_tb_s"] = (ex["read_bytes_per_pair"] + ex["write_bytes_per_pair"]) * pairs / kernel_s / 1e12 if kernel_s > 0 else None
            out["bound"] = "fp64 vector (VALU); no MFMA; workspace + spill traffic through the L2's memory side reported as `traffic`"
    return out


def make_arena_torch(num_images: int, feats: int, seed: int, device, overlap: str = "ring", stats: str = "l2"):
    """Seeded SIFT-like descriptors generated on the GPU (SURVEY.md section 8d recipe).

    Landmarks sit on a ring; image i views a window of the ring that overlaps its ~8 nearest
    neighbours on either side and nothing else — like an exhaustive match of a real capture,
    most of the N^2/2 pairs have no true overlap.  Per image: 60 % noisy copies of visible
    landmark prototypes + 40 % pure-noise features, shuffled, L2-normalised, x512, rounded,
    clamped to uint8 (COLMAP's storage convention).  stats="sift": what SIFT extractors really write - L2-normalise,
    clamp at 0.2, normalise again, x512 (Lowe's illumination clamp; COLMAP's and VLFeat's descriptors): nearly all
    bytes below 128, a handful above (the `sift_stats` leg)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sigma_d = 0.08
    k = int(0.6 * feats)            # landmark features per image
    W = 2 * k                       # window of the ring an image can see
    stride = max(1, W // 8)         # neighbours up to 8 apart share landmarks
    if overlap == "all":            # every image looks at the same window: every pair overlaps
        stride = 0
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
        if stats == "sift":
            d = d.clamp_max(0.2)
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
    # the same pairs by the AVX-512 VNNI variant of the matcher (oracle/match_vnni.c: vpdpbusd with the zero-point
    # identity, vectorised top-2): what the host's cores can do, beside what the literal restatement costs
    opt = None
    if oracle_lib.vnni_available():
        t1 = time.perf_counter()
        voff, vm = oracle_lib.match_pairs(imgs, a, b, threads=threads, variant="vnni")
        vdt = time.perf_counter() - t1
        opt = {"value": ndist / vdt, "unit": "distances/s", "cores": min(threads, len(idx)), "kind": "optimised",
               "sample": f"the same {len(idx)} pairs, oracle/match_vnni.c (AVX-512 VNNI vpdpbusd, -O3, OpenMP, one pair per "
                         f"thread), {vdt:.2f} s", "identical_to_port": bool(np.array_equal(voff, off) and np.array_equal(vm, m))}
    # COLMAP's DEFAULT CPU matcher is not the brute-force one but a FLANN k-d forest (approximate; SURVEY.md section 8 row
    # M4).  FLANN is not in this image; oracle/match_kdforest.cc restates it (4 trees, 128 checks, 2-NN, one index per
    # image built inside the timed call).  Rated in the distances the brute-force matcher would have computed.
    t2 = time.perf_counter()
    koff, km = oracle_lib.match_pairs(imgs, a, b, threads=threads, variant="kdforest")
    kdt = time.perf_counter() - t2
    port = {(k, int(x), int(y)) for k in range(len(idx)) for x, y in m[int(off[k]):int(off[k + 1])]}
    kdf = {(k, int(x), int(y)) for k in range(len(idx)) for x, y in km[int(koff[k]):int(koff[k + 1])]}
    default_cpu = {"value": ndist / kdt, "unit": "distances/s (brute-force equivalent)", "pairs_per_s": len(idx) / kdt,
                   "cores": min(threads, len(idx)), "kind": "port (approximate matcher, restated from the published algorithm)",
                   "sample": f"the same {len(idx)} pairs, oracle/match_kdforest.cc (FLANN-style k-d forest: 4 trees, 128 checks; "
                             f"{len(used)} indices built in the call), {kdt:.2f} s",
                   "port_matches_found": len(port & kdf) / max(1, len(port)),
                   "matches_that_are_port_matches": len(port & kdf) / max(1, len(kdf))}
    return ndist / dt, len(idx), dt, (idx, off, m), opt, default_cpu


def verify_leg(ctx_factory, device_index: int, npairs: int, steps: int, warmup: int, cpu_sample: int, distinct: int = 4096):
    """BASELINE's second metric: verified image-pairs/s.  Workload (configs[2] style): `npairs`
    synthetic calibrated two-view scenes (PINHOLE, prior focal length => E, F and H LO-RANSACs +
    model selection + watermark test all run), ~300 planted inliers + ~100 outliers each, a
    quarter of the scenes planar.  `distinct` seeded scenes (default 4096: different trial counts, branches and
    correspondences from pair to pair), reused round-robin to fill the 124,750 pairs."""
    quiet_gc()
    from pycolmap_amd import _capi, synth
    rng = np.random.default_rng(7)
    distinct = max(1, min(distinct, npairs))
    t_gen = time.perf_counter()
    scenes = [synth.two_view_scene(rng, num_inliers=int(rng.integers(150, 450)),
                                   num_outliers=int(rng.integers(50, 200)), planar=(k % 4 == 3))
              for k in range(distinct)]
    t_gen = time.perf_counter() - t_gen
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
    # The inputs are resident in HBM when the timed region starts, as for the match legs: keypoints and cameras above, the
    # match rows by amc_upload_matches (420 MB for 124,750 pairs) - amc_verify_pairs(matches = NULL) reads them where
    # they lie; the results (records and masks, 116 MB) come back to the host inside the timed region.  The call that
    # takes the rows over PCIe every time is timed beside it (`pcie_inclusive`), never as `value`.
    host_steps = max(1, min(3, steps))
    ctx.verify_pairs(s1, s2, off, matches, opts)
    t0 = time.perf_counter()
    for _ in range(host_steps):
        tvg = mask = st = None
        tvg, mask, st = ctx.verify_pairs(s1, s2, off, matches, opts, copy=False)
    hdt = time.perf_counter() - t0
    host_bits = (tvg.tobytes(), mask.copy())
    tvg = mask = st = None
    ctx.upload_matches(matches)
    for _ in range(warmup):
        w_ = ctx.verify_pairs(s1, s2, off, None, opts, copy=False)
        w_ = None
    t0 = time.perf_counter()
    kms, launches = 0.0, 0
    for _ in range(steps):
        tvg = mask = st = None                                    # release the previous result first
        tvg, mask, st = ctx.verify_pairs(s1, s2, off, None, opts, copy=False)   # views, as a C++ caller reads the result
        kms += st["kernel_ms"]
        launches += st["kernel_launches"]
    dt = time.perf_counter() - t0
    assert tvg.tobytes() == host_bits[0] and np.array_equal(mask, host_bits[1]), "resident and host-row verification differ"
    del host_bits
    out = {
        "metric": "verified image-pairs/sec (E+F+H LO-RANSAC, model selection, watermark test)",
        "value": npairs * steps / dt, "unit": "pairs/s", "pairs": npairs, "steps": steps,
        "ms_per_step": 1e3 * dt / steps, "kernel_ms_per_step": kms / steps,
        "inputs": "resident (amc_upload_matches); results downloaded inside the timed region",
        "pcie_inclusive": {"value": npairs * host_steps / hdt, "unit": "pairs/s", "ms_per_step": 1e3 * hdt / host_steps,
                           "steps": host_steps, "h2d_match_bytes_per_step": int(matches.nbytes)},
        "mean_matches_per_pair": float(counts.mean()), "dtype": "f64", "distinct_scenes": distinct,
        "scene_generation_s": t_gen,
        "configs": {_capi.CONFIG_NAMES[c]: int(n) for c, n in zip(*np.unique(tvg["config"], return_counts=True))},
        "mean_trials_E_F_H": [float(x) for x in tvg["num_trials"][:, :3].mean(axis=0)],
        "roofline": fp64_roofline([w * steps for w in st["work"]], kms * 1e-3, launches, pairs=npairs * steps),
    }
    # the same workload with TwoViewGeometryOptions.compute_relative_pose (pose.hip on the selected
    # inliers after the estimation): reported beside the metric, not as the metric
    popts = _capi.tvg_options(compute_relative_pose=1)
    w_ = ctx.verify_pairs(s1, s2, off, None, popts, copy=False)   # (the rows are still resident)
    w_ = None
    t0 = time.perf_counter()
    ptvg, pmask, pst = ctx.verify_pairs(s1, s2, off, None, popts, copy=False)   # (views, as in the leg's own loop)
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


def pipeline_leg(ctx_factory, steps: int, warmup: int, cpu_pairs: int, num_images: int, feats: int):
    """BASELINE.json configs[2] as it is stated: exhaustive match of the image set + F/E/H RANSAC verification of
    every pair with >= 15 matches, on ONE scene with real geometry (synth.tower_scene: an orbit capture, sparse
    overlap), the two stages chained on the device: amc_match_verify_pairs - the verification kernel reads the
    matches where the matcher left them in HBM."""
    quiet_gc()
    from pycolmap_amd import _capi, synth
    rng = np.random.default_rng(11)
    t0 = time.perf_counter()
    images = synth.tower_scene(rng, num_images=num_images, n_feats=feats)
    gen_s = time.perf_counter() - t0
    ctx = ctx_factory()
    ctx.reserve_slots(num_images)
    for k, im in enumerate(images):
        ctx.upload_descriptors(k, im["descriptors"])
        ctx.upload_keypoints(k, im["keypoints"])
        ctx.upload_camera(k, im["model"], im["width"], im["height"], im["params"], True)
    s1, s2 = synth.exhaustive_pairs(num_images)
    opts = _capi.tvg_options()
    for _ in range(warmup):   # (the timed loop's own call: result views, released before the next call)
        w_ = ctx.match_verify_pairs(s1, s2, opts, copy=False)
        w_ = None
    t0 = time.perf_counter()
    acc = dict(match_ms=0.0, scan_ms=0.0, cross_ms=0.0, verify_ms=0.0, verify_kernel_ms=0.0, launches=0)
    tl = dict(c_call_ms=0.0, verify_setup_ms=0.0, match_call_ms=0.0, close_and_launch_ms=0.0, verify_wait_pack_download_ms=0.0,
              batch_handover_host_ms_hidden=0.0, python_free_previous_ms=0.0, step_ms_max=0.0)
    for _ in range(steps):
        ts0 = time.perf_counter()
        off = m = tvg = mask = vst = None                          # release the previous results first
        tl["python_free_previous_ms"] += 1e3 * (time.perf_counter() - ts0)
        off, m, mst, tvg, mask, vst = ctx.match_verify_pairs(s1, s2, opts, copy=False)   # views, as a C++ caller reads the results
        tl["step_ms_max"] = max(tl["step_ms_max"], 1e3 * (time.perf_counter() - ts0))
        tl["steps_ms"] = tl.get("steps_ms", []) + [round(1e3 * (time.perf_counter() - ts0), 2)]
        if hasattr(ctx, "last_timeline"):   # the library's own clock inside the call (amc_ctx_last_timeline)
            t_ = ctx.last_timeline()
            tl["c_call_ms"] += t_["call_returned"]; tl["verify_setup_ms"] += t_["verify_setup_done"]
            tl["match_call_ms"] += t_["match_returned"] - t_["verify_setup_done"]
            tl["close_and_launch_ms"] += t_["verify_launched"] - t_["match_returned"]
            tl["verify_wait_pack_download_ms"] += t_["verify_results_on_host"] - t_["verify_launched"]
            tl["batch_handover_host_ms_hidden"] += t_["batch_handover_host_ms_hidden"]
        acc["match_ms"] += mst["device_ms"]; acc["scan_ms"] += mst["match_kernel_ms"]; acc["cross_ms"] += mst["cross_kernel_ms"]
        acc["verify_ms"] += vst["device_ms"]; acc["verify_kernel_ms"] += vst["kernel_ms"]; acc["launches"] += vst["kernel_launches"]
    dt = time.perf_counter() - t0
    counts = np.diff(off.astype(np.int64))
    ver = counts >= 15
    nver = int(ver.sum())
    out = {
        "metric": "verified image-pairs/sec, BASELINE configs[2]: exhaustive match + F/E/H verification chained on the device",
        "value": nver * steps / dt, "unit": "verified pairs/s",
        "workload": f"{num_images} images x {feats} features of one synthetic orbit scene (synth.tower_scene, PINHOLE "
                    f"cameras with prior focal length: E, F and H run), all {len(s1)} pairs matched, the {nver} pairs "
                    f"with >= 15 matches verified",
        "pairs_total": int(len(s1)), "pairs_verified": nver, "steps": steps, "ms_per_step": 1e3 * dt / steps,
        "distances_per_s_whole_pipeline": float(mst["num_distances"]) * steps / dt,
        "stage_ms_per_step": {k: v / steps for k, v in acc.items() if k != "launches"},
        # where a step's wall time goes on the HOST (the library's clock inside the call + Python around it): the call's
        # phases add up to c_call_ms; ms_per_step - c_call_ms is Python (argument conversion, result views, freeing the
        # previous results); step_ms_max is the slowest step (a stall shows there, not in the kernels' spans)
        "host_timeline_ms_per_step": {k: (v if k in ("step_ms_max", "steps_ms") else v / steps) for k, v in tl.items()},
        "matches_per_verified_pair": {"mean": float(counts[ver].mean()) if nver else 0.0,
                                      "max": int(counts.max()) if len(counts) else 0},
        "mean_inliers_per_verified_pair": float(tvg["num_inliers"][ver].mean()) if nver else 0.0,
        "configs": {_capi.CONFIG_NAMES[c]: int(n) for c, n in zip(*np.unique(tvg["config"][ver], return_counts=True))},
        "mean_trials_E_F_H": [float(x) for x in tvg["num_trials"][ver][:, :3].mean(axis=0)] if nver else [0, 0, 0],
        "roofline": fp64_roofline([w * steps for w in vst["work"]], acc["verify_kernel_ms"] * 1e-3, acc["launches"]),
        "scene_generation_s": gen_s, "dtype": "u8 -> int8 MFMA / int32 (match), f64 (verification)",
    }
    # guided matching (SiftMatchingOptions.guided_matching) of the verified pairs with the models just estimated:
    # the third stage of the same pipeline when the option is on, by the candidate-generation kernel
    gi = np.flatnonzero(ver & np.isin(tvg["config"], [2, 3, 4, 5, 6]))
    if len(gi):
        g1, g2, gt = s1[gi], s2[gi], tvg[gi]
        max_error = float(opts.ransac.max_error)
        ctx.match_guided_pairs(g1, g2, gt, max_error)
        tg = time.perf_counter()
        for _ in range(steps):
            goff, gm, gst = ctx.match_guided_pairs(g1, g2, gt, max_error)
        gdt = (time.perf_counter() - tg) / steps
        entries = 2.0 * float(gst["num_distances"])          # both directions of the cross check
        out["guided"] = {"metric": "guided-matching matrix entries/sec (MatchGuided: geometric filter + top-2 + ratio + cross check)",
                         "value": entries / gdt, "unit": "entries/s", "pairs": int(len(gi)), "ms_per_step": 1e3 * gdt,
                         "pairs_by_kernel": {"candidate_generation": gst["pairs_guided_grid"], "dense_filtered_scan": gst["pairs_dot4"]},
                         "matches_per_pair": float(len(gm)) / len(gi),
                         "note": "entries = 2 n1 n2 per pair, the size of the matrix MatchGuided scores; the candidate kernel "
                                 "evaluates the filter on ~4 % and the dot product on ~1 % of them"}
        if cpu_pairs > 0:
            sys.path.insert(0, str(ROOT / "tests"))
            import oracle_lib as o
            rs = np.random.default_rng(6)
            pick = np.sort(rs.choice(len(gi), size=min(len(gi), max(cpu_pairs // 8, 2)), replace=False))
            tg = time.perf_counter()
            gmis = 0
            for k in pick:
                a, b = int(g1[k]), int(g2[k])
                want = o.match_guided(images[a]["descriptors"], images[a]["keypoints"], images[b]["descriptors"],
                                      images[b]["keypoints"], gt[k]["config"], gt[k]["F"], gt[k]["H"], max_error)
                gmis += int(not np.array_equal(gm[int(goff[k]):int(goff[k + 1])], want))
            out["guided"]["cpu_baseline"] = {"value": 2.0 * feats * feats * len(pick) / (time.perf_counter() - tg), "unit": "entries/s",
                                             "cores": 1, "kind": "port", "sample": f"{len(pick)} seeded pairs, oracle_match_guided",
                                             "gpu_vs_oracle_mismatching_pairs": gmis}
    if cpu_pairs > 0:
        # the CPU oracles on the same pairs: a seeded sample of ALL pairs for the rate (most do not overlap, as in
        # the job itself), plus verified pairs only for the parity of the chained result
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as o
        cores = host_cores()
        rs = np.random.default_rng(5)
        idx = np.sort(rs.choice(len(s1), size=min(cpu_pairs, len(s1)), replace=False))
        vidx = np.flatnonzero(ver)
        vidx = np.sort(rs.choice(vidx, size=min(len(vidx), max(cpu_pairs // 4, 1)), replace=False)) if len(vidx) else vidx
        cam = o.make_camera("PINHOLE", images[0]["width"], images[0]["height"], images[0]["params"], prior=True)
        kps = [im["keypoints"][:, :2].astype(np.float64) for im in images]
        t0 = time.perf_counter()
        imgs = [im["descriptors"] for im in images]
        coff, cm = o.match_pairs(imgs, s1[idx], s2[idx], threads=cores)
        mm = [cm[int(coff[k]):int(coff[k + 1])] for k in range(len(idx))]
        todo = [k for k in range(len(idx)) if len(mm[k]) >= 15]
        want = o.estimate_two_view_geometry_batch([cam] * len(todo), [kps[int(s1[idx[k]])] for k in todo], [cam] * len(todo),
                                                  [kps[int(s2[idx[k]])] for k in todo], [mm[k] for k in todo], threads=cores)
        cdt = time.perf_counter() - t0
        mism = 0
        for k, p in enumerate(idx):
            if not np.array_equal(m[int(off[p]):int(off[p + 1])], mm[k]):
                mism += 1
        for k, w in zip(todo, want):
            p = idx[k]
            if tvg["config"][p] != w["config"] or not np.array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"]):
                mism += 1
        # parity on verified pairs (their matches come from the GPU result just compared pair by pair above)
        vw = o.estimate_two_view_geometry_batch([cam] * len(vidx), [kps[int(s1[p])] for p in vidx], [cam] * len(vidx),
                                                [kps[int(s2[p])] for p in vidx],
                                                [m[int(off[p]):int(off[p + 1])] for p in vidx], threads=cores)
        vmis = sum(1 for p, w in zip(vidx, vw)
                   if tvg["config"][p] != w["config"] or not np.array_equal(mask[int(off[p]):int(off[p + 1])], w["inlier_mask"])
                   or not np.array_equal(tvg["F"][p].view(np.uint64), w["F"].view(np.uint64)))
        out["cpu_baseline"] = {"value": len(idx) / cdt, "unit": "pairs/s (matched, and verified when >= 15 matches)",
                               "cores": min(cores, len(idx)), "kind": "port",
                               "sample": f"{len(idx)} seeded pairs of the same set ({len(todo)} of them verified), oracle/match_oracle.c "
                                         f"+ oracle/tvg_oracle.cc, OpenMP one pair per thread, {cdt:.1f} s",
                               "gpu_pairs_per_s_same_unit": len(s1) * steps / dt,
                               "gpu_vs_oracle_mismatching_pairs": mism,
                               "verified_pairs_checked": int(len(vidx)), "verified_pairs_mismatching": int(vmis)}
    ctx.close()
    return out


def dense_leg(ctx_factory, device, steps: int, warmup: int, num_images: int, feats: int, kernel: str, check: bool = True):
    """configs[1] on a set where EVERY pair overlaps (all images look at the same landmarks): the share of accepted
    rows is ~100x that of the sparse set, so the reverse scan of the candidate columns and the D2H of the match
    table stop being negligible.  Reported beside the headline, same unit."""
    quiet_gc()
    import torch
    arena = make_arena_torch(num_images, feats, seed=1, device=device, overlap="all")
    ctx = ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.reserve_slots(num_images)
    for i in range(num_images):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), feats)
    torch.cuda.synchronize()
    from pycolmap_amd import synth
    s1, s2 = synth.exhaustive_pairs(num_images)
    for _ in range(warmup):   # (the timed loop's own call: result views, released before the next call)
        w_ = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)
        w_ = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scan = cross = dev = 0.0
    t_free = t_call = 0.0
    for _ in range(steps):
        ta = time.perf_counter()
        off = m = None                       # give the previous result's pinned buffer back to the pool first
        tb = time.perf_counter()
        off, m, st = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)   # views of the result, as a C++ caller reads it
        tc = time.perf_counter()
        t_free += tb - ta; t_call += tc - tb
        scan += st["match_kernel_ms"]; cross += st["cross_kernel_ms"]; dev += st["device_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a seeded sample of the last step's pairs against the CPU oracle (~1,000 matches each)
    chk = spot_check(lambda u: arena[u].cpu().numpy(), s1, s2, off, m, sample=6) if check else {}
    ctx.close()
    nm = int(m.shape[0])
    return {"metric": "descriptor-pair distances/sec, every pair overlapping", "value": float(st["num_distances"]) * steps / dt,
            "unit": "distances/s", "workload": f"{num_images} images x {feats} descriptors, all images share their landmarks",
            "steps": steps, "ms_per_step": 1e3 * dt / steps, "matches_per_pair": nm / max(len(s1), 1),
            "match_table_bytes": nm * 8, **chk,
            "scan_kernel_ms": scan / steps, "resolve_select_reverse_scan_ms": cross / steps,
            "stage_ms_per_step": {"scan_kernel": scan / steps, "resolve_select_reverse_scan": cross / steps,
                                  "device_total_incl_d2h": dev / steps,
                                  "host_side_of_the_call": 1e3 * dt / steps - dev / steps,
                                  "python_free_previous_result": 1e3 * t_free / steps, "python_call": 1e3 * t_call / steps}}


def db_leg(num_images: int, feats: int, seed: int = 11):
    """What a pycolmap user sees: `pycolmap.match_exhaustive(database_path)` (the drop-in of
    /root/reference/pycolmap/pipeline/match_features.h:22-49, verification on, default options) on a COLMAP database
    holding BASELINE configs[2]'s image set - SQLite read, upload, match + verify on the device, SQLite write - with
    the controller's own breakdown (`last_run_stats()`), and the resume path (a second call finds everything there).
    The kernel legs above time the device work; this leg is the wall clock around it."""
    quiet_gc()
    import sys as _sys
    import tempfile
    _sys.path.insert(0, str(ROOT / "tests"))
    import colmap_db                      # writes the reference's schema with sqlite3 (test infrastructure: the fixture, not a checker)
    import pycolmap_amd as pc
    from pycolmap_amd import synth
    rng = np.random.default_rng(seed)
    images = synth.tower_scene(rng, num_images=num_images, n_feats=feats)
    for k, im in enumerate(images):
        im["name"] = f"im{k:05d}.jpg"
        im["prior"] = True
    with tempfile.TemporaryDirectory() as d:
        db = os.path.join(d, "bench.db")
        t0 = time.perf_counter()
        colmap_db.create(db, images)
        create_s = time.perf_counter() - t0
        size_in = os.path.getsize(db)
        t0 = time.perf_counter()
        pc.match_exhaustive(db)
        wall = time.perf_counter() - t0
        st = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in dict(pc.last_run_stats()).items()}
        dbo = pc.Database(db)
        matched, verified = int(dbo.num_matched_image_pairs), int(dbo.num_verified_image_pairs)
        nmatches, ninl = int(dbo.num_matches), int(dbo.num_inlier_matches)
        del dbo
        size_out = os.path.getsize(db)
        t0 = time.perf_counter()
        pc.match_exhaustive(db)
        rerun = time.perf_counter() - t0
    npairs = num_images * (num_images - 1) // 2
    return {"metric": "pycolmap.match_exhaustive(database) wall time, verification on", "value": npairs / wall,
            "unit": "image pairs/s through the API", "wall_s": wall, "rerun_wall_s": rerun,
            "workload": f"{num_images} images x {feats} features (synth.tower_scene, seed {seed}), {npairs} pairs, default "
                        f"SiftMatchingOptions / TwoViewGeometryOptions, COLMAP schema database on local disk",
            "stats": st, "db_bytes_before": size_in, "db_bytes_after": size_out, "create_db_s": create_s,
            "pairs_with_matches": matched, "pairs_verified": verified, "matches": nmatches, "inlier_matches": ninl}


def ragged_leg(ctx_factory, device, steps: int, warmup: int, num_images: int, lo: int, hi: int, kernel: str,
               uniform_value: float, check: bool = True):
    """configs[1] with image sizes a real capture has: n ~ U[lo, hi] descriptors per image (seeded), nothing a multiple
    of the kernel's 128-row segments or 256-row chunks.  Same generator, same sparse overlap as the headline; reported
    beside it in the same unit, with the scan kernel's own rate, so that a tiling that only suits 4096 = 4 x 1024 rows
    shows."""
    quiet_gc()
    import torch
    rng = np.random.default_rng(7)
    rows = rng.integers(lo, hi + 1, size=num_images)
    arena = make_arena_torch(num_images, hi, seed=2, device=device)   # rows are shuffled: a prefix is a random subset
    ctx = ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.reserve_slots(num_images)
    for i in range(num_images):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), int(rows[i]))
    torch.cuda.synchronize()
    from pycolmap_amd import synth
    s1, s2 = synth.exhaustive_pairs(num_images)
    for _ in range(warmup):   # (the timed loop's own call: result views, released before the next call)
        w_ = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)
        w_ = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scan = cross = 0.0
    launches = 0
    for _ in range(steps):
        off = m = None
        off, m, st = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)
        scan += st["match_kernel_ms"]; cross += st["cross_kernel_ms"]; launches += st["match_kernel_launches"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chk = spot_check(lambda u: arena[u, :int(rows[u])].cpu().numpy(), s1, s2, off, m, sample=8) if check else {}
    ctx.close()
    ndist = float(st["num_distances"])
    value = ndist * steps / dt
    scan_ops = ndist * steps * OPS_PER_DISTANCE / (scan * 1e-3) if scan > 0 else 0.0
    return {"metric": "descriptor-pair distances/sec, ragged image sizes", "value": value, "unit": "distances/s", **chk,
            "scan_kernel_ms": scan / steps, "resolve_select_reverse_scan_ms": cross / steps,
            "workload": f"{num_images} images, n ~ U[{lo}, {hi}] descriptors (seed 7; mean {float(rows.mean()):.0f}), "
                        f"exhaustive match + ratio test + cross-check",
            "steps": steps, "ms_per_step": 1e3 * dt / steps, "distances_per_step": ndist,
            "vs_uniform": value / uniform_value if uniform_value > 0 else None,
            "scan_frac_of_int8_peak": scan_ops / INT8_DENSE_PEAK_OPS,
            "stage_ms_per_step": {"scan_kernel": scan / steps, "resolve_select_reverse_scan": cross / steps},
            "scan_launches_per_step": launches // max(steps, 1), "matches": int(m.shape[0])}


def sift_stats_leg(ctx_factory, device, steps: int, warmup: int, num_images: int, feats: int, kernel: str,
                   uniform_value: float, check: bool = True):
    """configs[1] on descriptors with the byte statistics extractors really write (L2-normalise, clamp 0.2, renormalise,
    x512: make_arena_torch(stats="sift")) - the same scene recipe, the same sparse overlap as the headline.  The scan is
    bound by the chip's power budget and the clock it holds moves with the operand bytes: this leg
    says what the headline's kernel does on data shaped like a real capture's."""
    quiet_gc()
    import torch
    arena = make_arena_torch(num_images, feats, seed=3, device=device, stats="sift")
    hi_share = float((arena >= 128).float().mean().item())
    ctx = ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.reserve_slots(num_images)
    for i in range(num_images):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), feats)
    torch.cuda.synchronize()
    from pycolmap_amd import synth
    s1, s2 = synth.exhaustive_pairs(num_images)
    for _ in range(warmup):   # (the timed loop's own call: result views, released before the next call)
        w_ = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)
        w_ = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scan = cross = 0.0
    launches = 0
    for _ in range(steps):
        off = m = None
        off, m, st = ctx.match_pairs(s1, s2, kernel=kernel, copy=False)
        scan += st["match_kernel_ms"]; cross += st["cross_kernel_ms"]; launches += st["match_kernel_launches"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chk = spot_check(lambda u: arena[u].cpu().numpy(), s1, s2, off, m, sample=8) if check else {}
    ctx.close()
    ndist = float(st["num_distances"])
    value = ndist * steps / dt
    scan_ops = ndist * steps * OPS_PER_DISTANCE / (scan * 1e-3) if scan > 0 else 0.0
    return {"metric": "descriptor-pair distances/sec, extractor byte statistics", "value": value, "unit": "distances/s", **chk,
            "workload": f"{num_images} images x {feats} descriptors, L2-normalised, clamped at 0.2, renormalised, x512 "
                        f"(share of bytes >= 128: {hi_share:.5f}); exhaustive match + ratio test + cross-check",
            "steps": steps, "ms_per_step": 1e3 * dt / steps, "vs_headline": value / uniform_value if uniform_value > 0 else None,
            "scan_frac_of_int8_peak": scan_ops / INT8_DENSE_PEAK_OPS, "scan_kernel_ms": scan / steps,
            "resolve_select_reverse_scan_ms": cross / steps, "scan_launches_per_step": launches // max(steps, 1),
            "matches": int(m.shape[0])}


class _DryRunContext:
    """--cpu-dry-run only: stands where _capi.Context stands so that the multi-rank entry (sharding, exchange,
    reductions, the JSON line) can be exercised without a GPU.  The match kernels are replaced by the CPU oracle
    (tests/oracle_lib.py, test infrastructure): nothing measured through this class is reported as a value."""

    def __init__(self):
        self.imgs = {}

    def set_stream(self, stream):
        pass

    def reserve_slots(self, n):
        self.imgs = {}

    def upload_descriptors(self, slot, desc):
        self.imgs[int(slot)] = np.ascontiguousarray(desc, dtype=np.uint8)

    def match_pairs(self, s1, s2, kernel="auto", cross_check=True, copy=True):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib
        used = np.unique(np.concatenate([np.asarray(s1, np.int64), np.asarray(s2, np.int64)])) if len(s1) else np.zeros(0, np.int64)
        remap = {int(u): k for k, u in enumerate(used)}
        imgs = [self.imgs[int(u)] for u in used]
        a = np.array([remap[int(x)] for x in s1], np.uint32)
        b = np.array([remap[int(x)] for x in s2], np.uint32)
        t0 = time.perf_counter()
        off, m = oracle_lib.match_pairs(imgs, a, b, threads=1) if len(a) else (np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32))
        ms = 1e3 * (time.perf_counter() - t0)
        nd = int(sum(len(imgs[int(x)]) * len(imgs[int(y)]) for x, y in zip(a, b)))
        return off, m, {"num_distances": nd, "match_kernel_ms": ms, "match_kernel_launches": 1, "cross_kernel_ms": 0.0,
                        "device_ms": ms, "pairs_mfma": 0, "pairs_dot4": 0}

    def close(self):
        pass


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the
    driver's command line would (one process per GPU, rendezvous on 127.0.0.1)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def dist_setup(args):
    """(world, rank, local_rank, device, use_dist) for this process; initialises the process group when there is more
    than one rank (RCCL on the GPUs; gloo on the CPU under --cpu-dry-run)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.cpu_dry_run:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.cpu_dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, device_id=device, rank=rank, world_size=world)
    return world, rank, local_rank, device, use_dist


def device_sync(args):
    if not args.cpu_dry_run:
        import torch
        torch.cuda.synchronize()


def spot_check(get_image, s1, s2, off, m, sample: int = 6, seed: int = 99):
    """A seeded sample of the pairs a leg just matched, recomputed by the CPU oracle (oracle/match_oracle.c, the
    literal port) and compared row for row with the GPU's result: {"gpu_vs_oracle_mismatching_pairs", "pairs_checked"}."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    if len(s1) == 0:
        return {"gpu_vs_oracle_mismatching_pairs": 0, "pairs_checked": 0}
    rs = np.random.default_rng(seed)
    idx = np.sort(rs.choice(len(s1), size=min(sample, len(s1)), replace=False))
    used = np.unique(np.concatenate([s1[idx], s2[idx]]))
    remap = {int(u): k for k, u in enumerate(used)}
    imgs = [np.ascontiguousarray(get_image(int(u))) for u in used]
    a = np.array([remap[int(x)] for x in s1[idx]], np.uint32)
    b = np.array([remap[int(x)] for x in s2[idx]], np.uint32)
    coff, cm = oracle_lib.match_pairs(imgs, a, b, threads=min(host_cores(), len(idx)))
    mism = 0
    for k, p in enumerate(idx):
        g = m[int(off[p]):int(off[p + 1])]
        c = cm[int(coff[k]):int(coff[k + 1])]
        if g.shape != c.shape or not np.array_equal(g, c):
            mism += 1
    return {"gpu_vs_oracle_mismatching_pairs": int(mism), "pairs_checked": int(len(idx))}


def config34_leg(args, config: int, steps: int, warmup: int, dist_state):
    """BASELINE configs[3] (2000 x 8192 exhaustive) and configs[4] (10000 x 4096 sequential + loop): a FIXED
    workload whose pairs are sharded over the ranks by work (sum of n1 * n2), the descriptor arena replicated on
    every GPU, one all-gather of the match tables at the end of every step (inside the timed region) - through the
    library's own entry point (amc_allgather_match_tables: RCCL called behind the C ABI, rows straight from the
    device memory the kernels wrote).  Collective: every rank calls it; rank 0 gets the leg's dict, the others None."""
    quiet_gc()
    import torch
    import torch.distributed as dist
    from pycolmap_amd import _capi, synth
    from pycolmap_amd import distributed as D

    world, rank, local_rank, device, use_dist = dist_state
    dry = args.cpu_dry_run
    args = argparse.Namespace(**{**vars(args), "config": config, "steps": steps, "warmup": warmup})

    if args.config == 3:
        num_images = 2000 if args.images == 500 else args.images
        feats = 8192 if args.feats == 4096 else args.feats
    else:
        num_images = 10000 if args.images == 500 else args.images
        feats = args.feats
    arena = make_arena_torch(num_images, feats, seed=0, device=device)
    loop_feats = min(512, feats)
    if dry:
        ctx = _DryRunContext()
        ctx.reserve_slots(num_images * 2)
        for i in range(num_images):
            ctx.upload_descriptors(i, arena[i].numpy())
            if args.config == 4:
                ctx.upload_descriptors(num_images + i, arena[i, :loop_feats].numpy())
    else:
        ctx = _capi.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.reserve_slots(num_images * (2 if args.config == 4 else 1))
        for i in range(num_images):
            ctx.upload_descriptors_device(i, arena[i].data_ptr(), feats)
            if args.config == 4:   # the loop index: every image's first 512 descriptors (controller.cc SetupLoopIndex)
                ctx.upload_descriptors_device(num_images + i, arena[i].data_ptr(), loop_feats)
    device_sync(args)
    rows = np.full(num_images, feats)

    if args.config == 3:
        a_all, b_all = synth.exhaustive_pairs(num_images)
    else:
        # SequentialFeatureMatcher: offsets 1..49 and 2^k (quadratic overlap), self pairs and duplicates dropped
        import pycolmap_amd as pc
        blocks = pc._pycolmap._sequential_blocks(list(range(1, num_images + 1)), 50, True)
        seen, aa, bb = set(), [], []
        for blk in blocks:
            for i1, i2 in blk:
                if i1 == i2 or (i1, i2) in seen:
                    continue
                seen.add((i1, i2))
                aa.append(i1 - 1)
                bb.append(i2 - 1)
        a_all, b_all = np.array(aa, np.uint32), np.array(bb, np.uint32)
        del seen
    s1, s2, mine = D.shard_pairs(a_all, b_all, rank, world, rows=rows)
    queries = np.arange(0, num_images, 10)[rank::world] if args.config == 4 else np.zeros(0, np.int64)
    loop_num_images = min(50, num_images - 1)
    exch = {"ms": 0.0}

    # the exchange's communicator: the library's own (C ABI) on the GPUs - also with one rank, where the step is the
    # copy into the global order - and torch.distributed over gloo under --cpu-dry-run
    comm = None if dry else D.make_comm(ctx)
    do_exchange = use_dist or comm is not None
    abi = {"rows_ms": 0.0, "reorder_ms": 0.0, "rows_sent": 0}

    def exchange(fn):
        # the exchange step, timed on its own (host clock between device synchronisations): it waits for the slowest
        # rank's kernels, so it holds the load imbalance as well as the transfer
        device_sync(args)
        te = time.perf_counter()
        gen = None if dry else ctx.resident_generation
        r = fn()
        device_sync(args)
        # the collective read the library's own device memory (resident()): no match call may have touched it meanwhile
        assert dry or ctx.resident_view_valid(gen), "a match call ran while the exchange held the resident match table"
        exch["ms"] += 1e3 * (time.perf_counter() - te)
        if comm is not None:
            st_ = D.last_gather_stats()
            abi["rows_ms"] += st_["rows_ms"]; abi["reorder_ms"] += st_["reorder_ms"]; abi["rows_sent"] = st_["rows_sent"]
        return r

    def resident():
        # the match table where the kernels left it in HBM (no host round trip before the collective); with the
        # library's communicator the rows are read there by the library itself
        return None if dry else True

    def step():
        off, m, st = ctx.match_pairs(s1, s2, kernel=args.kernel, copy=False)
        nd = st["num_distances"]
        kms, kl = st["match_kernel_ms"], st["match_kernel_launches"]
        parts = [(mine, off, m)]
        extra = {}
        seq = (st["num_distances"], st["match_kernel_ms"], st["match_kernel_launches"], st["device_ms"])
        gathered = None
        if do_exchange:
            # sequential / exhaustive pairs have global positions: one all-gather of the tables, fed from device memory
            gathered = [exchange(lambda: D.all_gather_match_tables(mine, off, m, device=device, as_numpy=False,
                                                                   device_matches=resident(), comm=comm))]
        if args.config == 4 and len(queries):
            # loop closure (SequentialFeatureMatcher::RunLoopDetection with the vocabulary tree replaced by exact
            # feature voting): every 10th image against every other image on the first 512
            # descriptors, the 50 best-voted candidates are then matched at full size
            qs = queries.astype(np.uint32)
            q1 = np.repeat(qs, num_images - 1)
            base = np.arange(num_images - 1, dtype=np.uint32)
            q2 = np.tile(base, (len(qs), 1))                     # every image but the query itself, ascending
            q2 += base[None, :] >= qs[:, None]
            voff, _, vst = ctx.match_pairs(num_images + q1, num_images + q2.reshape(-1), kernel=args.kernel, copy=False)
            votes = np.diff(voff).reshape(len(qs), num_images - 1)
            # per query the loop_num_images best-voted candidates, more votes first, ties by image order, none without a
            # vote (= a stable sort of every row by -votes, cut, zero votes dropped: only the voted-for entries are sorted)
            r, c = np.nonzero(votes)
            so = np.lexsort((c, -votes[r, c].astype(np.int64), r))
            r, c = r[so], c[so]
            sel = (np.arange(len(r)) - np.searchsorted(r, np.arange(len(qs)))[r]) < loop_num_images
            voff = _ = None                                      # (the voting call's result goes back to the pool)
            l1 = qs[r[sel]]
            l2 = q2[r[sel], c[sel]]
            loff, lm, lst = ctx.match_pairs(l1, l2, kernel=args.kernel)
            nd += vst["num_distances"] + lst["num_distances"]
            kms += vst["match_kernel_ms"] + lst["match_kernel_ms"]
            kl += vst["match_kernel_launches"] + lst["match_kernel_launches"]
            def part(d, k, n, dev):   # one match call of the step: its forward scans against the int8 peak, and the call's device time
                return {"distances": int(d), "scan_ms": round(k, 2), "scan_launches": int(n), "device_ms": round(dev, 2),
                        "scan_frac": (d * OPS_PER_DISTANCE / (k * 1e-3) / INT8_DENSE_PEAK_OPS) if k > 0 else None}
            extra = dict(loop_queries=int(len(queries)), loop_scoring_pairs=int(len(q1)), loop_pairs=int(len(l1)),
                         loop_scoring_distances=int(vst["num_distances"]), loop_match_distances=int(lst["num_distances"]),
                         # where the step's scan time goes: 4096 x 4096 sequential pairs run at the headline's rate; the
                         # loop-closure VOTING matches 512 x 512 images (two 256-row chunks of Y per item: the item's fixed
                         # costs - X fragments, epilogue, 16 B of row table per row and pair - against a sixteenth of the scan)
                         parts={"sequential": part(*seq), "loop_voting_512x512": part(vst["num_distances"], vst["match_kernel_ms"],
                                                                                      vst["match_kernel_launches"], vst["device_ms"]),
                                "loop_matches": part(lst["num_distances"], lst["match_kernel_ms"], lst["match_kernel_launches"],
                                                     lst["device_ms"])})
            parts.append((None, loff, lm))
        if do_exchange and args.config == 4:
            # the loop pairs of a rank are numbered after those of the ranks before it (one small all-gather of the counts)
            have = len(parts) > 1
            lo_, lm_ = (parts[1][1], parts[1][2]) if have else (np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32))
            gathered.append(exchange(lambda: D.all_gather_appended_tables(lo_, lm_, device=device, as_numpy=False,
                                                                          device_matches=resident() if have else None,
                                                                          comm=comm)))
        return nd, kms, kl, int(sum(p[2].shape[0] for p in parts)), extra, gathered, (off, m)

    def fence():
        device_sync(args)
        if use_dist:
            dist.barrier()
            device_sync(args)

    for _ in range(args.warmup):
        step()
    fence()
    exch["ms"] = 0.0
    t0 = time.perf_counter()
    nd = kms = kl = nm = 0
    extra = {}
    last_tab = None
    for _ in range(args.steps):
        d_, k_, l_, nm, extra, _, last_tab = step()
        nd += d_; kms += k_; kl += l_
    fence()
    elapsed = time.perf_counter() - t0
    nd_rank = nd
    kernel_ms_by_rank = [kms / max(args.steps, 1)]
    exch_ms_by_rank = [exch["ms"] / max(args.steps, 1)]
    if use_dist:
        t = torch.tensor([elapsed, float(nd)], device=device, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, nd = float(tmax[0].item()), float(t[1].item())
        per = torch.tensor([kernel_ms_by_rank[0], exch_ms_by_rank[0]], device=device, dtype=torch.float64)
        allper = torch.empty(2 * world, device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(allper, per)
        allper = allper.cpu().view(world, 2)
        kernel_ms_by_rank = [float(x) for x in allper[:, 0]]
        exch_ms_by_rank = [float(x) for x in allper[:, 1]]
    out = None
    if rank == 0:
        avg_kernel_s = (kms / max(kl, 1)) * 1e-3
        ach = (nd_rank * OPS_PER_DISTANCE / max(kl, 1)) / avg_kernel_s if avg_kernel_s > 0 else 0.0
        name = ("2000 images x 8192 descriptors, exhaustive match, pairs sharded across the GPUs (BASELINE.json configs[3])"
                if args.config == 3 else
                "10000 images x 4096 descriptors, sequential (overlap 50, quadratic) + loop matching (BASELINE.json configs[4])")
        if (args.config == 3 and (num_images, feats) != (2000, 8192)) or (args.config == 4 and (num_images, feats) != (10000, 4096)):
            name = f"REDUCED {num_images} x {feats} variant of: " + name
        out = {
            "metric": "descriptor-pair distances/sec (exhaustive SIFT match: dot + top-2 + ratio + cross-check)",
            "value": None if dry else nd / elapsed, "unit": "distances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 descriptors, int8 MFMA / int32 accumulate", "data": "synthetic",
            "config": {"workload": name, "pairs_total": int(len(a_all)), "pairs_rank0": int(len(s1)),
                       "distances_per_step_all_ranks": nd / args.steps, "matches_rank0": nm,
                       "rccl_ranks": world if (comm is not None) else 0, "backend": ("gloo" if dry else args.backend) if use_dist else None,
                       "gather_path": D.last_gather_path() if do_exchange else None,
                       "kernel_ms_per_step_by_rank": kernel_ms_by_rank,
                       "exchange_ms_per_step": max(exch_ms_by_rank), "exchange_ms_per_step_by_rank": exch_ms_by_rank,
                       "exchange_rows_ms_per_step_rank0": abi["rows_ms"] / max(args.steps, 1),
                       "exchange_reorder_ms_per_step_rank0": abi["reorder_ms"] / max(args.steps, 1),
                       "exchange_rows_sent_rank0": abi["rows_sent"],
                       "sharding": "pairs sorted by image 2, contiguous slices of equal sum(n1*n2), arena replicated; "
                                   "one all-gather of the match tables per step (amc_allgather_match_tables: sizes, "
                                   "per-pair records, rows from the resident table by grouped ncclSend / ncclRecv)", **extra},
            "roofline": {"bound": "mfma", "achieved": ach / 1e12, "peak": INT8_DENSE_PEAK_OPS / 1e12,
                         "unit": "TOP/s (int8; 256 ops per descriptor-pair distance)", "frac": ach / INT8_DENSE_PEAK_OPS,
                         "frac_of_measured_i8_ceiling": ach / INT8_MEASURED_CEILING_OPS, "traffic": None,
                         "kernel": "match_mfma_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "launches_per_step": kl // max(args.steps, 1)},
        }
        if dry:
            out["dry_run"] = True
            out["metric"] = "DRY RUN (CPU oracle in place of the kernels, gloo in place of RCCL): NOT a measurement - " + out["metric"]
            out["data"] = "synthetic (cpu dry run)"
            out["roofline"] = None
        elif last_tab is not None:
            # rank 0's share of the last step against the CPU oracle, on a seeded sample of its pairs
            out["config"].update(spot_check(lambda u: arena[u].cpu().numpy(), s1, s2, last_tab[0], last_tab[1],
                                            sample=4 if args.config == 3 else 6))
    if comm is not None:
        comm.close()
    ctx.close()
    return out


def _short(v, n=60):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def _num(v):
    """floats to 5 significant digits (the line is read by people and by a driver that keeps its tail)"""
    if isinstance(v, float):
        return float(f"{v:.5g}")
    if isinstance(v, list):
        return [_num(x) for x in v]
    if isinstance(v, dict):
        return {k: _num(x) for k, x in v.items()}
    return v


def _pick(d, keys):
    return {k: _short(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out):
    """The line as printed: every contract key, `roofline`, `cpu_baseline` and every leg with its value and the few scalars
    a reader needs, strings cut to 60 characters, under 6 KB - the driver keeps known keys and the TAIL of stdout, and a
    15 KB line of workload descriptions lost `verify.value` there.  The explanations are DESIGN.md section 5;
    the detailed line is `--full-line` / `--detail-json`."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data") if k in out}
    if "dry_run" in out:
        c["dry_run"] = out["dry_run"]
    cfg = out.get("config") or {}
    c["config"] = {"workload": _short(cfg.get("workload", ""), 120), **_pick(cfg, ("pairs_total", "pairs_per_rank", "distances_total", "kernel",
                                                                               "matches_rank0", "gather_path"))}
    r = out.get("roofline")
    c["roofline"] = None if r is None else _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_written", "algorithmic_bytes",
                                                      "whole_step_frac", "avg_kernel_ms", "launches_per_step", "kernel", "traffic_source",
                                                      "traffic_source_current", "frac_of_measured_i8_ceiling"))
    if r is not None and "traffic" not in c["roofline"]:
        c["roofline"]["traffic"] = None
    b = out.get("cpu_baseline")
    if b is not None:
        c["cpu_baseline"] = _pick(b, ("value", "unit", "cores", "kind", "sample", "gpu_vs_oracle_mismatching_pairs", "optimised_value",
                                      "optimised_identical_to_port", "default_cpu_matcher_value", "default_cpu_matcher_pairs_per_s"))
    for name in ("config3", "config4"):
        leg = out.get(name)
        if leg is None:
            continue
        c[name] = _pick(leg, ("value", "unit", "scaling", "n_gpus", "steps", "ms_per_step", "exchange_ms_per_step", "per_rank_kernel_ms",
                              "max_rank_kernel_ms", "rccl_ranks", "gather_path", "pairs_total", "gpu_vs_oracle_mismatching_pairs",
                              "pairs_checked", "scan_frac_of_int8_peak", "dry_run"))
        c[name]["workload"] = _short(leg.get("workload", ""), 100)
        if leg.get("parts"):
            c[name]["parts"] = {k: _pick(v, ("distances", "scan_ms", "device_ms", "scan_frac")) for k, v in leg["parts"].items()}
    for k in ("config3_value", "config3_ms_per_step", "config3_exchange_ms_per_step", "config4_value"):
        if k in out:
            c[k] = out[k]
    v = out.get("verify")
    if v is not None:
        vr = v.get("roofline") or {}
        c["verify"] = {**_pick(v, ("value", "unit", "pairs", "steps", "ms_per_step", "kernel_ms_per_step", "mean_matches_per_pair",
                                   "mean_trials_E_F_H", "roofline_frac")),
                       "roofline": {**_pick(vr, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_written", "frac_of_no_fma_ceiling",
                                                 "flop_per_launch", "avg_kernel_ms")),
                                    "valu_issue_frac": (vr.get("executed") or {}).get("frac_of_valu_issue_peak"),
                                    "lane_utilisation": (vr.get("executed") or {}).get("lane_utilisation")},
                       "with_relative_pose_value": (v.get("with_relative_pose") or {}).get("value"),
                       "inputs": "resident", "pcie_inclusive_value": (v.get("pcie_inclusive") or {}).get("value"),
                       "pcie_inclusive_ms_per_step": (v.get("pcie_inclusive") or {}).get("ms_per_step"),
                       "gpu_vs_oracle_mismatching_pairs": (v.get("cpu_baseline") or {}).get("gpu_vs_oracle_mismatching_pairs"),
                       "cpu_value": (v.get("cpu_baseline") or {}).get("value")}
        if "traffic" not in c["verify"]["roofline"]:
            c["verify"]["roofline"]["traffic"] = None
    p_ = out.get("pipeline")
    if p_ is not None:
        sm = p_.get("stage_ms_per_step") or {}
        c["pipeline"] = {**_pick(p_, ("value", "unit", "pairs_total", "pairs_verified", "steps", "ms_per_step")),
                         **{k: sm.get(k) for k in ("match_ms", "scan_ms", "cross_ms", "verify_ms", "verify_kernel_ms")},
                         "non_scan_ms": (p_["ms_per_step"] - sm["scan_ms"]) if "scan_ms" in sm else None,
                         "host": _pick(p_.get("host_timeline_ms_per_step") or {}, ("c_call_ms", "verify_setup_ms", "match_call_ms",
                                                                                    "close_and_launch_ms", "verify_wait_pack_download_ms",
                                                                                    "batch_handover_host_ms_hidden", "python_free_previous_ms",
                                                                                    "step_ms_max")),
                         "verify_frac_fp64": (p_.get("roofline") or {}).get("frac"),
                         "guided_value": (p_.get("guided") or {}).get("value"),
                         "cpu_value": (p_.get("cpu_baseline") or {}).get("value"),
                         "gpu_vs_oracle_mismatching_pairs": (p_.get("cpu_baseline") or {}).get("gpu_vs_oracle_mismatching_pairs"),
                         "verified_pairs_mismatching": (p_.get("cpu_baseline") or {}).get("verified_pairs_mismatching")}
    for name, keys in (("ragged", ("value", "unit", "ms_per_step", "vs_uniform", "scan_frac_of_int8_peak", "scan_kernel_ms",
                                   "resolve_select_reverse_scan_ms", "gpu_vs_oracle_mismatching_pairs", "pairs_checked")),
                       ("sift_stats", ("value", "unit", "ms_per_step", "vs_headline", "scan_frac_of_int8_peak", "scan_kernel_ms",
                                       "gpu_vs_oracle_mismatching_pairs", "pairs_checked")),
                       ("dense", ("value", "unit", "ms_per_step", "matches_per_pair", "scan_kernel_ms", "resolve_select_reverse_scan_ms",
                                  "gpu_vs_oracle_mismatching_pairs", "pairs_checked")),
                       ("db", ("value", "unit", "wall_s", "rerun_wall_s", "pairs_with_matches", "pairs_verified", "error"))):
        if out.get(name) is not None:
            c[name] = _pick(out[name], keys)
    return _num(c)


def print_line(out, args=None):
    """The ONE JSON line, last thing on stdout (RCCL writes a version banner to the C stdout buffer, which would
    otherwise be flushed at exit AFTER it: drain that first)."""
    if out is None:
        return
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if args is not None and getattr(args, "detail_json", ""):
        Path(args.detail_json).write_text(json.dumps(out) + "\n")
    full = args is None or getattr(args, "full_line", False) or out.get("dry_run")
    print(json.dumps(out if full else compact_line(out)), flush=True)


def run_config34(args):
    """`--config 3 | 4`: that configuration alone, as the line."""
    import torch.distributed as dist
    state = dist_setup(args)
    out = config34_leg(args, args.config, args.steps, args.warmup, state)
    if state[4]:
        dist.barrier()
        dist.destroy_process_group()
    print_line(out, args)


def config3_summary(leg):
    """The strong-scaling leg every `python bench.py --gpus N` line carries (BASELINE configs[3], the workload north_star
    states its ">= 7x at 8 GPUs" on): the scalars the scaling curve is read from, then the leg's own line."""
    c = leg["config"]
    return {"metric": leg["metric"], "value": leg["value"], "unit": leg["unit"], "scaling": "strong", "n_gpus": leg["n_gpus"],
            "steps": leg["steps"], "warmup": leg["warmup"], "ms_per_step": leg["ms_per_step"],
            "exchange_ms_per_step": c["exchange_ms_per_step"], "per_rank_kernel_ms": c["kernel_ms_per_step_by_rank"],
            "max_rank_kernel_ms": max(c["kernel_ms_per_step_by_rank"]), "rccl_ranks": c["rccl_ranks"],
            "gather_path": c["gather_path"], "workload": c["workload"], "pairs_total": c["pairs_total"],
            "gpu_vs_oracle_mismatching_pairs": c.get("gpu_vs_oracle_mismatching_pairs"),
            "pairs_checked": c.get("pairs_checked"),
            "scan_frac_of_int8_peak": (leg["roofline"] or {}).get("frac") if leg.get("roofline") else None,
            "dry_run": bool(leg.get("dry_run")), "line": leg}


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
    ap.add_argument("--verify-scenes", type=int, default=4096, help="distinct seeded scenes in the verification leg")
    ap.add_argument("--no-cross-check", action="store_true",
                    help="diagnostic: one-way matching only (NOT the BASELINE workload)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: run the multi-GPU exchange path (process group + all-gather) even with 1 rank")
    ap.add_argument("--config", type=int, default=None, choices=[1, 3, 4],
                    help="1: BASELINE configs[1] (+ verify / pipeline / dense legs; the default at N=1); "
                         "3: configs[3], 2000 x 8192 fixed pair set sharded over the ranks (strong scaling; the default "
                         "at N>1); 4: configs[4], 10000 x 4096 sequential + loop matching, sharded")
    ap.add_argument("--weak", action="store_true", help="(accepted for older command lines: the headline at N>1 IS the weak-scaled configs[1])")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the exchange step (nccl = RCCL; gloo only with --cpu-dry-run)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU: run the multi-rank plumbing (sharding, exchange, reductions, JSON line) over gloo with the "
                         "CPU oracle in place of the kernels; prints value = null.  For tests/, with tiny --images/--feats")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the chained configs[2] leg")
    ap.add_argument("--no-config4", action="store_true", help="skip the one-step BASELINE configs[4] leg of the N=1 line")
    ap.add_argument("--full-line", action="store_true",
                    help="print the detailed line (every leg's workload text, notes and sub-objects: ~15 KB) instead of the "
                         "compact one; --detail-json PATH writes it to a file beside the compact line")
    ap.add_argument("--detail-json", default="", help="also write the detailed line to this file")
    ap.add_argument("--pipeline-steps", type=int, default=0, help="timed steps of the chained configs[2] leg (default: min(steps, 5))")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-overlap match leg")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged-size match leg (n ~ U[2000, 6000])")
    ap.add_argument("--no-sift-stats", action="store_true", help="skip the extractor-statistics match leg")
    ap.add_argument("--no-db", action="store_true", help="skip the database leg (pycolmap.match_exhaustive wall time)")
    ap.add_argument("--no-config3", action="store_true", help="skip the configs[3] strong-scaling leg of the default line")
    ap.add_argument("--config3-steps", type=int, default=0,
                    help="timed steps of the configs[3] leg (0 = one at N=1, where a step is ~11 s, two at N>1)")
    args = ap.parse_args()
    if args.config is None:
        args.config = 1
    if args.weak and args.config != 1:
        raise SystemExit("--weak is the N>1 variant of --config 1")
    if args.cpu_dry_run:
        args.backend = "gloo"
    elif args.backend != "nccl":
        raise SystemExit("--backend gloo is for --cpu-dry-run: the measured path exchanges over RCCL")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.config in (3, 4):
        return run_config34(args)

    import torch
    import torch.distributed as dist
    from pycolmap_amd import _capi, synth

    state = dist_setup(args)
    world, rank, local_rank, device, use_dist = state
    dry = args.cpu_dry_run
    quiet_gc()

    # ---- workload ------------------------------------------------------------------------
    num_images = args.images if world == 1 else int(round(args.images * math.sqrt(world)))
    arena = make_arena_torch(num_images, args.feats, seed=0, device=device)
    s1_all, s2_all = synth.exhaustive_pairs(num_images)
    # shard: order by image 2 (the kernel's L2-friendly order), deal contiguous slices to ranks
    from pycolmap_amd import distributed as D
    s1, s2, mine = D.shard_pairs(s1_all, s2_all, rank, world, rows=np.full(num_images, args.feats))

    if dry:
        ctx = _DryRunContext()
        ctx.reserve_slots(num_images)
        for i in range(num_images):
            ctx.upload_descriptors(i, arena[i].numpy())
    else:
        ctx = _capi.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.reserve_slots(num_images)
        for i in range(num_images):
            ctx.upload_descriptors_device(i, arena[i].data_ptr(), args.feats)
    device_sync(args)
    # the exchange step of the sharded headline: the library's own entry point (amc_allgather_match_tables) on the GPUs
    comm = D.make_comm(ctx) if (use_dist and not dry) else None

    def step():
        off, m, st = ctx.match_pairs(s1, s2, kernel=args.kernel, cross_check=not args.no_cross_check, copy=False)
        gathered = None
        if use_dist:
            # the exchange step: all-gather of the match tables (sizes, per-pair records, rows from the device memory
            # the kernels wrote); afterwards every rank holds the whole match graph (rank 0 would feed the SQLite writer)
            gen = None if dry else ctx.resident_generation
            gathered = D.all_gather_match_tables(mine, off, m, device=device, as_numpy=False,
                                                 device_matches=None if dry else True, comm=comm)
            device_sync(args)
            assert dry or ctx.resident_view_valid(gen)
        return off, m, st, gathered

    def fence():
        device_sync(args)
        if use_dist:
            dist.barrier()
            device_sync(args)

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
    headline_gather_path = D.last_gather_path() if use_dist else None

    # ---- BASELINE configs[3], fixed size, sharded: the strong-scaling leg of every line (collective) -----------------
    c3 = None
    if not args.no_config3:
        steps3 = args.config3_steps if args.config3_steps > 0 else (1 if world == 1 else 2)
        c3 = config34_leg(args, 3, steps3, 0 if world == 1 else 1, state)

    c4 = None
    if world == 1 and not dry and not args.no_config4 and args.images == 500 and args.feats == 4096:
        # BASELINE configs[4] (10,000 x 4096, sequential + loop matching) at N = 1, one step: what the driver's default
        # line would otherwise never show (`--config 4` alone is the full line)
        c4 = config34_leg(args, 4, 1, 1, state)   # (one warm-up step: the first call of a context allocates its tables)

    out = None
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
            "value": None if dry else value,
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
                            f"ratio test + cross-check (BASELINE.json configs[1] at N=1"
                            + ("" if world == 1 else f"; weak scaling: 500 x sqrt({world}) images, ~124,750 pairs per rank") + ")",
                "pairs_total": int(len(s1_all)),
                "pairs_per_rank": int(len(s1)),
                "distances_total": ndist_total,
                "kernel": args.kernel,
                "pairs_mfma": st["pairs_mfma"],
                "pairs_dot4": st["pairs_dot4"],
                "matches_rank0": int(m.shape[0]),
                "gather_path": headline_gather_path,
                "sharding": "pairs sorted by image 2, contiguous slice per rank, arena replicated"
                            + ("; all-gather of the match tables per step (amc_allgather_match_tables over RCCL)" if world > 1 else ""),
            },
            "roofline": None if dry else {
                "bound": "mfma",
                "achieved": achieved / 1e12,
                "peak": INT8_DENSE_PEAK_OPS / 1e12,
                "unit": "TOP/s (int8; 256 ops per descriptor-pair distance)",
                "frac": achieved / INT8_DENSE_PEAK_OPS,
                "frac_of_measured_i8_ceiling": achieved / INT8_MEASURED_CEILING_OPS,
                "measured_i8_ceiling": INT8_MEASURED_CEILING_OPS / 1e12,
                "whole_step_frac": value * OPS_PER_DISTANCE / INT8_DENSE_PEAK_OPS / max(world, 1),
                "traffic": None,  # filled below from the committed PMC pass when the launch shape is the same
                "kernel": "match_mfma_kernel" if st["pairs_mfma"] else "match_dot4_kernel",
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "launches_per_step": launches_per_step,
            },
        }
        if dry:
            out["dry_run"] = True
            out["metric"] = "DRY RUN (CPU oracle in place of the kernels, gloo in place of RCCL): NOT a measurement - " + out["metric"]
            out["data"] = "synthetic (cpu dry run)"
        if c3 is not None:
            out["config3"] = config3_summary(c3)
            # (the same scalars at the top level: a reader that keeps only one level of the line still has the curve)
            out["config3_value"] = c3["value"]
            out["config3_ms_per_step"] = c3["ms_per_step"]
            out["config3_exchange_ms_per_step"] = c3["config"]["exchange_ms_per_step"]
        if c4 is not None:
            out["config4"] = config3_summary(c4)
            out["config4"]["parts"] = c4["config"].get("parts")
            out["config4_value"] = c4["value"]
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be collected inside this process, so the
        # number comes from the committed rocprofv3 --pmc pass of this same command (profiles/*/pmc_hbm_*.json) - and
        # only while (i) this run launches the same shape and (ii) the kernel's source still hashes to what it was when
        # the counters were taken (roofline.traffic_source_sha); otherwise the field stays null.
        try:
            if dry:
                raise OSError("dry run")
            import hashlib
            pmc_path = sorted((ROOT / "profiles").glob("r*/pmc_hbm_r*.json"))[-1]
            pmc = json.loads(pmc_path.read_text())
            per_launch = int(len(s1)) // launches_per_step
            shas = pmc.get("kernel_source_sha256", {})
            same_source = bool(shas) and all(hashlib.sha256((ROOT / f).read_bytes()).hexdigest() == h for f, h in shas.items())
            out["roofline"]["traffic_source"] = str(pmc_path.relative_to(ROOT))
            out["roofline"]["traffic_source_sha"] = shas
            out["roofline"]["traffic_source_sha16"] = ",".join(h[:16] for h in shas.values())
            out["roofline"]["traffic_source_current"] = same_source
            out["roofline"]["traffic_note"] = ("HBM counters cannot be read inside this process: traffic is the committed rocprofv3 "
                                               "--pmc pass of this same command (traffic_source), per launch; null unless the "
                                               "kernel source still hashes to what it was when the pass was taken")
            if same_source and st["pairs_mfma"] and args.feats == 4096 and abs(per_launch - pmc["pairs_per_launch"]) <= 1:
                out["roofline"]["traffic"] = pmc["fetch_bytes_per_launch_corrected"]
                out["roofline"]["traffic_unit"] = ("bytes read from HBM per launch (FETCH_SIZE x 1024 x 2, gfx950 correction), from the "
                                                   "committed PMC pass named in traffic_source - not counted in this run")
                out["roofline"]["traffic_written"] = pmc.get("write_bytes_per_launch")
                out["roofline"]["algorithmic_bytes"] = float(per_launch) * 2 * args.feats * 128
        except (OSError, KeyError, ValueError, IndexError):
            pass
        if world == 1 and not args.no_cpu_baseline and not dry:
            cores = host_cores()
            arena_cpu = arena.cpu().numpy()
            sample = args.cpu_sample_pairs if args.cpu_sample_pairs > 0 else 8 * cores
            v, npairs, dt, (idx, coff, cm), cpu_opt, cpu_default = cpu_baseline(arena_cpu, s1, s2, sample, cores)
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
                # the two other CPU rates as scalars (the objects follow): the AVX-512 VNNI variant of the same matcher -
                # what the host's cores can really do - and COLMAP's DEFAULT CPU matcher (approximate k-d forest, restated)
                "optimised_value": cpu_opt["value"] if cpu_opt else None,
                "optimised_identical_to_port": cpu_opt["identical_to_port"] if cpu_opt else None,
                "default_cpu_matcher_value": cpu_default["value"],
                "default_cpu_matcher_pairs_per_s": cpu_default["pairs_per_s"],
                "default_cpu_matcher_port_matches_found": cpu_default["port_matches_found"],
                "default_cpu_matcher": cpu_default,   # the k-d forest COLMAP's CPU path uses by default (approximate)
                "optimised": cpu_opt,   # None on a host without AVX-512 VNNI
            }
        gpu_legs = world == 1 and not dry
        if args.verify_pairs > 0 and gpu_legs:
            out["verify"] = verify_leg(lambda: _capi.Context(local_rank), local_rank, args.verify_pairs,
                                       max(1, args.steps), min(1, args.warmup),
                                       0 if args.no_cpu_baseline else 256, distinct=args.verify_scenes)
            vr = out["verify"]["roofline"]
            out["verify"].update(roofline_frac=vr["frac"], roofline_achieved_tflops=vr["achieved"],
                                 executed_note="verify.roofline.executed comes from the committed SQ counter pass named in its "
                                               "`source` (instructions per pair), divided by this run's kernel time")
        def release_headline():   # the legs below bring their own contexts and arenas
            nonlocal arena, ctx
            if ctx is not None:
                ctx.close()
                ctx = arena = None
                torch.cuda.empty_cache()
        if gpu_legs and not args.no_pipeline:
            release_headline()
            # (five steps: with two, one 20 ms host-side stall - seen once in this round's six runs of the line - is 10 ms of the figure)
            out["pipeline"] = pipeline_leg(lambda: _capi.Context(local_rank), args.pipeline_steps or max(1, min(args.steps, 5)), min(2, args.warmup),
                                           0 if args.no_cpu_baseline else 4 * host_cores(), args.images, args.feats)
            sm = out["pipeline"]["stage_ms_per_step"]
            out["pipeline"].update(verify_ms=sm["verify_ms"], verify_kernel_ms=sm["verify_kernel_ms"], match_ms=sm["match_ms"],
                                   scan_ms=sm["scan_ms"])
        if gpu_legs and not args.no_ragged:
            release_headline()
            # (five steps: with two, one slow host-side moment - 16 ms once in this round's runs - moves vs_uniform by 8 %;
            #  two warm-up calls: that moment was the first timed call after ONE warm-up in profiles/r06's v8 and v11 lines)
            out["ragged"] = ragged_leg(lambda: _capi.Context(local_rank), device, max(1, min(args.steps, 5)),
                                       min(2, args.warmup), args.images, 2000, 6000, args.kernel, value,
                                       check=not args.no_cpu_baseline)
        if gpu_legs and not args.no_sift_stats:
            release_headline()
            out["sift_stats"] = sift_stats_leg(lambda: _capi.Context(local_rank), device, max(1, min(args.steps, 5)),
                                               min(2, args.warmup), args.images, args.feats, args.kernel, value,
                                               check=not args.no_cpu_baseline)
        if gpu_legs and not args.no_dense:
            release_headline()
            out["dense"] = dense_leg(lambda: _capi.Context(local_rank), device, max(1, min(args.steps, 3)),
                                     min(2, args.warmup), args.images, args.feats, args.kernel, check=not args.no_cpu_baseline)
        if gpu_legs and not args.no_db:
            release_headline()
            try:
                out["db"] = db_leg(args.images, args.feats)
            except Exception as e:  # the API leg must not take the kernel legs' numbers down with it
                out["db"] = {"error": f"{type(e).__name__}: {e}"}
    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    print_line(out, args)


if __name__ == "__main__":
    main()
