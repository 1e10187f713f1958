"""world_size-2 gloo tests of the N>1 path: pair sharding + the all-gather of match tables
(pycolmap_amd/distributed.py).  The per-rank compute is the CPU oracle here (no GPU in this
container); on the GPU box bench.py runs the same functions over RCCL with libamc.so."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    import oracle_lib
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)            # every rank builds the same replicated arena
    imgs = synth.scene_images(rng, 7, 96, num_landmarks=160, visible_frac=0.5)
    imgs.append(np.zeros((0, 128), np.uint8))
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    a, b, mine = D.shard_pairs(s1, s2, rank, world)
    off, m = oracle_lib.match_pairs(imgs, a, b, threads=1)
    g_off, g_m = D.all_gather_match_tables(mine, off, m)
    # the same exchange fed from a resident table (a tensor where the match kernels would have left it: here a CPU
    # int32 view of the rows) with only rank 0 downloading the result - the form bench.py / the SQLite writer rank use
    import torch
    resident = torch.from_numpy(np.ascontiguousarray(m, dtype=np.uint32).view(np.int32).reshape(-1, 2).copy())
    d_off, d_m = D.all_gather_match_tables(mine, off, None, device_matches=resident, download_rank=0)
    if rank == 0:
        assert np.array_equal(d_off, g_off) and np.array_equal(d_m, g_m)
    else:
        assert d_off is None and d_m is None
    np.savez(Path(out_dir) / f"rank{rank}.npz", g_off=g_off, g_m=g_m, mine=mine)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_match_tables_equal_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    from pycolmap_amd import synth
    rng = np.random.default_rng(0)
    imgs = synth.scene_images(rng, 7, 96, num_landmarks=160, visible_frac=0.5)
    imgs.append(np.zeros((0, 128), np.uint8))
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    want_off, want_m = oracle_lib.match_pairs(imgs, s1, s2, threads=2)
    assert want_off[-1] > 50
    covered = []
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(z["g_off"], want_off)     # every rank holds the whole graph
        np.testing.assert_array_equal(z["g_m"], want_m)
        covered.append(z["mine"])
    allidx = np.concatenate(covered)
    assert sorted(allidx.tolist()) == list(range(len(s1)))       # shards partition the pair list


def test_shard_pairs_groups_by_streamed_image():
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    s1, s2 = synth.exhaustive_pairs(40)
    parts = [D.shard_pairs(s1, s2, r, 4) for r in range(4)]
    sizes = [len(p[0]) for p in parts]
    assert sum(sizes) == len(s1) and max(sizes) - min(sizes) <= 1 + (len(s1) % 4 != 0) * 4
    for a, b, idx in parts:
        assert np.all(np.diff(b.astype(np.int64)) >= 0)          # sorted by image 2 within a shard
        np.testing.assert_array_equal(s1[idx], a)
    assert D.shard_pairs(s1[:0], s2[:0], 0, 2)[0].size == 0


# ---- the verify half: geometries + inlier matches of sharded pairs ---------------------------------
TVG_DT = np.dtype([("config", np.int32), ("num_inliers", np.int32), ("E", np.float64, (3, 3)),
                   ("F", np.float64, (3, 3)), ("H", np.float64, (3, 3))])


def _verify_scenes():
    from pycolmap_amd import synth
    rng = np.random.default_rng(5)
    return [synth.two_view_scene(rng, num_inliers=int(rng.integers(0, 120)), num_outliers=int(rng.integers(5, 60)),
                                 planar=bool(k % 3 == 0)) for k in range(11)]


def _oracle_verify(scenes, which):
    import oracle_lib as o
    cam = o.make_camera(prior=True)
    tvg = np.zeros(len(which), dtype=TVG_DT)
    masks, mm, off = [], [], [0]
    for i, k in enumerate(which):
        sc = scenes[int(k)]
        r = o.estimate_two_view_geometry(cam, sc["pts1"], cam, sc["pts2"], sc["matches"])
        tvg[i] = (r["config"], r["num_inliers"], r["E"], r["F"], r["H"])
        masks.append(r["inlier_mask"])
        mm.append(sc["matches"])
        off.append(off[-1] + len(sc["matches"]))
    cat = np.concatenate(mm) if mm else np.zeros((0, 2), np.uint32)
    return tvg, np.array(off), cat, (np.concatenate(masks) if masks else np.zeros(0, bool))


def _verify_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    from pycolmap_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scenes = _verify_scenes()
    n = len(scenes)
    per = (n + world - 1) // world
    mine = np.arange(n)[rank * per:(rank + 1) * per]
    tvg, off, m, mask = _oracle_verify(scenes, mine)
    g = D.all_gather_verification(mine, tvg, off, m, mask, n)
    np.savez(Path(out_dir) / f"v{rank}.npz", tvg=g[0], off=g[1], m=g[2], ioff=g[3], im=g[4])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_verification_equals_single_process(tmp_path, world):
    """Geometries, matches and inlier matches of pairs verified on different ranks, gathered: every rank ends up with
    what one process computes for all pairs (ragged shards; with 4 ranks and 11 pairs the last shard is short)."""
    import torch.multiprocessing as mp
    mp.spawn(_verify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT / "tests"))
    scenes = _verify_scenes()
    tvg, off, m, mask = _oracle_verify(scenes, np.arange(len(scenes)))
    assert len(set(tvg["config"].tolist())) >= 3
    inl_counts = np.array([mask[off[k]:off[k + 1]].sum() for k in range(len(scenes))])
    for r in range(world):
        z = np.load(tmp_path / f"v{r}.npz")
        assert z["tvg"].tobytes() == tvg.tobytes()
        np.testing.assert_array_equal(z["off"], off)
        np.testing.assert_array_equal(z["m"], m)
        np.testing.assert_array_equal(np.diff(z["ioff"]), inl_counts)
        np.testing.assert_array_equal(z["im"], m[mask])


def test_shard_pairs_balances_work_for_ragged_images():
    """shards are cut at equal sum(n1 * n2), not at equal pair counts."""
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    rng = np.random.default_rng(3)
    rows = rng.integers(50, 4000, size=60)
    rows[:5] = 0                                                   # images without descriptors cost nothing
    s1, s2 = synth.exhaustive_pairs(60)
    for world in (2, 3, 8):
        parts = [D.shard_pairs(s1, s2, r, world, rows=rows) for r in range(world)]
        idx = np.concatenate([p[2] for p in parts])
        assert sorted(idx.tolist()) == list(range(len(s1)))
        work = np.array([(rows[a].astype(np.int64) * rows[b]).sum() for a, b, _ in parts], dtype=np.float64)
        assert work.max() <= 1.15 * work.mean(), (world, work)
        by_count = np.array([(rows[a].astype(np.int64) * rows[b]).sum() for a, b, _ in
                             [D.shard_pairs(s1, s2, r, world) for r in range(world)]], dtype=np.float64)
        assert work.max() <= by_count.max()                        # never worse than dealing equal counts
        for a, b, _ in parts:
            assert np.all(np.diff(b.astype(np.int64)) >= 0)


def _worker_config4(rank, world, port, out_dir):
    """bench.py --config 4's exchange with the CPU oracle as the matcher: sequential pairs sharded by work, every
    rank's own loop-closure pairs (queries dealt round-robin) appended in rank order."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    import oracle_lib
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    imgs = synth.scene_images(rng, 12, 64, num_landmarks=120, visible_frac=0.5)
    n = len(imgs)
    seq = [(i, j) for i in range(n) for j in (i + 1, i + 2, i + 4) if j < n]
    a_all = np.array([p[0] for p in seq], np.uint32)
    b_all = np.array([p[1] for p in seq], np.uint32)
    s1, s2, mine = D.shard_pairs(a_all, b_all, rank, world, rows=np.full(n, 64))
    off, m = oracle_lib.match_pairs(imgs, s1, s2, threads=1)
    g_off, g_m = D.all_gather_match_tables(mine, off, m)
    queries = np.arange(0, n, 3)[rank::world]
    l1 = np.repeat(queries.astype(np.uint32), 2)
    l2 = ((l1 + np.tile([5, 7], len(queries))) % n).astype(np.uint32)
    loff, lm = oracle_lib.match_pairs(imgs, l1, l2, threads=1) if len(l1) else (np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32))
    a_off, a_m, base = D.all_gather_appended_tables(loff, lm)
    np.savez(Path(out_dir) / f"c4_rank{rank}.npz", g_off=g_off, g_m=g_m, a_off=a_off, a_m=a_m, base=base, l1=l1, l2=l2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config4_exchange_dry_run(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_config4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    from pycolmap_amd import synth
    rng = np.random.default_rng(1)
    imgs = synth.scene_images(rng, 12, 64, num_landmarks=120, visible_frac=0.5)
    n = len(imgs)
    seq = [(i, j) for i in range(n) for j in (i + 1, i + 2, i + 4) if j < n]
    want_off, want_m = oracle_lib.match_pairs(imgs, np.array([p[0] for p in seq], np.uint32),
                                              np.array([p[1] for p in seq], np.uint32), threads=2)
    z = [np.load(tmp_path / f"c4_rank{r}.npz") for r in range(world)]
    l1 = np.concatenate([x["l1"] for x in z])                      # rank order = the order of the appended table
    l2 = np.concatenate([x["l2"] for x in z])
    lw_off, lw_m = oracle_lib.match_pairs(imgs, l1, l2, threads=2)
    assert [int(x["base"]) for x in z] == np.cumsum([0] + [len(x["l1"]) for x in z[:-1]]).tolist()
    for x in z:
        np.testing.assert_array_equal(x["g_off"], want_off)
        np.testing.assert_array_equal(x["g_m"], want_m)
        np.testing.assert_array_equal(x["a_off"], lw_off)
        np.testing.assert_array_equal(x["a_m"], lw_m)


# ---- configs[2] sharded: shards cut by a matches-weighted cost, world 8 ------------------------------------------
def test_shard_pairs_by_cost_balances_verification_work():
    """A sharded match + VERIFY run: verification time follows the matches (a few overlapping pairs carry all of it), not
    n1 * n2.  Cuts at equal cumulative cost keep every rank's cost near the mean where equal-n1*n2 cuts leave one rank
    with most of the verification; the shards stay contiguous in the (image 2, image 1) order and partition the list."""
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    rng = np.random.default_rng(21)
    n = 64
    s1, s2 = synth.exhaustive_pairs(n)
    rows = np.full(n, 4096)
    matches = np.zeros(len(s1))
    near = np.abs(s1.astype(np.int64) - s2.astype(np.int64)) <= 2          # an orbit: only neighbours overlap ...
    matches[near] = rng.integers(200, 900, size=int(near.sum()))
    matches[(s2 > 48) & near] *= 6                                          # ... and one stretch of it much more
    scan, verify = 4096.0 * 4096.0 * 8.5e-14, matches * 4.0e-6              # seconds per pair: scan rate, per-match verify cost
    cost = scan + verify
    for world in (2, 4, 8):
        parts = [D.shard_pairs_by_cost(s1, s2, cost, r, world) for r in range(world)]
        idx = np.concatenate([p[2] for p in parts])
        assert sorted(idx.tolist()) == list(range(len(s1)))
        per_rank = np.array([cost[p[2]].sum() for p in parts])
        assert per_rank.max() <= 1.15 * per_rank.mean(), (world, per_rank)   # (one heavy pair is a third of a rank's share)
        by_rows = np.array([cost[D.shard_pairs(s1, s2, r, world, rows=rows)[2]].sum() for r in range(world)])
        assert per_rank.max() <= by_rows.max() + 1e-12
        if world == 8:
            assert by_rows.max() > 1.3 * by_rows.mean()                     # what the n1 * n2 cut would have cost
        for a, b, _ in parts:
            assert np.all(np.diff(b.astype(np.int64)) >= 0)
    assert D.shard_pairs_by_cost(s1[:0], s2[:0], cost[:0], 0, 2)[0].size == 0
    with pytest.raises(ValueError):
        D.shard_pairs_by_cost(s1, s2, -cost, 0, 2)
    # all-zero costs fall back to equal counts
    assert sum(len(D.shard_pairs_by_cost(s1, s2, np.zeros(len(s1)), r, 3)[2]) for r in range(3)) == len(s1)


def _config2_worker(rank, world, port, out_dir):
    """BASELINE configs[2] sharded, as a dry run: every rank matches + verifies its shard with the CPU oracles; the shards
    are cut by cost (matches-weighted, from a cheap first pass: here the true counts), and the exchange step hands every
    rank the whole result."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    import oracle_lib as o
    from pycolmap_amd import distributed as D
    from pycolmap_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    images = synth.multiview_scene(np.random.default_rng(17), num_images=7, n_feats=160, num_landmarks=260)
    s1, s2 = synth.exhaustive_pairs(len(images))
    descs = [im["descriptors"] for im in images]
    counts = np.load(Path(out_dir) / "counts.npy")
    a, b, mine = D.shard_pairs_by_cost(s1, s2, 160.0 * 160.0 * 1e-3 + counts, rank, world)
    off, m = o.match_pairs(descs, a, b, threads=1) if len(a) else (np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32))
    cam = o.make_camera("PINHOLE", images[0]["width"], images[0]["height"], images[0]["params"], prior=True)
    tvg = np.zeros(len(a), dtype=TVG_DT)
    mask = np.zeros(len(m), dtype=np.uint8)
    for k in range(len(a)):
        mm = m[int(off[k]):int(off[k + 1])]
        r = o.estimate_two_view_geometry(cam, images[int(a[k])]["keypoints"][:, :2].astype(np.float64), cam,
                                         images[int(b[k])]["keypoints"][:, :2].astype(np.float64), mm)
        tvg[k] = (r["config"], r["num_inliers"], r["E"], r["F"], r["H"])
        mask[int(off[k]):int(off[k + 1])] = r["inlier_mask"]
    g = D.all_gather_verification(mine, tvg, off, m, mask, len(s1))
    np.savez(Path(out_dir) / f"c2_{rank}.npz", tvg=g[0], off=g[1], m=g[2], ioff=g[3], im=g[4], n=len(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_config2_sharded_by_cost_dry_run_world8(tmp_path):
    """8 ranks (the node north_star names), 21 pairs of one scene, shards cut by matches-weighted cost: some ranks own
    one pair, every rank ends with the geometries, matches and inlier matches of ALL pairs = the single-process result."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as o
    from pycolmap_amd import synth
    images = synth.multiview_scene(np.random.default_rng(17), num_images=7, n_feats=160, num_landmarks=260)
    s1, s2 = synth.exhaustive_pairs(len(images))
    woff, wm = o.match_pairs([im["descriptors"] for im in images], s1, s2, threads=4)
    counts = np.diff(woff.astype(np.int64)).astype(np.float64)
    np.save(tmp_path / "counts.npy", counts)
    world = 8
    mp.spawn(_config2_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cam = o.make_camera("PINHOLE", images[0]["width"], images[0]["height"], images[0]["params"], prior=True)
    z = [np.load(tmp_path / f"c2_{r}.npz") for r in range(world)]
    assert sum(int(x["n"]) for x in z) == len(s1)
    cfgs = []
    for p in range(len(s1)):
        mm = wm[int(woff[p]):int(woff[p + 1])]
        r = o.estimate_two_view_geometry(cam, images[int(s1[p])]["keypoints"][:, :2].astype(np.float64), cam,
                                         images[int(s2[p])]["keypoints"][:, :2].astype(np.float64), mm)
        cfgs.append(r["config"])
        for x in z:
            assert x["tvg"]["config"][p] == r["config"] and x["tvg"]["num_inliers"][p] == r["num_inliers"]
            assert np.array_equal(x["tvg"]["F"][p].view(np.uint64), np.asarray(r["F"]).view(np.uint64))
            np.testing.assert_array_equal(x["im"][int(x["ioff"][p]):int(x["ioff"][p + 1])], mm[np.asarray(r["inlier_mask"], bool)])
    assert sum(c > 1 for c in cfgs) >= 3
    for x in z:
        np.testing.assert_array_equal(x["off"], woff)
        np.testing.assert_array_equal(x["m"], wm)
