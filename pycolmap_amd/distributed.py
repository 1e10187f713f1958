"""Multi-GPU sharding of the match + verify path.

Image pairs are independent units, so the path shards embarrassingly: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests), the
descriptor arena replicated on every GPU, the pair list dealt in contiguous slices, and ONE exchange
step at the end: an all-gather of the per-rank match tables (sizes first, then the per-pair records, then the rows,
fed from the device memory the match kernels wrote) so that every rank — in particular the rank that owns the SQLite
writer — holds the whole match graph.

On the GPUs the exchange is the library's own: `amc_allgather_match_tables` behind the C ABI (include/amc.h; RCCL
called from C++, grouped ncclSend / ncclRecv of exactly each rank's rows straight from the resident match table) -
`make_comm()` builds its communicator, `all_gather_match_tables(..., comm=comm)` calls it.  Without a `comm` the same
three-step protocol runs as torch.distributed collectives: that is what the gloo tests on CPU exercise (and what a
host without the library's communicator would use on RCCL).
"""
from __future__ import annotations

import numpy as np


def shard_pairs(slot1: np.ndarray, slot2: np.ndarray, rank: int, world: int, rows: np.ndarray | None = None):
    """Contiguous slice of the pair list, after ordering by image 2 (then image 1) so that pairs
    sharing the streamed image stay on one GPU and in one L2.  With `rows` (descriptors per slot) the cuts
    sit at equal shares of the WORK, sum of n1 * n2 (ragged image sizes would otherwise
    leave ranks with very different scan times); without it, at equal pair counts.  Returns (slot1, slot2, index)
    where `index` maps the shard back to positions in the caller's list."""
    s1 = np.asarray(slot1, dtype=np.uint32)
    s2 = np.asarray(slot2, dtype=np.uint32)
    order = np.lexsort((s1, s2))
    if rows is None or len(order) == 0:
        per = (len(order) + world - 1) // world
        mine = order[rank * per:(rank + 1) * per]
        return s1[mine], s2[mine], mine
    r = np.asarray(rows, dtype=np.float64)
    work = np.cumsum(r[s1[order]] * r[s2[order]])
    total = work[-1]
    if total <= 0:                                       # nothing to weigh: fall back to counts
        return shard_pairs(s1, s2, rank, world)
    # pair k goes to the rank whose share of the cumulative work its midpoint falls into
    mid = work - 0.5 * r[s1[order]] * r[s2[order]]
    owner = np.minimum((mid * world / total).astype(np.int64), world - 1)
    lo, hi = np.searchsorted(owner, [rank, rank + 1])
    mine = order[lo:hi]
    return s1[mine], s2[mine], mine


_last_gather_path = None   # "c-abi" | "uneven" | "padded": which path the last exchange took (tests and bench.py report it)
_last_gather_stats = None  # the C ABI's own phase timings of the last exchange through it


def last_gather_path():
    return _last_gather_path


def last_gather_stats():
    return _last_gather_stats


def make_comm(ctx, group=None):
    """The library's RCCL communicator for the ranks of `group` (amc_comm_create): rank 0 draws the unique id
    (ncclGetUniqueId), torch.distributed carries its 128 bytes to the others, every rank joins with its own context.
    Collective.  Works with one rank (no process group needed)."""
    from . import _capi
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return ctx.comm_create(1, 0, _capi.comm_unique_id())
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [_capi.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return ctx.comm_create(world, rank, box[0])


def _gather_rows(t, sizes, group=None):
    """All-gather of per-rank tensors with DIFFERENT numbers of rows: returns the list of every rank's rows.
    RCCL (backend "nccl"): ONE uneven all_gather - each rank sends exactly its rows, nothing is padded; a rank with no
    rows sends a single dummy row (dropped on arrival), so that the same call serves every distribution of the rows
    and no zero-byte collective is ever issued.  gloo (CPU tests) only gathers equal sizes: pad to the largest rank
    there."""
    global _last_gather_path
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [int(x) for x in sizes]
    t = t.contiguous()
    tail = tuple(t.shape[1:])
    if dist.get_backend(group) == "nccl":
        _last_gather_path = "uneven"
        send = t if t.shape[0] else torch.zeros((1,) + tail, dtype=t.dtype, device=t.device)
        outs = [torch.empty((max(n, 1),) + tail, dtype=t.dtype, device=t.device) for n in sizes]
        dist.all_gather(outs, send, group=group)
        return [o[:n] for o, n in zip(outs, sizes)]
    _last_gather_path = "padded"
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx,) + tail, dtype=t.dtype, device=t.device)
    if t.shape[0]:
        pad[:t.shape[0]] = t
    allp = torch.empty((world * mx,) + tail, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(allp, pad, group=group)
    return [allp[r * mx:r * mx + sizes[r]] for r in range(world)]


def all_gather_match_tables(pair_index: np.ndarray, offsets: np.ndarray, matches: np.ndarray, device=None,
                            group=None, as_numpy: bool = True, device_matches=None, download_rank=None, comm=None):
    """Exchange per-rank CSR match tables.  Every rank passes the global indices of its pairs, its
    CSR offsets and its (M, 2) uint32 matches; every rank gets back the table for ALL pairs as
    (global_offsets, global_matches) in the global pair order.

    `device_matches`: the rank's match rows as an int32 [M, 2] tensor already on `device` - the table where the match
    kernels left it (Context.resident_matches_tensor(): no host round trip, no copy); `matches` is then not read.
    Collectives: an all-gather of the (npairs, nmatches) sizes, then two uneven all-gathers (per-pair (index, count)
    and the match rows, 32-bit payloads, each rank sends exactly what it has) - few and large, which is what xGMI's
    point-to-point links want.  The reassembly into the global CSR is a handful of tensor ops on `device` (scatter of
    the counts, one cumulative sum, one indexed copy of the rows): no per-pair host work.
    as_numpy=False: device tensors are returned.  download_rank=r: only rank r copies the result to the host (the
    rank that owns the SQLite writer), the others return (None, None).

    comm: the library's communicator (make_comm): the exchange runs behind the C ABI (amc_allgather_match_tables) -
    with `device_matches` given the rows come from the context's resident table, otherwise from `matches`; returns
    (global offsets, global matches or None) as numpy arrays (as_numpy=False: (global offsets, None) - the table
    stays in the library's device memory, last_gather_stats()["device_ptr"])."""
    global _last_gather_path, _last_gather_stats
    if comm is not None:
        want = as_numpy and (download_rank is None or comm.rank == download_rank)
        g_off, g_m, st = comm.allgather_match_tables(pair_index, offsets, None if device_matches is not None else matches,
                                                     download=want)
        _last_gather_path, _last_gather_stats = "c-abi", st
        if as_numpy and not want:
            return None, None
        return g_off, g_m
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    counts = np.diff(np.asarray(offsets, dtype=np.int64))
    npairs = len(pair_index)
    nm = int(device_matches.shape[0]) if device_matches is not None else int(matches.shape[0])
    if int(counts.sum()) != nm:
        raise ValueError("offsets and matches disagree")
    # The per-pair records travel as int32: a rank whose positions or counts do not fit says so IN the size exchange,
    # so that every rank raises together instead of one rank leaving the others inside the next collective.
    pair_index = np.asarray(pair_index, dtype=np.int64)
    too_big = bool(npairs and (int(pair_index.max()) >= 2 ** 31 or int(pair_index.min()) < 0 or int(counts.max()) >= 2 ** 31))
    sizes = torch.tensor([npairs, nm, int(too_big)], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(world, 3).cpu()
    if int(all_sizes[:, 2].sum()):
        raise ValueError("all_gather_match_tables: pair indices and per-pair match counts travel as int32 "
                         "(8 bytes per pair); an index or a count of 2^31 or more does not fit (rank(s) %s)"
                         % [r for r in range(world) if int(all_sizes[r, 2])])
    meta_h = np.empty((npairs, 2), dtype=np.int32)          # (global index, count): 8 bytes per pair, the only H2D
    meta_h[:, 0] = pair_index
    meta_h[:, 1] = counts
    meta = torch.from_numpy(meta_h).to(dev)
    if device_matches is not None:
        rows = device_matches.view(torch.int32).reshape(-1, 2)
    else:
        rows = torch.from_numpy(np.ascontiguousarray(matches, dtype=np.uint32).view(np.int32).reshape(-1, 2)).to(dev)
    all_meta = torch.cat(_gather_rows(meta, all_sizes[:, 0].tolist(), group), 0)
    rows_cat = torch.cat(_gather_rows(rows, all_sizes[:, 1].tolist(), group), 0)   # rank-major, like all_meta

    # ---- global CSR, on `dev` ----
    idx = all_meta[:, 0].to(torch.int64)
    cnt = all_meta[:, 1].to(torch.int64)
    total_pairs = int(idx.numel())
    g_counts = torch.zeros(total_pairs, dtype=torch.int64, device=dev)
    g_counts[idx] = cnt
    g_off = torch.zeros(total_pairs + 1, dtype=torch.int64, device=dev)
    g_off[1:] = torch.cumsum(g_counts, 0)
    src_start = torch.cumsum(cnt, 0) - cnt                        # where each pair's rows start in rows_cat
    shift = torch.repeat_interleave(g_off[idx] - src_start, cnt)  # per row: destination - source position
    g_matches = torch.empty(int(rows_cat.shape[0]), 2, dtype=torch.int32, device=dev)
    if rows_cat.shape[0]:
        g_matches[torch.arange(rows_cat.shape[0], device=dev) + shift] = rows_cat
    if not as_numpy:
        return g_off, g_matches
    if download_rank is not None and dist.get_rank(group) != download_rank:
        return None, None
    return g_off.cpu().numpy().astype(np.uint64), g_matches.cpu().numpy().view(np.uint32)


def all_gather_appended_tables(offsets: np.ndarray, matches: np.ndarray, device=None, group=None, as_numpy: bool = True,
                               device_matches=None, comm=None):
    """all_gather_match_tables for pair lists that have no global numbering yet - e.g. the loop-closure pairs every
    rank retrieves for its own query images (BASELINE configs[4]): the lists are appended in rank order.  One more
    tiny all-gather (the per-rank pair counts) gives every rank its base position; returns what
    all_gather_match_tables returns plus this rank's base."""
    global _last_gather_path, _last_gather_stats
    if comm is not None:
        # The C ABI appends the lists itself (pair_index = NULL).  This rank's base = the pairs of the ranks before it:
        # one 8-byte record per rank through amc_allgather_pair_records (the tiny count all-gather of the torch path).
        # as_numpy=False: the table stays in the library's device memory (last_gather_stats()["device_ptr"]) and the
        # second value is None - there is no torch tensor to return on this path.
        n = len(offsets) - 1
        counts, _ = comm.allgather_pair_records(np.array([comm.rank], dtype=np.uint64), np.array([n], dtype=np.uint64))
        base = int(counts[:comm.rank].sum())
        g_off, g_m, st = comm.allgather_match_tables(None, offsets, None if device_matches is not None else matches,
                                                     download=as_numpy)
        _last_gather_path, _last_gather_stats = "c-abi", st
        return g_off, g_m, base
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    n = len(offsets) - 1
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, cnt, group=group)
    base = int(allc[:rank].sum().item())
    g_off, g_m = all_gather_match_tables(base + np.arange(n, dtype=np.int64), offsets, matches, device=device, group=group,
                                         as_numpy=as_numpy, device_matches=device_matches)
    return g_off, g_m, base


def all_gather_pair_records(pair_index: np.ndarray, records: np.ndarray, total_pairs: int, device=None, group=None):
    """All-gather of one fixed-size record per pair (e.g. the TwoViewGeometry of a verified pair: config, E, F, H,
    pose).  `records` is a 1-D structured / plain array with one element per local pair, `pair_index` their global
    positions; returns the array of all `total_pairs` records in global order (pairs nobody owns stay zero).  One
    size exchange + one padded byte all-gather, reassembled with a single indexed copy on `device`."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    rec = np.ascontiguousarray(records)
    width = rec.dtype.itemsize
    n = len(pair_index)
    if len(rec) != n:
        raise ValueError("one record per local pair")
    sizes = torch.tensor([n], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    max_n = max(int(all_sizes.max()), 1)
    idx = torch.full((max_n,), -1, dtype=torch.int64, device=dev)
    body = torch.zeros(max_n, width, dtype=torch.uint8, device=dev)
    if n:
        idx[:n] = torch.from_numpy(np.asarray(pair_index, dtype=np.int64)).to(dev)
        body[:n] = torch.from_numpy(rec.view(np.uint8).reshape(n, width)).to(dev)
    all_idx = torch.empty(world * max_n, dtype=torch.int64, device=dev)
    all_body = torch.empty(world * max_n, width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(all_idx, idx, group=group)
    dist.all_gather_into_tensor(all_body, body, group=group)
    valid = all_idx >= 0
    out = torch.zeros(total_pairs, width, dtype=torch.uint8, device=dev)
    out[all_idx[valid]] = all_body[valid]
    return out.cpu().numpy().reshape(-1).view(rec.dtype).copy()


def all_gather_verification(pair_index: np.ndarray, tvg: np.ndarray, match_offsets: np.ndarray, matches: np.ndarray,
                            inlier_mask: np.ndarray, total_pairs: int, device=None, group=None, comm=None,
                            resident: bool = False, download_rank=None):
    """The exchange step of a sharded match + verify run: every rank passes the two-view geometries of its pairs (the
    structured array `Context.verify_pairs` returns), their matches (CSR) and inlier mask; every rank gets back, in the
    global pair order, (tvg records, match offsets, matches, inlier-match offsets, inlier matches) - what rank 0 needs
    to write COLMAP's `matches` and `two_view_geometries` tables.

    comm: the library's communicator (make_comm): all three exchanges run behind the C ABI - amc_allgather_pair_records
    for the records, amc_allgather_match_tables for the matches, amc_allgather_inlier_tables for the inlier lists.
    resident=True (the ranks ran Context.match_verify_pairs and nothing since): records, match rows and inlier rows are
    all taken where the kernels left them in device memory (the inlier lists compacted there from the masks), `tvg` /
    `matches` / `inlier_mask` are not read; otherwise they travel from the host arrays.  download_rank=r: only rank r
    copies the results to the host, the others get Nones for the row arrays."""
    global _last_gather_path, _last_gather_stats
    off = np.asarray(match_offsets, dtype=np.int64)
    if comm is not None:
        want = download_rank is None or comm.rank == download_rank
        idx = np.asarray(pair_index, dtype=np.uint64)
        if resident:
            g_tvg, st_r = comm.allgather_pair_records(idx, None, download=want)
            g_off, g_m, st_m = comm.allgather_match_tables(idx, off, None, download=want)
            g_ioff, g_im, st_i = comm.allgather_inlier_tables(idx, download=want)
        else:
            m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
            mask = np.asarray(inlier_mask).astype(np.uint8)
            keep = np.flatnonzero(mask)
            csum = np.zeros(len(mask) + 1, dtype=np.int64)
            csum[1:] = np.cumsum(mask != 0)
            # per pair by (mask byte, position): ExtractInlierMatches' order, the per-geometry lists of a MULTIPLE
            # geometry one after the other (all bytes are 1 otherwise and this is the position order)
            pair_of = np.repeat(np.arange(len(off) - 1), np.diff(off))[keep]
            inl = m[keep[np.lexsort((keep, mask[keep], pair_of))]]
            g_tvg, st_r = comm.allgather_pair_records(idx, tvg, download=want)
            g_off, g_m, st_m = comm.allgather_match_tables(idx, off, m, download=want)
            g_ioff, g_im, st_i = comm.allgather_match_tables(idx, csum[off], inl, download=want)
        _last_gather_path = "c-abi"
        _last_gather_stats = dict(st_m, records=st_r, inliers=st_i,
                                  total_ms=st_r["total_ms"] + st_m["total_ms"] + st_i["total_ms"])
        return g_tvg, g_off, g_m, g_ioff, g_im
    m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
    mask = np.asarray(inlier_mask).astype(bool)
    csum = np.zeros(len(mask) + 1, dtype=np.int64)
    csum[1:] = np.cumsum(mask)
    inl_off = csum[off]            # inlier matches before each pair's first match: the CSR of the inlier lists
    g_tvg = all_gather_pair_records(pair_index, tvg, total_pairs, device=device, group=group)
    g_off, g_m = all_gather_match_tables(pair_index, off, m, device=device, group=group)
    g_ioff, g_im = all_gather_match_tables(pair_index, inl_off, m[mask], device=device, group=group)
    return g_tvg, g_off, g_m, g_ioff, g_im


def shard_pairs_by_cost(slot1: np.ndarray, slot2: np.ndarray, cost: np.ndarray, rank: int, world: int):
    """shard_pairs with an explicit per-pair cost instead of n1 * n2: a sharded match + VERIFY run is cut where the
    cumulative cost is equal - e.g. cost = n1 * n2 * c_scan + matches * c_verify, with the match counts of a previous
    pass (or a model of them) - because verification time follows the matches, not the image sizes (data-dependent
    trial counts on a few overlapping pairs, nothing on the many that do not overlap).  Same ordering (image 2, then
    image 1) and the same contiguous slices as shard_pairs; returns (slot1, slot2, index)."""
    s1 = np.asarray(slot1, dtype=np.uint32)
    s2 = np.asarray(slot2, dtype=np.uint32)
    order = np.lexsort((s1, s2))
    if len(order) == 0:
        return s1[order], s2[order], order
    c = np.asarray(cost, dtype=np.float64)[order]
    if not np.all(c >= 0):
        raise ValueError("costs must be >= 0")
    work = np.cumsum(c)
    total = work[-1]
    if total <= 0:
        return shard_pairs(s1, s2, rank, world)
    mid = work - 0.5 * c
    owner = np.minimum((mid * world / total).astype(np.int64), world - 1)
    lo, hi = np.searchsorted(owner, [rank, rank + 1])
    mine = order[lo:hi]
    return s1[mine], s2[mine], mine
