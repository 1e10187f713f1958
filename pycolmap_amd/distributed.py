"""Multi-GPU sharding of the match + verify path.

Image pairs are independent units, so the path shards embarrassingly: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests), the
descriptor arena replicated on every GPU, the pair list dealt in contiguous slices, and ONE exchange
step at the end: an all-gather of the per-rank match tables (counts first, then padded tables) so
that every rank — in particular the rank that owns the SQLite writer — holds the whole match graph.
The functions here are pure tensor plumbing and work unchanged on CPU tensors (gloo) and GPU
tensors (RCCL).
"""
from __future__ import annotations

import numpy as np


def shard_pairs(slot1: np.ndarray, slot2: np.ndarray, rank: int, world: int):
    """Contiguous slice of the pair list, after ordering by image 2 (then image 1) so that pairs
    sharing the streamed image stay on one GPU and in one L2.  Returns (slot1, slot2, index) where
    `index` maps the shard back to positions in the caller's list."""
    s1 = np.asarray(slot1, dtype=np.uint32)
    s2 = np.asarray(slot2, dtype=np.uint32)
    order = np.lexsort((s1, s2))
    per = (len(order) + world - 1) // world
    mine = order[rank * per:(rank + 1) * per]
    return s1[mine], s2[mine], mine


def all_gather_match_tables(pair_index: np.ndarray, offsets: np.ndarray, matches: np.ndarray, device=None,
                            group=None):
    """Exchange per-rank CSR match tables.  Every rank passes the global indices of its pairs, its
    CSR offsets and its (M, 2) uint32 matches; every rank gets back the table for ALL pairs as
    (global_offsets, global_matches) in the global pair order.

    Two collectives: an all-gather of the (npairs, nmatches) sizes, then padded all-gathers of the
    per-pair counts / indices and of the match rows."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    counts = np.diff(np.asarray(offsets, dtype=np.int64))
    npairs, nm = len(pair_index), int(matches.shape[0])
    sizes = torch.tensor([npairs, nm], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(world, 2).cpu().numpy()
    max_p, max_m = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())

    meta = torch.zeros(max(max_p, 1), 2, dtype=torch.int64, device=dev)   # (global index, count)
    if npairs:
        meta[:npairs, 0] = torch.from_numpy(np.asarray(pair_index, dtype=np.int64)).to(dev)
        meta[:npairs, 1] = torch.from_numpy(counts).to(dev)
    all_meta = torch.empty(world * max(max_p, 1), 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_meta, meta, group=group)

    rows = torch.zeros(max(max_m, 1), 2, dtype=torch.int32, device=dev)
    if nm:
        rows[:nm] = torch.from_numpy(np.ascontiguousarray(matches, dtype=np.uint32).view(np.int32)).to(dev)
    all_rows = torch.empty(world * max(max_m, 1), 2, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_rows, rows, group=group)

    all_meta = all_meta.view(world, max(max_p, 1), 2).cpu().numpy()
    all_rows = all_rows.view(world, max(max_m, 1), 2).cpu().numpy().view(np.uint32)
    total_pairs = int(all_sizes[:, 0].sum())
    g_counts = np.zeros(total_pairs, dtype=np.int64)
    per_rank = []
    for r in range(world):
        p, m = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        idx, cnt = all_meta[r, :p, 0], all_meta[r, :p, 1]
        g_counts[idx] = cnt
        per_rank.append((idx, cnt, all_rows[r, :m]))
    g_off = np.zeros(total_pairs + 1, dtype=np.uint64)
    g_off[1:] = np.cumsum(g_counts)
    g_matches = np.zeros((int(g_off[-1]), 2), dtype=np.uint32)
    for idx, cnt, rws in per_rank:
        src = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        for k in range(len(idx)):
            if cnt[k]:
                g_matches[int(g_off[idx[k]]):int(g_off[idx[k]]) + int(cnt[k])] = rws[src[k]:src[k + 1]]
    return g_off, g_matches
