#!/usr/bin/env python3
"""N ranks of the exchange step as THREADS of one process on ONE device, with tests/shim/fake_rccl.cc standing where
librccl.so.1 stands (AMC_RCCL_LIBRARY must name the built shim; tests/test_multigpu_gpu.py does that and runs this file
in a process of its own, because a process resolves RCCL once).  Everything above the transport is the product:
amc_comm_create, amc_allgather_match_tables with its displacements, per-rank exact counts, reorder into the global CSR,
appended lists, the poisoned size exchange - checked against the single-context result.

    AMC_RCCL_LIBRARY=tests/shim/_build/libfakerccl.so python tests/comm_threads.py <world> [--images N] [--verify]

--verify: the verification half (amc_allgather_pair_records / amc_allgather_inlier_tables: geometries and inlier lists of
a sharded match + verify run from device memory, records of any size, the allocation-failure agreement, collective
errors for what one rank alone gets wrong).
"""
import os
import sys
import threading
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pycolmap_amd import _capi, synth  # noqa: E402
from pycolmap_amd import distributed as D  # noqa: E402


def verify_main(world, num_images):
    """The verification half of the exchange (amc_allgather_pair_records, amc_allgather_inlier_tables) over `world` ranks:
    every rank runs match + verify on its shard, the geometries / matches / inlier lists of all ranks come back in the
    global order - from device memory (resident=True) and from the host arrays - and equal the single-context result."""
    rng = np.random.default_rng(9)
    images = synth.multiview_scene(rng, num_images=num_images, n_feats=384, num_landmarks=500)
    s1_all, s2_all = synth.exhaustive_pairs(num_images)
    rows = np.full(num_images, 384)
    opts = _capi.tvg_options()

    def fill(ctx):
        ctx.reserve_slots(num_images)
        for k, im in enumerate(images):
            ctx.upload_descriptors(k, im["descriptors"])
            ctx.upload_keypoints(k, im["keypoints"])
            ctx.upload_camera(k, im["model"], im["width"], im["height"], im["params"], True)

    ref = _capi.Context(0)
    fill(ref)
    woff, wm, _, wtvg, wmask, _ = ref.match_verify_pairs(s1_all, s2_all, opts)
    woff, wm, wtvg, wmask = woff.copy(), wm.copy(), wtvg.copy(), wmask.copy()
    assert int((wtvg["config"] > 1).sum()) >= 2, "the scene must verify a few pairs"
    wcs = np.zeros(len(wmask) + 1, dtype=np.int64)
    wcs[1:] = np.cumsum(wmask != 0)
    w_ioff, w_im = wcs[woff.astype(np.int64)], wm[wmask != 0]
    uid = _capi.comm_unique_id()
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            ctx = _capi.Context(0)
            fill(ctx)
            comm = ctx.comm_create(world, r, uid)
            s1, s2, mine = D.shard_pairs(s1_all, s2_all, r, world, rows=rows)
            off, m, _, tvg, mask, _ = ctx.match_verify_pairs(s1, s2, opts)
            off, m, tvg, mask = off.copy(), m.copy(), tvg.copy(), mask.copy()
            out = {"npairs": len(mine)}
            out["resident"] = D.all_gather_verification(mine, None, off, None, None, len(s1_all), comm=comm, resident=True)
            out["resident_stats"] = D.last_gather_stats()
            out["host"] = D.all_gather_verification(mine, tvg, off, m, mask, len(s1_all), comm=comm, download_rank=0)
            # any plain record travels (here: 16 bytes per pair), also without a global numbering (appended in rank order)
            rec = np.zeros(len(mine), dtype=[("pos", np.uint64), ("rank", np.uint64)])
            rec["pos"], rec["rank"] = mine, r
            out["plain"], _ = comm.allgather_pair_records(mine, rec)
            out["appended_rec"], _ = comm.allgather_pair_records(None, rec)
            _, _, out["base"] = D.all_gather_appended_tables(off, m, comm=comm)
            # a rank that cannot allocate after the size exchange: every rank returns AMC_E_NOMEM, nobody is stranded,
            # and (the agreement being collective) the communicator lives on
            os.environ["AMC_COMM_FAIL_ALLOC_RANK"] = str(world - 1)
            notes = []
            for call in (lambda: comm.allgather_match_tables(mine, off, m), lambda: comm.allgather_pair_records(mine, rec)):
                try:
                    call()
                    notes.append("RETURNED")
                except _capi.AmcError as e:
                    notes.append("nomem" if e.code == _capi.AMC_E_NOMEM else repr(e))
            out["alloc_failure"] = notes
            # (all ranks must have left the failing calls before the hook goes away: the next exchange is that barrier)
            try:
                comm.allgather_pair_records(mine, rec)
            except _capi.AmcError:
                pass
            os.environ.pop("AMC_COMM_FAIL_ALLOC_RANK", None)
            # a record size that differs between ranks, a wrapper-level mistake on one rank: collective errors
            try:
                comm.allgather_pair_records(mine, rec if r else rec["pos"].copy())
                out["size_mismatch"] = "accepted"
            except _capi.AmcError as e:
                out["size_mismatch"] = "rejected" if e.code == _capi.AMC_E_INVALID else repr(e)
            out["short_applied"] = bool(r == world - 1 and len(m))
            try:
                comm.allgather_match_tables(mine, off, m[:-1] if out["short_applied"] else m)
                out["short_rows"] = "accepted"
            except _capi.AmcError as e:
                out["short_rows"] = "rejected" if e.code == _capi.AMC_E_INVALID else repr(e)
            out["again"] = comm.allgather_inlier_tables(mine)     # still the last verification call's lists
            comm.close()
            ctx.close()
            results[r] = out
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            errors.append((r, repr(e)))
            raise

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=150)
    if any(t.is_alive() for t in threads) or errors:
        print("FAILED: stuck ranks", [r for r, t in enumerate(threads) if t.is_alive()], "errors", errors, flush=True)
        os._exit(1)
    sizes = [out["npairs"] for out in results]
    for r, out in enumerate(results):
        g_tvg, g_off, g_m, g_ioff, g_im = out["resident"]
        assert g_tvg.tobytes() == wtvg.tobytes(), (r, "tvg records from device memory")
        assert np.array_equal(g_off, woff) and np.array_equal(g_m, wm), (r, "matches")
        assert np.array_equal(g_ioff.astype(np.int64), w_ioff) and np.array_equal(g_im, w_im), (r, "inlier lists compacted on the device")
        assert out["resident_stats"]["records"]["world_size"] == world
        h_tvg, h_off, h_m, h_ioff, h_im = out["host"]
        assert np.array_equal(h_off, woff) and np.array_equal(h_ioff.astype(np.int64), w_ioff)
        if r == 0:
            assert h_tvg.tobytes() == wtvg.tobytes() and np.array_equal(h_m, wm) and np.array_equal(h_im, w_im)
        else:
            assert h_tvg is None and h_m is None and h_im is None      # only rank 0 asked for the rows on the host
        assert np.array_equal(out["plain"]["pos"], np.arange(len(s1_all)))
        owner = np.concatenate([np.full(n, q) for q, n in enumerate(sizes)]) if sum(sizes) else np.zeros(0)
        assert np.array_equal(out["appended_rec"]["rank"], owner)
        assert out["base"] == sum(sizes[:r]), (r, out["base"])
        assert out["alloc_failure"] == ["nomem", "nomem"], (r, out["alloc_failure"])
        assert out["size_mismatch"] == ("rejected" if world > 1 else "accepted"), (r, out["size_mismatch"])
        a_off, a_m, _ = out["again"]
        assert np.array_equal(a_off.astype(np.int64), w_ioff) and np.array_equal(a_m, w_im)
    # the last rank passed one row too few (when it had any): the wrapper's check became a collective error
    want_short = "rejected" if any(out["short_applied"] for out in results) else "accepted"
    assert {out["short_rows"] for out in results} == {want_short}, [out["short_rows"] for out in results]
    ref.close()
    print(f"comm threads verify ok: world {world}, {len(s1_all)} pairs as {sizes}, {len(wm)} matches, {len(w_im)} inlier matches, "
          f"{int((wtvg['config'] > 1).sum())} verified pairs")


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    num_images = int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 7
    assert "fakerccl" in os.environ.get("AMC_RCCL_LIBRARY", ""), "this harness is for the shim, not for RCCL"
    if "--verify" in sys.argv:
        return verify_main(world, num_images)
    rng = np.random.default_rng(4)
    imgs = synth.scene_images(rng, num_images, 500)
    s1_all, s2_all = synth.exhaustive_pairs(num_images)
    rows = np.full(num_images, 500)

    # the single-context reference
    ref = _capi.Context(0)
    ref.reserve_slots(num_images)
    for k, im in enumerate(imgs):
        ref.upload_descriptors(k, im)
    woff, wm, _ = ref.match_pairs(s1_all, s2_all)
    assert len(wm) > 10

    uid = _capi.comm_unique_id()
    results, errors = [None] * world, []
    shard_sizes = [len(D.shard_pairs(s1_all, s2_all, q, world, rows=rows)[2]) for q in range(world)]
    ranks_with_pairs = sum(1 for z in shard_sizes if z > 0)

    def rank_main(r):
        try:
            ctx = _capi.Context(0)
            ctx.reserve_slots(num_images)
            for k, im in enumerate(imgs):
                ctx.upload_descriptors(k, im)
            comm = ctx.comm_create(world, r, uid)                       # collective
            s1, s2, mine = D.shard_pairs(s1_all, s2_all, r, world, rows=rows)
            off, m, _ = ctx.match_pairs(s1, s2)
            out = {}
            # rows from the resident table of THIS rank's context; every rank downloads
            out["resident"] = comm.allgather_match_tables(mine, off, None)
            # rows from the host; only rank 0 downloads
            g_off, g_m, st = comm.allgather_match_tables(mine, off, m, download=(r == 0))
            out["host"] = (g_off, g_m, st)
            # lists without a global numbering: appended in rank order
            out["appended"] = comm.allgather_match_tables(None, off, m)
            out["local"] = (mine, off, m)
            # one rank with bad arguments: EVERY rank gets an error, nobody is left inside a collective
            try:
                bad = off.copy()
                if r == world - 1:
                    bad[-1] += 1
                comm.allgather_match_tables(mine, bad, None)
                out["poison_message"] = "RETURNED"
            except _capi.AmcError as e:
                out["poison_message"] = str(e) if e.code == _capi.AMC_E_INVALID else "WRONG CODE " + str(e)
            # positions that are no permutation (every rank claims positions 0 ..: duplicates as soon as two ranks have
            # pairs): every rank returns the error.  (No assertion may fire between two collectives of this thread:
            # the other ranks would wait for it - the outcome is noted and checked by the main thread.)
            try:
                comm.allgather_match_tables(np.arange(len(mine), dtype=np.uint64), off, m)
                out["dup_positions"] = "accepted"
            except _capi.AmcError as e:
                out["dup_positions"] = "rejected" if e.code == _capi.AMC_E_INVALID else repr(e)
            # and the communicator still works afterwards
            out["again"] = comm.allgather_match_tables(mine, off, None)
            comm.close()
            ctx.close()
            results[r] = out
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            errors.append((r, repr(e)))
            raise

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=90)
    if any(t.is_alive() for t in threads) or errors:   # (a stuck rank sits in a C call: leave without waiting for it)
        print("FAILED: stuck ranks", [r for r, t in enumerate(threads) if t.is_alive()], "errors", errors, flush=True)
        os._exit(1)
    total_sent = 0
    for r, out in enumerate(results):
        for key in ("resident", "again"):
            g_off, g_m, st = out[key]
            assert np.array_equal(g_off, woff) and np.array_equal(g_m, wm), (r, key)
            assert st["world_size"] == world and st["rank"] == r
        g_off, g_m, st = out["host"]
        assert (g_m is None) == (r != 0)            # only rank 0 asked for the rows on the host
        assert np.array_equal(g_off, woff) and (g_m is None or np.array_equal(g_m, wm))
        assert out["poison_message"].startswith("amc error") and "WRONG CODE" not in out["poison_message"], out["poison_message"]
        total_sent += out["resident"][2]["rows_sent"]
        assert out["resident"][2]["rows_received"] == len(wm) - len(out["local"][2])
        # appended: rank order, each rank's list as it passed it
        a_off, a_m, _ = out["appended"]
        base = sum(len(results[q]["local"][0]) for q in range(r))
        mine, off, m = out["local"]
        for k in range(len(mine)):
            got = a_m[int(a_off[base + k]):int(a_off[base + k + 1])]
            assert np.array_equal(got, m[int(off[k]):int(off[k + 1])]), (r, k)
        assert len(a_off) == len(woff) and int(a_off[-1]) == len(wm)
        if r != world - 1:
            assert "rank %d reported invalid arguments" % (world - 1) in out["poison_message"], out["poison_message"]
        assert out["dup_positions"] == ("rejected" if ranks_with_pairs >= 2 else "accepted"), (r, out["dup_positions"])
    assert total_sent == (world - 1) * len(wm)      # every row went to every other rank exactly once
    sizes = [len(out["local"][0]) for out in results]
    ref.close()
    print(f"comm threads ok: world {world}, {len(s1_all)} pairs as {sizes}, {len(wm)} matches, {total_sent} rows over the wire")


if __name__ == "__main__":
    main()
