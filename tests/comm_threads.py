#!/usr/bin/env python3
"""N ranks of the exchange step as THREADS of one process on ONE device, with tests/shim/fake_rccl.cc standing where
librccl.so.1 stands (AMC_RCCL_LIBRARY must name the built shim; tests/test_multigpu_gpu.py does that and runs this file
in a process of its own, because a process resolves RCCL once).  Everything above the transport is the product:
amc_comm_create, amc_allgather_match_tables with its displacements, per-rank exact counts, reorder into the global CSR,
appended lists, the poisoned size exchange - checked against the single-context result.

    AMC_RCCL_LIBRARY=tests/shim/_build/libfakerccl.so python tests/comm_threads.py <world> [--images N]
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


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    num_images = int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 7
    assert "fakerccl" in os.environ.get("AMC_RCCL_LIBRARY", ""), "this harness is for the shim, not for RCCL"
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
