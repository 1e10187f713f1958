"""A failed reset of a queue head or counter must fail the call, never launch a persistent kernel on a stale counter
(VERDICT r4 item 6).  libamc.so routes every memset / small copy whose target a kernel reads through
`memset_async` / `memcpy_async` (csrc/amc_internal.h); AMC_FAIL_NEXT_MEMSET=k / AMC_FAIL_NEXT_MEMCPY=k make the k-th one
report hipErrorInvalidValue without enqueuing anything.  Sweeping k over a call walks every such site: each must
surface as an error (C ABI: AMC_E_HIP; Python API: RuntimeError, /root/reference/pycolmap/log_exceptions.h:54-76's
convention of turning failures into Python exceptions), and the context must stay usable - the next clean call returns
exactly what the oracle says."""
import os

import numpy as np
import pytest

import colmap_db
import oracle_lib
import pycolmap_amd as pycolmap
from pycolmap_amd import _capi, synth

pytestmark = pytest.mark.gpu


def _sweep(var, call, clean, max_sites=64):
    """call() under var=k for k = 1, 2, ...: every k up to the number of sites must raise; the first k that does not
    raise has run past the last site and must return the clean result.  Returns the number of sites met."""
    sites = 0
    try:
        for k in range(1, max_sites + 1):
            os.environ[var] = str(k)
            try:
                got = call()
            except _capi.AmcError as e:
                assert e.code == _capi.AMC_E_HIP, e
                sites += 1
                os.environ.pop(var)          # (the count restarts when the variable goes away)
                clean_again = call()         # the context survives the failed call
                for a, b in zip(clean_again, clean):
                    np.testing.assert_array_equal(a, b)
                continue
            for a, b in zip(got, clean):
                np.testing.assert_array_equal(a, b)
            return sites
    finally:
        os.environ.pop(var, None)
    raise AssertionError(f"{var}: still failing after {max_sites} sites")


@pytest.mark.parametrize("batches", [1, 3])
def test_every_memset_of_a_match_call_is_checked(amc_ctx, batches, monkeypatch):
    rng = np.random.default_rng(3)
    imgs = synth.scene_images(rng, 5, 384)
    amc_ctx.reserve_slots(len(imgs))
    for k, im in enumerate(imgs):
        amc_ctx.upload_descriptors(k, im)
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    if batches > 1:   # several batches: the copy of a batch's matches rides in the next scan (its own counter)
        monkeypatch.setenv("AMC_MATCH_BATCH_ENTRIES", "2048")
        monkeypatch.setenv("AMC_D2H_FUSE_MIN_BYTES", "8")
    woff, wm = oracle_lib.match_pairs(imgs, s1, s2, 0.8, 0.7, True)

    def call():
        off, m, _ = amc_ctx.match_pairs(s1, s2, 0.8, 0.7, True, kernel="mfma")
        return off, m

    clean = call()
    np.testing.assert_array_equal(clean[0], woff)
    np.testing.assert_array_equal(clean[1], wm)
    sites = _sweep("AMC_FAIL_NEXT_MEMSET", call, clean)
    # per batch: cursor, error count, accept mask, two queue heads (forward + reverse scan) at least
    assert sites >= 5 * batches, sites
    if batches > 1:
        assert _sweep("AMC_FAIL_NEXT_MEMCPY", call, clean) >= 0


def test_every_memset_of_a_verification_call_is_checked(amc_ctx):
    rng = np.random.default_rng(4)
    scenes = [synth.two_view_scene(rng, num_inliers=120, num_outliers=40) for _ in range(3)]
    amc_ctx.reserve_slots(2 * len(scenes))
    offs, ms = [0], []
    for k, sc in enumerate(scenes):
        for side, pts in ((0, sc["pts1"]), (1, sc["pts2"])):
            amc_ctx.upload_keypoints(2 * k + side, pts.astype(np.float32))
            amc_ctx.upload_camera(2 * k + side, "PINHOLE", sc["width"], sc["height"],
                                  (sc["f"], sc["f"], sc["width"] / 2.0, sc["height"] / 2.0), True)
        ms.append(sc["matches"])
        offs.append(offs[-1] + len(sc["matches"]))
    s1 = np.arange(0, 2 * len(scenes), 2, dtype=np.uint32)
    s2 = s1 + 1
    offs = np.asarray(offs, dtype=np.uint64)
    ms = np.concatenate(ms).astype(np.uint32)

    def call():
        tvg, mask, _ = amc_ctx.verify_pairs(s1, s2, offs, ms, _capi.tvg_options(), seed=7)
        return tvg.copy(), mask.copy()

    clean = call()
    assert (clean[0]["config"] > 1).all()
    sites = _sweep("AMC_FAIL_NEXT_MEMSET", call, clean)
    assert sites >= 5, sites   # bad-index / stream counters, the records, both kernels' queue heads, the work sums


def test_the_python_api_raises(tmp_path):
    """The same failure through pycolmap.match_exhaustive: a RuntimeError carrying the library's message, and the
    database left as it was (nothing is written before the device calls have succeeded)."""
    rng = np.random.default_rng(6)
    images = synth.multiview_scene(rng, num_images=4, n_feats=256)
    db = tmp_path / "db.db"
    colmap_db.create(db, images)
    os.environ["AMC_FAIL_NEXT_MEMSET"] = "2"
    try:
        with pytest.raises(RuntimeError, match="memset|scan|amc_"):
            pycolmap.match_exhaustive(db)
    finally:
        os.environ.pop("AMC_FAIL_NEXT_MEMSET", None)
    got_m, got_t = colmap_db.read_all(db)
    assert got_m == {} and got_t == {}
    pycolmap.match_exhaustive(db)   # and the same call goes through afterwards
    got_m, _ = colmap_db.read_all(db)
    assert len(got_m) == 6
