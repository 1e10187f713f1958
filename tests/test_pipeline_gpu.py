"""End-to-end GPU tests of the drop-in API: COLMAP-schema SQLite in -> pycolmap_amd.match_exhaustive /
match_sequential / verify_matches -> SQLite out, compared row by row with the CPU oracles run in the
controller's own pair order (BASELINE.json configs[0] shape, scaled)."""
import numpy as np
import pytest

import colmap_db
import oracle_lib as o
import pycolmap_amd as pycolmap
from pycolmap_amd import synth

pytestmark = pytest.mark.gpu
MIN_INL = 15


def expected_rows(images, ids, blocks, sift=(0.8, 0.7, True), prior=False, tvg_kw=None, guided=False):
    """What COLMAP's controller would store for the given pair blocks (oracle matcher + oracle TVG)."""
    by_id = {i: im for i, im in zip(ids, images)}
    exp_m, exp_t, seen = {}, {}, set()
    for block in blocks:
        for id1, id2 in block:
            if id1 == id2:
                continue
            pid = colmap_db.pair_id(id1, id2)
            if pid in seen:
                continue
            seen.add(pid)
            a, b = by_id[id1], by_id[id2]
            m = o.match(a["descriptors"], b["descriptors"], *sift)
            tv = None
            if len(m) >= MIN_INL:
                cam1 = o.make_camera(a.get("model", 1), a["width"], a["height"], a["params"], prior=prior)
                cam2 = o.make_camera(b.get("model", 1), b["width"], b["height"], b["params"], prior=prior)
                r = o.estimate_two_view_geometry(cam1, a["keypoints"][:, :2].astype(np.float64), cam2,
                                                 b["keypoints"][:, :2].astype(np.float64), m,
                                                 o.tvg_default_options(**(tvg_kw or {})))
                inl = m[r["inlier_mask"]]
                if guided and len(inl) >= MIN_INL and r["config"] in (2, 3, 4, 5, 6):
                    # MatchGuided replaces the inlier matches; raw matches and models stay
                    max_error = (tvg_kw or {}).get("max_error", 4.0)
                    inl = o.match_guided(a["descriptors"], a["keypoints"], b["descriptors"], b["keypoints"],
                                         r["config"], r["F"], r["H"], max_error, *sift)
                if len(inl) >= MIN_INL:
                    tv = dict(config=r["config"], F=r["F"], E=r["E"], H=r["H"], inl=inl, qvec=r["qvec"], tvec=r["tvec"])
            else:
                m = np.zeros((0, 2), np.uint32)
            swap = id1 > id2
            exp_m[pid] = np.ascontiguousarray(m[:, ::-1]) if swap else m
            if tv is None:
                exp_t[pid] = dict(config=0, inl=np.zeros((0, 2), np.uint32), F=None, E=None, H=None)
            elif swap:
                w, x, y, z = tv["qvec"]                       # Inverse(cam2_from_cam1): R^T, -R^T t
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                              [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                              [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
                exp_t[pid] = dict(config=tv["config"], inl=np.ascontiguousarray(tv["inl"][:, ::-1]), F=tv["F"].T,
                                  E=tv["E"].T, H=np.linalg.inv(tv["H"]), H_approx=True,
                                  qvec=np.array([w, -x, -y, -z]), tvec=-R.T @ tv["tvec"])
            else:
                exp_t[pid] = tv
    return exp_m, exp_t


def compare(db_path, exp_m, exp_t):
    got_m, got_t = colmap_db.read_all(db_path)
    assert set(got_m) == set(exp_m) and set(got_t) == set(exp_t)
    nver = 0
    for pid in exp_m:
        np.testing.assert_array_equal(got_m[pid], exp_m[pid], err_msg=f"matches of pair {pid}")
        g, e = got_t[pid], exp_t[pid]
        assert g["config"] == e["config"], f"pair {pid}"
        np.testing.assert_array_equal(g["inlier_matches"], e["inl"], err_msg=f"inliers of pair {pid}")
        if len(e["inl"]) == 0:
            assert g["F"] is None and g["E"] is None and g["H"] is None   # empty blobs, like COLMAP
            continue
        nver += 1
        for k in "FE":
            np.testing.assert_array_equal(g[k].view(np.uint64), np.ascontiguousarray(e[k]).view(np.uint64))
        if e.get("H_approx"):     # stored inverted (id1 > id2): H^-1 and the inverse pose up to rounding
            np.testing.assert_allclose(g["H"], e["H"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(g["qvec"], e["qvec"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(g["tvec"], e["tvec"], rtol=0, atol=1e-12)
        else:
            np.testing.assert_array_equal(g["H"].view(np.uint64), e["H"].view(np.uint64))
            np.testing.assert_array_equal(g["qvec"].view(np.uint64), e["qvec"].view(np.uint64))
            np.testing.assert_array_equal(g["tvec"].view(np.uint64), e["tvec"].view(np.uint64))
    return nver


@pytest.mark.parametrize("prior", [False, True])
def test_match_exhaustive_end_to_end(tmp_path, prior):
    rng = np.random.default_rng(1 if prior else 0)
    images = synth.multiview_scene(rng, num_images=7, n_feats=512)
    for im in images:
        im["prior"] = prior
    db = tmp_path / "db.db"
    ids = colmap_db.create(db, images)
    pycolmap.match_exhaustive(db, matching_options={"block_size": 3})      # several blocks, swapped pairs
    blocks = pycolmap._pycolmap._exhaustive_blocks(ids, 3)
    exp_m, exp_t = expected_rows(images, ids, blocks, prior=prior)
    assert len(exp_m) == 21
    assert compare(db, exp_m, exp_t) >= 5
    st = pycolmap.last_run_stats()
    assert st["pairs_matched"] == 21 and st["pairs_skipped"] == 0
    # the fused calls' own host timeline (amc_ctx_last_timeline): its phases are there and fit inside the calls' wall time
    tl = st["fused_call_timeline_ms"]
    assert set(tl) == {"verify_setup", "match_call", "close_and_launch", "verify_wait_pack_download", "batch_handover_hidden"}
    assert all(v >= 0 for v in tl.values()) and tl["match_call"] > 0 and tl["verify_wait_pack_download"] > 0
    assert tl["verify_setup"] + tl["match_call"] + tl["close_and_launch"] + tl["verify_wait_pack_download"] <= st["match_call_ms"] + 0.5
    # resume: everything exists -> nothing recomputed, rows unchanged
    pycolmap.match_exhaustive(db, matching_options={"block_size": 3})
    st = pycolmap.last_run_stats()
    assert st["pairs_matched"] == 0 and st["pairs_skipped"] == 21
    compare(db, exp_m, exp_t)
    d = pycolmap.Database(db)
    assert d.num_matched_image_pairs == 21 and d.num_verified_image_pairs == 21


def test_match_sequential_and_options(tmp_path):
    rng = np.random.default_rng(2)
    images = synth.multiview_scene(rng, num_images=8, n_feats=400)
    db = tmp_path / "db.db"
    ids = colmap_db.create(db, images)
    sift = pycolmap.SiftMatchingOptions(max_ratio=0.9, cross_check=False)
    pycolmap.match_sequential(db, sift_options=sift, matching_options={"overlap": 3, "quadratic_overlap": True},
                              verification_options={"detect_watermark": False, "ransac": {"max_error": 3.0}})
    blocks = pycolmap._pycolmap._sequential_blocks(ids, 3, True)      # names are already ordered
    exp_m, exp_t = expected_rows(images, ids, blocks, sift=(0.9, 0.7, False),
                                 tvg_kw=dict(detect_watermark=0, max_error=3.0))
    assert compare(db, exp_m, exp_t) >= 4


def test_match_spatial(tmp_path):
    """match_spatial = SpatialFeatureMatcher::Run's pairs through the same match + verify path: every image is matched
    with its nearest neighbours by location prior; images without a prior are left alone."""
    rng = np.random.default_rng(21)
    images = synth.multiview_scene(rng, num_images=9, n_feats=400)
    for k, im in enumerate(images[:8]):                                # a line of cameras 10 m apart; the last has no prior
        im["prior_t"] = (10.0 * k + 1.0, 2.0, 7.0)
    db = tmp_path / "db.db"
    ids = colmap_db.create(db, images)
    mo = pycolmap.SpatialMatchingOptions(is_gps=False, ignore_z=False, max_num_neighbors=4, max_distance=25.0)
    pycolmap.match_spatial(db, matching_options=mo)
    pri = [list(im.get("prior_t", (np.nan,) * 3)) for im in images]
    blocks = pycolmap._pycolmap._spatial_blocks(ids, pri, mo)
    assert len(blocks) == 8 and all(ids[8] not in p for b in blocks for p in b)
    exp_m, exp_t = expected_rows(images, ids, blocks)
    # neighbours within 25 m: +-1 and +-2 along the line (knn = 4 includes the query itself, so at most 3 others)
    assert {tuple(sorted(p)) for b in blocks for p in b} == {(ids[a], ids[b]) for a in range(8) for b in range(a + 1, 8) if b - a <= 2}
    assert compare(db, exp_m, exp_t) >= 4
    st = pycolmap.last_run_stats()
    assert st["pairs_matched"] == len(exp_m)
    # GPS priors: same images a few metres apart on the ground
    db2 = tmp_path / "db2.db"
    for k, im in enumerate(images):
        im["prior_t"] = (48.0 + 1e-4 * k, 11.0, 500.0 + k)              # 1e-4 degrees of latitude = 11.1 m
    ids2 = colmap_db.create(db2, images)
    pycolmap.match_spatial(db2, matching_options=dict(max_num_neighbors=3, max_distance=15.0))
    got_m, _ = colmap_db.read_all(db2)
    assert set(got_m) == {colmap_db.pair_id(ids2[a], ids2[a + 1]) for a in range(8)}


def test_database_built_through_the_api(tmp_path):
    """A database populated with Database.write_camera / write_image / write_keypoints / write_descriptors (no sqlite3 on
    the caller's side) gives the rows a COLMAP-written file gives."""
    rng = np.random.default_rng(22)
    images = synth.multiview_scene(rng, num_images=4, n_feats=400)
    db = tmp_path / "api.db"
    d = pycolmap.Database(db)
    ids = []
    with pycolmap.DatabaseTransaction(d):
        for im in images:
            cid = d.write_camera(pycolmap.Camera(model=im.get("model", 1), width=im["width"], height=im["height"], params=list(im["params"])))
            iid = d.write_image(pycolmap.Image(name=im["name"], camera_id=cid))
            d.write_keypoints(iid, im["keypoints"])
            d.write_descriptors(iid, im["descriptors"])
            ids.append(iid)
    d.close()
    pycolmap.match_exhaustive(db)
    exp_m, exp_t = expected_rows(images, ids, pycolmap._pycolmap._exhaustive_blocks(ids, 50))
    assert compare(db, exp_m, exp_t) >= 2


def test_verify_matches_reads_stored_matches(tmp_path):
    rng = np.random.default_rng(3)
    images = synth.multiview_scene(rng, num_images=4, n_feats=512)
    db = tmp_path / "db.db"
    ids = colmap_db.create(db, images)
    pairs = [(ids[0], ids[1]), (ids[2], ids[1]), (ids[0], ids[3])]
    stored = {}
    for a, b in pairs:
        ia, ib = ids.index(a), ids.index(b)
        m = o.match(images[ia]["descriptors"], images[ib]["descriptors"])
        colmap_db.write_matches(db, a, b, m)
        stored[(a, b)] = m
    pairs_txt = tmp_path / "pairs.txt"
    pairs_txt.write_text("# comment\n\n" + "\n".join(f"{images[ids.index(a)]['name']} {images[ids.index(b)]['name']}"
                                                     for a, b in pairs) + "\n")
    pycolmap.verify_matches(db, pairs_txt)
    st = pycolmap.last_run_stats()
    assert st["pairs_matched"] == 0 and st["pairs_verified"] >= 2
    exp_m, exp_t = expected_rows(images, ids, [pairs])
    compare(db, exp_m, exp_t)


def test_failed_call_leaves_stored_rows_alone(tmp_path):
    """Imported matches (hloc style) with one index past the keypoints: verify_matches must raise and the database
    must still hold every stored row - the deletes of a recomputed pair happen in the transaction that writes
    its replacement rows, never before the device calls have succeeded."""
    rng = np.random.default_rng(5)
    images = synth.multiview_scene(rng, num_images=4, n_feats=512)
    db = tmp_path / "db.db"
    ids = colmap_db.create(db, images)
    pairs = [(ids[0], ids[1]), (ids[1], ids[2]), (ids[2], ids[3])]
    for k, (a, b) in enumerate(pairs):
        m = o.match(images[ids.index(a)]["descriptors"], images[ids.index(b)]["descriptors"]).copy()
        assert len(m) >= MIN_INL
        if k == 1:
            m[3, 1] = 100000            # past image b's 512 keypoints
        colmap_db.write_matches(db, a, b, m)
    before_m, before_t = colmap_db.read_all(db)
    pairs_txt = tmp_path / "pairs.txt"
    pairs_txt.write_text("\n".join(f"{images[ids.index(a)]['name']} {images[ids.index(b)]['name']}" for a, b in pairs))
    with pytest.raises(ValueError, match="keypoints"):
        pycolmap.verify_matches(db, pairs_txt)
    after_m, after_t = colmap_db.read_all(db)
    assert set(after_m) == set(before_m) and after_t == before_t == {}
    for pid in before_m:
        np.testing.assert_array_equal(after_m[pid], before_m[pid])
    # with the bad pair left out the same call goes through and replaces nothing it should not
    pairs_txt.write_text("\n".join(f"{images[ids.index(a)]['name']} {images[ids.index(b)]['name']}"
                                   for a, b in (pairs[0], pairs[2])))
    pycolmap.verify_matches(db, pairs_txt)
    final_m, final_t = colmap_db.read_all(db)
    assert set(final_m) == set(before_m) and len(final_t) == 2
    np.testing.assert_array_equal(final_m[colmap_db.pair_id(*pairs[1])], before_m[colmap_db.pair_id(*pairs[1])])


def test_match_exhaustive_with_relative_pose(tmp_path):
    """TwoViewGeometryOptions.compute_relative_pose: cam2_from_cam1 lands in the qvec / tvec columns (inverted
    for pairs stored in swapped order), PLANAR_OR_PANORAMIC is resolved, and read_two_view_geometry returns it."""
    rng = np.random.default_rng(77)
    images = synth.multiview_scene(rng, num_images=6, n_feats=512)
    for im in images:
        im["prior"] = True
    db = tmp_path / "pose.db"
    ids = colmap_db.create(db, images)
    pycolmap.match_exhaustive(db, matching_options={"block_size": 2},
                              verification_options={"compute_relative_pose": True})
    blocks = pycolmap._pycolmap._exhaustive_blocks(ids, 2)
    exp_m, exp_t = expected_rows(images, ids, blocks, prior=True, tvg_kw=dict(compute_relative_pose=1))
    assert compare(db, exp_m, exp_t) >= 5
    moved = [e for e in exp_t.values() if len(e["inl"]) and not np.array_equal(e["qvec"], [1, 0, 0, 0])]
    assert len(moved) >= 5 and any(e.get("H_approx") for e in moved)
    d = pycolmap.Database(db)
    pid, e = next((p, e) for p, e in exp_t.items() if len(e["inl"]) and not e.get("H_approx"))
    id1, id2 = pycolmap.Database.pair_id_to_image_pair(pid)
    g = d.read_two_view_geometry(id1, id2)
    np.testing.assert_array_equal(g.cam2_from_cam1.rotation.quat, e["qvec"][[1, 2, 3, 0]])
    np.testing.assert_array_equal(g.cam2_from_cam1.translation, e["tvec"])
    gi = d.read_two_view_geometry(id2, id1)                     # asked in the other order: inverted on the way out
    np.testing.assert_allclose(gi.cam2_from_cam1.rotation.quat, e["qvec"][[1, 2, 3, 0]] * [-1, -1, -1, 1], atol=1e-12)


@pytest.mark.parametrize("prior", [False, True])
def test_match_exhaustive_with_guided_matching(tmp_path, prior):
    """SiftMatchingOptions.guided_matching: verified pairs are matched again under the geometric filter
    and the guided matches become the stored inlier matches."""
    rng = np.random.default_rng(40 + int(prior))
    images = synth.multiview_scene(rng, num_images=5, n_feats=500, num_landmarks=700)
    for im in images:
        im["prior"] = prior
    db = tmp_path / "guided.db"
    ids = colmap_db.create(db, images)
    pycolmap.match_exhaustive(db, sift_options=dict(guided_matching=True))
    stats = pycolmap.last_run_stats()
    assert stats["pairs_guided"] > 0
    blocks = pycolmap._pycolmap._exhaustive_blocks(ids, 50)
    exp_m, exp_t = expected_rows(images, ids, blocks, prior=prior, guided=True)
    assert compare(db, exp_m, exp_t) >= 5


def test_match_sequential_loop_detection(tmp_path):
    """loop_detection=True: after the sequential pairs, every 10th image (in name order) is matched
    against the loop_detection_num_images images that share the most matches with it on the first
    loop_detection_max_num_features descriptors (feature voting)."""
    rng = np.random.default_rng(50)
    images = synth.multiview_scene(rng, num_images=21, n_feats=300, num_landmarks=420)
    db = tmp_path / "loop.db"
    ids = colmap_db.create(db, images)
    L, K = 96, 3
    pycolmap.match_sequential(db, matching_options=dict(overlap=2, quadratic_overlap=False, loop_detection=True,
                                                        loop_detection_num_images=K,
                                                        loop_detection_max_num_features=L))
    st = pycolmap.last_run_stats()
    assert st["loop_queries"] == 3 and st["loop_pairs_scored"] == 3 * 20
    blocks = pycolmap._pycolmap._sequential_blocks(ids, 2, False)
    # the voting, restated with the CPU oracle matcher
    loop_blocks = []
    for qi in range(0, len(ids), 10):
        votes = [(len(o.match(images[qi]["descriptors"][:L], images[j]["descriptors"][:L])), j)
                 for j in range(len(ids)) if j != qi]
        ranked = sorted(votes, key=lambda v: -v[0])           # stable: ties keep the candidate order
        loop_blocks.append([(ids[qi], ids[j]) for n, j in ranked[:K] if n > 0])
    assert sum(len(b) for b in loop_blocks) >= 6
    exp_m, exp_t = expected_rows(images, ids, list(blocks) + loop_blocks)
    assert compare(db, exp_m, exp_t) >= 10
    # the loop pairs are pairs the sequential pass did not cover
    seq_pairs = {colmap_db.pair_id(a, b) for blk in blocks for a, b in blk if a != b}
    assert any(colmap_db.pair_id(a, b) not in seq_pairs for blk in loop_blocks for a, b in blk)


def _dump_tables(db_path):
    import sqlite3
    con = sqlite3.connect(db_path)
    out = [list(con.execute(f"SELECT * FROM {t} ORDER BY pair_id")) for t in ("matches", "two_view_geometries")]
    con.close()
    return out


@pytest.mark.parametrize("gpu_index", ["0,0", "0,0,0", "-1"])
def test_multi_context_run_writes_the_same_database(tmp_path, gpu_index):
    """SiftMatchingOptions.gpu_index = "0,1,.." / "-1": one context + one host thread per entry, the group's pairs
    dealt to them by work.  Two (three) contexts on the one device of the test box must leave exactly the rows the
    single-context run leaves - every column of both tables, byte for byte, rows in the same order."""
    rng = np.random.default_rng(8)
    images = synth.multiview_scene(rng, num_images=5, n_feats=384) + synth.multiview_scene(rng, num_images=4, n_feats=700)
    for k, im in enumerate(images):
        im["name"] = f"im{k:03d}.jpg"
        im["prior"] = k % 2 == 0
    dumps = {}
    for tag, idx in (("single", "0"), ("multi", gpu_index)):
        db = tmp_path / f"{tag}.db"
        colmap_db.create(db, images)
        pycolmap.match_exhaustive(db, sift_options={"gpu_index": idx, "guided_matching": True},
                                  matching_options={"block_size": 4}, verification_options={"compute_relative_pose": True})
        dumps[tag] = _dump_tables(db)
        assert pycolmap.last_run_stats()["pairs_matched"] == 36
    assert dumps["single"] == dumps["multi"]
    assert len(dumps["single"][0]) == 36 and sum(r[4] >= 2 for r in dumps["single"][1]) >= 10
    # stored matches + verify_matches through several contexts
    db = tmp_path / "stored.db"
    ids = colmap_db.create(db, images)
    for a, b in ((0, 1), (1, 2), (5, 6), (7, 8), (0, 8)):
        colmap_db.write_matches(db, ids[a], ids[b], o.match(images[a]["descriptors"], images[b]["descriptors"]))
    pairs_txt = tmp_path / "pairs.txt"
    pairs_txt.write_text("\n".join(f"{images[a]['name']} {images[b]['name']}" for a, b in ((0, 1), (1, 2), (5, 6), (7, 8), (0, 8))))
    pycolmap.verify_matches(db, pairs_txt)
    assert pycolmap.last_run_stats()["pairs_verified"] >= 4


def test_bad_gpu_index_raises(tmp_path):
    rng = np.random.default_rng(9)
    images = synth.multiview_scene(rng, num_images=3, n_feats=128)
    db = tmp_path / "db.db"
    colmap_db.create(db, images)
    with pytest.raises(ValueError):
        pycolmap.match_exhaustive(db, sift_options={"gpu_index": "0,gpu1"})
    with pytest.raises(ValueError):
        pycolmap.match_exhaustive(db, sift_options={"gpu_index": "0,99"})      # no such device
    assert colmap_db.read_all(db) == ({}, {})
