"""CPU tests of the C++/pybind11 host layer (pycolmap_amd._pycolmap): API surface, option
"dataclass" protocol, device/error conventions, pair generators, SQLite schema round trips.
No GPU compute here (the drop-in boundary)."""
import copy
import itertools
import os
import pickle
from pathlib import Path

import numpy as np
import pytest

import colmap_db
import pycolmap_amd as pycolmap
from pycolmap_amd import synth

ROOT = Path(__file__).resolve().parent.parent


def test_option_defaults_match_the_reference():
    s = pycolmap.SiftMatchingOptions()
    assert (s.max_ratio, s.max_distance, s.cross_check, s.max_num_matches, s.guided_matching,
            s.num_threads, s.gpu_index) == (0.8, 0.7, True, 32768, False, -1, "-1")
    assert pycolmap.ExhaustiveMatchingOptions().block_size == 50
    q = pycolmap.SequentialMatchingOptions()
    assert (q.overlap, q.quadratic_overlap, q.loop_detection) == (10, True, False)
    # RANSACOptions(): Python-side overrides (/root/reference/pycolmap/optim/bindings.h:10-18)
    r = pycolmap.RANSACOptions()
    assert (r.max_error, r.min_inlier_ratio, r.confidence, r.min_num_trials, r.max_num_trials) == \
        (4.0, 0.01, 0.9999, 1000, 100000)
    # TwoViewGeometryOptions(): py::init<>() of the C++ struct -> C++ RANSAC defaults
    t = pycolmap.TwoViewGeometryOptions()
    assert (t.min_num_inliers, t.min_E_F_inlier_ratio, t.max_H_inlier_ratio, t.detect_watermark) == \
        (15, 0.95, 0.8, True)
    assert (t.ransac.min_inlier_ratio, t.ransac.confidence, t.ransac.min_num_trials, t.ransac.max_num_trials) == \
        (0.25, 0.999, 100, 10000)


def test_dataclass_protocol():
    o = pycolmap.TwoViewGeometryOptions({"min_num_inliers": 20, "ransac": {"max_error": 2.0}})
    assert o.min_num_inliers == 20 and o.ransac.max_error == 2.0 and o.ransac.confidence == 0.999
    o2 = pycolmap.SiftMatchingOptions(max_ratio=0.7, cross_check=False)
    assert o2.max_ratio == 0.7 and o2.cross_check is False
    d = o.todict()
    assert d["ransac"]["max_error"] == 2.0 and d["force_H_use"] is False
    o.mergedict({"ransac": {"confidence": 0.99}})
    assert o.ransac.confidence == 0.99 and o.ransac.max_error == 2.0
    assert "min_num_inliers = 20" in o.summary() and "max_error = 2.0" in o.summary()
    with pytest.raises(ValueError):
        pycolmap.SiftMatchingOptions({"no_such_option": 1})
    with pytest.raises((TypeError, ValueError)):
        pycolmap.SiftMatchingOptions(max_ratio="abc")
    for c in (copy.copy(o), copy.deepcopy(o), pickle.loads(pickle.dumps(o))):
        assert c.todict() == o.todict()
    c = copy.deepcopy(o)
    c.ransac.max_error = 9.0
    o.min_num_inliers = 33
    assert c.min_num_inliers == 20


def test_device_enum_and_error_conventions(tmp_path):
    assert pycolmap.Device("auto") == pycolmap.Device.auto and pycolmap.Device("cuda") == pycolmap.Device.cuda
    with pytest.raises(ValueError):
        pycolmap.Device("tpu")
    with pytest.raises(ValueError, match="does not exist"):
        pycolmap.match_exhaustive(tmp_path / "missing.db")
    db = tmp_path / "a.db"
    colmap_db.create(db, [])
    with pytest.raises(ValueError, match="no CPU fallback"):
        pycolmap.match_exhaustive(db, device="cpu")           # implicit str -> Device
    with pytest.raises(ValueError, match="does not exist"):
        pycolmap.verify_matches(db, tmp_path / "nopairs.txt")
    with pytest.raises((TypeError, ValueError)):   # failed implicit dict -> options conversion
        pycolmap.match_exhaustive(str(db), sift_options={"bogus": 1})
    with pytest.raises(ValueError):
        pycolmap.match_vocabtree(db)
    assert pycolmap.has_cuda is True


@pytest.mark.parametrize("n,B", [(1, 2), (5, 2), (7, 3), (20, 50), (23, 5), (50, 50), (51, 50)])
def test_exhaustive_blocks_cover_every_pair_exactly_once(n, B):
    ids = list(range(10, 10 + n))
    blocks = pycolmap._pycolmap._exhaustive_blocks(ids, B)
    nb = (n + B - 1) // B
    assert len(blocks) == nb * nb
    flat = [tuple(p) for b in blocks for p in b]
    assert all(a != b for a, b in flat)
    unordered = [tuple(sorted(p)) for p in flat]
    assert len(set(unordered)) == len(unordered) == n * (n - 1) // 2
    assert set(unordered) == set(itertools.combinations(ids, 2))


def test_sequential_blocks_follow_colmap():
    ids = list(range(1, 21))
    blocks = pycolmap._pycolmap._sequential_blocks(ids, 4, True)
    assert len(blocks) == 20
    # image 1 (idx 0): linear offsets 0..3 and quadratic 1,2,4,8
    assert [tuple(p) for p in blocks[0]] == [(1, 1), (1, 2), (1, 2), (1, 3), (1, 3), (1, 5), (1, 4), (1, 9)]
    assert [tuple(p) for p in blocks[18]] == [(19, 19), (19, 20), (19, 20)]
    lin = pycolmap._pycolmap._sequential_blocks(ids, 3, False)
    assert [tuple(p) for p in lin[5]] == [(6, 6), (6, 7), (6, 8)]


def test_database_binding_reads_colmap_schema(tmp_path):
    rng = np.random.default_rng(0)
    imgs = synth.multiview_scene(rng, num_images=3, n_feats=40, num_landmarks=60)
    db_path = tmp_path / "s.db"
    ids = colmap_db.create(db_path, imgs)
    m = np.array([[0, 5], [3, 7], [9, 1]], np.uint32)
    colmap_db.write_matches(db_path, ids[0], ids[1], m)
    colmap_db.write_matches(db_path, ids[2], ids[0], m)   # stored swapped
    db = pycolmap.Database(db_path)
    assert (db.num_cameras, db.num_images, db.num_keypoints, db.num_descriptors) == (3, 3, 120, 120)
    assert db.num_matched_image_pairs == 2 and db.num_matches == 6 and db.num_verified_image_pairs == 0
    pid = pycolmap.Database.image_pair_to_pair_id(ids[0], ids[1])
    assert pid == colmap_db.pair_id(ids[0], ids[1]) == pycolmap.Database.image_pair_to_pair_id(ids[1], ids[0])
    assert pycolmap.Database.pair_id_to_image_pair(pid) == (ids[0], ids[1])
    np.testing.assert_array_equal(db.read_matches(ids[0], ids[1]), m)
    np.testing.assert_array_equal(db.read_matches(ids[1], ids[0]), m[:, ::-1])
    np.testing.assert_array_equal(db.read_matches(ids[2], ids[0]), m)
    assert db.exists_matches(ids[0], ids[2]) and not db.exists_inlier_matches(ids[0], ids[1])
    g = db.read_two_view_geometry(ids[0], ids[1])
    assert g.config == pycolmap.TwoViewGeometryConfiguration.UNDEFINED and g.inlier_matches.shape == (0, 2)
    # Database::Open creates a missing file with COLMAP's empty tables (like the reference's Database(path))
    fresh = pycolmap.Database(tmp_path / "fresh.db")
    assert fresh.num_images == 0 and fresh.num_matches == 0 and fresh.num_cameras == 0
    del fresh
    import sqlite3
    tables = {r[0] for r in sqlite3.connect(tmp_path / "fresh.db").execute("SELECT name FROM sqlite_master WHERE type='table'")}
    assert {"cameras", "images", "keypoints", "descriptors", "matches", "two_view_geometries"} <= tables
    # keypoints / descriptors / matches accessors (unbound in the reference)
    np.testing.assert_array_equal(db.read_keypoints(ids[1]), np.ascontiguousarray(imgs[1]["keypoints"], np.float32))
    np.testing.assert_array_equal(db.read_descriptors(ids[2]), imgs[2]["descriptors"])
    assert db.read_keypoints(999).shape == (0, 0) and db.read_descriptors(999).shape == (0, 128)
    assert db.exists_keypoints(ids[0]) and not db.exists_keypoints(999) and db.exists_descriptors(ids[0])
    db.write_matches(ids[2], ids[1], m)                          # descending ids: stored swapped, read back as given
    np.testing.assert_array_equal(db.read_matches(ids[2], ids[1]), m)
    np.testing.assert_array_equal(db.read_matches(ids[1], ids[2]), m[:, ::-1])
    assert db.num_matched_image_pairs == 3
    db.delete_matches(ids[1], ids[2])
    assert not db.exists_matches(ids[2], ids[1]) and db.num_matched_image_pairs == 2
    with pytest.raises(ValueError):
        db.write_matches(ids[0], ids[1], np.zeros((3, 3), np.uint32))
    with pytest.raises(RuntimeError):
        db.write_keypoints(ids[0], np.zeros((4, 2), np.float32))     # the row exists: UNIQUE constraint, like COLMAP's CHECK
    with pytest.raises(ValueError):
        db.write_descriptors(ids[0], np.zeros((4, 64), np.uint8))
    import sqlite3
    con = sqlite3.connect(db_path)
    con.execute("INSERT INTO images(name, camera_id) VALUES ('extra.jpg', 1)")
    new_id = con.execute("SELECT image_id FROM images WHERE name = 'extra.jpg'").fetchone()[0]
    con.commit()
    con.close()
    kp = rng.uniform(0, 100, size=(7, 4)).astype(np.float32)
    de = rng.integers(0, 256, size=(7, 128), dtype=np.uint8)
    db.write_keypoints(new_id, kp)
    db.write_descriptors(new_id, de)
    np.testing.assert_array_equal(db.read_keypoints(new_id), kp)
    np.testing.assert_array_equal(db.read_descriptors(new_id), de)
    assert db.num_keypoints == 127 and db.num_descriptors == 127
    # bulk-write mode: rollback journal while a controller appends, WAL (COLMAP's mode) again afterwards
    assert db.set_bulk_write_mode(True) == "truncate"
    db.write_matches(ids[1], ids[2], m)
    assert db.set_bulk_write_mode(False) == "wal"
    np.testing.assert_array_equal(db.read_matches(ids[1], ids[2]), m)
    db2 = pycolmap.Database(db_path)
    db2.set_bulk_write_mode(True)
    db2.write_matches(ids[0], ids[2], m) if not db2.exists_matches(ids[0], ids[2]) else None
    del db2                                                     # closing restores WAL
    import gc
    gc.collect()
    con = sqlite3.connect(db_path)
    assert con.execute("PRAGMA journal_mode").fetchone()[0] == "wal"
    con.close()


# ---- Camera + single-pair estimator bindings (surface only: no GPU here) --------------------------
def test_camera_binding_surface():
    import pycolmap_amd as pc
    c = pc.Camera.create(3, "SIMPLE_PINHOLE", 1000.0, 1600, 1200)   # Camera::CreateFromModelId
    assert c.camera_id == 3 and c.model == pc.CameraModelId.SIMPLE_PINHOLE
    assert c.params.tolist() == [1000.0, 800.0, 600.0] and c.params_info == "f, cx, cy"
    assert c.mean_focal_length() == 1000.0 and c.focal_length == 1000.0
    assert c.cam_from_img_threshold(4.0) == 4.0 / 1000.0
    assert not c.has_prior_focal_length
    np.testing.assert_array_equal(c.calibration_matrix(), [[1000, 0, 800], [0, 1000, 600], [0, 0, 1]])
    p = pc.Camera(model="PINHOLE", width=640, height=480, params=[500.0, 520.0, 320.0, 240.0],
                  has_prior_focal_length=True)
    assert p.mean_focal_length() == 510.0 and p.focal_length_x == 500.0 and p.focal_length_y == 520.0
    assert p.principal_point_x == 320.0 and p.principal_point_y == 240.0 and p.has_prior_focal_length
    assert p.verify_params()
    p.params = [1.0, 2.0, 3.0]
    assert not p.verify_params()
    assert pc.CameraModelId.OPENCV == 4 and pc.CameraModelId["RADIAL"] == 3   # COLMAP's model ids
    with pytest.raises(ValueError):
        pc.Camera(model="PINHOLE", width=1, height=1, params=[1.0])
    with pytest.raises(ValueError):
        pc.Camera(model="NOT_A_MODEL", width=1, height=1, params=[])
    assert "SIMPLE_PINHOLE" in repr(c)


def test_estimator_bindings_argument_checks():
    """Pre-flight failures are ValueErrors, as THROW_CHECK_EQ is in the reference; with matching sizes
    the call reaches the accelerator and, on a box without one, fails loudly (no CPU fallback)."""
    import pycolmap_amd as pc
    a, b = np.zeros((5, 2)), np.zeros((6, 2))
    for fn in (pc.fundamental_matrix_estimation, pc.homography_matrix_estimation):
        with pytest.raises(ValueError, match="size"):
            fn(a, b)
        with pytest.raises(ValueError):
            fn(np.zeros((5, 3)), np.zeros((5, 3)))
    cam = pc.Camera.create(1, "SIMPLE_PINHOLE", 1000.0, 1600, 1200)
    with pytest.raises(ValueError, match="size"):
        pc.essential_matrix_estimation(a, b, cam, cam)
    with pytest.raises(ValueError, match="size"):
        pc.estimate_two_view_geometry(cam, a, cam, b)           # matches=None needs equal sizes
    with pytest.raises(ValueError, match="size"):
        pc.squared_sampson_error(a, b, np.eye(3))
    with pytest.raises(ValueError):
        pc.squared_sampson_error(a, a, np.eye(2))
    with pytest.raises(TypeError):
        pc.estimate_two_view_geometry_pose()                    # five positional arguments, like the reference
    g = pc.TwoViewGeometry()                                    # defaults of cam2_from_cam1 / tri_angle
    assert g.tri_angle == 0.0 and np.array_equal(g.cam2_from_cam1.rotation.quat, [0, 0, 0, 1])
    assert np.array_equal(g.cam2_from_cam1.translation, [0, 0, 0])
    assert np.array_equal(g.cam2_from_cam1.matrix(), np.c_[np.eye(3), np.zeros(3)])
    g.invert()                                                  # Invert() of the default geometry: still the identity pose
    assert np.allclose(g.cam2_from_cam1.matrix(), np.c_[np.eye(3), np.zeros(3)]) and g.inlier_matches.shape == (0, 2)
    assert set(g.todict()) == {"config", "E", "F", "H", "cam2_from_cam1", "inlier_matches", "tri_angle"}
    r = pc.Rotation3d([0.0, 0.0, np.sin(0.25), np.cos(0.25)])   # 0.5 rad about z
    assert np.allclose(r.matrix(), [[np.cos(0.5), -np.sin(0.5), 0], [np.sin(0.5), np.cos(0.5), 0], [0, 0, 1]])
    assert abs(r.norm() - 1.0) < 1e-15 and "Rotation3d" in repr(r) and "Rigid3d" in repr(pc.Rigid3d(r, [1, 2, 3]))
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="amc_ctx_create"):
            pc.fundamental_matrix_estimation(np.zeros((8, 2)), np.zeros((8, 2)))


def test_gpu_index_parsing():
    """SiftMatchingOptions.gpu_index (/root/reference/pycolmap/pipeline/match_features.h:76-81): a comma-separated
    device list, one context per entry; anything else raises instead of being silently ignored."""
    import pycolmap_amd as pc
    parse = pc._pycolmap._parse_gpu_index
    assert parse("0") == [0] and parse("0,1,2,3") == [0, 1, 2, 3] and parse(" 1 , 1 ") == [1, 1]
    for bad in ("", "x", "0,,1", "0,", "-2", "0,-1", "1.5", "0 1"):
        with pytest.raises(ValueError):
            parse(bad)
    with pytest.raises(RuntimeError):      # "-1" = all devices: there is none in the CPU container
        parse("-1")


def test_option_protocol_matches_the_reference_helpers():
    """/root/reference/pycolmap/helpers.h:159-283: summary(write_type), todict(recursive), defaults in docstrings,
    dict / kwargs construction and implicit conversion, copy and pickle."""
    import copy
    import pickle

    import pycolmap_amd as pc
    o = pc.TwoViewGeometryOptions(min_num_inliers=20, ransac=dict(max_error=2.0))
    lines = o.summary().splitlines()
    assert lines[0] == "TwoViewGeometryOptions:" and "    min_num_inliers = 20" in lines
    assert "    ransac: RANSACOptions:" in lines and "        max_error = 2.0" in lines           # nested, four deeper
    typed = o.summary(write_type=True).splitlines()
    assert "    min_num_inliers: int = 20" in typed and "        max_error: float = 2.0" in typed
    assert isinstance(o.todict()["ransac"], dict) and o.todict(recursive=True)["ransac"]["max_error"] == 2.0
    assert isinstance(o.todict(recursive=False)["ransac"], pc.RANSACOptions)
    assert "(int, default: 15)" in pc.TwoViewGeometryOptions.min_num_inliers.__doc__
    assert "(float, default: 0.8)" in pc.SiftMatchingOptions.max_ratio.__doc__
    assert pc.SiftMatchingOptions.max_ratio.__doc__.startswith("Maximum distance ratio")
    assert "(str, default: -1)" in pc.SiftMatchingOptions.gpu_index.__doc__
    for c in (copy.copy(o), copy.deepcopy(o), pickle.loads(pickle.dumps(o)), pc.TwoViewGeometryOptions(o.todict())):
        assert c.todict() == o.todict()
    with pytest.raises(ValueError, match=r"^\[module\.cc:\d+\] Check Failed: ExistsFile\(db_path\) : File .* does not exist\.$"):
        pc.match_exhaustive("/nonexistent/database.db")
    with pytest.raises(ValueError, match=r"^\[estimators\.h:\d+\] Check Failed: .*\(5 vs\. 6\)$"):
        pc.fundamental_matrix_estimation(np.zeros((5, 2)), np.zeros((6, 2)))


def test_logging_surface_and_pycolmap_alias(tmp_path, capfd):
    """/root/reference/pycolmap/main.cc:39-89 (logging) and :91-118 (the module is called pycolmap)."""
    import pycolmap
    import pycolmap_amd as pc
    assert pycolmap.match_exhaustive is pc.match_exhaustive and pycolmap.TwoViewGeometryOptions is pc.TwoViewGeometryOptions
    assert pycolmap.has_cuda and isinstance(pycolmap.COLMAP_version, str) and isinstance(pycolmap.COLMAP_build, str)
    with pytest.raises(AttributeError, match="outside pycolmap_amd's scope"):
        pycolmap.incremental_mapping
    lg = pycolmap.logging
    assert lg.Level.INFO == lg.INFO and int(lg.WARNING) == 1 and int(lg.FATAL) == 3
    lg.set_log_destination(lg.INFO, str(tmp_path / "info.log"))
    lg.info("hello from the test")
    lg.warning("careful")
    err = capfd.readouterr().err
    assert "hello from the test" in err and err.lstrip().startswith("I") and "test_logging_surface_and_pycolmap_alias" in err
    text = (tmp_path / "info.log").read_text()
    assert "hello from the test" in text and "careful" in text
    lg.minloglevel = 2
    lg.info("suppressed")
    assert "suppressed" not in capfd.readouterr().err
    lg.minloglevel = 0
    with pytest.raises(RuntimeError):
        lg.fatal("stop")
    lg.set_log_destination(lg.INFO, "")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_concurrent_database_use_under_sanitizers(tmp_path, sanitizer):
    """The writer-thread / reader overlap of the grouped runners (controller.cc RunGrouped) and the rollback of an
    abandoned transaction, compiled with TSan and with ASan + UBSan (tests/shim/db_threads.cc); any report fails."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    inc = next((p for p in ("/usr/include", "/opt/conda/include") if (Path(p) / "sqlite3.h").exists()), None)
    if inc is None:
        pytest.skip("no sqlite3.h")
    exe = tmp_path / "db_threads"
    cmd = ["g++", "-O1", "-g", "-std=c++17", f"-fsanitize={sanitizer}", f"-I{inc}", str(ROOT / "tests" / "shim" / "db_threads.cc"),
           str(ROOT / "pycolmap_amd" / "csrc" / "host" / "database.cc"), "-l:libsqlite3.so.0", "-lpthread", "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(exe), str(tmp_path / "san.db")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok rows="), (r.returncode, r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("seed", [1, 2])
def test_slot_arena_bookkeeping_under_asan(tmp_path, seed):
    """The sub-allocator behind the image slots' device memory (pycolmap_amd/csrc/slot_arena.h: best fit, split, merge
    with free neighbours, idle slabs released at trim) over a fake raw allocator, 20,000 random alloc / free / trim
    steps under ASan + UBSan (tests/shim/slot_arena_fuzz.cc): live blocks disjoint and aligned, free blocks tile the rest
    of every slab with nothing left unmerged, the failed-slab fallback takes the exact size, nothing leaks."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "arena_fuzz"
    b = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                        str(ROOT / "tests" / "shim" / "slot_arena_fuzz.cc"), "-o", str(exe)], capture_output=True, text=True)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe), str(seed)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "slot arena fuzz ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_out_of_scope_matcher_keeps_its_option_class():
    """match_vocabtree is out of scope, but a script written for the reference can still construct its options
    (reference match_features.h:177-214) and gets the reason at the call; match_spatial's options are live."""
    import pycolmap_amd as pc
    sp = pc.SpatialMatchingOptions(max_num_neighbors=20)
    assert sp.todict() == dict(is_gps=True, ignore_z=True, max_num_neighbors=20, max_distance=100.0)
    vt = pc.VocabTreeMatchingOptions({"num_images": 7})
    assert vt.num_images == 7 and vt.num_checks == 256 and vt.vocab_tree_path == ""
    with pytest.raises(ValueError, match="outside pycolmap_amd's scope"):
        pc.match_vocabtree("nowhere.db", matching_options=vt)


def test_camera_parameter_helpers():
    """Camera::FocalLengthIdxs / PrincipalPointIdxs / ExtraParamsIdxs, HasBogusParams, Rescale, SetParamsFromString
    (/root/reference/pycolmap/scene/camera.h) - metadata and arithmetic on the parameter vector."""
    import pycolmap_amd as pc
    c = pc.Camera(model="OPENCV", width=640, height=480, params=[500, 510, 320, 240, 0.1, 0.01, 0.001, 0.002])
    assert (c.focal_length_idxs(), c.principal_point_idxs(), c.extra_params_idxs()) == ([0, 1], [2, 3], [4, 5, 6, 7])
    s = pc.Camera(model="SIMPLE_RADIAL", width=100, height=50, params=[80, 50, 25, 0.1])
    assert (s.focal_length_idxs(), s.principal_point_idxs(), s.extra_params_idxs()) == ([0], [1, 2], [3])
    assert pc.Camera(model="PINHOLE", width=1, height=1, params=[1, 1, 0, 0]).extra_params_idxs() == []
    assert not c.has_bogus_params(0.1, 10, 1.0)
    assert c.has_bogus_params(0.1, 10, 0.05)            # |k1| = 0.1 > 0.05
    assert c.has_bogus_params(1.0, 10, 1.0)             # 500 / 640 < 1
    off = pc.Camera(model="PINHOLE", width=10, height=10, params=[10, 10, 11, 5])
    assert off.has_bogus_params(0.1, 10, 1.0)           # principal point outside the image
    c.rescale(0.5)
    assert (c.width, c.height, list(c.params[:4])) == (320, 240, [250.0, 255.0, 160.0, 120.0])
    c.rescale(1280, 960)
    assert (c.width, c.height, list(c.params[:4])) == (1280, 960, [1000.0, 1020.0, 640.0, 480.0])
    s.rescale(0.33)                                     # 33 x 17 (rounded): the scales differ per axis, one focal length takes their mean
    assert (s.width, s.height) == (33, 17)
    np.testing.assert_allclose(list(s.params), [80 * (0.33 + 0.34) / 2, 50 * 0.33, 25 * 0.34, 0.1], rtol=1e-15)
    with pytest.raises(ValueError):
        s.rescale(0.0)
    assert s.set_params_from_string("1, 2,3 ,4") and list(s.params) == [1.0, 2.0, 3.0, 4.0]
    assert not s.set_params_from_string("1,2") and not s.set_params_from_string("a,b,c,d") and list(s.params) == [1.0, 2.0, 3.0, 4.0]


def test_rotation_and_rigid_algebra():
    """Rotation3d / Rigid3d operators of /root/reference/pycolmap/geometry/bindings.h:24-104 (Eigen's quaternion algebra,
    colmap/geometry/rigid3.h), against scipy's rotations and plain matrix arithmetic."""
    import pycolmap_amd as pc
    from scipy.spatial.transform import Rotation as SR, Slerp
    rng = np.random.default_rng(5)
    for _ in range(20):
        qa, qb = SR.random(random_state=int(rng.integers(1 << 30))), SR.random(random_state=int(rng.integers(1 << 30)))
        a, b = pc.Rotation3d(qa.as_quat()), pc.Rotation3d(qb.as_quat())
        np.testing.assert_allclose(a.matrix(), qa.as_matrix(), atol=1e-14)
        np.testing.assert_allclose((a * b).matrix(), qa.as_matrix() @ qb.as_matrix(), atol=1e-14)
        v = rng.normal(size=3)
        np.testing.assert_allclose(a * v, qa.as_matrix() @ v, atol=1e-14)
        P = rng.normal(size=(7, 3))
        np.testing.assert_allclose(a * P, P @ qa.as_matrix().T, atol=1e-14)
        np.testing.assert_allclose((a * a.inverse()).matrix(), np.eye(3), atol=1e-14)
        assert abs(a.angle() - min(qa.magnitude(), 2 * np.pi - qa.magnitude())) < 1e-12
        assert abs(a.angle_to(b) - (qa * qb.inv()).magnitude()) < 1e-12
        # the other constructors
        np.testing.assert_allclose(pc.Rotation3d(qa.as_matrix()).matrix(), qa.as_matrix(), atol=1e-14)
        np.testing.assert_allclose(pc.Rotation3d(qa.as_rotvec()).matrix(), qa.as_matrix(), atol=1e-14)
        # rigid transforms
        ta, tb = rng.normal(size=3), rng.normal(size=3)
        A, B = pc.Rigid3d(a, ta), pc.Rigid3d(b, tb)
        Ma = np.vstack([A.matrix(), [0, 0, 0, 1]])
        Mb = np.vstack([B.matrix(), [0, 0, 0, 1]])
        np.testing.assert_allclose(np.vstack([(A * B).matrix(), [0, 0, 0, 1]]), Ma @ Mb, atol=1e-13)
        np.testing.assert_allclose(np.vstack([A.inverse().matrix(), [0, 0, 0, 1]]), np.linalg.inv(Ma), atol=1e-13)
        np.testing.assert_allclose(A * v, qa.as_matrix() @ v + ta, atol=1e-14)
        np.testing.assert_allclose(A * P, P @ qa.as_matrix().T + ta, atol=1e-14)
        np.testing.assert_allclose(pc.Rigid3d(A.matrix()).matrix(), A.matrix(), atol=1e-14)
        tn = ta / np.linalg.norm(ta)
        tx = np.array([[0, -tn[2], tn[1]], [tn[2], 0, -tn[0]], [-tn[1], tn[0], 0]])
        np.testing.assert_allclose(A.essential_matrix(), tx @ qa.as_matrix(), atol=1e-14)
        s = float(rng.uniform())
        I = pc.Rigid3d.interpolate(A, B, s)
        want = Slerp([0, 1], SR.concatenate([qa, qb]))([s])[0].as_matrix()
        np.testing.assert_allclose(I.rotation.matrix(), want, atol=1e-12)
        np.testing.assert_allclose(I.translation, ta + (tb - ta) * s, atol=1e-14)
    r = pc.Rotation3d([0.0, 0.0, 0.6, 0.8])
    r.quat = [0.0, 0.0, 3.0, 4.0]
    r.normalize()
    np.testing.assert_allclose(r.quat, [0, 0, 0.6, 0.8], atol=1e-15)
    assert pc.Rotation3d([0.0, 0.0, 0.0]).quat.tolist() == [0.0, 0.0, 0.0, 1.0]     # zero axis-angle: identity
    with pytest.raises(ValueError):
        pc.Rotation3d(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        pc.Rigid3d(np.zeros((3, 3)))
