"""The Database surface of the reference (/root/reference/pycolmap/scene/database.h:9-46: open / close, cameras and images
read and written as objects, DatabaseTransaction) on a COLMAP-schema SQLite file - host logic, no GPU."""
import sqlite3

import numpy as np
import pytest

import pycolmap_amd as pycolmap


def make(tmp_path):
    db = pycolmap.Database(tmp_path / "a.db")
    cam = pycolmap.Camera(model="SIMPLE_RADIAL", width=640, height=480, params=[500.0, 320.0, 240.0, 0.01])
    cid = db.write_camera(cam)
    return db, cam, cid


def test_cameras_round_trip(tmp_path):
    db, cam, cid = make(tmp_path)
    assert cid == 1 and db.num_cameras == 1 and db.exists_camera(1) and not db.exists_camera(2)
    cam7 = pycolmap.Camera(model="PINHOLE", width=64, height=48, params=[50.0, 51.0, 32.0, 24.0], camera_id=7,
                           has_prior_focal_length=True)
    assert db.write_camera(cam7, use_camera_id=True) == 7
    back = db.read_camera(7)
    assert (back.camera_id, back.model.name, back.width, back.height, list(back.params), back.has_prior_focal_length) == \
        (7, "PINHOLE", 64, 48, [50.0, 51.0, 32.0, 24.0], True)
    assert [c.camera_id for c in db.read_all_cameras()] == [1, 7]
    assert list(db.read_camera(1).params) == [500.0, 320.0, 240.0, 0.01]
    with pytest.raises(ValueError, match="exists already"):
        db.write_camera(cam7, use_camera_id=True)
    with pytest.raises(ValueError, match="does not exist"):
        db.read_camera(99)
    # the row is what COLMAP's own tools would read: model id, float64 params blob
    con = sqlite3.connect(tmp_path / "a.db")
    model, w, h, blob, prior = con.execute("SELECT model, width, height, params, prior_focal_length FROM cameras WHERE camera_id = 7").fetchone()
    assert (model, w, h, prior) == (1, 64, 48, 1) and np.frombuffer(blob, np.float64).tolist() == [50.0, 51.0, 32.0, 24.0]


def test_images_round_trip_and_priors(tmp_path):
    db, cam, cid = make(tmp_path)
    im = pycolmap.Image(name="a.jpg", camera_id=cid)
    assert repr(im) == 'Image(image_id=Invalid, camera_id=1, name="a.jpg", triangulated=0/0)'
    assert im.has_camera() and not pycolmap.Image().has_camera()
    assert np.isnan(im.cam_from_world_prior.translation).all()           # unknown until set
    im.cam_from_world_prior = pycolmap.Rigid3d(pycolmap.Rotation3d([0.0, 0.0, 0.0, 1.0]), [1.0, 2.0, 3.0])
    iid = db.write_image(im)
    assert iid == 1 and db.exists_image(1) and db.num_images == 1
    im2 = pycolmap.Image("b.jpg", [[1.0, 2.0], [3.0, 4.0]], pycolmap.Rigid3d(), cid, 42)
    assert im2.num_points2D() == 2 and im2.image_id == 42
    assert db.write_image(im2, use_image_id=True) == 42
    r = db.read_image(1)
    assert (r.image_id, r.camera_id, r.name) == (1, cid, "a.jpg")
    assert list(r.cam_from_world_prior.translation) == [1.0, 2.0, 3.0] and list(r.cam_from_world_prior.rotation.quat) == [0.0, 0.0, 0.0, 1.0]
    r2 = db.read_image_with_name("b.jpg")
    assert r2.image_id == 42 and np.isnan(r2.cam_from_world_prior.translation).all()     # NULL columns
    assert [i.name for i in db.read_all_images()] == ["a.jpg", "b.jpg"]
    con = sqlite3.connect(tmp_path / "a.db")
    assert con.execute("SELECT prior_qw, prior_qx, prior_tx, prior_tz FROM images WHERE image_id = 1").fetchone() == (1.0, 0.0, 1.0, 3.0)
    assert con.execute("SELECT prior_qw, prior_tx FROM images WHERE image_id = 42").fetchone() == (None, None)
    with pytest.raises(ValueError, match="exists already"):
        db.write_image(im2, use_image_id=True)
    with pytest.raises(ValueError, match="does not exist"):
        db.read_image(7)
    with pytest.raises(ValueError, match="does not exist"):
        db.read_image_with_name("nope")
    with pytest.raises(RuntimeError, match="FOREIGN KEY"):
        db.write_image(pycolmap.Image(name="c.jpg", camera_id=555))     # no such camera
    with pytest.raises(ValueError):
        im.camera_id = 0xFFFFFFFF
    # per-image counts
    assert db.num_keypoints_for_image(1) == 0
    db.write_keypoints(1, np.zeros((5, 4), np.float32))
    db.write_descriptors(1, np.zeros((5, 128), np.uint8))
    assert db.num_keypoints_for_image(1) == 5 and db.num_descriptors_for_image(1) == 5 and db.num_descriptors_for_image(42) == 0


def test_transactions(tmp_path):
    db, cam, cid = make(tmp_path)
    t = pycolmap.DatabaseTransaction(db)                # COLMAP's scope guard: END when the object goes away
    db.write_image(pycolmap.Image(name="a", camera_id=cid))
    other = sqlite3.connect(tmp_path / "a.db")
    assert other.execute("SELECT COUNT(*) FROM images").fetchone()[0] == 0        # not committed yet
    del t
    assert other.execute("SELECT COUNT(*) FROM images").fetchone()[0] == 1
    with pycolmap.DatabaseTransaction(db):
        db.write_image(pycolmap.Image(name="b", camera_id=cid))
    assert db.num_images == 2
    with pytest.raises(KeyError):
        with pycolmap.DatabaseTransaction(db):
            db.write_image(pycolmap.Image(name="c", camera_id=cid))
            raise KeyError("rolled back")
    assert db.num_images == 2


def test_open_and_close(tmp_path):
    db, cam, cid = make(tmp_path)
    db.close()
    db.close()                                          # idempotent
    with pytest.raises(RuntimeError, match="closed"):
        db.num_images
    with pytest.raises(RuntimeError, match="closed"):
        db.write_camera(cam)
    db.open(tmp_path / "b.db")                          # a new file: COLMAP's tables are created
    assert db.num_cameras == 0
    db.open(tmp_path / "a.db")
    assert db.num_cameras == 1


def test_spatial_pairs_from_written_priors(tmp_path):
    """The priors match_spatial reads are the ones write_image stores."""
    db, cam, cid = make(tmp_path)
    ids = []
    for k in range(4):
        im = pycolmap.Image(name=f"{k}.jpg", camera_id=cid)
        if k < 3:
            im.cam_from_world_prior = pycolmap.Rigid3d(pycolmap.Rotation3d(), [10.0 * k + 1.0, 1.0, 0.0])
        ids.append(db.write_image(im))
    pri = [list(i.cam_from_world_prior.translation) for i in db.read_all_images()]
    o = pycolmap.SpatialMatchingOptions(is_gps=False, max_distance=15.0)
    blocks = pycolmap._pycolmap._spatial_blocks(ids, pri, o)
    assert [sorted(b) for b in blocks] == [[(1, 2)], [(2, 1), (2, 3)], [(3, 2)]]
