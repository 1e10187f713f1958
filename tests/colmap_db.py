"""Writer/reader for COLMAP's SQLite schema (mirrors COLMAP's
scripts/python/database.py) used to build test fixtures with Python's sqlite3."""
from __future__ import annotations

import sqlite3

import numpy as np

MAX_IMAGE_ID = 2 ** 31 - 1
SCHEMA = """
CREATE TABLE IF NOT EXISTS cameras (camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, model INTEGER NOT NULL,
    width INTEGER NOT NULL, height INTEGER NOT NULL, params BLOB, prior_focal_length INTEGER NOT NULL);
CREATE TABLE IF NOT EXISTS images (image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, name TEXT NOT NULL UNIQUE,
    camera_id INTEGER NOT NULL, prior_qw REAL, prior_qx REAL, prior_qy REAL, prior_qz REAL, prior_tx REAL,
    prior_ty REAL, prior_tz REAL, CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < 2147483647),
    FOREIGN KEY(camera_id) REFERENCES cameras(camera_id));
CREATE UNIQUE INDEX IF NOT EXISTS index_name ON images(name);
CREATE TABLE IF NOT EXISTS keypoints (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
    cols INTEGER NOT NULL, data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS descriptors (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
    cols INTEGER NOT NULL, data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS matches (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
    cols INTEGER NOT NULL, data BLOB);
CREATE TABLE IF NOT EXISTS two_view_geometries (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
    cols INTEGER NOT NULL, data BLOB, config INTEGER NOT NULL, F BLOB, E BLOB, H BLOB, qvec BLOB, tvec BLOB);
"""


def pair_id(id1: int, id2: int) -> int:
    a, b = (id2, id1) if id1 > id2 else (id1, id2)
    return MAX_IMAGE_ID * a + b


def create(path, images):
    """images: list of dict(name, keypoints [n,c] float32, descriptors [n,128] uint8, model, width, height,
    params, prior, prior_t = (tx, ty, tz) location prior or absent = NULL columns). One camera per image.
    Returns the image ids (1-based, like COLMAP)."""
    con = sqlite3.connect(path)
    con.executescript(SCHEMA)
    ids = []
    for im in images:
        cur = con.execute("INSERT INTO cameras(model, width, height, params, prior_focal_length) VALUES (?,?,?,?,?)",
                          (im.get("model", 1), im.get("width", 1600), im.get("height", 1200),
                           np.asarray(im.get("params", (1200.0, 1200.0, 800.0, 600.0)), np.float64).tobytes(),
                           int(im.get("prior", False))))
        cam_id = cur.lastrowid
        tx, ty, tz = im.get("prior_t", (None, None, None))
        cur = con.execute("INSERT INTO images(name, camera_id, prior_tx, prior_ty, prior_tz) VALUES (?, ?, ?, ?, ?)",
                          (im["name"], cam_id, tx, ty, tz))
        iid = cur.lastrowid
        ids.append(iid)
        kp = np.ascontiguousarray(im["keypoints"], np.float32)
        con.execute("INSERT INTO keypoints VALUES (?,?,?,?)", (iid, kp.shape[0], kp.shape[1], kp.tobytes()))
        d = np.ascontiguousarray(im["descriptors"], np.uint8)
        con.execute("INSERT INTO descriptors VALUES (?,?,?,?)", (iid, d.shape[0], 128, d.tobytes()))
    con.commit()
    con.close()
    return ids


def write_matches(path, id1, id2, matches):
    m = np.ascontiguousarray(matches, np.uint32).reshape(-1, 2)
    if id1 > id2:
        m = np.ascontiguousarray(m[:, ::-1])
    con = sqlite3.connect(path)
    con.execute("INSERT INTO matches VALUES (?,?,?,?)", (pair_id(id1, id2), m.shape[0], 2, m.tobytes()))
    con.commit()
    con.close()


def read_all(path):
    """Returns (matches {pair_id: [n,2] uint32}, tvgs {pair_id: dict})."""
    con = sqlite3.connect(path)
    matches = {}
    for pid, rows, cols, data in con.execute("SELECT pair_id, rows, cols, data FROM matches"):
        matches[pid] = np.frombuffer(data or b"", np.uint32).reshape(rows, 2).copy()
    tvgs = {}
    for pid, rows, cols, data, config, F, E, H, q, t in con.execute(
            "SELECT pair_id, rows, cols, data, config, F, E, H, qvec, tvec FROM two_view_geometries"):
        def mat(b):
            return np.frombuffer(b, np.float64).reshape(3, 3).copy() if b else None
        tvgs[pid] = dict(inlier_matches=np.frombuffer(data or b"", np.uint32).reshape(rows, 2).copy(),
                         config=config, F=mat(F), E=mat(E), H=mat(H),
                         qvec=np.frombuffer(q, np.float64).copy() if q else None,
                         tvec=np.frombuffer(t, np.float64).copy() if t else None)
    con.close()
    return matches, tvgs
