#!/usr/bin/env python3
"""End-to-end run of the pycolmap-level pipeline on a synthetic COLMAP database (BASELINE config 3
shape): pycolmap_amd.match_exhaustive = SQLite read -> upload -> match -> verify -> SQLite write.

    python tools/pipeline_bench.py --images 500 --feats 4096

Prints one JSON line with the wall time and the controller's own breakdown (device time of the
match and verification stages, SQLite time).  This is a plumbing / scale check of the host layer; the
judged kernel numbers come from bench.py.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--prior", type=int, default=1, help="cameras have a prior focal length (E + F + H)")
    ap.add_argument("--block-size", type=int, default=50)
    args = ap.parse_args()

    import colmap_db
    import pycolmap_amd as pc
    from pycolmap_amd import synth

    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    images = synth.multiview_scene(rng, num_images=args.images, n_feats=args.feats,
                                   num_landmarks=int(args.feats * 1.5))
    for im in images:
        im["prior"] = bool(args.prior)
    t_gen = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as d:
        db = os.path.join(d, "bench.db")
        t0 = time.perf_counter()
        colmap_db.create(db, images)
        t_create = time.perf_counter() - t0
        t0 = time.perf_counter()
        pc.match_exhaustive(db, matching_options=pc.ExhaustiveMatchingOptions(block_size=args.block_size))
        wall = time.perf_counter() - t0
        st = dict(pc.last_run_stats())
        dbo = pc.Database(db)
        out = dict(images=args.images, feats=args.feats, pairs=args.images * (args.images - 1) // 2,
                   wall_s=wall, generate_s=t_gen, create_db_s=t_create, stats=st,
                   matched_pairs=dbo.num_matched_image_pairs, verified_pairs=dbo.num_verified_image_pairs,
                   num_matches=dbo.num_matches, num_inlier_matches=dbo.num_inlier_matches,
                   db_bytes=os.path.getsize(db))
        # second run: everything exists -> resume path skips every pair
        t0 = time.perf_counter()
        pc.match_exhaustive(db, matching_options=pc.ExhaustiveMatchingOptions(block_size=args.block_size))
        out["rerun_wall_s"] = time.perf_counter() - t0
        out["rerun_stats"] = dict(pc.last_run_stats())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
