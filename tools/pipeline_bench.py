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
    ap.add_argument("--sequential", action="store_true", help="match_sequential instead of match_exhaustive")
    ap.add_argument("--overlap", type=int, default=10)
    ap.add_argument("--loop", action="store_true", help="sequential matching with loop detection")
    ap.add_argument("--loop-features", type=int, default=256)
    ap.add_argument("--guided", action="store_true", help="SiftMatchingOptions.guided_matching")
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
        sift = pc.SiftMatchingOptions(guided_matching=args.guided)

        def run():
            if args.sequential or args.loop:
                pc.match_sequential(db, sift_options=sift, matching_options=pc.SequentialMatchingOptions(
                    overlap=args.overlap, loop_detection=args.loop, loop_detection_max_num_features=args.loop_features))
            else:
                pc.match_exhaustive(db, sift_options=sift,
                                    matching_options=pc.ExhaustiveMatchingOptions(block_size=args.block_size))

        t0 = time.perf_counter()
        run()
        wall = time.perf_counter() - t0
        st = dict(pc.last_run_stats())
        dbo = pc.Database(db)
        out = dict(images=args.images, feats=args.feats, mode=("sequential+loop" if args.loop else "sequential"
                                                                if args.sequential else "exhaustive"),
                   guided=args.guided,
                   wall_s=wall, generate_s=t_gen, create_db_s=t_create, stats=st,
                   matched_pairs=dbo.num_matched_image_pairs, verified_pairs=dbo.num_verified_image_pairs,
                   num_matches=dbo.num_matches, num_inlier_matches=dbo.num_inlier_matches,
                   db_bytes=os.path.getsize(db))
        # second run: everything exists -> resume path skips every pair
        t0 = time.perf_counter()
        run()
        out["rerun_wall_s"] = time.perf_counter() - t0
        out["rerun_stats"] = dict(pc.last_run_stats())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
