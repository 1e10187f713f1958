#!/usr/bin/env python3
"""Reference pin kit: run the REAL pycolmap (0.6.x on COLMAP 3.9.1) on this repo's seeded scenes and store what it
returns, so that the oracle and the HIP path can be compared with the reference itself instead of with each other.

The reference holds no golden vectors for the matching / verification path and cannot be built in the development
container, so the parity of this repo is "unpinned": bit-exact to oracle/tvg_oracle.cc, which is a
restatement.  This script is the route to a real pin.  On any machine with pycolmap installed:

    pip install pycolmap==0.6.1            # wheels bundle COLMAP 3.9.1
    python tests/golden/make_reference_golden.py          # -> tests/golden/reference_v1.npz
    python -m pytest tests/test_reference_golden.py -q    # CPU: the oracle against the file;  -m gpu: the HIP path

It needs numpy, the module under test and pycolmap_amd/synth.py (pure numpy, loaded by path through scenes.py) - nothing
of this repo has to be built.  `--module pycolmap_amd` runs the same calls against this repo's API-compatible module
(a dry run that proves the script and the consuming tests execute; its output is NOT a reference and is never committed
as reference_v1.npz).

What is recorded per scene (floating-point results as raw uint64 bit patterns):
  * fundamental_matrix_estimation, homography_matrix_estimation on the matched points
    (/root/reference/pycolmap/estimators/fundamental_matrix.h:17-39, homography_matrix.h:16-37: SetPRNGSeed(0), one
    LO-RANSAC), essential_matrix_estimation (essential_matrix.h:19-83) - with TwoViewGeometryOptions().ransac's values
    and with pycolmap's own RANSACOptions() defaults;
  * estimate_two_view_geometry and estimate_calibrated_two_view_geometry (two_view_geometry.h:95-151) with the scene's
    options.  These bindings do NOT reseed COLMAP's thread-local generator, so each call is preceded by a
    fundamental_matrix_estimation on a single point: SetPRNGSeed(0) runs, LORANSAC::Estimate returns before drawing;
  * squared_sampson_error (two_view_geometry.h:161-175) of the matched points under the estimated F;
  * homography_decomposition (geometry/homography_matrix.h:13-40) on six seeded planes (`record_homography_decomposition`).

The MATCHER pin (round 4; `record_matching`, keys "mg_*").  COLMAP's default CPU matcher is FLANN - approximate,
randomised, not a parity target - but `SiftCPUFeatureMatcher::MatchGuided` is an EXACT brute-force scan
(ComputeSiftDistanceMatrix with the geometric filter, FindBestMatchesBruteForce), so it is the one place where the
reference itself exercises M1-M3 on the CPU.  The kit writes a tiny seeded COLMAP database
(tests/colmap_db.py: sqlite3 only), runs
    pycolmap.match_exhaustive(db, sift_options=SiftMatchingOptions(guided_matching=True), device=cpu)
(/root/reference/pycolmap/pipeline/match_features.h:95-98, 219-226) and records, for every stored two-view geometry,
the configuration, the reference's OWN F / E / H and its guided inlier matches, next to the descriptors and float32
keypoints.  The consuming tests re-run only the guided match - with the reference's models as input - through
oracle_match_guided and through amc_match_guided_pairs and demand identical rows: whatever FLANN and the reference's
RANSAC did upstream is input, not something to reproduce.
"""
import argparse
import importlib
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import scenes  # noqa: E402

TVG_RANSAC = dict(max_error=4.0, min_inlier_ratio=0.25, confidence=0.999, dyn_num_trials_multiplier=3.0,
                  min_num_trials=100, max_num_trials=10000)      # TwoViewGeometryOptions().ransac (C++ defaults)
PY_RANSAC = dict(max_error=4.0, min_inlier_ratio=0.01, confidence=0.9999, dyn_num_trials_multiplier=3.0,
                 min_num_trials=1000, max_num_trials=100000)     # pycolmap.RANSACOptions() (optim/bindings.h:10-18)
TVG_OPTION_KEYS = ("min_num_inliers", "min_E_F_inlier_ratio", "max_H_inlier_ratio", "watermark_min_inlier_ratio",
                   "watermark_border_size", "detect_watermark", "multiple_ignore_watermark", "force_H_use",
                   "compute_relative_pose", "multiple_models")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1).view(np.uint64).copy()


def make_camera(pc, cam, width, height, prior):
    model, params = cam
    c = pc.Camera(model=model, width=int(width), height=int(height), params=[float(x) for x in params])
    c.has_prior_focal_length = bool(prior)
    return c


def ransac_options(pc, kw):
    o = pc.RANSACOptions()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def tvg_options(pc, scene_opts, **extra):
    o = pc.TwoViewGeometryOptions()
    for k, v in TVG_RANSAC.items():
        setattr(o.ransac, k, v)
    for k, v in {**scene_opts, **extra}.items():
        if k in scenes.RANSAC_OPTION_KEYS:
            setattr(o.ransac, k, type(getattr(o.ransac, k))(v))
        else:
            setattr(o, k, type(getattr(o, k))(v))
    return o


def reseed(pc):
    """SetPRNGSeed(0) without consuming a draw: one correspondence is fewer than any minimal sample."""
    assert pc.fundamental_matrix_estimation(np.zeros((1, 2)), np.ones((1, 2))) is None


def record_ransac(out, tag, res, key, n):
    out[f"{tag}_success"] = np.int64(res is not None)
    out[f"{tag}_model"] = bits(res[key]) if res is not None else np.zeros(9, np.uint64)
    out[f"{tag}_num_inliers"] = np.int64(res["num_inliers"] if res is not None else 0)
    out[f"{tag}_mask"] = np.asarray(res["inliers"], dtype=bool) if res is not None else np.zeros(n, bool)


def record_tvg(out, tag, g):
    out[f"{tag}_config"] = np.int64(int(g.config))
    for k in "EFH":
        out[f"{tag}_{k}"] = bits(getattr(g, k))
    out[f"{tag}_inlier_matches"] = np.ascontiguousarray(g.inlier_matches, dtype=np.uint32).reshape(-1, 2)
    out[f"{tag}_tri_angle"] = bits([g.tri_angle])
    out[f"{tag}_quat_xyzw"] = bits(g.cam2_from_cam1.rotation.quat)
    out[f"{tag}_tvec"] = bits(g.cam2_from_cam1.translation)


def record_all(pc, module_name, limit=0, verbose=True):
    """Every call of the kit against module `pc`; returns the dict that main() stores."""
    is_reference = module_name == "pycolmap" and not hasattr(pc, "has_hip")
    out = {"module": np.array(module_name), "is_reference": np.int64(is_reference),
           "module_version": np.array(str(getattr(pc, "__version__", "?"))),
           "colmap_version": np.array(str(getattr(pc, "COLMAP_version", "?")))}
    names = []
    for n, sc in enumerate(scenes.all_scenes()):
        if limit and n >= limit:
            break
        name = sc["name"]
        names.append(name)
        m = sc["matches"]
        p1, p2 = sc["pts1"][m[:, 0]], sc["pts2"][m[:, 1]]
        cam1 = make_camera(pc, sc["cam1"], sc["width"], sc["height"], sc["prior"])
        cam2 = make_camera(pc, sc["cam2"], sc["width"], sc["height"], sc["prior"])
        for oname, okw in (("tvgopts", TVG_RANSAC), ("pyopts", PY_RANSAC)):
            if oname == "pyopts" and not name.startswith("golden"):
                continue                      # the 100,000-trial defaults on the golden scenes only (minutes on a CPU)
            ro = ransac_options(pc, okw)
            record_ransac(out, f"{name}_F_{oname}", pc.fundamental_matrix_estimation(p1, p2, ro), "F", len(m))
            record_ransac(out, f"{name}_H_{oname}", pc.homography_matrix_estimation(p1, p2, ro), "H", len(m))
            if len(m):
                rE = pc.essential_matrix_estimation(p1, p2, cam1, cam2, ro)
                record_ransac(out, f"{name}_E_{oname}", rE, "E", len(m))
                if rE is not None:
                    out[f"{name}_E_{oname}_quat_xyzw"] = bits(rE["cam2_from_cam1"].rotation.quat)
                    out[f"{name}_E_{oname}_tvec"] = bits(rE["cam2_from_cam1"].translation)
        for pose in (0, 1):
            reseed(pc)
            g = pc.estimate_two_view_geometry(cam1, sc["pts1"], cam2, sc["pts2"], m,
                                              tvg_options(pc, sc["opts"], compute_relative_pose=bool(pose)))
            record_tvg(out, f"{name}_tvg_p{pose}", g)
        reseed(pc)
        gc = pc.estimate_calibrated_two_view_geometry(cam1, sc["pts1"], cam2, sc["pts2"], m, tvg_options(pc, sc["opts"]))
        record_tvg(out, f"{name}_ctvg", gc)
        Fm = np.frombuffer(out[f"{name}_F_tvgopts_model"].tobytes(), dtype=np.float64).reshape(3, 3)
        out[f"{name}_sampson"] = bits(pc.squared_sampson_error(p1, p2, Fm)) if len(m) else np.zeros(0, np.uint64)
        if verbose:
            print(name, len(m), "matches: tvg config", int(g.config), "inlier matches", len(g.inlier_matches), flush=True)
    out["names"] = np.array(names)
    return out


MG_IMAGES, MG_FEATS, MG_SEED = 6, 320, 424242


def matching_scene():
    """The tiny seeded image set of the matcher pin (pure numpy: pycolmap_amd/synth.py loaded by path)."""
    synth = scenes.synth
    rng = np.random.default_rng(MG_SEED)
    images = synth.multiview_scene(rng, num_images=MG_IMAGES, n_feats=MG_FEATS, num_landmarks=480)
    for k, im in enumerate(images):
        im["name"] = f"mg{k:03d}.jpg"
        im["prior"] = True
    return images


def record_matching(pc, module_name, workdir=None):
    """match_exhaustive(guided_matching=True) on the tiny database through module `pc`; the rows it leaves."""
    import tempfile
    sys.path.insert(0, str(scenes.ROOT / "tests"))
    import colmap_db
    images = matching_scene()
    out = {"mg_num_images": np.int64(len(images))}
    for k, im in enumerate(images):
        out[f"mg_desc_{k}"] = np.ascontiguousarray(im["descriptors"], dtype=np.uint8)
        out[f"mg_kp_{k}"] = np.ascontiguousarray(im["keypoints"], dtype=np.float32)
    sift = pc.SiftMatchingOptions()
    sift.guided_matching = True
    for key in ("max_ratio", "max_distance", "cross_check"):
        out[f"mg_opt_{key}"] = np.float64(float(getattr(sift, key)))
    # MatchGuided's threshold is the verification's TwoViewGeometryOptions.ransac.max_error (4 px by default)
    out["mg_opt_max_error"] = np.float64(float(pc.TwoViewGeometryOptions().ransac.max_error))
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        db = str(Path(d) / "mg.db")
        ids = colmap_db.create(db, images)
        is_amd = hasattr(pc, "has_hip") or module_name == "pycolmap_amd"
        # the real module: the CPU matcher (MatchGuided's brute force is what is being pinned); this repo's module has
        # no CPU path and raises for device=cpu - its dry run takes the device it has
        pc.match_exhaustive(db, sift_options=sift, device=pc.Device.auto if is_amd else pc.Device.cpu)
        _, tvgs = colmap_db.read_all(db)
    pairs = []
    for a in range(len(images)):
        for b in range(a + 1, len(images)):
            g = tvgs.get(colmap_db.pair_id(ids[a], ids[b]))
            if g is None or g["F"] is None:
                continue
            n = len(pairs)
            pairs.append((a, b))
            out[f"mg_config_{n}"] = np.int64(g["config"])
            for key in "FEH":
                out[f"mg_{key}_{n}"] = bits(g[key] if g[key] is not None else np.zeros((3, 3)))
            out[f"mg_inlier_matches_{n}"] = np.ascontiguousarray(g["inlier_matches"], dtype=np.uint32).reshape(-1, 2)
    out["mg_pairs"] = np.array(pairs, dtype=np.int64).reshape(-1, 2)
    return out


HD_SEEDS = (7001, 7002, 7003, 7004, 7005, 7006)


def hd_scene(seed):
    """A seeded plane seen by two PINHOLE cameras: (H in pixels, K1, K2, the correspondences in camera coordinates).
    Pure numpy.  Seed 7006 is a pure rotation (one candidate); the second image's points carry a little noise, so
    not every correspondence passes the cheirality test of every candidate."""
    rng = np.random.default_rng(seed)
    K1 = np.array([[900.0, 0, 640], [0, 910.0, 360], [0, 0, 1]])
    K2 = np.array([[1100.0, 0, 600], [0, 1090.0, 400], [0, 0, 1]])
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    ang = float(rng.uniform(0.05, 0.5))
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    t = np.zeros(3) if seed == HD_SEEDS[-1] else rng.normal(size=3) * 0.3
    nrm = np.array([0.1, -0.2, 1.0])
    nrm /= np.linalg.norm(nrm)
    d = 5.0
    n = int(rng.integers(20, 200))
    rays = np.column_stack([rng.uniform(-0.4, 0.4, (n, 2)), np.ones(n)])
    X = rays * (d / (rays @ nrm))[:, None]
    X2 = X @ R.T + t
    p1 = X[:, :2] / X[:, 2:]
    p2 = X2[:, :2] / X2[:, 2:] + rng.normal(0, 1e-3, (n, 2))
    H = K2 @ (R + np.outer(t, nrm) / d) @ np.linalg.inv(K1) * float(rng.uniform(0.5, 2.0))
    return H, K1, K2, np.ascontiguousarray(p1), np.ascontiguousarray(p2)


def record_homography_decomposition(pc):
    """homography_decomposition (/root/reference/pycolmap/geometry/homography_matrix.h:13-40 = COLMAP's
    PoseFromHomographyMatrix) on the seeded planes: keys "hd_*" - deterministic, no RANSAC, so against a real pin the
    results agree to rounding (Eigen's SVD versus the oracle's Jacobi in the normalisation and the triangulation)."""
    out = {"hd_seeds": np.array(HD_SEEDS, dtype=np.int64)}
    for s in HD_SEEDS:
        H, K1, K2, p1, p2 = hd_scene(s)
        r = pc.homography_decomposition(H, K1, K2, p1, p2)
        out[f"hd_R_{s}"] = bits(np.asarray(r["R"], dtype=np.float64))
        out[f"hd_t_{s}"] = bits(np.asarray(r["t"], dtype=np.float64))
        out[f"hd_n_{s}"] = bits(np.asarray(r["n"], dtype=np.float64))
        out[f"hd_points3D_{s}"] = bits(np.asarray(r["points3D"], dtype=np.float64).reshape(-1, 3))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--module", default="pycolmap", help="module to record (the real pycolmap; pycolmap_amd for a dry run)")
    ap.add_argument("--out", default=str(HERE / "reference_v1.npz"))
    ap.add_argument("--limit", type=int, default=0, help="only the first N scenes (dry runs)")
    ap.add_argument("--no-matching", action="store_true", help="skip the matcher pin (match_exhaustive with guided matching)")
    args = ap.parse_args()
    if args.module == "pycolmap_amd":
        sys.path.insert(0, str(scenes.ROOT))
    pc = importlib.import_module(args.module)
    out = record_all(pc, args.module, args.limit)
    if not args.no_matching:
        out.update(record_matching(pc, args.module))
    if hasattr(pc, "homography_decomposition"):
        out.update(record_homography_decomposition(pc))
    if not int(out["is_reference"]) and Path(args.out).name == "reference_v1.npz":
        raise SystemExit("refusing to write reference_v1.npz from a module that is not the real pycolmap: pass --out")
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "from", args.module, "(reference)" if int(out["is_reference"]) else "(NOT a reference: dry run)")


if __name__ == "__main__":
    main()
