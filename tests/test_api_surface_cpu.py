"""The names the reference registers for the match + verify path and its neighbours (collected once from
/root/reference/pycolmap: pipeline/match_features.h, estimators/{two_view_geometry,essential_matrix,fundamental_matrix,
homography_matrix}.h, geometry/{bindings,homography_matrix}.h, scene/{database,camera}.h, optim/bindings.h, utils.h) exist
on pycolmap_amd with the members a script written for the reference touches.  Hard-coded: the reference tree is not
shipped with the tests."""
import pycolmap_amd as pc

MODULE = ["match_exhaustive", "match_sequential", "match_spatial", "match_vocabtree", "verify_matches",
          "SiftMatchingOptions", "ExhaustiveMatchingOptions", "SequentialMatchingOptions", "SpatialMatchingOptions",
          "VocabTreeMatchingOptions", "TwoViewGeometryOptions", "TwoViewGeometryConfiguration", "TwoViewGeometry",
          "RANSACOptions", "fundamental_matrix_estimation", "essential_matrix_estimation", "homography_matrix_estimation",
          "estimate_two_view_geometry", "estimate_calibrated_two_view_geometry", "estimate_two_view_geometry_pose",
          "squared_sampson_error", "homography_decomposition", "Database", "DatabaseTransaction", "Camera", "CameraModelId",
          "Image", "Rotation3d", "Rigid3d", "Device", "logging", "has_cuda", "COLMAP_version", "COLMAP_build"]
MEMBERS = {
    "SiftMatchingOptions": ["num_threads", "gpu_index", "max_ratio", "max_distance", "cross_check", "max_num_matches", "guided_matching"],
    "ExhaustiveMatchingOptions": ["block_size"],
    "SequentialMatchingOptions": ["overlap", "quadratic_overlap", "loop_detection", "loop_detection_num_images",
                                  "loop_detection_num_nearest_neighbors", "loop_detection_num_checks",
                                  "loop_detection_num_images_after_verification", "loop_detection_max_num_features", "vocab_tree_path"],
    "SpatialMatchingOptions": ["is_gps", "ignore_z", "max_num_neighbors", "max_distance"],
    "TwoViewGeometryOptions": ["min_num_inliers", "min_E_F_inlier_ratio", "max_H_inlier_ratio", "watermark_min_inlier_ratio",
                               "watermark_border_size", "detect_watermark", "multiple_ignore_watermark", "force_H_use",
                               "compute_relative_pose", "multiple_models", "ransac"],
    "RANSACOptions": ["max_error", "min_inlier_ratio", "confidence", "dyn_num_trials_multiplier", "min_num_trials", "max_num_trials"],
    "TwoViewGeometry": ["config", "E", "F", "H", "cam2_from_cam1", "inlier_matches", "tri_angle", "invert"],
    "Database": ["open", "close", "num_cameras", "num_images", "num_keypoints", "num_keypoints_for_image", "num_descriptors",
                 "num_descriptors_for_image", "num_matches", "num_inlier_matches", "num_matched_image_pairs",
                 "num_verified_image_pairs", "image_pair_to_pair_id", "pair_id_to_image_pair", "read_camera",
                 "read_all_cameras", "read_image", "read_image_with_name", "read_all_images", "read_two_view_geometry",
                 "write_camera", "write_image"],
    "Camera": ["camera_id", "model", "width", "height", "params", "params_info", "has_prior_focal_length", "focal_length",
               "focal_length_x", "focal_length_y", "principal_point_x", "principal_point_y", "mean_focal_length",
               "focal_length_idxs", "principal_point_idxs", "extra_params_idxs", "calibration_matrix", "cam_from_img",
               "cam_from_img_threshold", "img_from_cam", "verify_params", "has_bogus_params", "params_to_string", "set_params_from_string",
               "rescale", "create"],
    "Image": ["image_id", "camera_id", "name", "cam_from_world", "cam_from_world_prior", "has_camera", "num_points2D"],
    "Rotation3d": ["quat", "matrix", "norm", "normalize", "angle", "angle_to", "inverse", "__mul__"],
    "Rigid3d": ["rotation", "translation", "matrix", "essential_matrix", "inverse", "interpolate", "__mul__"],
}
# known gaps, named so that closing one is a visible edit
NOT_BUILT = {}


def test_module_names():
    missing = [n for n in MODULE if not hasattr(pc, n)]
    assert not missing, missing


def test_class_members():
    missing = [f"{cls}.{m}" for cls, ms in MEMBERS.items() for m in ms if not hasattr(getattr(pc, cls), m)]
    assert not missing, missing
    for cls, ms in NOT_BUILT.items():
        assert all(not hasattr(getattr(pc, cls), m) for m in ms), "a gap was closed: move it to MEMBERS"


def test_configuration_enum_values():
    c = pc.TwoViewGeometryConfiguration
    assert [int(getattr(c, n)) for n in ("UNDEFINED", "DEGENERATE", "CALIBRATED", "UNCALIBRATED", "PLANAR", "PANORAMIC",
                                         "PLANAR_OR_PANORAMIC", "WATERMARK", "MULTIPLE")] == list(range(9))
    assert pc.Device.auto is not None and pc.Device.cpu is not None and pc.Device.cuda is not None
