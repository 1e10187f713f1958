"""pycolmap_amd — MI355X-native exhaustive/sequential SIFT matching + two-view verification behind
the pycolmap API.

    import pycolmap_amd as pycolmap
    pycolmap.match_exhaustive(database_path)        # same signature as the reference
    pycolmap.verify_matches(database_path, pairs_path)
    pycolmap.fundamental_matrix_estimation(points2D1, points2D2)   # ... and the other estimator bindings

The compiled host layer (`_pycolmap`, C++/pybind11) drives `libamc.so` (HIP kernels behind a C ABI,
include/amc.h).  There is no CPU fallback: importing works anywhere, computing needs a gfx950 GPU.
"""
from __future__ import annotations

__version__ = "0.1.0"

try:  # the compiled host layer; absent only before `python -m pycolmap_amd.build`
    from ._pycolmap import (  # noqa: F401
        COLMAP_build, COLMAP_version, Camera, CameraModelId, Database, DatabaseTransaction, Device, ExhaustiveMatchingOptions, Image, RANSACOptions, Rigid3d,
        Rotation3d,
        SequentialMatchingOptions, SiftMatchingOptions, SpatialMatchingOptions, TwoViewGeometry, TwoViewGeometryConfiguration,
        VocabTreeMatchingOptions,
        TwoViewGeometryOptions, essential_matrix_estimation, estimate_calibrated_two_view_geometry,
        estimate_two_view_geometry, estimate_two_view_geometry_pose, fundamental_matrix_estimation, has_cuda,
        has_hip, homography_decomposition, homography_matrix_estimation, last_run_stats, logging, match_exhaustive, match_sequential,
        match_spatial, match_vocabtree, squared_sampson_error, verify_matches,
    )
    _HOST_LAYER_ERROR = None
except ImportError as _e:  # pragma: no cover - exercised only on an unbuilt tree
    _HOST_LAYER_ERROR = _e

    def __getattr__(name):
        # submodules (`from pycolmap_amd import build`, `_capi`, `synth`, ...) must stay importable on an unbuilt tree:
        # the import machinery asks the package for the attribute first and imports the submodule on AttributeError
        import importlib.util
        if name.startswith("__") or importlib.util.find_spec(f"{__name__}.{name}") is not None:
            raise AttributeError(name)
        raise ImportError(
            f"pycolmap_amd.{name}: the compiled host layer is missing ({_HOST_LAYER_ERROR}); run "
            "`python -m pycolmap_amd.build` (needs hipcc + g++). pycolmap_amd has no Python/CPU fallback.")
