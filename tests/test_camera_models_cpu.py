"""Camera::CamFromImg / CamFromImgThreshold / CalibrationMatrix for COLMAP's eleven camera models
(/root/reference/pycolmap/scene/camera.h:136-165, /root/reference/pycolmap/estimators/essential_matrix.h:33-46).

Three independent statements are held against each other, no GPU needed:
  * the oracle's restatement of the upstream templates (oracle/tvg_oracle.cc),
  * a numpy forward projection written separately (pycolmap_amd/synth.py img_from_cam): the lift must invert it,
  * the product's camera_math.h compiled for the host (tests/shim): bit-identical to the oracle, so a GPU
    mismatch could only come from the device build of the same source."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from pycolmap_amd import synth
from test_tvg_math_host import shim  # noqa: F401  (fixture: builds tests/shim/_build/libtvgshim.so)

MODELS = list(synth.EXAMPLE_CAMERAS)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("model", MODELS)
def test_lift_inverts_forward_projection(model):
    rng = np.random.default_rng(7)
    prm = synth.EXAMPLE_CAMERAS[model]
    uv = rng.uniform(-0.65, 0.65, size=(4000, 2)) * np.array([1.0, 0.75])
    uv[:8] = [[0, 0], [1e-9, 0], [0, -1e-9], [0.3, 0], [0, 0.3], [-0.5, 0.4], [1e-4, 1e-4], [-1e-5, 2e-5]]
    xy = synth.img_from_cam(model, prm, uv)
    back = o.cam_from_img(o.make_camera(model, 1600, 1200, prm), xy)
    # IterativeUndistortion stops when the squared Newton step drops below 1e-10
    assert np.abs(back - uv).max() < 2e-9


def test_pinhole_lift_and_threshold_known_answers():
    cam = o.make_camera("PINHOLE", 1600, 1200, (1000.0, 500.0, 800.0, 600.0))
    np.testing.assert_array_equal(o.cam_from_img(cam, [[1800.0, 100.0]]), [[1.0, -1.0]])
    assert o.cam_from_img_threshold(cam, 3.0) == 3.0 / 750.0
    np.testing.assert_array_equal(o.calibration_matrix(cam), [[1000, 0, 800], [0, 500, 600], [0, 0, 1]])
    cam = o.make_camera("SIMPLE_RADIAL", 1600, 1200, (800.0, 10.0, 20.0, 0.0))   # k = 0: plain pinhole
    np.testing.assert_array_equal(o.cam_from_img(cam, [[810.0, 420.0]]), [[1.0, 0.5]])
    assert o.cam_from_img_threshold(cam, 4.0) == 4.0 / 800.0
    np.testing.assert_array_equal(o.calibration_matrix(cam), [[800, 0, 10], [0, 800, 20], [0, 0, 1]])


@pytest.mark.parametrize("model", MODELS)
def test_product_camera_math_matches_oracle_bit_for_bit(shim, model):
    rng = np.random.default_rng(11)
    prm = np.array(synth.EXAMPLE_CAMERAS[model], dtype=np.float64)
    mid = synth.CAMERA_MODEL_IDS[model]
    xy = np.ascontiguousarray(rng.uniform([0, 0], [1600, 1200], size=(3000, 2)).astype(np.float32).astype(np.float64))
    got = np.empty_like(xy)
    shim.shim_cam_from_img(C.c_int(mid), _p(prm), _p(xy), C.c_int(len(xy)), _p(got))
    want = o.cam_from_img(o.make_camera(model, 1600, 1200, prm), xy)
    np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
    shim.shim_cam_from_img_threshold.restype = C.c_double
    t = shim.shim_cam_from_img_threshold(C.c_int(mid), _p(prm), C.c_double(4.0))
    assert t == o.cam_from_img_threshold(o.make_camera(model, 1600, 1200, prm), 4.0)


def test_strong_distortion_still_converges():
    """Large radial coefficients and points in the image corners: the Newton iteration must still land
    on the pre-image (this is where 100 iterations and the numerical Jacobian matter)."""
    rng = np.random.default_rng(3)
    for model, prm in (("SIMPLE_RADIAL", (700.0, 800.0, 600.0, -0.25)), ("RADIAL", (700.0, 800.0, 600.0, -0.3, 0.09)),
                       ("OPENCV", (700.0, 710.0, 800.0, 600.0, -0.28, 0.08, 0.004, -0.003)),
                       ("OPENCV_FISHEYE", (500.0, 500.0, 800.0, 600.0, 0.05, -0.02, 0.005, -0.001))):
        uv = rng.uniform(-0.8, 0.8, size=(2000, 2)) * np.array([1.0, 0.75])   # inside the monotone range
        xy = synth.img_from_cam(model, prm, uv)
        back = o.cam_from_img(o.make_camera(model, 1600, 1200, prm), xy)
        assert np.abs(back - uv).max() < 1e-8, model
