// estimators.h — pycolmap's single-pair estimator bindings over libamc.so:
//   Camera (subset)                      /root/reference/pycolmap/scene/camera.h:44-160
//   fundamental_matrix_estimation        /root/reference/pycolmap/estimators/fundamental_matrix.h:17-49
//   homography_matrix_estimation         /root/reference/pycolmap/estimators/homography_matrix.h:16-47
//   essential_matrix_estimation          /root/reference/pycolmap/estimators/essential_matrix.h:19-102
//   estimate_two_view_geometry, estimate_calibrated_two_view_geometry, squared_sampson_error
//                                        /root/reference/pycolmap/estimators/two_view_geometry.h:95-175
// Same names, argument order and failure behaviour (None on a failed RANSAC, ValueError on
// mismatched sizes).  Every function runs on the GPU through the C ABI (amc_ransac_pairs,
// amc_verify_pairs, amc_squared_sampson_error); nothing is computed here.
//
// Differences, all documented in DESIGN.md section 7: results are deterministic (seed 0 per call,
// also for estimate_two_view_geometry, which in the reference inherits the calling thread's PRNG
// state); the relative pose (cam2_from_cam1, tri_angle) comes from the pose kernel (amc_pose_pairs).
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "controller.h"
#include "py_types.h"

namespace amchost {
using namespace pybind11::literals;

// COLMAP camera models: id, name, number of parameters, focal / principal indices
struct CameraModelInfo {
    int id;
    const char* name;
    int num_params;
    int num_focal;  // 1: f at params[0]; 2: fx, fy at params[0], params[1]
    const char* params_info;
};
inline const std::vector<CameraModelInfo>& CameraModels() {
    static const std::vector<CameraModelInfo> k = {
        {0, "SIMPLE_PINHOLE", 3, 1, "f, cx, cy"},
        {1, "PINHOLE", 4, 2, "fx, fy, cx, cy"},
        {2, "SIMPLE_RADIAL", 4, 1, "f, cx, cy, k"},
        {3, "RADIAL", 5, 1, "f, cx, cy, k1, k2"},
        {4, "OPENCV", 8, 2, "fx, fy, cx, cy, k1, k2, p1, p2"},
        {5, "OPENCV_FISHEYE", 8, 2, "fx, fy, cx, cy, k1, k2, k3, k4"},
        {6, "FULL_OPENCV", 12, 2, "fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6"},
        {7, "FOV", 5, 2, "fx, fy, cx, cy, omega"},
        {8, "SIMPLE_RADIAL_FISHEYE", 4, 1, "f, cx, cy, k"},
        {9, "RADIAL_FISHEYE", 5, 1, "f, cx, cy, k1, k2"},
        {10, "THIN_PRISM_FISHEYE", 12, 2, "fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1"},
    };
    return k;
}
inline const CameraModelInfo* FindCameraModel(int id) {
    for (const auto& m : CameraModels())
        if (m.id == id) return &m;
    return nullptr;
}

struct PyCamera {
    uint32_t camera_id = 0xFFFFFFFFu;  // kInvalidCameraId
    int model = -1;                    // CameraModelId::kInvalid
    uint64_t width = 0, height = 0;
    std::vector<double> params;
    bool has_prior_focal_length = false;

    const CameraModelInfo& Info() const {
        const CameraModelInfo* m = FindCameraModel(model);
        if (!m) throw py::value_error("Camera: invalid camera model");
        return *m;
    }
    // the parameter side of Camera::Rescale: principal point by axis, one focal length by the mean scale, two by axis
    void RescaleParams(double sx, double sy) {
        const int nf = Info().num_focal;
        params[nf] *= sx;
        params[nf + 1] *= sy;
        if (nf == 1) {
            params[0] *= (sx + sy) / 2.0;
        } else {
            params[0] *= sx;
            params[1] *= sy;
        }
    }
    void CheckParams() const {
        if (static_cast<int>(params.size()) != Info().num_params)
            throw py::value_error(std::string("Camera: model ") + Info().name + " takes " +
                                  std::to_string(Info().num_params) + " parameters (" + Info().params_info + ")");
    }
    double MeanFocalLength() const {
        CheckParams();
        return Info().num_focal == 1 ? params[0] : (params[0] + params[1]) / 2.0;
    }
    bool IsPinhole() const { return model == 0 || model == 1; }
    std::string Repr() const {
        std::ostringstream ss;
        const CameraModelInfo* m = FindCameraModel(model);
        ss << "Camera(camera_id=" << (camera_id == 0xFFFFFFFFu ? std::string("Invalid") : std::to_string(camera_id))
           << ", model=" << (m ? m->name : "Invalid") << ", width=" << width << ", height=" << height << ", params=[";
        for (size_t i = 0; i < params.size(); ++i) ss << (i ? ", " : "") << params[i];
        ss << "] (" << (m ? m->params_info : "?") << "))";
        return ss.str();
    }
};

inline int ParseCameraModel(const py::object& o) {
    if (py::isinstance<py::str>(o)) {
        const std::string s = o.cast<std::string>();
        for (const auto& m : CameraModels())
            if (s == m.name) return m.id;
        throw py::value_error("Invalid string value " + s + " for enum CameraModelId");
    }
    return o.cast<int>();
}

py::object CamFromImgPoints(const PyCamera& c, const py::object& points);  // below (uses the estimator context)
py::object ImgFromCamPoints(const PyCamera& c, const py::object& points);

inline void BindCamera(py::module_& m) {
    py::dict members("INVALID"_a = -1);
    for (const auto& cm : CameraModels()) members[py::str(cm.name)] = cm.id;
    py::object model_enum = py::module_::import("enum").attr("IntEnum")("CameraModelId", members);
    m.attr("CameraModelId") = model_enum;

    py::class_<PyCamera> cam(m, "Camera");
    cam.def(py::init<>())
        .def(py::init([](const py::object& model, uint64_t width, uint64_t height, std::vector<double> params,
                         uint32_t camera_id, bool has_prior) {
                 PyCamera c;
                 c.model = ParseCameraModel(model);
                 c.width = width;
                 c.height = height;
                 c.params = std::move(params);
                 c.camera_id = camera_id;
                 c.has_prior_focal_length = has_prior;
                 c.CheckParams();
                 return c;
             }),
             "model"_a, "width"_a, "height"_a, "params"_a, "camera_id"_a = 0xFFFFFFFFu,
             "has_prior_focal_length"_a = false)
        .def_static(
            "create",
            [](uint32_t camera_id, const py::object& model, double focal_length, uint64_t width, uint64_t height) {
                PyCamera c;  // Camera::CreateFromModelId: focal(s) = focal_length, principal point = centre, extras 0
                c.camera_id = camera_id;
                c.model = ParseCameraModel(model);
                c.width = width;
                c.height = height;
                const CameraModelInfo& info = c.Info();
                c.params.assign(info.num_params, 0.0);
                for (int i = 0; i < info.num_focal; ++i) c.params[i] = focal_length;
                c.params[info.num_focal] = width / 2.0;
                c.params[info.num_focal + 1] = height / 2.0;
                return c;
            },
            "camera_id"_a, "model"_a, "focal_length"_a, "width"_a, "height"_a)
        .def_readwrite("camera_id", &PyCamera::camera_id, "Unique identifier of the camera.")
        .def_property(
            "model", [model_enum](const PyCamera& c) { return model_enum(c.model); },
            [](PyCamera& c, const py::object& v) { c.model = ParseCameraModel(v); }, "Camera model.")
        .def_readwrite("width", &PyCamera::width, "Width of camera sensor.")
        .def_readwrite("height", &PyCamera::height, "Height of camera sensor.")
        .def_readwrite("has_prior_focal_length", &PyCamera::has_prior_focal_length)
        .def_property(
            "params",
            [](const PyCamera& c) {
                py::array_t<double> a(static_cast<py::ssize_t>(c.params.size()));
                if (!c.params.empty()) std::memcpy(a.mutable_data(), c.params.data(), c.params.size() * sizeof(double));
                return a;
            },
            [](PyCamera& c, const std::vector<double>& p) { c.params = p; }, "Camera parameters.")
        .def_property_readonly("params_info", [](const PyCamera& c) { return std::string(c.Info().params_info); })
        .def("mean_focal_length", &PyCamera::MeanFocalLength)
        .def_property(
            "focal_length",
            [](const PyCamera& c) {
                if (c.Info().num_focal != 1) throw py::value_error("Camera: model has two focal lengths");
                c.CheckParams();
                return c.params[0];
            },
            [](PyCamera& c, double f) {
                c.CheckParams();
                for (int i = 0; i < c.Info().num_focal; ++i) c.params[i] = f;
            })
        .def_property_readonly("focal_length_x", [](const PyCamera& c) { c.CheckParams(); return c.params[0]; })
        .def_property_readonly("focal_length_y",
                               [](const PyCamera& c) { c.CheckParams(); return c.params[c.Info().num_focal - 1]; })
        .def_property_readonly("principal_point_x",
                               [](const PyCamera& c) { c.CheckParams(); return c.params[c.Info().num_focal]; })
        .def_property_readonly("principal_point_y",
                               [](const PyCamera& c) { c.CheckParams(); return c.params[c.Info().num_focal + 1]; })
        .def("calibration_matrix",
             [](const PyCamera& c) {
                 c.CheckParams();
                 const int nf = c.Info().num_focal;
                 return Mat3({c.params[0], 0, c.params[nf], 0, c.params[nf - 1], c.params[nf + 1], 0, 0, 1});
             })
        .def("cam_from_img_threshold",
             [](const PyCamera& c, double threshold) { return threshold / c.MeanFocalLength(); },
             "Convert pixel threshold in image plane to world space.")
        .def("cam_from_img", &CamFromImgPoints, "image_points"_a,
             "Project point(s) in image plane to world / infinity: one point (2,) or an N x 2 array.")
        .def("img_from_cam", &ImgFromCamPoints, "world_points"_a,
             "Project point(s) from the camera frame to image coordinates: N x 2 points of the normalised image plane, or "
             "N x 3 points in front of the camera (divided by their depth first).")
        .def("verify_params",
             [](const PyCamera& c) {
                 const CameraModelInfo* mi = FindCameraModel(c.model);
                 return mi && static_cast<int>(c.params.size()) == mi->num_params;
             })
        .def("params_to_string",
             [](const PyCamera& c) {
                 std::ostringstream ss;
                 for (size_t i = 0; i < c.params.size(); ++i) ss << (i ? ", " : "") << c.params[i];
                 return ss.str();
             })
        // Camera::FocalLengthIdxs / PrincipalPointIdxs / ExtraParamsIdxs: positions in `params` (every model keeps the
        // order focal length(s), principal point, extra parameters)
        .def("focal_length_idxs",
             [](const PyCamera& c) {
                 std::vector<size_t> v;
                 for (int i = 0; i < c.Info().num_focal; ++i) v.push_back(i);
                 return v;
             },
             "Indices of focal length parameters in params property.")
        .def("principal_point_idxs",
             [](const PyCamera& c) {
                 const size_t nf = c.Info().num_focal;
                 return std::vector<size_t>{nf, nf + 1};
             },
             "Indices of principal point parameters in params property.")
        .def("extra_params_idxs",
             [](const PyCamera& c) {
                 std::vector<size_t> v;
                 for (int i = c.Info().num_focal + 2; i < c.Info().num_params; ++i) v.push_back(i);
                 return v;
             },
             "Indices of extra parameters in params property.")
        // Camera::HasBogusParams = bogus principal point || bogus focal length || bogus extra parameters
        .def("has_bogus_params",
             [](const PyCamera& c, double min_focal_length_ratio, double max_focal_length_ratio, double max_extra_param) {
                 c.CheckParams();
                 const int nf = c.Info().num_focal;
                 const double cx = c.params[nf], cy = c.params[nf + 1];
                 if (cx < 0 || cx > static_cast<double>(c.width) || cy < 0 || cy > static_cast<double>(c.height)) return true;
                 const double max_size = static_cast<double>(std::max(c.width, c.height));
                 for (int i = 0; i < nf; ++i) {
                     const double ratio = c.params[i] / max_size;
                     if (ratio < min_focal_length_ratio || ratio > max_focal_length_ratio) return true;
                 }
                 for (int i = nf + 2; i < c.Info().num_params; ++i)
                     if (std::abs(c.params[i]) > max_extra_param) return true;
                 return false;
             },
             "min_focal_length_ratio"_a, "max_focal_length_ratio"_a, "max_extra_param"_a,
             "Check whether camera has bogus parameters.")
        // Camera::Rescale(scale) / Rescale(width, height)
        .def("rescale",
             [](PyCamera& c, double scale) {
                 if (!(scale > 0.0)) throw py::value_error(CheckMessage(__FILE__, __LINE__, "scale > 0.0"));
                 c.CheckParams();
                 const double sx = std::round(scale * c.width) / c.width, sy = std::round(scale * c.height) / c.height;
                 c.width = static_cast<uint64_t>(std::round(scale * c.width));
                 c.height = static_cast<uint64_t>(std::round(scale * c.height));
                 c.RescaleParams(sx, sy);
             },
             "scale"_a, "Rescale camera dimensions by given factor and accordingly the focal length and the principal point.")
        .def("rescale",
             [](PyCamera& c, uint64_t new_width, uint64_t new_height) {
                 c.CheckParams();
                 const double sx = static_cast<double>(new_width) / c.width, sy = static_cast<double>(new_height) / c.height;
                 c.width = new_width;
                 c.height = new_height;
                 c.RescaleParams(sx, sy);
             },
             "new_width"_a, "new_height"_a,
             "Rescale camera dimensions to given size and accordingly the focal length and the principal point.")
        .def("set_params_from_string",
             [](PyCamera& c, const std::string& text) {  // CSVToVector<double>; false (and no change) on a wrong count
                 std::vector<double> v;
                 std::stringstream ss(text);
                 std::string item;
                 while (std::getline(ss, item, ',')) {
                     const size_t a = item.find_first_not_of(" \t\r\n");
                     if (a == std::string::npos) continue;
                     try {
                         v.push_back(std::stod(item.substr(a)));
                     } catch (const std::exception&) {
                         return false;
                     }
                 }
                 if (static_cast<int>(v.size()) != c.Info().num_params) return false;
                 c.params = std::move(v);
                 return true;
             },
             "params"_a, "Set camera parameters from comma-separated list.")
        .def("__repr__", &PyCamera::Repr)
        .def("__copy__", [](const PyCamera& c) { return PyCamera(c); })
        .def("__deepcopy__", [](const PyCamera& c, const py::dict&) { return PyCamera(c); });
}

// ---- one lazily created context for the single-pair calls (device 0) ---------------------------
struct EstimatorCtx {
    std::mutex mu;
    amc_ctx* ctx = nullptr;
    amc_ctx* Get() {
        if (!ctx) {
            const int rc = amc_ctx_create(0, &ctx);
            if (rc != AMC_OK) throw std::runtime_error(std::string("amc_ctx_create: ") + amc_last_error());
            if (amc_ctx_reserve_slots(ctx, 2) != AMC_OK)
                throw std::runtime_error(std::string("amc_ctx_reserve_slots: ") + amc_last_error());
        }
        return ctx;
    }
};
inline EstimatorCtx& TheEstimatorCtx() {
    static EstimatorCtx* e = new EstimatorCtx();  // intentionally leaked: no HIP calls at interpreter exit
    return *e;
}
inline void EstCheck(int rc, const char* what) {
    if (rc == AMC_OK) return;
    const std::string msg = std::string(what) + ": " + amc_last_error();
    if (rc == AMC_E_INVALID) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

using PointsArray = py::array_t<double, py::array::c_style | py::array::forcecast>;
inline size_t CheckPoints(const PointsArray& a, const char* name) {
    if (a.ndim() != 2 || a.shape(1) != 2) {
        if (a.size() == 0) return 0;
        throw py::value_error(std::string(name) + " must be an N x 2 float64 array");
    }
    return static_cast<size_t>(a.shape(0));
}
// THROW_CHECK_EQ(points2D1.size(), points2D2.size()) (/root/reference/pycolmap/estimators/fundamental_matrix.h:22):
// ValueError "[file:line] Check Failed: a == b (x vs. y)"
#define CheckSameSize(a, b, what) CheckSameSizeAt(__FILE__, __LINE__, (a), (b), (what))
inline void CheckSameSizeAt(const char* file, int line, size_t a, size_t b, const char* what) {
    if (a != b)
        throw py::value_error(CheckMessage(file, line, std::string(what) + " (" + std::to_string(a) + " vs. " +
                                                             std::to_string(b) + ")"));
}
// Camera.cam_from_img (/root/reference/pycolmap/scene/camera.h:136-150) through amc_cam_from_img
inline py::object CamFromImgPoints(const PyCamera& c, const py::object& points) {
    c.CheckParams();
    auto arr = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(points);
    if (!arr) throw py::value_error("cam_from_img: points must be convertible to a float64 array");
    const bool single = arr.ndim() == 1 && arr.shape(0) == 2;
    if (!single && !(arr.ndim() == 2 && arr.shape(1) == 2) && arr.size() != 0)
        throw py::value_error("cam_from_img: expected one point (2,) or an N x 2 array");
    const size_t n = single ? 1 : static_cast<size_t>(arr.size() / 2);
    py::array_t<double> out({static_cast<py::ssize_t>(n), static_cast<py::ssize_t>(2)});
    {
        py::gil_scoped_release release;
        EstimatorCtx& E = TheEstimatorCtx();
        std::lock_guard<std::mutex> lock(E.mu);
        EstCheck(amc_cam_from_img(E.Get(), c.model, c.params.data(), static_cast<int32_t>(c.params.size()), arr.data(), n,
                                  out.mutable_data()),
                 "amc_cam_from_img");
    }
    if (single) return out.attr("reshape")(2);
    return std::move(out);
}
// Camera::ImgFromCam on N x 2 (normalised plane) or N x 3 (hnormalized first) points: /root/reference/pycolmap/scene/camera.h:166-196
inline py::object ImgFromCamPoints(const PyCamera& c, const py::object& points) {
    c.CheckParams();
    auto arr = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(points);
    if (!arr) throw py::value_error("img_from_cam: points must be convertible to a float64 array");
    const bool single = arr.ndim() == 1 && (arr.shape(0) == 2 || arr.shape(0) == 3);
    if (!single && !(arr.ndim() == 2 && (arr.shape(1) == 2 || arr.shape(1) == 3)) && arr.size() != 0)
        throw py::value_error("img_from_cam: expected one point, an N x 2 or an N x 3 array");
    const py::ssize_t cols = arr.size() == 0 ? 2 : (single ? arr.shape(0) : arr.shape(1));
    const size_t n = static_cast<size_t>(arr.size() / cols);
    std::vector<double> uv(2 * n);
    for (size_t i = 0; i < n; ++i) {
        const double* q = arr.data() + cols * i;
        uv[2 * i] = cols == 3 ? q[0] / q[2] : q[0];
        uv[2 * i + 1] = cols == 3 ? q[1] / q[2] : q[1];
    }
    py::array_t<double> out({static_cast<py::ssize_t>(n), static_cast<py::ssize_t>(2)});
    {
        py::gil_scoped_release release;
        EstimatorCtx& E = TheEstimatorCtx();
        std::lock_guard<std::mutex> lock(E.mu);
        EstCheck(amc_img_from_cam(E.Get(), c.model, c.params.data(), static_cast<int32_t>(c.params.size()), uv.data(), n,
                                  out.mutable_data()),
                 "amc_img_from_cam");
    }
    if (single) return out.attr("reshape")(2);
    return std::move(out);
}
inline amc_ransac_opts ToAmcRansac(const RANSACOptions& o) {
    amc_ransac_opts r;
    r.max_error = o.max_error;
    r.min_inlier_ratio = o.min_inlier_ratio;
    r.confidence = o.confidence;
    r.dyn_num_trials_multiplier = o.dyn_num_trials_multiplier;
    r.min_num_trials = static_cast<int64_t>(o.min_num_trials);
    r.max_num_trials = static_cast<int64_t>(std::min<size_t>(o.max_num_trials, size_t(1) << 30));
    return r;
}
inline void UploadCamera(amc_ctx* ctx, uint32_t slot, const PyCamera& c, bool force_prior) {
    c.CheckParams();
    EstCheck(amc_upload_camera(ctx, slot, c.model, c.width, c.height, c.params.data(),
                               static_cast<int32_t>(c.params.size()),
                               (force_prior || c.has_prior_focal_length) ? 1 : 0),
             "amc_upload_camera");
}
inline RANSACOptions PyRansacDefaults() {  // /root/reference/pycolmap/optim/bindings.h:10-18
    RANSACOptions o;
    o.max_error = 4.0;
    o.min_inlier_ratio = 0.01;
    o.confidence = 0.9999;
    o.min_num_trials = 1000;
    o.max_num_trials = 100000;
    return o;
}

inline PyRigid3d RigidFromPose(const amc_pose& q) {
    PyRigid3d r;
    r.rotation.xyzw = {{q.qvec[1], q.qvec[2], q.qvec[3], q.qvec[0]}};
    r.translation = {{q.tvec[0], q.tvec[1], q.tvec[2]}};
    return r;
}

// LORANSAC<...>::Estimate(points1, points2) through amc_ransac_pairs -> dict or None
inline py::object RansacEstimate(int kind, const char* key, const PointsArray& p1, const PointsArray& p2,
                                 const PyCamera* cam1, const PyCamera* cam2, const RANSACOptions& opts) {
    const size_t n1 = CheckPoints(p1, "points2D1"), n2 = CheckPoints(p2, "points2D2");
    CheckSameSize(n1, n2, "points2D1.size() == points2D2.size()");
    amc_ransac_report rep{};
    std::vector<uint8_t> mask(n1, 0);
    amc_pose pose{};
    {
        py::gil_scoped_release release;
        EstimatorCtx& E = TheEstimatorCtx();
        std::lock_guard<std::mutex> lock(E.mu);
        amc_ctx* ctx = E.Get();
        EstCheck(amc_upload_points_f64(ctx, 0, p1.data(), static_cast<uint32_t>(n1)), "amc_upload_points_f64");
        EstCheck(amc_upload_points_f64(ctx, 1, p2.data(), static_cast<uint32_t>(n2)), "amc_upload_points_f64");
        if (cam1 && cam2) {
            UploadCamera(ctx, 0, *cam1, false);
            UploadCamera(ctx, 1, *cam2, false);
        }
        std::vector<uint32_t> matches(2 * n1);
        for (size_t i = 0; i < n1; ++i) matches[2 * i] = matches[2 * i + 1] = static_cast<uint32_t>(i);
        const uint32_t s1 = 0, s2 = 1;
        const uint64_t off[2] = {0, n1};
        const amc_ransac_opts ro = ToAmcRansac(opts);
        amc_ransac_result res;
        EstCheck(amc_ransac_pairs(ctx, kind, &s1, &s2, 1, off, matches.data(), &ro, /*seed=*/0, &res),
                 "amc_ransac_pairs");
        rep = res.reports[0];
        if (n1) std::memcpy(mask.data(), res.inlier_mask, n1);
        amc_ransac_result_free(&res);
        if (kind == AMC_RANSAC_E && rep.success) {
            // PoseFromEssentialMatrix on the inlier correspondences
            // (/root/reference/pycolmap/estimators/essential_matrix.h:63-83)
            std::vector<uint32_t> inl;
            for (size_t i = 0; i < n1; ++i)
                if (mask[i]) {
                    inl.push_back(static_cast<uint32_t>(i));
                    inl.push_back(static_cast<uint32_t>(i));
                }
            amc_tvg g{};
            g.config = AMC_TVG_CALIBRATED;
            std::memcpy(g.E, rep.model, sizeof g.E);
            const uint64_t ioff[2] = {0, inl.size() / 2};
            EstCheck(amc_pose_pairs(ctx, &s1, &s2, 1, ioff, inl.data(), &g, &pose), "amc_pose_pairs");
        }
    }
    if (!rep.success) return py::none();
    std::array<double, 9> model;
    std::memcpy(model.data(), rep.model, sizeof rep.model);
    py::list inliers;
    for (size_t i = 0; i < n1; ++i) inliers.append(py::bool_(mask[i] != 0));
    py::dict d;
    d[py::str(key)] = Mat3(model);
    if (kind == AMC_RANSAC_E) d["cam2_from_cam1"] = RigidFromPose(pose);
    d["num_inliers"] = static_cast<size_t>(rep.num_inliers);
    d["inliers"] = inliers;
    return std::move(d);
}

inline PyTwoViewGeometry EstimateTvg(const PyCamera& cam1, const PointsArray& p1, const PyCamera& cam2,
                                     const PointsArray& p2, const py::object& matches_obj,
                                     const TwoViewGeometryOptions& options, bool calibrated_entry) {
    const size_t n1 = CheckPoints(p1, "points1"), n2 = CheckPoints(p2, "points2");
    std::vector<uint32_t> matches;
    if (matches_obj.is_none()) {
        CheckSameSize(n1, n2, "points1.size() == points2.size()");
        matches.resize(2 * n1);
        for (size_t i = 0; i < n1; ++i) matches[2 * i] = matches[2 * i + 1] = static_cast<uint32_t>(i);
    } else {
        auto arr = py::array_t<uint32_t, py::array::c_style | py::array::forcecast>::ensure(matches_obj);
        if (!arr || (arr.size() != 0 && (arr.ndim() != 2 || arr.shape(1) != 2)))
            throw py::value_error("matches must be an M x 2 unsigned integer array");
        matches.assign(arr.data(), arr.data() + arr.size());
    }
    TwoViewGeometryOptions opt = options;
    if (calibrated_entry) opt.force_H_use = false;  // EstimateCalibratedTwoViewGeometry is called directly
    PyTwoViewGeometry g;
    const size_t M = matches.size() / 2;
    std::vector<uint8_t> mask(M, 0);
    amc_tvg t{};
    amc_pose pose{};
    bool have_pose = false;
    {
        py::gil_scoped_release release;
        EstimatorCtx& E = TheEstimatorCtx();
        std::lock_guard<std::mutex> lock(E.mu);
        amc_ctx* ctx = E.Get();
        EstCheck(amc_upload_points_f64(ctx, 0, p1.data(), static_cast<uint32_t>(n1)), "amc_upload_points_f64");
        EstCheck(amc_upload_points_f64(ctx, 1, p2.data(), static_cast<uint32_t>(n2)), "amc_upload_points_f64");
        UploadCamera(ctx, 0, cam1, calibrated_entry);
        UploadCamera(ctx, 1, cam2, calibrated_entry);
        const uint32_t s1 = 0, s2 = 1;
        const uint64_t off[2] = {0, M};
        const amc_tvg_opts to = ToAmc(opt);
        amc_verify_result vr;
        EstCheck(amc_verify_pairs(ctx, &s1, &s2, 1, off, matches.data(), &to, /*seed=*/0, &vr), "amc_verify_pairs");
        t = vr.tvg[0];
        if (M) std::memcpy(mask.data(), vr.inlier_mask, M);
        if (vr.pose) {
            pose = vr.pose[0];
            have_pose = true;
        }
        amc_verify_result_free(&vr);
    }
    if (have_pose) {
        g.cam2_from_cam1 = RigidFromPose(pose);
        g.tri_angle = pose.tri_angle;
    }
    g.config = t.config;
    std::memcpy(g.E.data(), t.E, sizeof t.E);
    std::memcpy(g.F.data(), t.F, sizeof t.F);
    std::memcpy(g.H.data(), t.H, sizeof t.H);
    AppendInlierMatches(mask.data(), matches.data(), M, &g.inlier_matches);
    return g;
}

inline void BindEstimators(py::module_& m) {
    const RANSACOptions est_options = PyRansacDefaults();
    m.def(
        "fundamental_matrix_estimation",
        [](const PointsArray& p1, const PointsArray& p2, const RANSACOptions& o) {
            return RansacEstimate(AMC_RANSAC_F, "F", p1, p2, nullptr, nullptr, o);
        },
        "points2D1"_a, "points2D2"_a, "estimation_options"_a = est_options, "LORANSAC + 7-point algorithm.");
    m.def(
        "homography_matrix_estimation",
        [](const PointsArray& p1, const PointsArray& p2, const RANSACOptions& o) {
            return RansacEstimate(AMC_RANSAC_H, "H", p1, p2, nullptr, nullptr, o);
        },
        "points2D1"_a, "points2D2"_a, "estimation_options"_a = est_options,
        "LORANSAC + normalized DLT homography estimation.");
    m.def(
        "essential_matrix_estimation",
        [](const PointsArray& p1, const PointsArray& p2, const PyCamera& c1, const PyCamera& c2,
           const RANSACOptions& o) {
            return RansacEstimate(AMC_RANSAC_E, "E", p1, p2, &c1, &c2, o);
        },
        "points2D1"_a, "points2D2"_a, "camera1"_a, "camera2"_a, "estimation_options"_a = est_options,
        "LORANSAC + 5-point algorithm.");
    m.def(
        "estimate_two_view_geometry",
        [](const PyCamera& c1, const PointsArray& p1, const PyCamera& c2, const PointsArray& p2,
           const py::object& matches, const TwoViewGeometryOptions& o) {
            return EstimateTvg(c1, p1, c2, p2, matches, o, false);
        },
        "camera1"_a, "points1"_a, "camera2"_a, "points2"_a, "matches"_a = py::none(),
        "options"_a = TwoViewGeometryOptions());
    m.def(
        "estimate_calibrated_two_view_geometry",
        [](const PyCamera& c1, const PointsArray& p1, const PyCamera& c2, const PointsArray& p2,
           const py::object& matches, const TwoViewGeometryOptions& o) {
            return EstimateTvg(c1, p1, c2, p2, matches, o, true);
        },
        "camera1"_a, "points1"_a, "camera2"_a, "points2"_a, "matches"_a = py::none(),
        "options"_a = TwoViewGeometryOptions());
    m.def(
        "estimate_two_view_geometry_pose",
        [](const PyCamera& c1, const PointsArray& p1, const PyCamera& c2, const PointsArray& p2,
           PyTwoViewGeometry& g) {
            // /root/reference/pycolmap/estimators/two_view_geometry.h:153-159; updates `geometry` in place
            const size_t n1 = CheckPoints(p1, "points1"), n2 = CheckPoints(p2, "points2");
            amc_pose pose{};
            {
                py::gil_scoped_release release;
                EstimatorCtx& E = TheEstimatorCtx();
                std::lock_guard<std::mutex> lock(E.mu);
                amc_ctx* ctx = E.Get();
                EstCheck(amc_upload_points_f64(ctx, 0, p1.data(), static_cast<uint32_t>(n1)), "amc_upload_points_f64");
                EstCheck(amc_upload_points_f64(ctx, 1, p2.data(), static_cast<uint32_t>(n2)), "amc_upload_points_f64");
                UploadCamera(ctx, 0, c1, false);
                UploadCamera(ctx, 1, c2, false);
                amc_tvg t{};
                t.config = g.config;
                std::memcpy(t.E, g.E.data(), sizeof t.E);
                std::memcpy(t.H, g.H.data(), sizeof t.H);
                const uint32_t s1 = 0, s2 = 1;
                const uint64_t off[2] = {0, g.inlier_matches.size() / 2};
                EstCheck(amc_pose_pairs(ctx, &s1, &s2, 1, off, g.inlier_matches.data(), &t, &pose), "amc_pose_pairs");
            }
            if (!pose.ok) return false;
            g.config = pose.config;
            g.cam2_from_cam1 = RigidFromPose(pose);
            g.tri_angle = pose.tri_angle;
            return true;
        },
        "camera1"_a, "points1"_a, "camera2"_a, "points2"_a, "geometry"_a);
    m.def(
        "squared_sampson_error",
        [](const PointsArray& p1, const PointsArray& p2, const PointsArray& E) {
            const size_t n1 = CheckPoints(p1, "points2D1"), n2 = CheckPoints(p2, "points2D2");
            CheckSameSize(n1, n2, "points1.size() == points2.size()");
            if (E.size() != 9) throw py::value_error("E must be a 3 x 3 matrix");
            std::vector<double> out(n1);
            {
                py::gil_scoped_release release;
                EstimatorCtx& ctx = TheEstimatorCtx();
                std::lock_guard<std::mutex> lock(ctx.mu);
                EstCheck(amc_squared_sampson_error(ctx.Get(), p1.data(), p2.data(), n1, E.data(), out.data()),
                         "amc_squared_sampson_error");
            }
            return out;
        },
        "points2D1"_a, "points2D2"_a, "E"_a,
        "Calculate the squared Sampson error for a given essential or fundamental matrix.");
    // PyPoseFromHomographyMatrix (/root/reference/pycolmap/geometry/homography_matrix.h:13-40)
    m.def(
        "homography_decomposition",
        [](const PointsArray& H, const PointsArray& K1, const PointsArray& K2, const PointsArray& p1, const PointsArray& p2) {
            if (H.size() != 9 || K1.size() != 9 || K2.size() != 9) throw py::value_error("H, K1, K2 must be 3 x 3 matrices");
            const size_t n1 = CheckPoints(p1, "points1"), n2 = CheckPoints(p2, "points2");
            CheckSameSize(n1, n2, "points1.size() == points2.size()");
            std::array<double, 9> R{};
            std::array<double, 3> t{}, n{};
            std::vector<double> X(3 * std::max<size_t>(n1, 1));
            uint64_t m3 = 0;
            {
                py::gil_scoped_release release;
                EstimatorCtx& ctx = TheEstimatorCtx();
                std::lock_guard<std::mutex> lock(ctx.mu);
                EstCheck(amc_homography_decomposition(ctx.Get(), H.data(), K1.data(), K2.data(), p1.data(), p2.data(), n1,
                                                      R.data(), t.data(), n.data(), X.data(), &m3),
                         "amc_homography_decomposition");
            }
            py::list pts;
            for (uint64_t i = 0; i < m3; ++i) {
                py::array_t<double> x(3);
                std::memcpy(x.mutable_data(), X.data() + 3 * i, 3 * sizeof(double));
                pts.append(x);
            }
            py::array_t<double> tv(3), nv(3);
            std::memcpy(tv.mutable_data(), t.data(), sizeof(double) * 3);
            std::memcpy(nv.mutable_data(), n.data(), sizeof(double) * 3);
            return py::dict("R"_a = Mat3(R), "t"_a = tv, "n"_a = nv, "points3D"_a = pts);
        },
        "H"_a, "K1"_a, "K2"_a, "points1"_a, "points2"_a, "Analytical Homography Decomposition.");
}

}  // namespace amchost
