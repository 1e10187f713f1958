// py_types.h — small Python-facing value types shared by the binding files.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <array>
#include <cstring>
#include <string>
#include <vector>

namespace amchost {
namespace py = pybind11;

// The reference's exception text (/root/reference/pycolmap/log_exceptions.h:29-76): "[file:line] Check Failed: expr"
// or "... expr : message"; file is the base name of the throwing source file.
inline std::string CheckMessage(const char* file, int line, const std::string& expr, const std::string& msg = std::string()) {
    const char* base = std::strrchr(file, '/');
    std::string out = std::string("[") + (base ? base + 1 : file) + ":" + std::to_string(line) + "] Check Failed: " + expr;
    if (!msg.empty()) out += " : " + msg;
    return out;
}

// Rotation3d / Rigid3d value types (/root/reference/pycolmap/geometry/bindings.h:24-104): only what a
// TwoViewGeometry's cam2_from_cam1 needs - the quaternion in Eigen's (x, y, z, w) coefficient order,
// the translation, and the matrix forms.
struct PyRotation3d {
    std::array<double, 4> xyzw{{0, 0, 0, 1}};
    // Eigen::Quaterniond::toRotationMatrix
    std::array<double, 9> Matrix() const {
        const double x = xyzw[0], y = xyzw[1], z = xyzw[2], w = xyzw[3];
        const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        return {{1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy,
                 tyz + twx, 1.0 - (txx + tyy)}};
    }
};
struct PyRigid3d {
    PyRotation3d rotation;
    std::array<double, 3> translation{{0, 0, 0}};
};

// TwoViewGeometry as pycolmap exposes it (/root/reference/pycolmap/estimators/two_view_geometry.h:79-93)
struct PyTwoViewGeometry {
    int config = 0;
    std::array<double, 9> E{}, F{}, H{};
    std::vector<uint32_t> inlier_matches;
    PyRigid3d cam2_from_cam1;
    double tri_angle = 0.0;
};
inline py::array_t<double> Mat3(const std::array<double, 9>& m) {
    py::array_t<double> a({3, 3});
    std::memcpy(a.mutable_data(), m.data(), sizeof(double) * 9);
    return a;
}
inline py::array_t<uint32_t> MatchesArray(const std::vector<uint32_t>& m) {
    py::array_t<uint32_t> a({static_cast<py::ssize_t>(m.size() / 2), static_cast<py::ssize_t>(2)});
    if (!m.empty()) std::memcpy(a.mutable_data(), m.data(), m.size() * sizeof(uint32_t));
    return a;
}

}  // namespace amchost
