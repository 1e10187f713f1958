// py_types.h — small Python-facing value types shared by the binding files.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <array>
#include <cmath>
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
    // The algebra below follows Eigen's Quaternion (the type behind the reference's Rotation3d,
    // /root/reference/pycolmap/geometry/bindings.h:24-73): product, rotation of a vector, inverse, angle.
    static PyRotation3d FromMatrix(const std::array<double, 9>& m) {  // Eigen::Quaterniond(Matrix3d)
        PyRotation3d r;
        double q[4];  // w, x, y, z
        double t = m[0] + m[4] + m[8];
        if (t > 0.0) {
            t = std::sqrt(t + 1.0);
            q[0] = 0.5 * t;
            t = 0.5 / t;
            q[1] = (m[7] - m[5]) * t;
            q[2] = (m[2] - m[6]) * t;
            q[3] = (m[3] - m[1]) * t;
        } else {
            int i = 0;
            if (m[4] > m[0]) i = 1;
            if (m[8] > m[4 * i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
            q[1 + i] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
            q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
            q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
        }
        r.xyzw = {{q[1], q[2], q[3], q[0]}};
        return r;
    }
    static PyRotation3d FromAxisAngle(const std::array<double, 3>& v) {  // AngleAxis(|v|, v.normalized())
        const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const double angle = std::sqrt(n2);
        std::array<double, 3> axis = v;
        if (n2 > 0.0)
            for (double& a : axis) a /= angle;
        const double s = std::sin(0.5 * angle);
        PyRotation3d r;
        r.xyzw = {{s * axis[0], s * axis[1], s * axis[2], std::cos(0.5 * angle)}};
        return r;
    }
    PyRotation3d Mul(const PyRotation3d& o) const {
        const double ax = xyzw[0], ay = xyzw[1], az = xyzw[2], aw = xyzw[3];
        const double bx = o.xyzw[0], by = o.xyzw[1], bz = o.xyzw[2], bw = o.xyzw[3];
        PyRotation3d r;
        r.xyzw = {{aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                   aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz}};
        return r;
    }
    std::array<double, 3> Rotate(const std::array<double, 3>& v) const {  // v + w uv + q x uv, uv = 2 (q x v)
        const double x = xyzw[0], y = xyzw[1], z = xyzw[2], w = xyzw[3];
        const double ux = 2.0 * (y * v[2] - z * v[1]), uy = 2.0 * (z * v[0] - x * v[2]), uz = 2.0 * (x * v[1] - y * v[0]);
        return {{v[0] + w * ux + (y * uz - z * uy), v[1] + w * uy + (z * ux - x * uz), v[2] + w * uz + (x * uy - y * ux)}};
    }
    double SquaredNorm() const { return xyzw[0] * xyzw[0] + xyzw[1] * xyzw[1] + xyzw[2] * xyzw[2] + xyzw[3] * xyzw[3]; }
    PyRotation3d Conjugate() const {
        PyRotation3d r;
        r.xyzw = {{-xyzw[0], -xyzw[1], -xyzw[2], xyzw[3]}};
        return r;
    }
    PyRotation3d Inverse() const {  // conjugate / squared norm (all zero for the zero quaternion)
        const double n2 = SquaredNorm();
        PyRotation3d r;
        if (n2 > 0.0) {
            r = Conjugate();
            for (double& c : r.xyzw) c /= n2;
        } else {
            r.xyzw = {{0, 0, 0, 0}};
        }
        return r;
    }
    double Angle() const {  // AngleAxis(q).angle()
        const double n = std::sqrt(xyzw[0] * xyzw[0] + xyzw[1] * xyzw[1] + xyzw[2] * xyzw[2]);
        return n != 0.0 ? 2.0 * std::atan2(n, std::fabs(xyzw[3])) : 0.0;
    }
    double AngleTo(const PyRotation3d& o) const {  // angularDistance
        const PyRotation3d d = Mul(o.Conjugate());
        return 2.0 * std::atan2(std::sqrt(d.xyzw[0] * d.xyzw[0] + d.xyzw[1] * d.xyzw[1] + d.xyzw[2] * d.xyzw[2]),
                                std::fabs(d.xyzw[3]));
    }
    PyRotation3d Slerp(double t, const PyRotation3d& o) const {
        const double d = xyzw[0] * o.xyzw[0] + xyzw[1] * o.xyzw[1] + xyzw[2] * o.xyzw[2] + xyzw[3] * o.xyzw[3];
        const double ad = std::fabs(d);
        double s0, s1;
        if (ad >= 1.0 - 2.220446049250313e-16) {
            s0 = 1.0 - t;
            s1 = t;
        } else {
            const double theta = std::acos(ad), st = std::sin(theta);
            s0 = std::sin((1.0 - t) * theta) / st;
            s1 = std::sin(t * theta) / st;
        }
        if (d < 0.0) s1 = -s1;
        PyRotation3d r;
        for (int i = 0; i < 4; ++i) r.xyzw[i] = s0 * xyzw[i] + s1 * o.xyzw[i];
        return r;
    }
};
struct PyRigid3d {
    PyRotation3d rotation;
    std::array<double, 3> translation{{0, 0, 0}};
    // colmap/geometry/rigid3.h: composition, application to a point, Inverse
    PyRigid3d Mul(const PyRigid3d& o) const {
        PyRigid3d r;
        r.rotation = rotation.Mul(o.rotation);
        const std::array<double, 3> rt = rotation.Rotate(o.translation);
        r.translation = {{translation[0] + rt[0], translation[1] + rt[1], translation[2] + rt[2]}};
        return r;
    }
    std::array<double, 3> Apply(const std::array<double, 3>& v) const {
        const std::array<double, 3> rv = rotation.Rotate(v);
        return {{rv[0] + translation[0], rv[1] + translation[1], rv[2] + translation[2]}};
    }
    PyRigid3d Inverse() const {
        PyRigid3d r;
        r.rotation = rotation.Inverse();
        r.translation = r.rotation.Rotate({{-translation[0], -translation[1], -translation[2]}});
        return r;
    }
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
