// py_types.h — small Python-facing value types shared by the binding files.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <array>
#include <cstring>
#include <vector>

namespace amchost {
namespace py = pybind11;

// TwoViewGeometry as pycolmap exposes it (/root/reference/pycolmap/estimators/two_view_geometry.h:79-93)
struct PyTwoViewGeometry {
    int config = 0;
    std::array<double, 9> E{}, F{}, H{};
    std::vector<uint32_t> inlier_matches;
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
