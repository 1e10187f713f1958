// camera_math.h — Camera::CamFromImg / CamFromImgThreshold / CalibrationMatrix for COLMAP 3.9.1's
// eleven camera models (colmap/sensor/models.h), as the calibrated two-view path needs them:
//   /root/reference/pycolmap/estimators/essential_matrix.h:33-46   camera.CamFromImg(point) per
//        correspondence, max_error = 0.5 * (CamFromImgThreshold(e)_1 + CamFromImgThreshold(e)_2)
//   /root/reference/pycolmap/scene/camera.h:136-165                 cam_from_img, cam_from_img_threshold
//
// Lifting a pixel to the normalised image plane is (x - c) / f followed, for every model with
// distortion parameters, by BaseCameraModel::IterativeUndistortion: a Newton iteration on
// x + Distortion(x) = x0 with a central-difference Jacobian (100 iterations at most, relative
// step 1e-6, stops when the squared step drops below 1e-10).  FOV has a closed form;
// THIN_PRISM_FISHEYE undoes the equidistant projection after the iteration.
//
// Where it runs.  The lift depends on the keypoint alone, not on the pair, so it is done ONCE PER
// KEYPOINT when an image first takes part in a calibrated estimation (COLMAP redoes it per match of
// every pair) and the result is kept beside the pixel coordinates in HBM (amc_api.hip
// ensure_normalized).  The polynomial models (SIMPLE_RADIAL, RADIAL, OPENCV, FULL_OPENCV) run in
// undistort_kernel on the device: only + - * / and no contraction, so the bits equal the host's.
// The models whose distortion calls atan / tan / sin / cos (the fisheye family, FOV) are lifted with
// the HOST libm, for the reason the matcher's acos table is: device
// transcendentals do not round like glibc's, and these values feed inlier decisions.
//
// Operation order is that of the upstream templates, written out for doubles; FP contraction is
// off for every translation unit that includes this header.
#pragma once

#include "tvg_math.h"  // AMC_HD, dabs, dsqrt

namespace amc {
namespace cam {

enum : int {
    SIMPLE_PINHOLE = 0, PINHOLE = 1, SIMPLE_RADIAL = 2, RADIAL = 3, OPENCV = 4, OPENCV_FISHEYE = 5,
    FULL_OPENCV = 6, FOV = 7, SIMPLE_RADIAL_FISHEYE = 8, RADIAL_FISHEYE = 9, THIN_PRISM_FISHEYE = 10,
    kNumModels = 11
};
constexpr int kMaxParams = 12;

AMC_HD int num_params(int model) {
    switch (model) {
        case SIMPLE_PINHOLE: return 3;
        case PINHOLE: return 4;
        case SIMPLE_RADIAL: return 4;
        case RADIAL: return 5;
        case OPENCV: return 8;
        case OPENCV_FISHEYE: return 8;
        case FULL_OPENCV: return 12;
        case FOV: return 5;
        case SIMPLE_RADIAL_FISHEYE: return 4;
        case RADIAL_FISHEYE: return 5;
        case THIN_PRISM_FISHEYE: return 12;
    }
    return -1;
}
// one focal length (params[0]) or two (params[0], params[1]); the principal point follows
AMC_HD int num_focal(int model) {
    return (model == SIMPLE_PINHOLE || model == SIMPLE_RADIAL || model == RADIAL || model == SIMPLE_RADIAL_FISHEYE ||
            model == RADIAL_FISHEYE)
               ? 1
               : 2;
}
AMC_HD bool is_pinhole(int model) { return model == SIMPLE_PINHOLE || model == PINHOLE; }
// the lift needs libm (atan / tan / sin / cos): host only
AMC_HD bool needs_libm(int model) {
    return model == OPENCV_FISHEYE || model == FOV || model == SIMPLE_RADIAL_FISHEYE || model == RADIAL_FISHEYE ||
           model == THIN_PRISM_FISHEYE;
}

// Camera::MeanFocalLength: sum over focal_length_idxs (starting from 0) / their number
AMC_HD double mean_focal_length(int model, const double* p) {
    double f = 0.0;
    const int nf = num_focal(model);
    for (int i = 0; i < nf; ++i) f += p[i];
    return f / (double)nf;
}
// Camera::CamFromImgThreshold
AMC_HD double cam_from_img_threshold(int model, const double* p, double threshold) {
    return threshold / mean_focal_length(model, p);
}
// Camera::CalibrationMatrix, row-major
AMC_HD void calibration_matrix(int model, const double* p, double* K) {
    for (int i = 0; i < 9; ++i) K[i] = 0.0;
    const int nf = num_focal(model);
    K[0] = p[0];
    K[4] = p[nf - 1];
    K[2] = p[nf];
    K[5] = p[nf + 1];
    K[8] = 1.0;
}

// ---- Distortion(extra_params, u, v, &du, &dv) of the polynomial models ---------------------------
AMC_HD void distortion_simple_radial(const double* e, double u, double v, double& du, double& dv) {
    const double k = e[0];
    const double u2 = u * u, v2 = v * v;
    const double r2 = u2 + v2;
    const double radial = k * r2;
    du = u * radial;
    dv = v * radial;
}
AMC_HD void distortion_radial(const double* e, double u, double v, double& du, double& dv) {
    const double k1 = e[0], k2 = e[1];
    const double u2 = u * u, v2 = v * v;
    const double r2 = u2 + v2;
    const double radial = k1 * r2 + k2 * r2 * r2;
    du = u * radial;
    dv = v * radial;
}
AMC_HD void distortion_opencv(const double* e, double u, double v, double& du, double& dv) {
    const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3];
    const double u2 = u * u, uv = u * v, v2 = v * v;
    const double r2 = u2 + v2;
    const double radial = k1 * r2 + k2 * r2 * r2;
    du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
    dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
}
AMC_HD void distortion_full_opencv(const double* e, double u, double v, double& du, double& dv) {
    const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], k5 = e[6], k6 = e[7];
    const double u2 = u * u, uv = u * v, v2 = v * v;
    const double r2 = u2 + v2;
    const double r4 = r2 * r2;
    const double r6 = r4 * r2;
    const double radial = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6);
    du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) - u;
    dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) - v;
}
AMC_HD void distortion_thin_prism(const double* e, double u, double v, double& du, double& dv) {
    const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], sx1 = e[6], sy1 = e[7];
    const double u2 = u * u, uv = u * v, v2 = v * v;
    const double r2 = u2 + v2;
    const double r4 = r2 * r2;
    const double r6 = r4 * r2;
    const double r8 = r6 * r2;
    const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
    du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) + sx1 * r2;
    dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) + sy1 * r2;
}

#if !defined(__HIP_DEVICE_COMPILE__)
}  // namespace cam
}  // namespace amc
#include <cmath>
namespace amc {
namespace cam {
// equidistant fisheye distortions: theta = atan(r), thetad = theta * (1 + k1 theta^2 + ...)
inline void distortion_fisheye(int nk, const double* e, double u, double v, double& du, double& dv) {
    const double r = std::sqrt(u * u + v * v);
    if (r > 2.220446049250313e-16) {
        const double theta = std::atan(r);
        const double theta2 = theta * theta;
        double thetad;
        if (nk == 1) {
            thetad = theta * (1.0 + e[0] * theta2);
        } else if (nk == 2) {
            const double theta4 = theta2 * theta2;
            thetad = theta * (1.0 + e[0] * theta2 + e[1] * theta4);
        } else {
            const double theta4 = theta2 * theta2;
            const double theta6 = theta4 * theta2;
            const double theta8 = theta4 * theta4;
            thetad = theta * (1.0 + e[0] * theta2 + e[1] * theta4 + e[2] * theta6 + e[3] * theta8);
        }
        du = u * thetad / r - u;
        dv = v * thetad / r - v;
    } else {
        du = 0.0;
        dv = 0.0;
    }
}
// FOVCameraModel::Undistortion (closed form)
inline void undistortion_fov(const double* e, double u, double v, double& ou, double& ov) {
    const double omega = e[0];
    const double kEpsilon = 1e-4;
    const double radius2 = u * u + v * v;
    const double omega2 = omega * omega;
    double factor;
    if (omega2 < kEpsilon) {
        factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
    } else if (radius2 < kEpsilon) {
        factor = (omega * (omega * omega * radius2 + 3.0)) / (6.0 * std::tan(omega / 2.0));
    } else {
        const double radius = std::sqrt(radius2);
        const double numerator = std::tan(radius * omega);
        factor = numerator / (radius * 2.0 * std::tan(omega / 2.0));
    }
    ou = u * factor;
    ov = v * factor;
}
// FOVCameraModel::Distortion: returns the distorted point itself (not an offset)
inline void distortion_fov(const double* e, double u, double v, double& ou, double& ov) {
    const double omega = e[0];
    const double kEpsilon = 1e-4;
    const double radius2 = u * u + v * v;
    const double omega2 = omega * omega;
    double factor;
    if (omega2 < kEpsilon) {
        factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
    } else if (radius2 < kEpsilon) {
        const double tan_half_omega = std::tan(omega / 2.0);
        factor = (-2.0 * tan_half_omega * (4.0 * radius2 * tan_half_omega * tan_half_omega - 3.0)) / (3.0 * omega);
    } else {
        const double radius = std::sqrt(radius2);
        const double numerator = std::atan(radius * 2.0 * std::tan(omega / 2.0));
        factor = numerator / (radius * omega);
    }
    ou = u * factor;
    ov = v * factor;
}
#endif

AMC_HD void distortion(int model, const double* e, double u, double v, double& du, double& dv) {
    switch (model) {
        case SIMPLE_RADIAL: distortion_simple_radial(e, u, v, du, dv); return;
        case RADIAL: distortion_radial(e, u, v, du, dv); return;
        case OPENCV: distortion_opencv(e, u, v, du, dv); return;
        case FULL_OPENCV: distortion_full_opencv(e, u, v, du, dv); return;
        case THIN_PRISM_FISHEYE: distortion_thin_prism(e, u, v, du, dv); return;
#if !defined(__HIP_DEVICE_COMPILE__)
        case OPENCV_FISHEYE: distortion_fisheye(4, e, u, v, du, dv); return;
        case SIMPLE_RADIAL_FISHEYE: distortion_fisheye(1, e, u, v, du, dv); return;
        case RADIAL_FISHEYE: distortion_fisheye(2, e, u, v, du, dv); return;
#endif
        default: du = 0.0; dv = 0.0; return;
    }
}

// BaseCameraModel::IterativeUndistortion
AMC_HD void iterative_undistortion(int model, const double* e, double& u, double& v) {
    const int kNumIterations = 100;
    const double kMaxStepNorm = 1e-10;
    const double kRelStepSize = 1e-6;
    const double kEps = 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
    const double x0_0 = u, x0_1 = v;
    double x_0 = u, x_1 = v;
    for (int i = 0; i < kNumIterations; ++i) {
        const double step0 = tvg::dmax(kEps, tvg::dabs(kRelStepSize * x_0));
        const double step1 = tvg::dmax(kEps, tvg::dabs(kRelStepSize * x_1));
        double dx_0, dx_1, dx_0b_0, dx_0b_1, dx_0f_0, dx_0f_1, dx_1b_0, dx_1b_1, dx_1f_0, dx_1f_1;
        distortion(model, e, x_0, x_1, dx_0, dx_1);
        distortion(model, e, x_0 - step0, x_1, dx_0b_0, dx_0b_1);
        distortion(model, e, x_0 + step0, x_1, dx_0f_0, dx_0f_1);
        distortion(model, e, x_0, x_1 - step1, dx_1b_0, dx_1b_1);
        distortion(model, e, x_0, x_1 + step1, dx_1f_0, dx_1f_1);
        const double J00 = 1.0 + (dx_0f_0 - dx_0b_0) / (2.0 * step0);
        const double J01 = (dx_1f_0 - dx_1b_0) / (2.0 * step1);
        const double J10 = (dx_0f_1 - dx_0b_1) / (2.0 * step0);
        const double J11 = 1.0 + (dx_1f_1 - dx_1b_1) / (2.0 * step1);
        // step_x = J.inverse() * (x + dx - x0); 2 x 2 inverse = adjugate * (1 / det)
        const double invdet = 1.0 / (J00 * J11 - J10 * J01);
        const double i00 = J11 * invdet, i01 = -J01 * invdet, i10 = -J10 * invdet, i11 = J00 * invdet;
        const double r_0 = x_0 + dx_0 - x0_0, r_1 = x_1 + dx_1 - x0_1;
        const double s_0 = i00 * r_0 + i01 * r_1;
        const double s_1 = i10 * r_0 + i11 * r_1;
        x_0 -= s_0;
        x_1 -= s_1;
        if (s_0 * s_0 + s_1 * s_1 < kMaxStepNorm) break;
    }
    u = x_0;
    v = x_1;
}

// Camera::CamFromImg: pixel (x, y) -> normalised image plane (u, v).  On the device only the models
// with !needs_libm(model) may be passed.
AMC_HD void cam_from_img(int model, const double* p, double x, double y, double& u, double& v) {
    const int nf = num_focal(model);
    const double f1 = p[0], f2 = p[nf - 1], c1 = p[nf], c2 = p[nf + 1];
    u = (x - c1) / f1;
    v = (y - c2) / f2;
    if (is_pinhole(model)) return;
    const double* e = p + nf + 2;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (model == FOV) {
        const double uu = u, vv = v;
        undistortion_fov(e, uu, vv, u, v);
        return;
    }
#endif
    iterative_undistortion(model, e, u, v);
#if !defined(__HIP_DEVICE_COMPILE__)
    if (model == THIN_PRISM_FISHEYE) {
        const double theta = std::sqrt(u * u + v * v);
        // sin and cos of one argument: GCC (-O2, what COLMAP and the oracle are built with) merges the two
        // calls into glibc's sincos, clang (this file's host pass) does not, and the two entry points are not
        // bit-identical in the last ulp - call sincos outright on both sides
        double sin_theta, cos_theta;
        ::sincos(theta, &sin_theta, &cos_theta);
        const double theta_cos_theta = theta * cos_theta;
        if (theta_cos_theta > 2.220446049250313e-16) {
            const double scale = sin_theta / theta_cos_theta;
            u *= scale;
            v *= scale;
        }
    }
#endif
}

// Camera::ImgFromCam: normalised image plane (u, v) -> pixel (x, y).  On the device only the models with
// !needs_libm(model) may be passed.
AMC_HD void img_from_cam(int model, const double* p, double u, double v, double& x, double& y) {
    const int nf = num_focal(model);
    const double f1 = p[0], f2 = p[nf - 1], c1 = p[nf], c2 = p[nf + 1];
    if (is_pinhole(model)) {
        x = f1 * u + c1;
        y = f2 * v + c2;
        return;
    }
    const double* e = p + nf + 2;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (model == FOV) {
        double du, dv;
        distortion_fov(e, u, v, du, dv);
        x = f1 * du + c1;
        y = f2 * dv + c2;
        return;
    }
    if (model == THIN_PRISM_FISHEYE) {
        const double r = std::sqrt(u * u + v * v);
        if (r > 2.220446049250313e-16) {
            const double theta = std::atan(r);
            u = theta * u / r;
            v = theta * v / r;
        }
    }
#endif
    double du, dv;
    distortion(model, e, u, v, du, dv);
    x = f1 * (u + du) + c1;
    y = f2 * (v + dv) + c2;
}

}  // namespace cam
}  // namespace amc
