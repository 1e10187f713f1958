// Test-only host build of pycolmap_amd/csrc/tvg_math.h (the lane-local device numerics), so the
// CPU suite can compare them bit-for-bit with the oracle without a GPU.
#include "../../pycolmap_amd/csrc/tvg_math.h"
#include "../../pycolmap_amd/csrc/pose_math.h"
#include "../../pycolmap_amd/csrc/camera_math.h"
#include <vector>
using namespace amc::tvg;
extern "C" {
int shim_estimate_f7(const double* p1, const double* p2, double* models) {
    double x1[7], y1[7], x2[7], y2[7];
    for (int i = 0; i < 7; ++i) { x1[i] = p1[2 * i]; y1[i] = p1[2 * i + 1]; x2[i] = p2[2 * i]; y2[i] = p2[2 * i + 1]; }
    for (int i = 0; i < 27; ++i) models[i] = 0.0;
    return estimate_f7(x1, y1, x2, y2, models);
}
int shim_estimate_h4(const double* p1, const double* p2, double* models) {
    double x1[4], y1[4], x2[4], y2[4];
    for (int i = 0; i < 4; ++i) { x1[i] = p1[2 * i]; y1[i] = p1[2 * i + 1]; x2[i] = p2[2 * i]; y2[i] = p2[2 * i + 1]; }
    estimate_h4(x1, y1, x2, y2, models);
    return 1;
}
int shim_estimate_e5(const double* p1, const double* p2, double* models) {
    double x1[5], y1[5], x2[5], y2[5];
    for (int i = 0; i < 5; ++i) { x1[i] = p1[2 * i]; y1[i] = p1[2 * i + 1]; x2[i] = p2[2 * i]; y2[i] = p2[2 * i + 1]; }
    return estimate_e5_minimal(x1, y1, x2, y2, models);
}
int shim_real_roots(const double* c, int deg, double* roots) { return real_roots(c, deg, roots); }
void shim_jacobi(int n, double* a, double* v) { jacobi_eigen(n, a, v); }
void shim_residuals(int kind, const double* m, const double* p1, const double* p2, int n, double* out) {
    for (int i = 0; i < n; ++i)
        out[i] = kind == 0 ? sampson(m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1])
                           : h_residual(m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
}
unsigned shim_temper(unsigned y) { return mt_temper(y); }
// The counting loop's FP32 pre-filter for homographies (tvg_math.h h32_prepare / h32_point): per point 1 inlier,
// 0 outlier, -1 undecided, with the operands prepared as the kernel prepares them (coordinates and scaled
// coordinates rounded to float; C = largest |coordinate| of the set).
void shim_h32_decisions(const double* model, double max_res, const double* p1, const double* p2, int n, signed char* out) {
    double C = 0.0;
    for (int i = 0; i < n; ++i) {
        C = dmax(C, dmax(dabs(p1[2 * i]), dabs(p1[2 * i + 1])));
        C = dmax(C, dmax(dabs(p2[2 * i]), dabs(p2[2 * i + 1])));
    }
    const double s = 1.0 / dsqrt(max_res);
    const H32Model h = h32_prepare(model, s, C);
    for (int i = 0; i < n; ++i)
        out[i] = (signed char)h32_point(h, (float)p1[2 * i], (float)p1[2 * i + 1], (float)(p2[2 * i] * s), (float)(p2[2 * i + 1] * s));
}

// the folded outlier test of every point (tvg_math.h h32_outlier_q): q
void shim_h32_q(const double* model, double max_res, const double* p1, const double* p2, int n, float* q) {
    double C = 0.0;
    for (int i = 0; i < n; ++i) {
        C = dmax(C, dmax(dabs(p1[2 * i]), dabs(p1[2 * i + 1])));
        C = dmax(C, dmax(dabs(p2[2 * i]), dabs(p2[2 * i + 1])));
    }
    const double s = 1.0 / dsqrt(max_res);
    const H32Model h = h32_prepare(model, s, C);
    for (int i = 0; i < n; ++i)
        q[i] = h32_outlier_q<float, H32Ops>(h, (float)p1[2 * i], (float)p1[2 * i + 1], (float)(p2[2 * i] * s), (float)(p2[2 * i + 1] * s));
}
// the FP32 Sampson outlier test of every point (tvg_math.h s32_outlier_q): q
void shim_s32_q(const double* model, double max_res, const double* p1, const double* p2, int n, float* q) {
    double C = 0.0;
    for (int i = 0; i < n; ++i) {
        C = dmax(C, dmax(dabs(p1[2 * i]), dabs(p1[2 * i + 1])));
        C = dmax(C, dmax(dabs(p2[2 * i]), dabs(p2[2 * i + 1])));
    }
    const S32Model h = s32_prepare(model, max_res, C);
    for (int i = 0; i < n; ++i)
        q[i] = s32_outlier_q<float, H32Ops>(h, (float)p1[2 * i], (float)p1[2 * i + 1], (float)p2[2 * i], (float)p2[2 * i + 1]);
}
// the loop's bound and the reference bound of every point (tvg_math.h: h32_eval's band >= h32_band_ref)
void shim_h32_bands(const double* model, double max_res, const double* p1, const double* p2, int n, float* band, float* band_ref) {
    double C = 0.0;
    for (int i = 0; i < n; ++i) {
        C = dmax(C, dmax(dabs(p1[2 * i]), dabs(p1[2 * i + 1])));
        C = dmax(C, dmax(dabs(p2[2 * i]), dabs(p2[2 * i + 1])));
    }
    const double s = 1.0 / dsqrt(max_res);
    const H32Model h = h32_prepare(model, s, C);
    for (int i = 0; i < n; ++i) {
        const float a = (float)p1[2 * i], b = (float)p1[2 * i + 1], cs = (float)(p2[2 * i] * s), ds = (float)(p2[2 * i + 1] * s);
        float t;
        h32_eval<float, H32Ops>(h, a, b, cs, ds, t, band[i]);
        band_ref[i] = h32_band_ref(h, a, b, cs, ds);
    }
}

// The relative-pose kernel's per-pair algorithm (csrc/pose.hip) run serially with the same
// lane-local functions: cameras (model id + params), n inlier correspondences in image
// coordinates, the geometry's config / E / H.  out: ok, config, num_points3D, then
// R[9] t[3] q[4] tri_angle as doubles.
static void shim_norm(int model, const double* prm, double x, double y, double* nx, double* ny) {
    amc::cam::cam_from_img(model, prm, x, y, *nx, *ny);
}
// Camera::CamFromImg / CamFromImgThreshold / CalibrationMatrix of camera_math.h (host build)
void shim_cam_from_img(int model, const double* prm, const double* xy, int n, double* uv) {
    for (int i = 0; i < n; ++i) amc::cam::cam_from_img(model, prm, xy[2 * i], xy[2 * i + 1], uv[2 * i], uv[2 * i + 1]);
}
double shim_cam_from_img_threshold(int model, const double* prm, double t) {
    return amc::cam::cam_from_img_threshold(model, prm, t);
}
static double shim_select(const std::vector<double>& c, size_t rank) {
    uint64_t K = 0;
    for (int bit = 62; bit >= 0; --bit) {
        const uint64_t T = K | (1ull << bit);
        size_t cnt = 0;
        for (double x : c) cnt += cosine_key(x) >= T;
        if (cnt >= rank + 1) K = T;
    }
    for (double x : c)
        if (cosine_key(x) == K) return x;
    return 0.0;
}
void shim_pose(int model1, const double* prm1, int model2, const double* prm2, const double* p1, const double* p2,
               int n, int config, const double* E, const double* H, int* iout, double* dout) {
    iout[0] = 0; iout[1] = config; iout[2] = 0;
    const double ident[17] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0};
    for (int i = 0; i < 17; ++i) dout[i] = ident[i];
    if (config != 2 && config != 3 && config != 4 && config != 5 && config != 6) return;
    std::vector<double> x1(n), y1(n), x2(n), y2(n);
    for (int i = 0; i < n; ++i) {
        shim_norm(model1, prm1, p1[2 * i], p1[2 * i + 1], &x1[i], &y1[i]);
        shim_norm(model2, prm2, p2[2 * i], p2[2 * i + 1], &x2[i], &y2[i]);
    }
    PoseCands c;
    if (config == 2 || config == 3) {
        pose_candidates_E(E, c);
    } else {
        double K1[9], K2[9];
        amc::cam::calibration_matrix(model1, prm1, K1);
        amc::cam::calibration_matrix(model2, prm2, K2);
        pose_candidates_H(H, K1, K2, c);
    }
    int best = 0;
    size_t best_count = 0;
    for (int k = 0; k < c.n; ++k) {
        const CheiralityBounds b = cheirality_bounds(c.R[k], c.t[k]);
        size_t cnt = 0;
        double X[3];
        for (int i = 0; i < n; ++i) cnt += cheirality_point(c.R[k], c.t[k], b, x1[i], y1[i], x2[i], y2[i], X);
        if (cnt >= best_count) { best = k; best_count = cnt; }
    }
    const double* R = c.R[best];
    const double* t = c.t[best];
    const CheiralityBounds b = cheirality_bounds(R, t);
    double c2[3], baseline2;
    second_centre(R, t, c2, &baseline2);
    std::vector<double> cs;
    for (int i = 0; i < n; ++i) {
        double X[3];
        if (cheirality_point(R, t, b, x1[i], y1[i], x2[i], y2[i], X)) cs.push_back(triangulation_cosine(c2, baseline2, X));
    }
    double cmed[2] = {1.0, 1.0};
    if (!cs.empty()) {
        cmed[0] = shim_select(cs, cs.size() / 2);
        if (cs.size() % 2 == 0) cmed[1] = shim_select(cs, cs.size() / 2 - 1);
    }
    double tri = median_angle_host((uint32_t)cs.size(), cmed);
    if (config == 6) {
        if (vec3_norm(t) == 0.0) { config = 5; tri = 0.0; }
        else config = 4;
    }
    iout[0] = 1; iout[1] = config; iout[2] = (int)cs.size();
    for (int i = 0; i < 9; ++i) dout[i] = R[i];
    for (int i = 0; i < 3; ++i) dout[9 + i] = t[i];
    rotation_to_quaternion(R, dout + 12);
    dout[16] = tri;
}
}
