// Test-only host build of pycolmap_amd/csrc/tvg_math.h (the lane-local device numerics), so the
// CPU suite can compare them bit-for-bit with the oracle without a GPU.
#include "../../pycolmap_amd/csrc/tvg_math.h"
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
}
