// guided_region.h — the geometry of guided matching's candidate generation (match_guided.hip), shared between the
// kernel and the host (amc_api.hip's per-pair setup; tests/shim/guided_shim.cc, which checks on the CPU that the
// regions computed here contain every pairing the float32 filter accepts).
//
// A pair's filter model is F (Sampson error) or H (forward transfer error); `dir` 0 searches image 2 for the
// partners of an image-1 point, `dir` 1 image 1 for the partners of an image-2 point.  The searched image's
// keypoints sit on a kGridDim x kGridDim grid over their bounding box; guided_row_region gives, for one point and
// one grid row (its y interval), the x interval a conservative superset of the filter's acceptance region covers.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AMC_GR_HD __host__ __device__ inline
#else
#define AMC_GR_HD inline
#endif

namespace amc {
namespace guided {

constexpr int kGridDim = 64;
enum : int { kNone = 0, kF = 1, kH = 2 };

// the cell coordinate of v along one axis: the same float operations on the host (grid build) and in the kernel
// (range lookup), monotone in v, so a coordinate interval maps to the cell interval of its end points
AMC_GR_HD int grid_cell(float v, float v0, float inv) {
    const float t = floorf((v - v0) * inv);
    return t < 0.f ? 0 : (t > (float)(kGridDim - 1) ? kGridDim - 1 : (int)t);
}

// reciprocal and square root: exact on the host; in the kernel the hardware approximations (relative error ~1e-8),
// far inside the 2 % and one pixel of slack the regions carry
#if defined(__HIP_DEVICE_COMPILE__)
AMC_GR_HD double arcp(double v) { return __builtin_amdgcn_rcp(v); }
AMC_GR_HD double asqrt(double v) { return __builtin_amdgcn_sqrt(v); }
#else
AMC_GR_HD double arcp(double v) { return 1.0 / v; }
AMC_GR_HD double asqrt(double v) { return sqrt(v); }
#endif
AMC_GR_HD bool finite_d(double v) { return v - v == 0.0; }
AMC_GR_HD double inf_d() {
    union { uint64_t u; double d; } x;
    x.u = 0x7FF0000000000000ull;
    return x.d;
}

// One point p = (px, py) against one grid row whose keypoints have y in [ylo, yhi].
//   m     the float filter model as doubles (row-major), minv its inverse (H, dir 1 only)
//   T     max_residual = (float)(max_error^2);  bound: F only, the maximum of |F^T x2|_12^2 (dir 0) or |F x1|_12^2
//         (dir 1) over the searched image's keypoint box
//   cell  cw + ch of the searched grid (slack of the H^-1 box)
// Returns false when the row holds no candidate, else [xa, xb] (possibly infinite: `full` rows list everything).
AMC_GR_HD bool guided_row_region(int kind, int dir, const double* m, const double* minv, double T, double bound,
                                 double px, double py, double ylo, double yhi, double cell, double& xa, double& xb) {
    const double kInf = inf_d();
    bool full = false, none = false;
    xa = -kInf;
    xb = kInf;
    if (kind == kF) {
        double a, b, c;
        if (dir == 0) {  // l = F p: the line of p in image 2
            a = m[0] * px + m[1] * py + m[2];
            b = m[3] * px + m[4] * py + m[5];
            c = m[6] * px + m[7] * py + m[8];
        } else {         // l = F^T p: the line of p in image 1
            a = m[0] * px + m[3] * py + m[6];
            b = m[1] * px + m[4] * py + m[7];
            c = m[2] * px + m[5] * py + m[8];
        }
        const double L2 = a * a + b * b;
        // Sampson <= T  =>  (l . q)^2 <= T (|l|^2 + |other|^2) <= T (L2 + bound); one pixel on top
        const double W = asqrt(T * (L2 + bound)) * 1.02 + asqrt(L2);
        if (!(L2 > 1e-30) || !(L2 < 1e30) || !(W < 1e300)) {
            full = true;
        } else {
            const double t0 = -(b * ylo + c), t1 = -(b * yhi + c);  // a x in [t - W, t + W]
            const double lo = fmin(t0, t1) - W, hi = fmax(t0, t1) + W;
            const double ia = arcp(a);  // (+-inf for a = +-0; not used then)
            if (a > 0.0) { xa = lo * ia; xb = hi * ia; }
            else if (a < 0.0) { xa = hi * ia; xb = lo * ia; }
            else if (!(lo <= 0.0 && hi >= 0.0)) none = true;
            if (xa != xa || xb != xb) { xa = -kInf; xb = kInf; }  // 0 * inf of a denormal a: every cell of the row
        }
    } else if (dir == 0) {  // box around hnormalized(H p)
        const double wq = m[6] * px + m[7] * py + m[8];
        const double wmag = fabs(m[6] * px) + fabs(m[7] * py) + fabs(m[8]);
        const double iw = arcp(wq);
        const double cx = (m[0] * px + m[1] * py + m[2]) * iw;
        const double cy = (m[3] * px + m[4] * py + m[5]) * iw;
        const double r = asqrt(T) * 1.02 + 1.0;
        if (!(fabs(wq) > 1e-4 * wmag) || !(wmag > 1e-30) || !finite_d(cx) || !finite_d(cy) || !(r < 1e300)) {
            full = true;
        } else if (yhi < cy - r || ylo > cy + r) {
            none = true;
        } else {
            xa = cx - r;
            xb = cx + r;
        }
    } else {  // p is an image-2 point: candidates are the image-1 points H maps into the box around p
        const double r = asqrt(T) * 1.02 + 1.0;
        double xmin = kInf, xmax = -kInf, ymin = kInf, ymax = -kInf;
        bool pos = false, neg = false, bad = !(r < 1e300);
        for (int k = 0; k < 4; ++k) {
            const double qx = px + ((k & 1) ? r : -r), qy = py + ((k & 2) ? r : -r);
            const double wk = minv[6] * qx + minv[7] * qy + minv[8];
            const double wmag = fabs(minv[6] * qx) + fabs(minv[7] * qy) + fabs(minv[8]);
            const double iw = arcp(wk);
            const double ux = (minv[0] * qx + minv[1] * qy + minv[2]) * iw;
            const double uy = (minv[3] * qx + minv[4] * qy + minv[5]) * iw;
            pos |= wk > 0.0;
            neg |= wk < 0.0;
            bad |= !(fabs(wk) > 1e-4 * wmag) || !finite_d(ux) || !finite_d(uy);
            xmin = fmin(xmin, ux); xmax = fmax(xmax, ux);
            ymin = fmin(ymin, uy); ymax = fmax(ymax, uy);
        }
        if (bad || (pos && neg)) {
            full = true;
        } else {
            const double sl = 1e-3 * cell;
            if (yhi < ymin - sl || ylo > ymax + sl) none = true;
            xa = xmin - sl;
            xb = xmax + sl;
        }
    }
    if (full) {
        none = false;
        xa = -kInf;
        xb = kInf;
    }
    return !none;
}

// Whether a pair may take the candidate-generation kernel at all, and what that kernel needs beyond the float model
// (bound[2], minv[9]): it is exact as long as the float32 filter cannot return NaN (NaN > t is false = "not rejected"
// for EVERY pairing, which no geometric candidate set contains) and the regions above hold - finite, sanely scaled
// models and keypoints.  box1 / box2: x0, y0, x1, y1 of the two images' keypoints.
inline bool guided_pair_setup(int kind, const float* mf, float max_residual, const float* box1, const float* box2,
                              double* bound, double* minv) {
    bound[0] = bound[1] = 0.0;
    for (int k = 0; k < 9; ++k) minv[k] = 0.0;
    double m[9], mx = 0.0;
    for (int k = 0; k < 9; ++k) {
        m[k] = (double)mf[k];
        if (!finite_d(m[k])) return false;
        mx = fmax(mx, fabs(m[k]));
    }
    if (!(mx > 1e-12) || !(mx < 1e12) || !finite_d((double)max_residual) || !(max_residual >= 0.f)) return false;
    const double c1[4][2] = {{box1[0], box1[1]}, {box1[2], box1[1]}, {box1[0], box1[3]}, {box1[2], box1[3]}};
    const double c2[4][2] = {{box2[0], box2[1]}, {box2[2], box2[1]}, {box2[0], box2[3]}, {box2[2], box2[3]}};
    for (int k = 0; k < 4; ++k)
        if (fabs(c1[k][0]) > 1e7 || fabs(c1[k][1]) > 1e7 || fabs(c2[k][0]) > 1e7 || fabs(c2[k][1]) > 1e7) return false;
    if (kind == kF) {
        // |F^T x2|_12^2 and |F x1|_12^2 are convex in the point: their maxima over a box are at its corners
        for (int k = 0; k < 4; ++k) {
            const double u0 = m[0] * c2[k][0] + m[3] * c2[k][1] + m[6], u1 = m[1] * c2[k][0] + m[4] * c2[k][1] + m[7];
            const double v0 = m[0] * c1[k][0] + m[1] * c1[k][1] + m[2], v1 = m[3] * c1[k][0] + m[4] * c1[k][1] + m[5];
            bound[0] = fmax(bound[0], u0 * u0 + u1 * u1);
            bound[1] = fmax(bound[1], v0 * v0 + v1 * v1);
        }
        bound[0] *= 1.001;  // a little room for the filter's float32 rounding of these terms
        bound[1] *= 1.001;
        return bound[0] < 1e30 && bound[1] < 1e30;
    }
    if (kind != kH) return false;
    // H: the projective division must keep one sign, well away from zero, over image 1's keypoint box (then the
    // filter never divides by zero and the points H maps into a box are the H^-1 image of that box)
    bool pos = false, neg = false;
    for (int k = 0; k < 4; ++k) {
        const double w = m[6] * c1[k][0] + m[7] * c1[k][1] + m[8];
        const double wmag = fabs(m[6] * c1[k][0]) + fabs(m[7] * c1[k][1]) + fabs(m[8]);
        if (!(fabs(w) > 1e-3 * wmag) || !(wmag > 1e-30)) return false;
        pos |= w > 0.0;
        neg |= w < 0.0;
    }
    if (pos && neg) return false;
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (!(fabs(det) > 1e-9 * mx * mx * mx)) return false;
    const double adj[9] = {m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
                           m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
                           m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3]};
    for (int k = 0; k < 9; ++k) {
        minv[k] = adj[k] / det;
        if (!finite_d(minv[k])) return false;
    }
    return true;
}

// The grid of one image's float32 keypoints (host only): box, cell sizes, and the keypoints' order by cell.
struct GridGeom {
    float x0, y0, cw, ch, inv_cw, inv_ch, bx1, by1;
};
// false: no grid (a coordinate is not finite, or the extent overflows float).  sidx: the original indices in cell
// order; cell_start: kGridDim^2 + 1 CSR entries.
template <class VecU32>
inline bool build_grid(const float* xy, uint32_t rows, GridGeom& g, VecU32& sidx, VecU32& cell_start) {
    if (rows == 0) return false;
    float x0 = xy[0], y0 = xy[1], x1 = xy[0], y1 = xy[1];
    for (uint32_t i = 0; i < rows; ++i) {
        const float x = xy[2 * (size_t)i], y = xy[2 * (size_t)i + 1];
        if (!(x - x == 0.f) || !(y - y == 0.f)) return false;
        x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1;
        y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
    }
    g.x0 = x0;
    g.y0 = y0;
    g.cw = (x1 - x0) / (float)kGridDim;
    g.ch = (y1 - y0) / (float)kGridDim;
    if (!(g.cw >= 1e-3f)) g.cw = 1e-3f;
    if (!(g.ch >= 1e-3f)) g.ch = 1e-3f;
    if (!(g.cw - g.cw == 0.f) || !(g.ch - g.ch == 0.f)) return false;  // (extent overflows float)
    g.inv_cw = 1.0f / g.cw;
    g.inv_ch = 1.0f / g.ch;
    g.bx1 = x1;
    g.by1 = y1;
    const size_t ncell = (size_t)kGridDim * kGridDim;
    VecU32 cell(rows), cur(ncell);
    cell_start.assign(ncell + 1, 0);
    sidx.assign(rows, 0);
    for (uint32_t i = 0; i < rows; ++i) {
        const int gx = grid_cell(xy[2 * (size_t)i], g.x0, g.inv_cw), gy = grid_cell(xy[2 * (size_t)i + 1], g.y0, g.inv_ch);
        cell[i] = (uint32_t)(gy * kGridDim + gx);
        ++cell_start[cell[i] + 1];
    }
    for (size_t k = 0; k < ncell; ++k) cell_start[k + 1] += cell_start[k];
    for (size_t k = 0; k < ncell; ++k) cur[k] = cell_start[k];
    for (uint32_t i = 0; i < rows; ++i) sidx[cur[cell[i]]++] = i;
    return true;
}
// the y interval of the keypoints of grid row gy (1 % of a cell on either side covers the float rounding of grid_cell)
AMC_GR_HD void grid_row_interval(float y0, float ch, int gy, double& ylo, double& yhi) {
    ylo = (double)y0 + ((double)gy - 0.01) * (double)ch;
    yhi = (double)y0 + ((double)gy + 1.01) * (double)ch;
}
// the cells [gx0, gx1] of a grid row that [xa, xb] covers; false: none
AMC_GR_HD bool grid_cells_of(double xa, double xb, float x0, float bx1, float inv_cw, int& gx0, int& gx1) {
    const double gx_lo = (double)x0 - 1.0, gx_hi = (double)bx1 + 1.0;
    if (!(xb >= gx_lo && xa <= gx_hi)) return false;
    gx0 = grid_cell((float)fmax(xa, gx_lo), x0, inv_cw);
    gx1 = grid_cell((float)fmin(xb, gx_hi), x0, inv_cw);
    return true;
}

}  // namespace guided
}  // namespace amc
