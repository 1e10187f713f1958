// pose_math.h — lane-local FP64 numerics of the relative-pose kernel (csrc/pose.hip).
//
// COLMAP 3.9.1 EstimateTwoViewGeometryPose (colmap/estimators/two_view_geometry.cc) and what it calls:
// DecomposeEssentialMatrix / PoseFromEssentialMatrix (geometry/essential_matrix.cc),
// DecomposeHomographyMatrix / PoseFromHomographyMatrix (geometry/homography_matrix.cc),
// TriangulatePoint / CalculateTriangulationAngles (geometry/triangulation.cc), CheckCheirality
// (geometry/pose.cc), Eigen::Quaterniond(Matrix3d).  Same conventions as tvg_math.h: straight-line
// scalar code, callable from the host (tests/shim) so the CPU suite can compare it bit-for-bit
// with oracle/tvg_oracle.cc; SVDs are taken from the round-robin Jacobi eigen-decomposition of
// A^T A.  All sizes are compile-time constants so that on the GPU every matrix
// lives in registers.
#pragma once

#include <cmath>

#include "tvg_math.h"

namespace amc {
namespace tvg {

// jacobi_eigen(N, a, v) of tvg_math.h with the matrix size a template parameter: the same
// operations in the same order, every index a compile-time constant after unrolling.
template <int N>
AMC_HD void jacobi_eigen_t(double (&a)[N * N], double (&v)[N * N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) v[i * N + j] = (i == j) ? 1.0 : 0.0;
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < N * N; ++i) total += a[i] * a[i];
    const double tol = total * 1e-32;
    constexpr int rounds = (N & 1) ? N : N - 1, np = N / 2;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) off += a[p * N + q] * a[p * N + q];
        if (!(off > tol)) break;
#pragma unroll
        for (int r = 0; r < rounds; ++r) {
            double c[np], s[np];
            bool act[np];
#pragma unroll
            for (int e = 0; e < np; ++e) {
                int p, q;
                jacobi_pair(N, r, e, p, q);
                act[e] = jacobi_rotation(a[p * N + p], a[q * N + q], a[p * N + q], c[e], s[e]);
            }
#pragma unroll
            for (int e = 0; e < np; ++e) {
                if (!act[e]) continue;
                int p, q;
                jacobi_pair(N, r, e, p, q);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double akp = a[k * N + p], akq = a[k * N + q];
                    a[k * N + p] = c[e] * akp - s[e] * akq;
                    a[k * N + q] = s[e] * akp + c[e] * akq;
                    const double vkp = v[k * N + p], vkq = v[k * N + q];
                    v[k * N + p] = c[e] * vkp - s[e] * vkq;
                    v[k * N + q] = s[e] * vkp + c[e] * vkq;
                }
            }
#pragma unroll
            for (int e = 0; e < np; ++e) {
                if (!act[e]) continue;
                int p, q;
                jacobi_pair(N, r, e, p, q);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double apk = a[p * N + k], aqk = a[q * N + k];
                    a[p * N + k] = c[e] * apk - s[e] * aqk;
                    a[q * N + k] = s[e] * apk + c[e] * aqk;
                }
            }
        }
    }
}

AMC_HD double mat3_det(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
AMC_HD void mat3_inv(const double* m, double* r) {
    const double d = mat3_det(m);
    r[0] = (m[4] * m[8] - m[5] * m[7]) / d; r[1] = (m[2] * m[7] - m[1] * m[8]) / d; r[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    r[3] = (m[5] * m[6] - m[3] * m[8]) / d; r[4] = (m[0] * m[8] - m[2] * m[6]) / d; r[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    r[6] = (m[3] * m[7] - m[4] * m[6]) / d; r[7] = (m[1] * m[6] - m[0] * m[7]) / d; r[8] = (m[0] * m[4] - m[1] * m[3]) / d;
}
AMC_HD void mat3_vec(const double* a, const double* x, double* r) {
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = a[3 * i] * x[0] + a[3 * i + 1] * x[1] + a[3 * i + 2] * x[2];
}
AMC_HD double vec3_norm(const double* a) { return dsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
AMC_HD void vec3_normalize(const double* a, double* r) {
    const double n = vec3_norm(a);
    r[0] = a[0] / n; r[1] = a[1] / n; r[2] = a[2] / n;
}

// A = U diag(S) V^T, S descending.  u_k = A v_k / s_k for the two largest singular values (the unit
// vector e_k when s_k is zero: U = I for the all-zero matrix, as Eigen returns), u_2 = u_0 x u_1.
AMC_HD void svd3(const double* A, double* U, double* S, double* V) {
    double At[9], ata[9], ev[9];
    mat3_t(A, At);
    mat3_mul(At, A, ata);
    jacobi_eigen_t<3>(ata, ev);
    int ord[3] = {0, 1, 2};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            // select by value instead of indexing `ata` with a run-time subscript
            const double di = ord[i] == 0 ? ata[0] : ord[i] == 1 ? ata[4] : ata[8];
            const double dj = ord[j] == 0 ? ata[0] : ord[j] == 1 ? ata[4] : ata[8];
            if (dj > di) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
        }
    double vc[3][3], uc[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double lam = ord[k] == 0 ? ata[0] : ord[k] == 1 ? ata[4] : ata[8];
        S[k] = dsqrt(lam < 0.0 ? 0.0 : lam);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            vc[k][i] = ord[k] == 0 ? ev[3 * i] : ord[k] == 1 ? ev[3 * i + 1] : ev[3 * i + 2];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (S[k] == 0.0) {
            uc[k][0] = k == 0 ? 1.0 : 0.0; uc[k][1] = k == 1 ? 1.0 : 0.0; uc[k][2] = 0.0;
            continue;
        }
        const double inv = 1.0 / S[k];
        double av[3];
        mat3_vec(A, vc[k], av);
        uc[k][0] = av[0] * inv; uc[k][1] = av[1] * inv; uc[k][2] = av[2] * inv;
    }
    uc[2][0] = uc[0][1] * uc[1][2] - uc[0][2] * uc[1][1];
    uc[2][1] = uc[0][2] * uc[1][0] - uc[0][0] * uc[1][2];
    uc[2][2] = uc[0][0] * uc[1][1] - uc[0][1] * uc[1][0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) { U[3 * i + k] = uc[k][i]; V[3 * i + k] = vc[k][i]; }
}

// Candidate poses (R, t) in the order COLMAP tries them; a later candidate wins a tie in the
// cheirality count.
struct PoseCands {
    int n;
    double R[4][9];
    double t[4][3];
    double nrm[4][3];  // plane normals of a homography's candidates (pose_candidates_H with normals = true)
};

// DecomposeEssentialMatrix + the four combinations of PoseFromEssentialMatrix
AMC_HD void pose_candidates_E(const double* E, PoseCands& c) {
    double U[9], S[3], V[9], Vt[9];
    svd3(E, U, S, V);
    mat3_t(V, Vt);
    if (mat3_det(U) < 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (mat3_det(Vt) < 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Wt[9], UW[9], R1[9], R2[9];
    mat3_t(W, Wt);
    mat3_mul(U, W, UW);
    mat3_mul(UW, Vt, R1);
    mat3_mul(U, Wt, UW);
    mat3_mul(UW, Vt, R2);
    const double u2[3] = {U[2], U[5], U[8]};
    double t[3];
    vec3_normalize(u2, t);
    c.n = 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 9; ++i) c.R[k][i] = (k & 1) ? R2[i] : R1[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) c.t[k][i] = k < 2 ? t[i] : t[i] * -1.0;
    }
}

AMC_HD int sign_of(double x) { return (0.0 < x) - (x < 0.0); }
AMC_HD double opposite_of_minor(const double* S, int row, int col) {
    const int col1 = col == 0 ? 1 : 0, col2 = col == 2 ? 1 : 2;
    const int row1 = row == 0 ? 1 : 0, row2 = row == 2 ? 1 : 2;
    return S[3 * row1 + col2] * S[3 * row2 + col1] - S[3 * row1 + col1] * S[3 * row2 + col2];
}
AMC_HD void homography_rotation(const double* Hn, const double* tstar, const double* n, double v, double* R) {
    double M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[3 * i + j] = (i == j ? 1.0 : 0.0) - (2.0 / v) * tstar[i] * n[j];
    mat3_mul(Hn, M, R);
}
// DecomposeHomographyMatrix: one candidate (pure rotation) or four
template <bool NORMALS = false>
AMC_HD void pose_candidates_H(const double* H, const double* K1, const double* K2, PoseCands& c) {
    double K2i[9], T[9], Hn[9];
    mat3_inv(K2, K2i);
    mat3_mul(K2i, H, T);
    mat3_mul(T, K1, Hn);
    {
        double U[9], Sv[3], V[9];
        svd3(Hn, U, Sv, V);
#pragma unroll
        for (int i = 0; i < 9; ++i) Hn[i] /= Sv[1];
    }
    if (mat3_det(Hn) < 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) Hn[i] *= -1.0;
    double Ht[9], S[9];
    mat3_t(Hn, Ht);
    mat3_mul(Ht, Hn, S);
    S[0] -= 1.0; S[4] -= 1.0; S[8] -= 1.0;
    double inf_norm = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double x = dabs(S[i]); inf_norm = inf_norm < x ? x : inf_norm; }  // std::max(inf_norm, x)
    if (inf_norm < 1e-3) {
        c.n = 1;
#pragma unroll
        for (int i = 0; i < 9; ++i) c.R[0][i] = Hn[i];
        c.t[0][0] = 0.0; c.t[0][1] = 0.0; c.t[0][2] = 0.0;
        if (NORMALS) { c.nrm[0][0] = 0.0; c.nrm[0][1] = 0.0; c.nrm[0][2] = 0.0; }
        return;
    }
    const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    const double rtM00 = dsqrt(M00), rtM11 = dsqrt(M11), rtM22 = dsqrt(M22);
    const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    const int e12 = sign_of(M12), e02 = sign_of(M02), e01 = sign_of(M01);
    const double nS0 = dabs(S[0]), nS1 = dabs(S[4]), nS2 = dabs(S[8]);
    int idx = 0;
    double nmax = nS0;
    if (nS1 > nmax) { idx = 1; nmax = nS1; }
    if (nS2 > nmax) { idx = 2; nmax = nS2; }
    double np1[3], np2[3];
    double Sii;
    if (idx == 0) {
        np1[0] = S[0]; np1[1] = S[1] + rtM22; np1[2] = S[2] + e12 * rtM11;
        np2[0] = S[0]; np2[1] = S[1] - rtM22; np2[2] = S[2] - e12 * rtM11;
        Sii = S[0];
    } else if (idx == 1) {
        np1[0] = S[1] + rtM22; np1[1] = S[4]; np1[2] = S[5] - e02 * rtM00;
        np2[0] = S[1] - rtM22; np2[1] = S[4]; np2[2] = S[5] + e02 * rtM00;
        Sii = S[4];
    } else {
        np1[0] = S[2] + e01 * rtM11; np1[1] = S[5] + rtM00; np1[2] = S[8];
        np2[0] = S[2] - e01 * rtM11; np2[1] = S[5] - rtM00; np2[2] = S[8];
        Sii = S[8];
    }
    const double traceS = S[0] + S[4] + S[8];
    const double v = 2.0 * dsqrt(1.0 + traceS - M00 - M11 - M22);
    const double ESii = sign_of(Sii);
    const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
    const double r = dsqrt(r_2), n_t = dsqrt(nt_2);
    double n1[3], n2[3];
    vec3_normalize(np1, n1);
    vec3_normalize(np2, n2);
    const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
    double t1s[3], t2s[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t1s[i] = half_nt * (esii_t_r * n2[i] - n_t * n1[i]);
        t2s[i] = half_nt * (esii_t_r * n1[i] - n_t * n2[i]);
    }
    double R1[9], R2[9], t1[3], t2[3];
    homography_rotation(Hn, t1s, n1, v, R1);
    mat3_vec(R1, t1s, t1);
    homography_rotation(Hn, t2s, n2, v, R2);
    mat3_vec(R2, t2s, t2);
    c.n = 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 9; ++i) c.R[k][i] = k < 2 ? R1[i] : R2[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double tv = k < 2 ? t1[i] : t2[i];
            c.t[k][i] = (k & 1) ? tv * -1.0 : tv;
            if (NORMALS) {  // n = {-n1, n1, -n2, n2}
                const double nv = k < 2 ? n1[i] : n2[i];
                c.nrm[k][i] = (k & 1) ? nv : nv * -1.0;
            }
        }
    }
}

// TriangulatePoint with P1 = [I | 0], P2 = [R | t]: the right singular vector of the smallest
// singular value of the 4 x 4 DLT system, dehomogenised
AMC_HD void triangulate_point(const double* R, const double* t, double x1, double y1, double x2, double y2, double* X) {
    const double P1[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    double P2[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) P2[i][j] = R[3 * i + j];
        P2[i][3] = t[i];
    }
    double A[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[0][j] = x1 * P1[2][j] - P1[0][j];
        A[1][j] = y1 * P1[2][j] - P1[1][j];
        A[2][j] = x2 * P2[2][j] - P2[0][j];
        A[3][j] = y2 * P2[2][j] - P2[1][j];
    }
    double ata[16], ev[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += A[k][i] * A[k][j];
            ata[4 * i + j] = sum;
        }
    jacobi_eigen_t<4>(ata, ev);
    // first minimum of the diagonal, selecting values rather than indexing with a run-time subscript
    double dmin = ata[0];
    double w = ev[12], e0 = ev[0], e1 = ev[4], e2 = ev[8];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (ata[5 * i] < dmin) { dmin = ata[5 * i]; e0 = ev[i]; e1 = ev[4 + i]; e2 = ev[8 + i]; w = ev[12 + i]; }
    X[0] = e0 / w; X[1] = e1 / w; X[2] = e2 / w;
}

// the per-candidate constants of CheckCheirality
struct CheiralityBounds { double max_depth, n2; };
AMC_HD CheiralityBounds cheirality_bounds(const double* R, const double* t) {
    double Rt[9], c[3];
    mat3_t(R, Rt);
    mat3_vec(Rt, t, c);
    CheiralityBounds b;
    b.max_depth = 1000.0 * vec3_norm(c);
    b.n2 = dsqrt(R[2] * R[2] + R[5] * R[5] + R[8] * R[8]);
    return b;
}
// one correspondence of CheckCheirality: triangulate, then both depths in (eps, max_depth)
AMC_HD bool cheirality_point(const double* R, const double* t, const CheiralityBounds& b, double x1, double y1,
                             double x2, double y2, double* X) {
    const double kMinDepth = 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
    triangulate_point(R, t, x1, y1, x2, y2, X);
    const double depth1 = (0.0 * X[0] + 0.0 * X[1] + 1.0 * X[2] + 0.0 * 1.0) * 1.0;
    if (depth1 > kMinDepth && depth1 < b.max_depth) {
        const double depth2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2] * 1.0) * b.n2;
        if (depth2 > kMinDepth && depth2 < b.max_depth) return true;
    }
    return false;
}

// Projection centre of the second camera, -R^T t, and the squared baseline to the first (origin)
AMC_HD void second_centre(const double* R, const double* t, double* c2, double* baseline2) {
    double Rt[9], v[3];
    mat3_t(R, Rt);
    mat3_vec(Rt, t, v);
    c2[0] = v[0] * -1.0; c2[1] = v[1] * -1.0; c2[2] = v[2] * -1.0;
    *baseline2 = (0.0 - c2[0]) * (0.0 - c2[0]) + (0.0 - c2[1]) * (0.0 - c2[1]) + (0.0 - c2[2]) * (0.0 - c2[2]);
}
// CalculateTriangulationAngles: the cosine whose acos is the point's angle (1.0, i.e. angle 0, for a
// zero denominator).  The angle itself is min(|acos(c)|, pi - |acos(c)|), taken on the host (libm).
AMC_HD double triangulation_cosine(const double* c2, double baseline2, const double* X) {
    double r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r1 += (X[k] - 0.0) * (X[k] - 0.0);
        r2 += (X[k] - c2[k]) * (X[k] - c2[k]);
    }
    const double den = 2.0 * dsqrt(r1 * r2);
    if (den == 0.0) return 1.0;
    const double nom = r1 + r2 - baseline2;
    return nom / den;
}

// Eigen::Quaterniond(rotation matrix) -> (w, x, y, z)
AMC_HD void rotation_to_quaternion(const double* m, double* q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = dsqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[7] - m[5]) * t;
        q[2] = (m[2] - m[6]) * t;
        q[3] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > (i == 0 ? m[0] : m[4])) i = 2;
        if (i == 0) {         // j = 1, k = 2
            t = dsqrt(m[0] - m[4] - m[8] + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[7] - m[5]) * t;
            q[2] = (m[3] + m[1]) * t;
            q[3] = (m[6] + m[2]) * t;
        } else if (i == 1) {  // j = 2, k = 0
            t = dsqrt(m[4] - m[8] - m[0] + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[2] - m[6]) * t;
            q[3] = (m[7] + m[5]) * t;
            q[1] = (m[1] + m[3]) * t;
        } else {              // j = 0, k = 1
            t = dsqrt(m[8] - m[0] - m[4] + 1.0);
            q[3] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[3] - m[1]) * t;
            q[1] = (m[2] + m[6]) * t;
            q[2] = (m[5] + m[7]) * t;
        }
    }
}

// ---- median by selection -----------------------------------------------------------------------
// The angle min(|acos(c)|, pi - |acos(c)|) falls as |c| grows, so the median angle belongs to the
// middle element(s) of the cosines ordered by |c|, largest first.  key(c) = the bit pattern of |c|
// (monotone for non-negative doubles).
AMC_HD uint64_t cosine_key(double c) {
    union { double d; uint64_t u; } b;
    b.d = c;
    return b.u & 0x7fffffffffffffffull;
}

// host side of the median: libm acos on the one or two selected cosines (cmed[0] = rank n/2,
// cmed[1] = rank n/2 - 1 of the cosines ordered by |c|, largest first), Median()'s rule for even n
inline double triangulation_angle_host(double c) {
    const double a = std::fabs(std::acos(c));
    const double b = 3.14159265358979323846 - a;
    return b < a ? b : a;  // std::min(a, b)
}
inline double median_angle_host(uint32_t n, const double* cmed) {
    if (n == 0) return 0.0;
    if (n % 2 == 0) return 0.5 * triangulation_angle_host(cmed[0]) + 0.5 * triangulation_angle_host(cmed[1]);
    return triangulation_angle_host(cmed[0]);
}

}  // namespace tvg
}  // namespace amc
