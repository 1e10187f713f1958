// tvg_math.h — lane-local FP64 numerics of the two-view verification kernels (csrc/tvg_core.h, included by tvg_e.hip and tvg_fh.hip).
//
// Every function here is straight-line scalar code one GPU lane runs on its own data (one
// RANSAC trial per lane for the minimal solvers; lane 0 for the local-optimisation solvers).
// AMC_HD makes them callable from the host as well, so tests can compare them bit-for-bit with
// the CPU oracle without a GPU.  No fast-math, no FMA contraction (-ffp-contract=off): IEEE
// double throughout, operation order fixed by the source.
//
// Algorithms (own restatement; COLMAP uses Eigen):
//   null spaces      Gauss-Jordan with full pivoting (minimal solvers)
//   least squares    smallest eigenvector(s) of A^T A by round-robin Jacobi
//   rank-2           project out the smallest right singular vector
//   roots            bottom-up bracketing over the derivative chain + bisection
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AMC_HD __host__ __device__ inline
#else
#define AMC_HD inline
#endif

#include <stdint.h>

namespace amc {
namespace tvg {

AMC_HD double dabs(double x) { return x < 0.0 ? -x : x; }
AMC_HD double dmax(double a, double b) { return a > b ? a : b; }  // std::max(a, b) semantics
#if defined(__HIP_DEVICE_COMPILE__)
AMC_HD double dsqrt(double x) { return __dsqrt_rn(x); }
#else
}  // namespace tvg
}  // namespace amc
#include <cmath>
namespace amc {
namespace tvg {
AMC_HD double dsqrt(double x) { return std::sqrt(x); }
#endif

AMC_HD void mat3_mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
AMC_HD void mat3_t(const double* a, double* c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * j + i];
}

// ---- residuals (colmap/estimators/utils.cc ComputeSquaredSampsonError; homography_matrix.cc;
//      translation_transform.h), exact operation order of SURVEY.md A.3 -------------------------
AMC_HD double sampson(const double* e, double x1_0, double x1_1, double x2_0, double x2_1) {
    const double Ex1_0 = e[0] * x1_0 + e[1] * x1_1 + e[2];
    const double Ex1_1 = e[3] * x1_0 + e[4] * x1_1 + e[5];
    const double Ex1_2 = e[6] * x1_0 + e[7] * x1_1 + e[8];
    const double Etx2_0 = e[0] * x2_0 + e[3] * x2_1 + e[6];
    const double Etx2_1 = e[1] * x2_0 + e[4] * x2_1 + e[7];
    const double x2tEx1 = x2_0 * Ex1_0 + x2_1 * Ex1_1 + Ex1_2;
    return x2tEx1 * x2tEx1 / (Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1 + Etx2_0 * Etx2_0 + Etx2_1 * Etx2_1);
}
AMC_HD double h_residual(const double* H, double s_0, double s_1, double d_0, double d_1) {
    const double pd_0 = H[0] * s_0 + H[1] * s_1 + H[2];
    const double pd_1 = H[3] * s_0 + H[4] * s_1 + H[5];
    const double pd_2 = H[6] * s_0 + H[7] * s_1 + H[8];
    const double inv_pd_2 = 1.0 / pd_2;
    const double dd_0 = d_0 - pd_0 * inv_pd_2;
    const double dd_1 = d_1 - pd_1 * inv_pd_2;
    return dd_0 * dd_0 + dd_1 * dd_1;
}
AMC_HD double t_residual(const double* t, double s_0, double s_1, double d_0, double d_1) {
    const double d0 = d_0 - s_0 - t[0];
    const double d1 = d_1 - s_1 - t[1];
    return d0 * d0 + d1 * d1;
}

// ---- Jacobi eigen-decomposition, symmetric n x n (n <= 9), round-robin ordering -----------------
// A sweep is `m` rounds (m = n if n is odd, n - 1 otherwise); the pairs of a round are disjoint,
// their rotations are computed from the matrix at the start of the round and applied as one
// similarity transform: columns (A and V) first, rows second.  See oracle/tvg_oracle.cc.
AMC_HD int jacobi_num_rounds(int n) { return (n & 1) ? n : n - 1; }
AMC_HD int jacobi_pairs_per_round(int n) { return n / 2; }
// e-th pair of round r -> (p, q), p < q
AMC_HD void jacobi_pair(int n, int r, int e, int& p, int& q) {
    const int m = jacobi_num_rounds(n);
    int x, y;
    if (e < (m - 1) / 2) {
        x = (r + e + 1) % m;
        y = (r - (e + 1) + m) % m;
    } else {
        x = r;
        y = n - 1;
    }
    p = x < y ? x : y;
    q = x < y ? y : x;
}
// rotation that annihilates a[p][q]; returns false (no rotation) when it is already zero
AMC_HD bool jacobi_rotation(double app, double aqq, double apq, double& c, double& s) {
    if (apq == 0.0) return false;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (dabs(theta) + dsqrt(theta * theta + 1.0));
    c = 1.0 / dsqrt(t * t + 1.0);
    s = t * c;
    return true;
}
AMC_HD void jacobi_eigen(int n, double* a, double* v) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    double total = 0.0;
    for (int i = 0; i < n * n; ++i) total += a[i] * a[i];
    const double tol = total * 1e-32;
    const int rounds = jacobi_num_rounds(n), np = jacobi_pairs_per_round(n);
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (!(off > tol)) break;
        for (int r = 0; r < rounds; ++r) {
            double c[5], s[5];
            bool act[5];
            for (int e = 0; e < np; ++e) {
                int p, q;
                jacobi_pair(n, r, e, p, q);
                act[e] = jacobi_rotation(a[p * n + p], a[q * n + q], a[p * n + q], c[e], s[e]);
            }
            for (int e = 0; e < np; ++e) {
                if (!act[e]) continue;
                int p, q;
                jacobi_pair(n, r, e, p, q);
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c[e] * akp - s[e] * akq;
                    a[k * n + q] = s[e] * akp + c[e] * akq;
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c[e] * vkp - s[e] * vkq;
                    v[k * n + q] = s[e] * vkp + c[e] * vkq;
                }
            }
            for (int e = 0; e < np; ++e) {
                if (!act[e]) continue;
                int p, q;
                jacobi_pair(n, r, e, p, q);
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c[e] * apk - s[e] * aqk;
                    a[q * n + k] = s[e] * apk + c[e] * aqk;
                }
            }
        }
    }
}
AMC_HD void smallest_eigvec9(double* ata, double* x) {
    double v[81];
    jacobi_eigen(9, ata, v);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (ata[i * 9 + i] < ata[best * 9 + best]) best = i;
    for (int i = 0; i < 9; ++i) x[i] = v[i * 9 + best];
}

// ---- null space of an R x 9 matrix (R <= 8), Gauss-Jordan with full pivoting -------------------
AMC_HD void nullspace9(int R, double* a, double* ns) {
    int perm[9];
    for (int j = 0; j < 9; ++j) perm[j] = j;
    for (int r = 0; r < R; ++r) {
        int pi = r, pj = r;
        double pv = -1.0;
        for (int i = r; i < R; ++i)
            for (int j = r; j < 9; ++j) {
                const double v = dabs(a[i * 9 + j]);
                if (v > pv) { pv = v; pi = i; pj = j; }
            }
        if (pi != r)
            for (int j = 0; j < 9; ++j) { const double t = a[r * 9 + j]; a[r * 9 + j] = a[pi * 9 + j]; a[pi * 9 + j] = t; }
        if (pj != r) {
            for (int i = 0; i < R; ++i) { const double t = a[i * 9 + r]; a[i * 9 + r] = a[i * 9 + pj]; a[i * 9 + pj] = t; }
            const int t = perm[r]; perm[r] = perm[pj]; perm[pj] = t;
        }
        const double inv = 1.0 / a[r * 9 + r];
        for (int j = 0; j < 9; ++j) a[r * 9 + j] = a[r * 9 + j] * inv;
        for (int i = 0; i < R; ++i) {
            if (i == r) continue;
            const double f = a[i * 9 + r];
            for (int j = 0; j < 9; ++j) a[i * 9 + j] = a[i * 9 + j] - f * a[r * 9 + j];
        }
    }
    for (int k = 0; k < 9 - R; ++k) {
        double* x = ns + k * 9;
        for (int j = 0; j < 9; ++j) x[j] = 0.0;
        x[perm[R + k]] = 1.0;
        for (int i = 0; i < R; ++i) x[perm[i]] = -a[i * 9 + (R + k)];
    }
}

// Same algorithm, fully unrolled with compile-time indices: the matrix lives in registers, row and
// column swaps are predicated selects (no dynamic indexing => no scratch memory on the GPU).
// Identical arithmetic and identical pivot choice (first maximum in row-major scan order).
template <int R>
AMC_HD void nullspace_reg(double (&a)[R][9], double (&ns)[9 - R][9]) {
    int perm[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) perm[j] = j;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int pi = r, pj = r;
        double pv = -1.0;
#pragma unroll
        for (int i = r; i < R; ++i)
#pragma unroll
            for (int j = r; j < 9; ++j) {
                const double v = dabs(a[i][j]);
                const bool g = v > pv;
                pv = g ? v : pv;
                pi = g ? i : pi;
                pj = g ? j : pj;
            }
#pragma unroll
        for (int i = r + 1; i < R; ++i) {
            const bool sw = (pi == i);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double t = a[r][j], u = a[i][j];
                a[r][j] = sw ? u : t;
                a[i][j] = sw ? t : u;
            }
        }
#pragma unroll
        for (int j = r + 1; j < 9; ++j) {
            const bool sw = (pj == j);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double t = a[i][r], u = a[i][j];
                a[i][r] = sw ? u : t;
                a[i][j] = sw ? t : u;
            }
            const int tp = perm[r], up = perm[j];
            perm[r] = sw ? up : tp;
            perm[j] = sw ? tp : up;
        }
        const double inv = 1.0 / a[r][r];
#pragma unroll
        for (int j = 0; j < 9; ++j) a[r][j] = a[r][j] * inv;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i == r) continue;
            const double f = a[i][r];
#pragma unroll
            for (int j = 0; j < 9; ++j) a[i][j] = a[i][j] - f * a[r][j];
        }
    }
#pragma unroll
    for (int k = 0; k < 9 - R; ++k)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double v = (perm[R + k] == j) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) v = (perm[i] == j) ? -a[i][R + k] : v;
            ns[k][j] = v;
        }
}

// ---- real roots, ascending --------------------------------------------------------------------
// Everything below is written with compile-time degrees and fully unrolled loops: coefficient
// arrays are then indexed statically and live in registers on the GPU (a run-time degree puts them
// in scratch memory, one dependent memory round trip per Horner step).
template <int DEG>
AMC_HD double poly_eval_t(const double (&c)[DEG + 1], double x) {
    double v = c[DEG];
#pragma unroll
    for (int i = DEG - 1; i >= 0; --i) v = v * x + c[i];
    return v;
}
AMC_HD double poly_eval(const double* c, int deg, double x) {
    double v = c[deg];
    for (int i = deg - 1; i >= 0; --i) v = v * x + c[i];
    return v;
}
// real roots of c (degree DEG >= 2, c[DEG] != 0) given the real roots `crit` of its derivative (coefficients dc):
// every monotone interval with a sign change holds one root - bisected until the bracket is narrower than 2^-26 of
// where it sits, then three Newton steps accepted only strictly inside the bracket (oracle: bracket_root) - ascending.
// Written for a SIMT lane: every array is indexed with compile-time indices (slots are filled by predicated
// selects), so nothing lives in scratch memory, and the intervals are first classified and the brackets then
// solved over the lane's OWN sign-change intervals only, two at a time on independent dependency chains - a wave's
// lanes solve different polynomials, and a loop over all DEG + 1 intervals would make every lane wait for a bracket
// whenever any lane has a sign change in that slot.  Per interval the arithmetic is that of the plain loop (for each
// interval: f(lo) == 0 -> root lo unless it repeats the previous root; f(hi) == 0 or equal signs -> nothing; else
// bracket_root): same roots, same bits, same order.
constexpr double kRootRelWidth = 1.4901161193847656e-08;  // 2^-26
// One level of the chain, as a lane holds it: the interval edges, the polynomial's values there, what each interval
// [edges[i], edges[i + 1]] is (1 exact root at the lower edge, 2 sign change, 0 nothing), the sign-change intervals
// still to be solved (bit i of todo) and their roots (val).
template <int DEG>
struct RootLevel {
    double edges[DEG + 1], f[DEG + 1], val[DEG];
    int kind[DEG];
    int ne;
    unsigned todo;
};
template <int DEG>
AMC_HD void roots_classify(const double (&c)[DEG + 1], const double (&crit)[DEG - 1], int nc, RootLevel<DEG>& L) {
    double bound = 0.0;
#pragma unroll
    for (int i = 0; i < DEG; ++i) bound = dmax(bound, dabs(c[i] / c[DEG]));
    bound = 1.0 + bound;
    // edges: -bound, the critical points inside (-bound, bound), bound
#pragma unroll
    for (int s = 0; s <= DEG; ++s) L.edges[s] = 0.0;
    L.edges[0] = -bound;
    int ne = 1;
#pragma unroll
    for (int j = 0; j < DEG - 1; ++j) {
        const bool put = j < nc && crit[j] > -bound && crit[j] < bound;
#pragma unroll
        for (int s = 1; s < DEG; ++s) L.edges[s] = (put && ne == s) ? crit[j] : L.edges[s];
        ne += put ? 1 : 0;
    }
#pragma unroll
    for (int s = 1; s <= DEG; ++s) L.edges[s] = (ne == s) ? bound : L.edges[s];
    ne += 1;
    L.ne = ne;
#pragma unroll
    for (int s = 0; s <= DEG; ++s) L.f[s] = poly_eval_t<DEG>(c, L.edges[s]);
    unsigned todo = 0u;
#pragma unroll
    for (int i = 0; i < DEG; ++i) {
        const bool valid = i + 1 < ne;
        const bool zlo = L.f[i] == 0.0;
        const bool chg = !zlo && L.f[i + 1] != 0.0 && ((L.f[i] < 0.0) != (L.f[i + 1] < 0.0));
        L.kind[i] = valid ? (zlo ? 1 : (chg ? 2 : 0)) : 0;
        todo |= (valid && chg) ? (1u << i) : 0u;
        L.val[i] = 0.0;
    }
    L.todo = todo;
}
// one bisection step of a bracket, given f(mid); a bracket that is not active is left as it is
AMC_HD void bracket_step(double mid, double fm, double& lo, double& hi, bool neg_lo, bool& act, bool& zero) {
    // (& and |, not && and ||: nothing here is worth a branch.)  The plain loop keeps f(lo) and replaces it by f(mid)
    // whenever lo moves; it only ever looks at its sign, and lo moves exactly when f(mid) has that sign - so the sign
    // of f(lo) never changes and is all a bracket carries: neg_lo.  |x| through fabs (one source modifier on the GPU;
    // x < 0 ? -x : x differs from it only in the sign of a zero, which the comparison below cannot see).
    const bool hit = act & (fm == 0.0);
    const bool move = act & !hit;
    const bool left = (fm < 0.0) == neg_lo;   // f(mid) has the sign of f(lo): the root is to the right of mid
    const bool set_lo = hit | (move & left), set_hi = hit | (move & !left);
    lo = set_lo ? mid : lo;
    hi = set_hi ? mid : hi;
    zero = zero | hit;
    const bool wide = !(hi - lo <= kRootRelWidth * (__builtin_fabs(lo) + __builtin_fabs(hi)));
    act = move & wide;
}
// Two sign-change brackets at once (oracle: bracket_root, once per bracket): bisection until the bracket is narrower
// than 2^-26 of where it sits, then three Newton steps accepted only strictly inside it.  The two brackets may belong
// to different polynomials (c0 / c1, derivatives dc0 / dc1): a lane solving its own polynomial passes it twice, the
// wave-balanced solver of tvg_core.h hands a lane brackets of other lanes' polynomials.  Each bracket sees exactly the
// plain loop's arithmetic and its own 200-step cap; the second one is idle when `two` is false.
template <int DEG>
AMC_HD void bracket_pair_root(const double (&c0)[DEG + 1], const double (&dc0)[DEG], const double (&c1)[DEG + 1],
                              const double (&dc1)[DEG], double lo0, double hi0, double flo0, double lo1, double hi1,
                              double flo1, bool one, bool two, double& r0_out, double& r1_out) {
    bool act0 = one, act1 = two, zero0 = false, zero1 = false;
    const bool neg0 = flo0 < 0.0, neg1 = flo1 < 0.0;
    for (int it = 0; it < 200 && (act0 || act1); ++it) {
        const double mid0 = 0.5 * (lo0 + hi0), mid1 = 0.5 * (lo1 + hi1);
        act0 = act0 && !(mid0 == lo0 || mid0 == hi0);
        act1 = act1 && !(mid1 == lo1 || mid1 == hi1);
        const double fm0 = poly_eval_t<DEG>(c0, mid0), fm1 = poly_eval_t<DEG>(c1, mid1);
        // the plain loop's step ("root hit: stop; same sign as f(lo): lo = mid, else hi = mid; narrow enough: stop")
        // written with selects only, so that on the GPU the two brackets' Horner chains - the step's latency - run
        // interleaved instead of one after the other inside two branches
        bracket_step(mid0, fm0, lo0, hi0, neg0, act0, zero0);
        bracket_step(mid1, fm1, lo1, hi1, neg1, act1, zero1);
    }
    double r0 = 0.5 * (lo0 + hi0), r1 = 0.5 * (lo1 + hi1);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const double f0 = poly_eval_t<DEG>(c0, r0), f1 = poly_eval_t<DEG>(c1, r1);
        const double d0 = poly_eval_t<DEG - 1>(dc0, r0), d1 = poly_eval_t<DEG - 1>(dc1, r1);
        const double n0 = r0 - f0 / d0, n1 = r1 - f1 / d1;
        r0 = (!zero0 && n0 > lo0 && n0 < hi0) ? n0 : r0;
        r1 = (!zero1 && n1 > lo1 && n1 < hi1) ? n1 : r1;
    }
    r0_out = r0;
    r1_out = r1;
}
// the lane's own sign-change brackets, two at a time
template <int DEG>
AMC_HD void roots_solve_own(const double (&c)[DEG + 1], const double (&dc)[DEG], RootLevel<DEG>& L) {
    unsigned todo = L.todo;
    while (todo) {
        const int cur0 = __builtin_ctz(todo);
        todo &= todo - 1u;
        const bool two = todo != 0u;
        const int cur1 = two ? __builtin_ctz(todo) : cur0;
        todo = two ? (todo & (todo - 1u)) : todo;
        double lo0 = L.edges[0], hi0 = L.edges[1], flo0 = L.f[0];
        double lo1 = L.edges[0], hi1 = L.edges[1], flo1 = L.f[0];
#pragma unroll
        for (int i = 1; i < DEG; ++i) {
            const bool me0 = cur0 == i, me1 = cur1 == i;
            lo0 = me0 ? L.edges[i] : lo0;
            hi0 = me0 ? L.edges[i + 1] : hi0;
            flo0 = me0 ? L.f[i] : flo0;
            lo1 = me1 ? L.edges[i] : lo1;
            hi1 = me1 ? L.edges[i + 1] : hi1;
            flo1 = me1 ? L.f[i] : flo1;
        }
        double r0, r1;
        bracket_pair_root<DEG>(c, dc, c, dc, lo0, hi0, flo0, lo1, hi1, flo1, true, two, r0, r1);
#pragma unroll
        for (int i = 0; i < DEG; ++i) L.val[i] = (cur0 == i) ? r0 : ((two && cur1 == i) ? r1 : L.val[i]);
    }
    L.todo = 0u;
}
// the ordered root list of a solved level
template <int DEG>
AMC_HD int roots_assemble(const RootLevel<DEG>& L, double (&roots)[DEG]) {
    int nr = 0;
    double last = 0.0;
#pragma unroll
    for (int s = 0; s < DEG; ++s) roots[s] = 0.0;
#pragma unroll
    for (int i = 0; i < DEG; ++i) {
        const bool push = (L.kind[i] == 1 && (nr == 0 || last != L.edges[i])) || L.kind[i] == 2;
        const double v = L.kind[i] == 1 ? L.edges[i] : L.val[i];
#pragma unroll
        for (int s = 0; s < DEG; ++s) roots[s] = (push && nr == s) ? v : roots[s];
        last = push ? v : last;
        nr += push ? 1 : 0;
    }
    {   // the upper end itself
        double fe = L.f[1], ee = L.edges[1];
#pragma unroll
        for (int s = 2; s <= DEG; ++s) {
            fe = (L.ne - 1 == s) ? L.f[s] : fe;
            ee = (L.ne - 1 == s) ? L.edges[s] : ee;
        }
        const bool push = fe == 0.0 && (nr == 0 || last != ee);
#pragma unroll
        for (int s = 0; s < DEG; ++s) roots[s] = (push && nr == s) ? ee : roots[s];
        nr += push ? 1 : 0;
    }
    return nr;
}
template <int DEG>
AMC_HD int roots_between_t(const double (&c)[DEG + 1], const double (&dc)[DEG], const double (&crit)[DEG - 1], int nc,
                           double (&roots)[DEG]) {
    RootLevel<DEG> L;
    roots_classify<DEG>(c, crit, nc, L);
    roots_solve_own<DEG>(c, dc, L);
    return roots_assemble<DEG>(L, roots);
}
// J-th derivative of c (degree DEG), coefficient by coefficient as the chain of successive
// derivatives produces it: d[m] = (...((c[m+J] * (m+J)) * (m+J-1)) ... * (m+1))
template <int DEG, int J>
AMC_HD void poly_derivative_t(const double (&c)[DEG + 1], double (&d)[DEG - J + 1]) {
#pragma unroll
    for (int m = 0; m <= DEG - J; ++m) {
        double v = c[m + J];
#pragma unroll
        for (int f = m + J; f > m; --f) v = v * f;
        d[m] = v;
    }
}
// roots of the derivative of c that has degree R (the (DEG-R)-th), recursively from the roots of
// the next one; the linear one is solved directly
template <int DEG, int R>
struct RootChain {
    static AMC_HD int run(const double (&c)[DEG + 1], double (&roots)[R]) {
        double crit[R - 1];
        const int nc = RootChain<DEG, R - 1>::run(c, crit);
        double d[R + 1], dd[R];
        poly_derivative_t<DEG, DEG - R>(c, d);
        poly_derivative_t<DEG, DEG - R + 1>(c, dd);  // the derivative of d (= the next level down)
        return roots_between_t<R>(d, dd, crit, nc, roots);
    }
};
template <int DEG>
struct RootChain<DEG, 1> {
    static AMC_HD int run(const double (&c)[DEG + 1], double (&roots)[1]) {
        double d[2];
        poly_derivative_t<DEG, DEG - 1>(c, d);
        roots[0] = -d[0] / d[1];
        return 1;
    }
};
// all real roots of a polynomial of degree <= DEG (low -> high coefficients), ascending;
// leading zero coefficients lower the degree
template <int DEG>
struct RealRoots {
    static AMC_HD int run(const double (&c)[DEG + 1], double* roots) {
        if (c[DEG] == 0.0) {
            double lower[DEG];
#pragma unroll
            for (int i = 0; i < DEG; ++i) lower[i] = c[i];
            return RealRoots<DEG - 1>::run(lower, roots);
        }
        double r[DEG];
        const int nr = RootChain<DEG, DEG>::run(c, r);
#pragma unroll
        for (int i = 0; i < DEG; ++i) roots[i] = r[i];
        return nr;
    }
};
template <>
struct RealRoots<0> {
    static AMC_HD int run(const double (&)[1], double*) { return 0; }
};
template <int DEG>
AMC_HD int real_roots_t(const double (&c)[DEG + 1], double* roots) { return RealRoots<DEG>::run(c, roots); }
// run-time degree (<= 10) front end
AMC_HD int real_roots(const double* c_in, int deg_in, double* roots) {
#define AMC_RR_CASE(D)                                   \
    case D: {                                            \
        double cc[D + 1];                                \
        for (int i = 0; i <= D; ++i) cc[i] = c_in[i];    \
        return real_roots_t<D>(cc, roots);               \
    }
    switch (deg_in) {
        AMC_RR_CASE(1) AMC_RR_CASE(2) AMC_RR_CASE(3) AMC_RR_CASE(4) AMC_RR_CASE(5)
        AMC_RR_CASE(6) AMC_RR_CASE(7) AMC_RR_CASE(8) AMC_RR_CASE(9) AMC_RR_CASE(10)
        default: return 0;
    }
#undef AMC_RR_CASE
}

// ---- 7-point fundamental matrix (FundamentalMatrixSevenPointEstimator::Estimate) ---------------
// x1/y1: image-1 coords of the 7 samples, x2/y2: image-2.  Returns #models (<= 3), row-major.
AMC_HD int estimate_f7(const double* x1s, const double* y1s, const double* x2s, const double* y2s,
                       double* models /* 3 x 9 */) {
    double A[7][9];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double x0 = x1s[i], y0 = y1s[i], x1 = x2s[i], y1 = y2s[i];
        A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
        A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
        A[i][6] = x0; A[i][7] = y0; A[i][8] = 1;
    }
    double ns[2][9];
    nullspace_reg<7>(A, ns);
    double f1[9], f2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { f2[i] = ns[1][i]; f1[i] = ns[0][i] - f2[i]; }
    // det(lambda f1 + f2): entries e = f2 + f1 lambda; 2x2 minors (degree 2), then expansion
    double mn[3][3];
    const int mi[3][4] = {{4, 8, 5, 7}, {3, 8, 5, 6}, {3, 7, 4, 6}};
    for (int m = 0; m < 3; ++m) {
        const int i = mi[m][0], j = mi[m][1], k = mi[m][2], l = mi[m][3];
        const double u0 = f2[i] * f2[j], u1 = f2[i] * f1[j] + f1[i] * f2[j], u2 = f1[i] * f1[j];
        const double w0 = f2[k] * f2[l], w1 = f2[k] * f1[l] + f1[k] * f2[l], w2 = f1[k] * f1[l];
        mn[m][0] = u0 - w0; mn[m][1] = u1 - w1; mn[m][2] = u2 - w2;
    }
    double c[4] = {0, 0, 0, 0};
    const double sg[3] = {1.0, -1.0, 1.0};
    for (int m = 0; m < 3; ++m) {
        const double e0 = f2[m], e1 = f1[m];
        c[0] += sg[m] * (e0 * mn[m][0]);
        c[1] += sg[m] * (e0 * mn[m][1] + e1 * mn[m][0]);
        c[2] += sg[m] * (e0 * mn[m][2] + e1 * mn[m][1]);
        c[3] += sg[m] * (e1 * mn[m][2]);
    }
    double roots[3] = {0.0, 0.0, 0.0};
    const int nr = real_roots_t<3>(c, roots);
    // models are the accepted roots in ascending order; compaction with predicated static slots
    int nm = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double lambda = roots[i];
        double F[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) F[k] = lambda * f1[k] + f2[k];
        const bool ok = (i < nr) && !(dabs(F[8]) < 1e-10);
        const double inv = F[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) F[k] = F[k] / inv;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const bool put = ok && (nm == sl);
#pragma unroll
            for (int k = 0; k < 9; ++k) models[9 * sl + k] = put ? F[k] : models[9 * sl + k];
        }
        nm += ok ? 1 : 0;
    }
    return nm;
}

// ---- 8-point tail: from A^T A of the normalised design matrix to F ----------------------------
AMC_HD void f8_from_vec(const double* f, const double* T1, const double* T2, double* F) {
    double Fh[9], Ft[9], ftf[9], v[9];
    for (int k = 0; k < 9; ++k) Fh[k] = f[k];
    mat3_t(Fh, Ft);
    mat3_mul(Ft, Fh, ftf);
    jacobi_eigen(3, ftf, v);
    int b = 0;
    for (int i = 1; i < 3; ++i)
        if (ftf[i * 3 + i] < ftf[b * 3 + b]) b = i;
    const double v3[3] = {v[0 * 3 + b], v[1 * 3 + b], v[2 * 3 + b]};
    double Fr[9];
    for (int i = 0; i < 3; ++i) {
        const double fv = Fh[3 * i] * v3[0] + Fh[3 * i + 1] * v3[1] + Fh[3 * i + 2] * v3[2];
        for (int j = 0; j < 3; ++j) Fr[3 * i + j] = Fh[3 * i + j] - fv * v3[j];
    }
    double T2t[9], tmp[9];
    mat3_t(T2, T2t);
    mat3_mul(T2t, Fr, tmp);
    mat3_mul(tmp, T1, F);
}
AMC_HD void f8_from_ata(double* ata, const double* T1, const double* T2, double* F) {
    double f[9];
    smallest_eigvec9(ata, f);
    f8_from_vec(f, T1, T2, F);
}

// ---- homography tail: H = T2^-1 * Hhat * T1 ---------------------------------------------------
AMC_HD void h_denormalize(const double* h, const double* T1, const double* T2, double* H) {
    double T2i[9];
    const double inv_nf = 1.0 / T2[0];
    T2i[0] = inv_nf; T2i[1] = 0; T2i[2] = -T2[2] * inv_nf;
    T2i[3] = 0; T2i[4] = inv_nf; T2i[5] = -T2[5] * inv_nf;
    T2i[6] = 0; T2i[7] = 0; T2i[8] = 1;
    double tmp[9];
    mat3_mul(T2i, h, tmp);
    mat3_mul(tmp, T1, H);
}

// CenterAndNormalizeImagePoints for exactly 4 points in the oracle's det_sum64 order, which for
// n = 4 reduces to (v0 + v2) + (v1 + v3).  Writes normalised points and T (row-major).
AMC_HD void normalize4(const double* x, const double* y, double* nx, double* ny, double* T) {
    const double cx = ((x[0] + x[2]) + (x[1] + x[3])) / 4;
    const double cy = ((y[0] + y[2]) + (y[1] + y[3])) / 4;
    double d[4];
    for (int i = 0; i < 4; ++i) {
        const double dx = x[i] - cx, dy = y[i] - cy;
        d[i] = dx * dx + dy * dy;
    }
    double rms = (d[0] + d[2]) + (d[1] + d[3]);
    rms = dsqrt(rms / 4);
    const double nf = dsqrt(2.0) / rms;
    T[0] = nf; T[1] = 0; T[2] = -nf * cx;
    T[3] = 0; T[4] = nf; T[5] = -nf * cy;
    T[6] = 0; T[7] = 0; T[8] = 1;
    for (int i = 0; i < 4; ++i) {
        const double np0 = T[0] * x[i] + T[1] * y[i] + T[2];
        const double np1 = T[3] * x[i] + T[4] * y[i] + T[5];
        const double np2 = T[6] * x[i] + T[7] * y[i] + T[8];
        const double inv = 1.0 / np2;
        nx[i] = np0 * inv;
        ny[i] = np1 * inv;
    }
}

// minimal 4-point homography in closed form (same operation order as the oracle's h4_closed_form): with
// S = [s0 s1 s2], D = [d0 d1 d2] (homogeneous columns), adj(S) has rows s1 x s2, s2 x s0, s0 x s1, and
//   Hhat = D * diag(mu_k / lam_k) * adj(S),   lam = adj(S) s3,  mu = adj(D) d3
// maps s_k -> d_k for all four points (projective basis change); scaled to unit Frobenius norm.
AMC_HD void h4_closed_form(const double* sx, const double* sy, const double* dx, const double* dy, double* h) {
    double a[3][3], b[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        a[k][0] = sy[p] - sy[q]; a[k][1] = sx[q] - sx[p]; a[k][2] = sx[p] * sy[q] - sy[p] * sx[q];
        b[k][0] = dy[p] - dy[q]; b[k][1] = dx[q] - dx[p]; b[k][2] = dx[p] * dy[q] - dy[p] * dx[q];
    }
    double c[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double lam = (a[k][0] * sx[3] + a[k][1] * sy[3]) + a[k][2];
        const double mu = (b[k][0] * dx[3] + b[k][1] * dy[3]) + b[k][2];
        const double r = mu / lam;
        c[k][0] = r * a[k][0]; c[k][1] = r * a[k][1]; c[k][2] = r * a[k][2];
    }
    double n2 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        h[j] = (dx[0] * c[0][j] + dx[1] * c[1][j]) + dx[2] * c[2][j];
        h[3 + j] = (dy[0] * c[0][j] + dy[1] * c[1][j]) + dy[2] * c[2][j];
        h[6 + j] = (c[0][j] + c[1][j]) + c[2][j];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) n2 = n2 + h[j] * h[j];
    const double inv = 1.0 / dsqrt(n2);
#pragma unroll
    for (int j = 0; j < 9; ++j) h[j] = h[j] * inv;
}

AMC_HD void estimate_h4(const double* x1, const double* y1, const double* x2, const double* y2, double* H) {
    double n1x[4], n1y[4], n2x[4], n2y[4], T1[9], T2[9];
    normalize4(x1, y1, n1x, n1y, T1);
    normalize4(x2, y2, n2x, n2y, T2);
    double h[9];
    h4_closed_form(n1x, n1y, n2x, n2y, h);
    h_denormalize(h, T1, T2, H);
}

// ---- 5-point essential matrix -------------------------------------------------------------------
// Monomial tables shared with the oracle's formulation (P1: x y z 1; P2: x^2 y^2 xy xz x yz y z^2 z 1;
// P3: x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy xz^2 xz x yz^2 yz y z^3 z^2 z 1).
// kM11[i][j] = P2 index of P1[i]*P1[j]; kM21[i][j] = P3 index of P2[i]*P1[j].
#define AMC_M11 {{0, 2, 3, 4}, {2, 1, 5, 6}, {3, 5, 7, 8}, {4, 6, 8, 9}}
#define AMC_M21 {{0, 2, 4, 5}, {3, 1, 6, 7}, {2, 3, 8, 9}, {4, 8, 10, 11}, {5, 9, 11, 12}, \
                 {8, 6, 13, 14}, {9, 7, 14, 15}, {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}}

// index tables as constexpr functions so that, with the loops unrolled, every array index below is a
// compile-time constant (registers instead of scratch memory on the GPU)
AMC_HD constexpr int e5_m11(int i, int j) {
    constexpr int M[4][4] = AMC_M11;
    return M[i][j];
}
AMC_HD constexpr int e5_m21(int i, int j) {
    constexpr int M[10][4] = AMC_M21;
    return M[i][j];
}
// r = a * b for a, b linear in (x, y, z, 1): 10 quadratic monomial coefficients
AMC_HD void e5_mul11(const double (&a)[4], const double (&b)[4], double (&r)[10]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) r[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[e5_m11(i, j)] += a[i] * b[j];
}
// r = a * b for a quadratic, b linear: 20 cubic monomial coefficients
AMC_HD void e5_mul21(const double (&a)[10], const double (&b)[4], double (&r)[20]) {
#pragma unroll
    for (int i = 0; i < 20; ++i) r[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[e5_m21(i, j)] += a[i] * b[j];
}
// r = a * b for univariate polynomials of degree DA, DB (low -> high), i outer / j inner
template <int DA, int DB>
AMC_HD void e5_pmul(const double (&a)[DA + 1], const double (&b)[DB + 1], double (&r)[DA + DB + 1]) {
#pragma unroll
    for (int i = 0; i <= DA + DB; ++i) r[i] = 0.0;
#pragma unroll
    for (int i = 0; i <= DA; ++i)
#pragma unroll
        for (int j = 0; j <= DB; ++j) r[i + j] += a[i] * b[j];
}

// The 5-point solver in three steps, so that the root finding in the middle can be swapped for a
// wave-cooperative version where one problem is solved by a whole wave (tvg_core.h, local optimisation):
//   e5_build:   nsp (4 x 9 basis; rows: x, y, z, 1 directions) -> B(z) (3 x 3 polynomial matrix) and
//               det B(z), degree 10
//   real roots of det B
//   e5_models:  roots -> essential matrices (row-major), in root order
struct E5Polys {
    double B[3][3][5];  // B[k][0..1]: degree 3, B[k][2]: degree 4 (low -> high)
    double det[11];
};
// rows 4..9, columns 10..19 of the eliminated constraint matrix -> B(z) and det B(z)
AMC_HD void e5_finish(const double (&hl)[6][10], E5Polys& P) {
    double (&B)[3][3][5] = P.B;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double (&hi)[10] = hl[2 * k];
        const double (&lo)[10] = hl[2 * k + 1];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int d = 0; d < 5; ++d) B[k][t][d] = 0.0;
        double (&a)[5] = B[k][0];
        double (&b)[5] = B[k][1];
        double (&c)[5] = B[k][2];
        a[2] += hi[0]; a[1] += hi[1]; a[0] += hi[2];
        b[2] += hi[3]; b[1] += hi[4]; b[0] += hi[5];
        c[3] += hi[6]; c[2] += hi[7]; c[1] += hi[8]; c[0] += hi[9];
        a[3] -= lo[0]; a[2] -= lo[1]; a[1] -= lo[2];
        b[3] -= lo[3]; b[2] -= lo[4]; b[1] -= lo[5];
        c[4] -= lo[6]; c[3] -= lo[7]; c[2] -= lo[8]; c[1] -= lo[9];
    }
    // det B(z) with the oracle's accumulation order: pz_mul (i outer, j inner), sub, add.  B[k][0]
    // and B[k][1] are cubics stored in 5 slots: the products below use their 4 coefficients.
    double (&det)[11] = P.det;
    {
        double b00[4], b01[4], b10[4], b11[4], b20[4], b21[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b00[i] = B[0][0][i]; b01[i] = B[0][1][i]; b10[i] = B[1][0][i];
            b11[i] = B[1][1][i]; b20[i] = B[2][0][i]; b21[i] = B[2][1][i];
        }
        double u8[8], w8[8], u7[7], w7[7], m0[8], m1[8], m2[7];
        e5_pmul<3, 4>(b11, B[2][2], u8); e5_pmul<4, 3>(B[1][2], b21, w8);
#pragma unroll
        for (int i = 0; i <= 7; ++i) m0[i] = u8[i] - w8[i];
        e5_pmul<3, 4>(b10, B[2][2], u8); e5_pmul<4, 3>(B[1][2], b20, w8);
#pragma unroll
        for (int i = 0; i <= 7; ++i) m1[i] = u8[i] - w8[i];
        e5_pmul<3, 3>(b10, b21, u7); e5_pmul<3, 3>(b11, b20, w7);
#pragma unroll
        for (int i = 0; i <= 6; ++i) m2[i] = u7[i] - w7[i];
        double q0[11], q1[11], q2[11];
        e5_pmul<3, 7>(b00, m0, q0);
        e5_pmul<3, 7>(b01, m1, q1);
        e5_pmul<4, 6>(B[0][2], m2, q2);
#pragma unroll
        for (int i = 0; i <= 10; ++i) det[i] = (q0[i] - q1[i]) + q2[i];
    }
}
// The 10 x 20 constraint matrix (rows: det E = 0, then the nine entries of 2 E E^T E - tr(E E^T) E = 0; columns: the
// cubic monomials in e5_m21's order), one row at a time: sink(r, row).  A row is handed over as soon as it is
// complete, so that a caller which stores it elsewhere (the kernel's minimal solver: tvg_core.h) never holds the
// whole matrix.
template <class Sink>
AMC_HD void e5_constraint_rows(const double* nsp, Sink& sink) {
    double e[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int d = 0; d < 4; ++d) e[k][d] = nsp[d * 9 + k];
    {   // det(E) -> row 0: (t0 - t1) + t2
        double a[10], b[10], d[10], t[20], row[20];
        e5_mul11(e[4], e[8], a); e5_mul11(e[5], e[7], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[0], row);
        e5_mul11(e[3], e[8], a); e5_mul11(e[5], e[6], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[1], t);
#pragma unroll
        for (int i = 0; i < 20; ++i) row[i] = row[i] - t[i];
        e5_mul11(e[3], e[7], a); e5_mul11(e[4], e[6], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[2], t);
#pragma unroll
        for (int i = 0; i < 20; ++i) row[i] = row[i] + t[i];
        sink(0, row);
    }
    double tr[10];
    {   // tr(E E^T) = (EEt[0][0] + EEt[1][1]) + EEt[2][2]
        double dg[3][10];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a[10], b[10], c[10];
            e5_mul11(e[3 * i], e[3 * i], a);
            e5_mul11(e[3 * i + 1], e[3 * i + 1], b);
            e5_mul11(e[3 * i + 2], e[3 * i + 2], c);
#pragma unroll
            for (int t = 0; t < 10; ++t) dg[i][t] = (a[t] + b[t]) + c[t];
        }
#pragma unroll
        for (int t = 0; t < 10; ++t) tr[t] = (dg[0][t] + dg[1][t]) + dg[2][t];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double eet[3][10];  // row i of E E^T
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double a[10], b[10], c[10];
            e5_mul11(e[3 * i], e[3 * j], a);
            e5_mul11(e[3 * i + 1], e[3 * j + 1], b);
            e5_mul11(e[3 * i + 2], e[3 * j + 2], c);
#pragma unroll
            for (int t = 0; t < 10; ++t) eet[j][t] = (a[t] + b[t]) + c[t];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // ((a + b) + c) * 2 - d
            double row[20], t[20];
            e5_mul21(eet[0], e[j], row);
            e5_mul21(eet[1], e[3 + j], t);
#pragma unroll
            for (int q = 0; q < 20; ++q) row[q] = row[q] + t[q];
            e5_mul21(eet[2], e[6 + j], t);
#pragma unroll
            for (int q = 0; q < 20; ++q) row[q] = row[q] + t[q];
            e5_mul21(tr, e[3 * i + j], t);
#pragma unroll
            for (int q = 0; q < 20; ++q) row[q] = row[q] * 2.0 - t[q];
            sink(1 + 3 * i + j, row);
        }
    }
}
struct E5MatrixSink {  // the rows into a 10 x 20 array
    double (*G)[20];
    AMC_HD void operator()(int r, const double (&row)[20]) {
#pragma unroll
        for (int c = 0; c < 20; ++c) G[r][c] = row[c];
    }
};
AMC_HD void e5_build(const double* nsp, E5Polys& P) {
    double G[10][20];
    {
        E5MatrixSink sink{G};
        e5_constraint_rows(nsp, sink);
    }
    // Gauss-Jordan with partial pivoting on the left 10 x 10 block.  The only run-time index is the
    // pivot row of the swap.
#pragma unroll
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        double pv = dabs(G[col][col]);
#pragma unroll
        for (int r = col + 1; r < 10; ++r)
            if (dabs(G[r][col]) > pv) { pv = dabs(G[r][col]); piv = r; }
        if (piv != col) {
#pragma unroll
            for (int r = col + 1; r < 10; ++r)
                if (r == piv) {
#pragma unroll
                    for (int c = 0; c < 20; ++c) { const double t = G[col][c]; G[col][c] = G[r][c]; G[r][c] = t; }
                }
        }
        const double inv = 1.0 / G[col][col];
#pragma unroll
        for (int c = 0; c < 20; ++c) G[col][c] = G[col][c] * inv;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = G[r][col];
#pragma unroll
            for (int c = 0; c < 20; ++c) G[r][c] = G[r][c] - f * G[col][c];
        }
    }
    double hl[6][10];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 10; ++c) hl[r][c] = G[4 + r][10 + c];
    e5_finish(hl, P);
}
// returns the number of models written (a root whose (x, y) blows up is skipped, as upstream skips
// |X(2)| < 1e-10 of the unit null vector of B(z)); every E is scaled to unit Frobenius norm
// one root z of det B -> its essential matrix (false: skipped)
AMC_HD bool e5_model_from_root(const double* nsp, const E5Polys& P, double z, double* E) {
    const double (&B)[3][3][5] = P.B;
    // null vector of B(z) = the longest cross product of two of its rows (the oracle's statement of
    // upstream's JacobiSVD null vector X, skipped when |X(2)| < 1e-10)
    double Bz[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Bz[k][0] = poly_eval(B[k][0], 3, z);
        Bz[k][1] = poly_eval(B[k][1], 3, z);
        Bz[k][2] = poly_eval(B[k][2], 4, z);
    }
    double X0 = 0.0, X1 = 0.0, X2 = 0.0, best_n2 = -1.0;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int iu = pr == 2 ? 1 : 0, iw = pr == 0 ? 1 : 2;
        const double c0 = Bz[iu][1] * Bz[iw][2] - Bz[iu][2] * Bz[iw][1];
        const double c1 = Bz[iu][2] * Bz[iw][0] - Bz[iu][0] * Bz[iw][2];
        const double c2 = Bz[iu][0] * Bz[iw][1] - Bz[iu][1] * Bz[iw][0];
        const double n2 = c0 * c0 + c1 * c1 + c2 * c2;
        if (n2 > best_n2) { best_n2 = n2; X0 = c0; X1 = c1; X2 = c2; }
    }
    const double nn = dsqrt(best_n2);
    X0 = X0 / nn; X1 = X1 / nn; X2 = X2 / nn;
    if (!(dabs(X2) >= 1e-10)) return false;
    const double x = X0 / X2;
    const double y = X1 / X2;
    for (int k = 0; k < 9; ++k) E[k] = x * nsp[k] + y * nsp[9 + k] + z * nsp[18 + k] + nsp[27 + k];
    double n2 = 0.0;
    for (int k = 0; k < 9; ++k) n2 += E[k] * E[k];
    const double nrm = dsqrt(n2);
    for (int k = 0; k < 9; ++k) E[k] = E[k] / nrm;
    return true;
}
AMC_HD int e5_models(const double* nsp, const E5Polys& P, const double* roots, int nr, double* models) {
    int nm = 0;
    for (int i = 0; i < nr; ++i)
        if (e5_model_from_root(nsp, P, roots[i], models + 9 * nm)) ++nm;
    return nm;
}
// nsp -> #models (<= 10), row-major
AMC_HD int e5_from_nullspace(const double* nsp, double* models /* 10 x 9 */) {
    E5Polys P;
    e5_build(nsp, P);
    double roots[10];
    const int nr = real_roots_t<10>(P.det, roots);
    return e5_models(nsp, P, roots, nr, models);
}

// minimal 5-point
AMC_HD int estimate_e5_minimal(const double* x1, const double* y1, const double* x2, const double* y2,
                               double* models) {
    double A[5][9];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        A[i][0] = x2[i] * x1[i]; A[i][1] = x2[i] * y1[i]; A[i][2] = x2[i];
        A[i][3] = y2[i] * x1[i]; A[i][4] = y2[i] * y1[i]; A[i][5] = y2[i];
        A[i][6] = x1[i]; A[i][7] = y1[i]; A[i][8] = 1;
    }
    double ns[4][9];
    nullspace_reg<5>(A, ns);
    double nsp[4 * 9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 9; ++j) nsp[k * 9 + j] = ns[k][j];
    return e5_from_nullspace(nsp, models);
}
// least-squares 5-point (local optimisation): 4 smallest eigenvectors of A^T A
// 4-D "null space" of the least-squares 5-point: the 4 smallest eigenvectors of A^T A, given its
// eigen-decomposition (ata: eigenvalues on the diagonal, v: eigenvectors in columns)
template <class PA, class PV>
AMC_HD void e5_nullspace_from_eig(PA ata, PV v, double* nsp) {
    int order[9];
    for (int i = 0; i < 9; ++i) order[i] = i;
    for (int i = 0; i < 9; ++i)
        for (int j = i + 1; j < 9; ++j)
            if (ata[order[j] * 9 + order[j]] < ata[order[i] * 9 + order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 9; ++i) nsp[k * 9 + i] = v[i * 9 + order[3 - k]];
}
AMC_HD int e5_from_ata(double* ata, double* models) {
    double v[81], nsp[4 * 9];
    jacobi_eigen(9, ata, v);
    e5_nullspace_from_eig(ata, v, nsp);
    return e5_from_nullspace(nsp, models);
}

// ---- homography inlier decisions in FP32, with a bound on their own error -------------------------
// The counting loop's pre-filter (tvg_core.h: count_lanes_h32 evaluates exactly this, two points per packed instruction).
// With s = 1 / sqrt(T) folded into rows 0 / 1 of the model and into the image-2 coordinates, a point is an inlier
// iff t = u'^2 + v'^2 - w^2 <= 0 (u' = c' w - p0', v' = d' w - p1').  Every operation below is one FP32 rounding
// (u = 2^-24); C = largest |coordinate| of the pair; A0 = (|m0| + |m1|) C + |m2|, A1, Aw likewise (m = scaled model):
//   |p32 - p| <= 5u A,   |u32 - u'| <= E0 + u |u32| with E0 = 5u max(A0, A1) + 6u (C s) Aw,   Ew = 5u Aw
//   |t32 - t| <= 6u |t32| + 6u R32 + 2 E0 (|u32| + |v32|) + 2 Ew |w32| + 2 E0^2 + Ew^2
// The FP64 test is itself only trusted outside |t| <= 2e-8 R, so a point is decided iff
//   |t32| > 4.2e-7 R32 + kE (|u32| + |v32|) + kW |w32| + K0,   kE = 2.05 E0, kW = 2.05 Ew, K0 = 2.05 (2 E0^2 + Ew^2)
// (> 2 % of slack for the rounding of the bound itself and the 6u |t32| term; NaN / inf compare false = undecided).
// The loop evaluates a bound that is no smaller and costs two fused multiply-adds on values it has anyway
// (S32 = u32^2 + v32^2, R32): for any beta > 0,  kE (|u| + |v|) <= kE sqrt(2 S) <= beta S + kE^2 / (2 beta)  and
// kW |w| = kW sqrt(R) <= beta R + kW^2 / (4 beta)  (2ab <= a^2 + b^2), hence
//   band = bS S32 + cR R32 + K1 >= 4.2e-7 R32 + kE (|u32| + |v32|) + kW |w32| + K0,
//   bS = beta (1 + 1e-4), cR = (4.2e-7 + beta)(1 + 1e-4), K1 = (K0 + kE^2 / (2 beta) + kW^2 / (4 beta))(1 + 1e-4)
// (the 1e-4 covers S32 / R32 versus the exact squares and the two roundings of the sum of positive terms).  The
// two sides meet where beta = kE / (2 |w|); beta is chosen per model for |w| ~ max(|m8|, Aw / 8) and kept within
// [2^-22, 2^-6], so even a wild model decides everything farther than ~3 % from the threshold circle.
struct H32Model {
    float m[9];
    float kE, kW, K0;   // the reference bound (h32_band_ref: tests compare the loop's bound against it)
    float bS, cR, K1;   // the loop's bound
    float qR, qK;       // the same test folded into one expression: q = u32^2 + v32^2 + qR R32 + qK > 0  =>  t32 - band > 0
};
AMC_HD float f32_up(double v) {  // a float >= v (v >= 0)
    return (float)(v * (1.0 + 1e-6));
}
AMC_HD H32Model h32_prepare(const double* model, double s, double C) {
    const double U = 5.9604644775390625e-08;  // 2^-24
    double msc[9];
    for (int i = 0; i < 9; ++i) msc[i] = i < 6 ? model[i] * s : model[i];
    const double A0 = (dabs(msc[0]) + dabs(msc[1])) * C + dabs(msc[2]);
    const double A1 = (dabs(msc[3]) + dabs(msc[4])) * C + dabs(msc[5]);
    const double Aw = (dabs(msc[6]) + dabs(msc[7])) * C + dabs(msc[8]);
    const double E0 = 5.0 * U * dmax(A0, A1) + 6.0 * U * (C * s) * Aw;
    const double Ew = 5.0 * U * Aw;
    H32Model h;
    for (int i = 0; i < 9; ++i) h.m[i] = (float)msc[i];
    const double kE = 2.05 * E0, kW = 2.05 * Ew, K0 = 2.05 * (2.0 * E0 * E0 + Ew * Ew) + 1e-30;
    h.kE = f32_up(kE);
    h.kW = f32_up(kW);
    h.K0 = f32_up(K0);
    const double wt = dmax(dabs(msc[8]), 0.125 * Aw);
    double beta = kE / (2.0 * wt);
    if (!(beta >= 2.384185791015625e-07)) beta = 2.384185791015625e-07;  // 2^-22 (also catches NaN: wt = 0 / inf)
    if (beta > 0.015625) beta = 0.015625;                                 // 2^-6
    const double bs = (double)(float)(beta * 1.0001);   // the float the loop multiplies S32 by; everything else follows it
    const double b0 = bs / 1.0001;
    h.bS = (float)bs;
    h.cR = f32_up((4.2e-7 + b0) * 1.0001);
    h.K1 = f32_up((K0 + kE * kE / (2.0 * b0) + kW * kW / (4.0 * b0)) * 1.0001);
    // q > 0 is meant to imply S - R > band + (rounding of q itself), S = u32^2 + v32^2: the inequality
    // (1 - bS) S - (1 + cR) R - K1 > 0 divided by its first coefficient, with 1e-6 of slack on every coefficient for
    // the three roundings of the folded expression (<= 3 x 2^-24 of the largest term)
    const double qs = (1.0 - (double)h.bS) * (1.0 - 1e-6);
    h.qR = -f32_up((1.0 + (double)h.cR) * (1.0 + 1e-6) / qs);
    h.qK = -f32_up((double)h.K1 * (1.0 + 1e-6) / qs);
    return h;
}
// t = u'^2 + v'^2 - w^2 and the bound on its error, for one point (V = float) or for two at once (V = a two-float
// vector: the kernel's v_pk_fma_f32 form) - the same expression either way
struct H32Ops {
    static AMC_HD float fma(float a, float b, float c) { return fmaf(a, b, c); }
    static AMC_HD float abs(float a) { return fabsf(a); }
    static AMC_HD float splat(float a) { return a; }
};
template <class V, class Ops>
AMC_HD void h32_eval(const H32Model& h, V a, V b, V cs, V ds, V& t, V& band) {
    const V p0 = Ops::fma(Ops::splat(h.m[0]), a, Ops::fma(Ops::splat(h.m[1]), b, Ops::splat(h.m[2])));
    const V p1 = Ops::fma(Ops::splat(h.m[3]), a, Ops::fma(Ops::splat(h.m[4]), b, Ops::splat(h.m[5])));
    const V w = Ops::fma(Ops::splat(h.m[6]), a, Ops::fma(Ops::splat(h.m[7]), b, Ops::splat(h.m[8])));
    const V u = Ops::fma(cs, w, -p0), v = Ops::fma(ds, w, -p1);
    const V R = w * w;
    const V S = Ops::fma(u, u, v * v);
    t = S - R;
    band = Ops::fma(Ops::splat(h.bS), S, Ops::fma(Ops::splat(h.cR), R, Ops::splat(h.K1)));
}
// The counting loop only needs an UPPER bound of a model's inlier count (a model whose bound reaches the best count so
// far is re-scored exactly anyway), i.e. the points that are outliers beyond doubt: q > 0 with
//   q = u32^2 + v32^2 + qR R32 + qK,  qR <= -(1 + cR) / (1 - bS),  qK <= -K1 / (1 - bS)   =>   t32 > band, an outlier by h32_eval.
// NaN / inf make q NaN or -inf: not an outlier beyond doubt.
template <class V, class Ops>
AMC_HD V h32_outlier_q(const H32Model& h, V a, V b, V cs, V ds) {
    const V p0 = Ops::fma(Ops::splat(h.m[0]), a, Ops::fma(Ops::splat(h.m[1]), b, Ops::splat(h.m[2])));
    const V p1 = Ops::fma(Ops::splat(h.m[3]), a, Ops::fma(Ops::splat(h.m[4]), b, Ops::splat(h.m[5])));
    const V w = Ops::fma(Ops::splat(h.m[6]), a, Ops::fma(Ops::splat(h.m[7]), b, Ops::splat(h.m[8])));
    const V u = Ops::fma(cs, w, -p0), v = Ops::fma(ds, w, -p1);
    const V R = w * w;
    return Ops::fma(u, u, Ops::fma(v, v, Ops::fma(Ops::splat(h.qR), R, Ops::splat(h.qK))));
}
// the reference bound of the same point (the right-hand side of the inequality above)
AMC_HD float h32_band_ref(const H32Model& h, float a, float b, float cs, float ds) {
    const float p0 = fmaf(h.m[0], a, fmaf(h.m[1], b, h.m[2]));
    const float p1 = fmaf(h.m[3], a, fmaf(h.m[4], b, h.m[5]));
    const float w = fmaf(h.m[6], a, fmaf(h.m[7], b, h.m[8]));
    const float u = fmaf(cs, w, -p0), v = fmaf(ds, w, -p1);
    const float R = w * w;
    const float auv = fabsf(u) + fabsf(v);
    return fmaf(4.2e-7f, R, fmaf(h.kE, auv, fmaf(h.kW, fabsf(w), h.K0)));
}
// one point (a, b: image-1 coordinates; cs, ds: image-2 coordinates times s, all rounded to float):
// 1 inlier, 0 outlier, -1 undecided
AMC_HD int h32_point(const H32Model& h, float a, float b, float cs, float ds) {
    float t, band;
    h32_eval<float, H32Ops>(h, a, b, cs, ds, t, band);
    if (!(fabsf(t) > band)) return -1;
    return t < 0.0f ? 1 : 0;
}

// ---- Sampson outliers in FP32, with a bound on the evaluation's own error -----------------------------------------
// The counting loop of the fundamental / essential RANSACs needs, like the homography one, only an upper bound of a
// model's inlier count: the correspondences that are outliers beyond doubt.  With u = 2^-24, C = largest |coordinate|
// of the pair, M the FP64 model, (A, B), (C', D') a correspondence, and in exact arithmetic
//   E1_k = M_k0 A + M_k1 B + M_k2,  T2_k = M_0k C' + M_1k D' + M_2k,  cc = C' E1_0 + D' E1_1 + E1_2,
//   den = E1_0^2 + E1_1^2 + T2_0^2 + T2_1^2                       (the point is an outlier iff cc^2 > T den),
// the FP32 evaluation below (operands rounded to float, one rounding per fused multiply-add) satisfies
//   |e1_k - E1_k| <= 4.01 u Ah_k,  Ah_k = (|M_k0| + |M_k1|) C + |M_k2|     (likewise t2_k with Bh_k)
//   |cc32 - cc|   <= Ec = 8 u (C (Ah_0 + Ah_1) + Ah_2)
//   |den32 - den| <= 8.03 u Dh + 4.1 u den32,  Dh = Ah_0^2 + Ah_1^2 + Bh_0^2 + Bh_1^2
// and, since (|x| - E)^2 >= (1 - beta) x^2 - E^2 (1 / beta - 1)  (2 E |x| <= beta x^2 + E^2 / beta),
//   q = cc32^2 + qR den32 + qK > 0,   with  qS = (1 - beta)(1 - u)  divided out:
//   qR <= -T (1 + 1e-7)(1 + 4.1 u) / qS,  qK <= -(Ec^2 (1 / beta - 1) + T (1 + 1e-7) 8.03 u Dh) / qS
// implies cc^2 > T (1 + 1e-7) den: an outlier for the reference's FP64 residual as well (when q > 0, |cc| exceeds
// 16 Ec ~ 7.6e-6 x the sum of the magnitudes of its terms, so the FP64 evaluation is good to ~1e-11 - far inside the
// 1e-7 margin).  beta = 2^-8; every coefficient carries 1e-6 of slack for the three roundings of q itself.  NaN / inf
// anywhere make q NaN or -inf: not an outlier beyond doubt.  A model FP32 cannot resolve (Ec^2 / beta beyond T den)
// simply decides nothing.
struct S32Model {
    float m[9];
    float qR, qK;
};
AMC_HD S32Model s32_prepare(const double* model, double T, double C) {
    const double U = 5.9604644775390625e-08;  // 2^-24
    const double* M = model;
    const double A0 = (dabs(M[0]) + dabs(M[1])) * C + dabs(M[2]);
    const double A1 = (dabs(M[3]) + dabs(M[4])) * C + dabs(M[5]);
    const double A2 = (dabs(M[6]) + dabs(M[7])) * C + dabs(M[8]);
    const double B0 = (dabs(M[0]) + dabs(M[3])) * C + dabs(M[6]);
    const double B1 = (dabs(M[1]) + dabs(M[4])) * C + dabs(M[7]);
    const double Ec = 8.0 * U * (C * (A0 + A1) + A2);
    const double Dh = A0 * A0 + A1 * A1 + B0 * B0 + B1 * B1;
    const double beta = 0.00390625;  // 2^-8
    const double Tq = T * (1.0 + 1e-7);
    S32Model h;
    for (int i = 0; i < 9; ++i) h.m[i] = (float)M[i];
    const double qs = (1.0 - beta) * (1.0 - U) * (1.0 - 1e-6);
    h.qR = -f32_up(Tq * (1.0 + 4.1 * U) * (1.0 + 1e-6) / qs);
    h.qK = -f32_up(((Ec * Ec * (1.0 / beta - 1.0) + Tq * 8.03 * U * Dh) * (1.0 + 1e-6) + 1e-30) / qs);
    return h;
}
template <class V, class Ops>
AMC_HD V s32_outlier_q(const S32Model& h, V a, V b, V c, V d) {
    const V e0 = Ops::fma(Ops::splat(h.m[0]), a, Ops::fma(Ops::splat(h.m[1]), b, Ops::splat(h.m[2])));
    const V e1 = Ops::fma(Ops::splat(h.m[3]), a, Ops::fma(Ops::splat(h.m[4]), b, Ops::splat(h.m[5])));
    const V e2 = Ops::fma(Ops::splat(h.m[6]), a, Ops::fma(Ops::splat(h.m[7]), b, Ops::splat(h.m[8])));
    const V t0 = Ops::fma(Ops::splat(h.m[0]), c, Ops::fma(Ops::splat(h.m[3]), d, Ops::splat(h.m[6])));
    const V t1 = Ops::fma(Ops::splat(h.m[1]), c, Ops::fma(Ops::splat(h.m[4]), d, Ops::splat(h.m[7])));
    const V cc = Ops::fma(c, e0, Ops::fma(d, e1, e2));
    const V den = Ops::fma(e0, e0, Ops::fma(e1, e1, Ops::fma(t0, t0, t1 * t1)));
    return Ops::fma(cc, cc, Ops::fma(Ops::splat(h.qR), den, Ops::splat(h.qK)));
}

// ---- mt19937 tempering + libstdc++ uniform_int_distribution<uint32_t> (Lemire) ------------------
AMC_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

}  // namespace tvg
}  // namespace amc
