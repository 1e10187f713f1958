/*
 * oracle/tvg_oracle.cc — CPU restatement of COLMAP 3.9.1's two-view geometric verification
 * (EstimateTwoViewGeometry: LO-RANSAC over F (7-pt / 8-pt), H (DLT), E (5-pt), model selection,
 * watermark test), SURVEY.md Appendix A.3.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pycolmap_amd/ may link, import or call this file.
 *
 * PARITY UNPINNED.  The arithmetic lives in the un-vendored COLMAP 3.9.1
 * (/root/reference/CMakeLists.txt:17) which is not available here, and the reference has no
 * tests for this path.  This file restates the published control flow literally —
 *   colmap/optim/loransac.h  LORANSAC::Estimate            -> lo_ransac()
 *   colmap/optim/ransac.h    RANSAC ctor, ComputeNumTrials  -> compute_num_trials(), ransac_max_trials()
 *   colmap/optim/support_measurement.cc InlierSupportMeasurer -> evaluate_support(), better()
 *   colmap/optim/random_sampler.cc, colmap/math/random.cc   -> Prng, Sampler (std::mt19937 +
 *        std::uniform_int_distribution<uint32_t>, exactly what COLMAP calls)
 *   colmap/estimators/fundamental_matrix.cc  7-point / 8-point
 *   colmap/estimators/homography_matrix.cc   normalised DLT
 *   colmap/estimators/utils.cc   CenterAndNormalizeImagePoints, ComputeSquaredSampsonError
 *   colmap/estimators/translation_transform.h
 *   colmap/estimators/two_view_geometry.cc   Estimate{Calibrated,Uncalibrated}TwoViewGeometry,
 *        DetectWatermark
 * anchored on the reference-side facts that ARE verifiable:
 *   estimator pairs 7pt/8pt, H/H, 5pt/5pt   /root/reference/pycolmap/estimators/fundamental_matrix.h:26-28,
 *        homography_matrix.h:25, essential_matrix.h:48-49
 *   SetPRNGSeed(0) before each single-pair estimate   .../fundamental_matrix.h:21
 *   E threshold = mean of max_error / focal            .../essential_matrix.h:41-46
 *   option names + Python-side RANSAC defaults         /root/reference/pycolmap/optim/bindings.h:10-25,
 *        .../estimators/two_view_geometry.h:41-63; config enum order ...:67-77
 *
 * Deliberate, documented deviations (COLMAP leans on Eigen, absent here):
 *   D1  Null spaces come from Gauss-Jordan elimination with full pivoting (minimal solvers) or
 *       the smallest eigenvector of A^T A by round-robin Jacobi (least-squares solvers) instead of
 *       Eigen::JacobiSVD; rank-2 enforcement projects out the smallest right singular vector.
 *   D2  Polynomial roots on the real line instead of companion-matrix eigenvalues: every monotone stretch (between
 *       the roots of the derivative, found the same way one degree down) with a sign change is bisected to 2^-26 of
 *       its position and finished with three bracketed Newton steps (round 3; rounds 1-2 bisected to the last ulp);
 *       models are tried in ascending root order.
 *   D3  Sums over correspondences (centroids, A^T A, inlier residual sums) use a fixed 64-way
 *       strided + butterfly order (det_sum64) so that a 64-lane wavefront reproduces them
 *       bit-for-bit; COLMAP sums sequentially.  Same values up to rounding.
 *   D4  Per-pair reseed: the PRNG is re-created with seed 0 at the start of every image pair
 *       (COLMAP's pipeline carries a thread_local PRNG across pairs and is therefore not
 *       reproducible run-to-run).
 * Compile with -ffp-contract=off (oracle/Makefile): no FMA contraction, IEEE double throughout.
 *
 * FROZEN (round 4).  ORACLE_TVG_VERSION below names this arithmetic - D1..D4 as stated above.  It is the parity
 * target of the HIP kernels and of the committed fixtures (tests/golden/tvg_golden_v4.npz,
 * tests/ref2/deviation_budget.json); tests/test_oracle_frozen_cpu.py regenerates both from this file and fails on any
 * difference.  A change of the arithmetic here - in particular another restatement of D1 / D2 to suit a kernel, as
 * round 3 did with D2 - is a new version, new fixtures and a line in DESIGN.md section 2; it is not done for speed.
 *
 * DEVIATION TOGGLES (oracle/Makefile builds one extra library per flag; tests/ref2 measures what each
 * deviation does to configs / masks / trial counts):
 *   -DORACLE_SEQ_SUMS         D3 off: every det_sum64 becomes the sequential sum COLMAP writes
 *   -DORACLE_LAPACK_SVD       D1 off: null spaces, least-squares null vectors, the rank-2 projection and the
 *                             pose SVDs come from LAPACK dgesvd (an actual SVD of the design matrix, as
 *                             Eigen::JacobiSVD is) instead of Gauss-Jordan / Jacobi on A^T A
 *   -DORACLE_COMPANION_ROOTS  D2 off: polynomial roots are the eigenvalues of the companion matrix (LAPACK
 *                             dgeev), |imag| <= 1e-10 kept, in the solver's order - COLMAP's
 *                             FindPolynomialRootsCompanionMatrix
 * LAPACK is the LAPACKE interface of the OpenBLAS that scipy bundles, loaded at run time from the path in the
 * environment variable ORACLE_LAPACK_LIB (tests/ref2/variants.py sets it).  The default build uses none of it.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#if defined(ORACLE_LAPACK_SVD) || defined(ORACLE_COMPANION_ROOTS)
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#define ORACLE_USES_LAPACK 1
#endif

namespace {

#ifdef ORACLE_USES_LAPACK
// LAPACKE (row-major layout = 101) entry points of scipy's OpenBLAS
struct Lapack {
    int (*dgesvd)(int, char, char, int, int, double*, int, double*, double*, int, double*, int, double*) = nullptr;
    int (*dgeev)(int, char, char, int, double*, int, double*, double*, double*, int, double*, int) = nullptr;
};
Lapack& lapack() {
    static Lapack L = [] {
        Lapack l;
        const char* path = std::getenv("ORACLE_LAPACK_LIB");
        void* h = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
        if (!h) {
            std::fprintf(stderr, "oracle: ORACLE_LAPACK_LIB not set or not loadable (%s)\n", path ? dlerror() : "unset");
            std::abort();
        }
        for (const char* prefix : {"scipy_", ""}) {
            if (!l.dgesvd) l.dgesvd = reinterpret_cast<decltype(l.dgesvd)>(dlsym(h, (std::string(prefix) + "LAPACKE_dgesvd").c_str()));
            if (!l.dgeev) l.dgeev = reinterpret_cast<decltype(l.dgeev)>(dlsym(h, (std::string(prefix) + "LAPACKE_dgeev").c_str()));
        }
        if (!l.dgesvd || !l.dgeev) {
            std::fprintf(stderr, "oracle: LAPACKE_dgesvd / LAPACKE_dgeev not found in %s\n", path);
            std::abort();
        }
        return l;
    }();
    return L;
}
// A (m x n, row-major, preserved) = U diag(S) V^T, S descending; Vt n x n row-major (row k = k-th right singular
// vector), U m x m row-major when asked for
void lapack_svd(int m, int n, const double* A, double* S, double* U, double* Vt) {
    std::vector<double> a(A, A + static_cast<size_t>(m) * n), superb(std::max(1, std::min(m, n))), udummy(1);
    const int info = lapack().dgesvd(101, U ? 'A' : 'N', 'A', m, n, a.data(), n, S, U ? U : udummy.data(), U ? m : 1, Vt, n,
                                     superb.data());
    if (info != 0) std::fprintf(stderr, "oracle: dgesvd info %d\n", info);
}
#endif

// ----------------------------------------------------------------------------------------------
// options / types (mirrors of COLMAP structs; field names as pycolmap exposes them)
// ----------------------------------------------------------------------------------------------
struct RansacOptions {
    double max_error;
    double min_inlier_ratio;
    double confidence;
    double dyn_num_trials_multiplier;
    int64_t min_num_trials;
    int64_t max_num_trials;
};

struct TvgOptions {
    int32_t min_num_inliers;
    double min_E_F_inlier_ratio;
    double max_H_inlier_ratio;
    double watermark_min_inlier_ratio;
    double watermark_border_size;
    int32_t detect_watermark;
    int32_t multiple_ignore_watermark;
    int32_t force_H_use;
    int32_t compute_relative_pose;
    int32_t multiple_models;
    RansacOptions ransac;
};

struct Camera {
    int32_t model_id;  // COLMAP camera model id, 0..10 (kModels below)
    int32_t has_prior_focal_length;
    uint64_t width, height;
    double params[12];
};

enum Config { UNDEFINED = 0, DEGENERATE = 1, CALIBRATED = 2, UNCALIBRATED = 3, PLANAR = 4,
              PANORAMIC = 5, PLANAR_OR_PANORAMIC = 6, WATERMARK = 7, MULTIPLE = 8 };

struct Pt { double x, y; };
struct Mat3 { double m[9]; };  // row-major

struct Support {
    size_t num_inliers = 0;
    double residual_sum = std::numeric_limits<double>::max();
};

struct Report {
    bool success = false;
    size_t num_trials = 0;
    Support support;
    std::vector<char> inlier_mask;
    Mat3 model{};
};

// ----------------------------------------------------------------------------------------------
// D3: deterministic 64-way sum.  partial[k & 63] accumulates term k in increasing k; then a
// butterfly (xor 32,16,8,4,2,1).  A 64-lane wave computes exactly this.
// ----------------------------------------------------------------------------------------------
template <typename F>
double det_sum64(size_t n, F term) {
#ifdef ORACLE_SEQ_SUMS
    double acc = 0.0;  // D3 off: COLMAP's plain loop
    for (size_t k = 0; k < n; ++k) acc += term(k);
    return acc;
#endif
    double p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0;
    for (size_t k = 0; k < n; ++k) p[k & 63] += term(k);
    for (int m = 32; m >= 1; m >>= 1) {
        double q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ m];
        for (int l = 0; l < 64; ++l) p[l] = q[l];
    }
    return p[0];
}

// ----------------------------------------------------------------------------------------------
// PRNG + sampler: colmap/math/random.{h,cc}, colmap/optim/random_sampler.cc
// ----------------------------------------------------------------------------------------------
struct Prng {
    std::mt19937 gen;
    explicit Prng(uint32_t seed) : gen(seed) {}
    // RandomUniformInteger<uint32_t>(lo, hi): a fresh distribution object per call
    uint32_t uniform(uint32_t lo, uint32_t hi) {
        std::uniform_int_distribution<uint32_t> d(lo, hi);
        return d(gen);
    }
};

struct Sampler {
    size_t k;
    std::vector<size_t> idx;
    explicit Sampler(size_t num_samples) : k(num_samples) {}
    void initialize(size_t total) {
        idx.resize(total);
        for (size_t i = 0; i < total; ++i) idx[i] = i;
    }
    // Shuffle(k, &idx): partial Fisher-Yates; the permutation persists across trials
    void sample(Prng& prng, size_t* out) {
        const uint32_t last = static_cast<uint32_t>(idx.size() - 1);
        for (uint32_t i = 0; i < static_cast<uint32_t>(k); ++i) {
            const uint32_t j = prng.uniform(i, last);
            std::swap(idx[i], idx[j]);
        }
        for (size_t i = 0; i < k; ++i) out[i] = idx[i];
    }
};

// ----------------------------------------------------------------------------------------------
// RANSAC bookkeeping: colmap/optim/ransac.h
// ----------------------------------------------------------------------------------------------
size_t compute_num_trials(size_t num_inliers, size_t num_samples, double confidence,
                          double multiplier, int min_num_samples) {
    const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
    const double nom = 1 - confidence;
    if (nom <= 0) return std::numeric_limits<size_t>::max();
    const double denom = 1 - std::pow(inlier_ratio, min_num_samples);
    if (denom <= 0) return 1;
    if (denom == 1.0) return std::numeric_limits<size_t>::max();
    return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom) * multiplier));
}

// RANSAC constructor: clamp max_num_trials by the trials needed at min_inlier_ratio
size_t ransac_max_trials(const RansacOptions& o, int min_num_samples) {
    const size_t kNumSamples = 100000;
    const size_t dyn = compute_num_trials(static_cast<size_t>(o.min_inlier_ratio * kNumSamples),
                                          kNumSamples, o.confidence, o.dyn_num_trials_multiplier,
                                          min_num_samples);
    return std::min<size_t>(static_cast<size_t>(o.max_num_trials), dyn);
}

// InlierSupportMeasurer::Evaluate (sum in det_sum64 order, D3)
Support evaluate_support(const std::vector<double>& r, double max_residual) {
    Support s;
    s.num_inliers = 0;
    for (double v : r)
        if (v <= max_residual) s.num_inliers += 1;
    s.residual_sum = det_sum64(r.size(), [&](size_t k) { return r[k] <= max_residual ? r[k] : 0.0; });
    return s;
}
bool better(const Support& a, const Support& b) {
    if (a.num_inliers > b.num_inliers) return true;
    return a.num_inliers == b.num_inliers && a.residual_sum < b.residual_sum;
}

// ----------------------------------------------------------------------------------------------
// small dense linear algebra (D1, D2)
// ----------------------------------------------------------------------------------------------
Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
    Mat3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c.m[3 * i + j] = a.m[3 * i + 0] * b.m[0 + j] + a.m[3 * i + 1] * b.m[3 + j] +
                             a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
Mat3 mat3_t(const Mat3& a) {
    Mat3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * j + i];
    return c;
}

// Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9) with the round-robin
// ("tournament") ordering: a sweep is m rounds (m = n for odd n, n - 1 for even n); round r
// rotates the disjoint index pairs {(r+k) mod m, (r-k) mod m}, k = 1..(m-1)/2 (plus {r, n-1} for
// even n).  All rotations of a round are computed from the matrix as it stands at the start of
// the round, then applied together: A <- A J (columns), then A <- J^T A (rows), V <- V J.  Disjoint
// pairs touch disjoint columns in the first step and disjoint rows in the second, so the result
// does not depend on the order in which the pairs of a round are processed - which is what lets
// a GPU wave apply them concurrently and still match this bit for bit.
// a is destroyed (diagonal = eigenvalues); v = eigenvectors in columns (row-major n x n).
int jacobi_round_pairs(int n, int r, int (*pq)[2]) {
    const int m = (n & 1) ? n : n - 1;
    int cnt = 0;
    for (int k = 1; k <= (m - 1) / 2; ++k) {
        const int x = (r + k) % m, y = (r - k + m) % m;
        pq[cnt][0] = x < y ? x : y;
        pq[cnt][1] = x < y ? y : x;
        ++cnt;
    }
    if (!(n & 1)) {
        pq[cnt][0] = r;
        pq[cnt][1] = n - 1;
        ++cnt;
    }
    return cnt;
}
void jacobi_eigen(int n, double* a, double* v) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    double total = 0.0;
    for (int i = 0; i < n * n; ++i) total += a[i] * a[i];
    const double tol = total * 1e-32;
    const int rounds = (n & 1) ? n : n - 1;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (!(off > tol)) break;
        for (int r = 0; r < rounds; ++r) {
            int pq[5][2];
            const int np = jacobi_round_pairs(n, r, pq);
            double cs[5][2];
            bool act[5];
            for (int e = 0; e < np; ++e) {
                const int p = pq[e][0], q = pq[e][1];
                const double apq = a[p * n + q];
                act[e] = apq != 0.0;
                if (!act[e]) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                cs[e][0] = 1.0 / std::sqrt(t * t + 1.0);
                cs[e][1] = t * cs[e][0];
            }
            for (int e = 0; e < np; ++e) {  // columns of A and of V
                if (!act[e]) continue;
                const int p = pq[e][0], q = pq[e][1];
                const double c = cs[e][0], sn = cs[e][1];
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - sn * akq;
                    a[k * n + q] = sn * akp + c * akq;
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - sn * vkq;
                    v[k * n + q] = sn * vkp + c * vkq;
                }
            }
            for (int e = 0; e < np; ++e) {  // rows of A
                if (!act[e]) continue;
                const int p = pq[e][0], q = pq[e][1];
                const double c = cs[e][0], sn = cs[e][1];
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - sn * aqk;
                    a[q * n + k] = sn * apk + c * aqk;
                }
            }
        }
    }
}

// eigenvector of the smallest eigenvalue of the symmetric 9x9 matrix ata (destroyed)
void smallest_eigvec9(double* ata, double* x) {
    double v[81];
    jacobi_eigen(9, ata, v);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (ata[i * 9 + i] < ata[best * 9 + best]) best = i;
    for (int i = 0; i < 9; ++i) x[i] = v[i * 9 + best];
}

// null space of an R x 9 matrix (R <= 8) by Gauss-Jordan elimination with full pivoting.
// Writes 9-R basis vectors into ns (row k = k-th basis vector).  a is destroyed.
void nullspace9(int R, double* a /* R x 9 */, double* ns /* (9-R) x 9 */) {
#ifdef ORACLE_LAPACK_SVD
    {   // D1 off: the right singular vectors of the 9 - R smallest singular values, V.col(R + k) -> ns row k
        double S[9], Vt[81];
        lapack_svd(R, 9, a, S, nullptr, Vt);
        for (int k = 0; k < 9 - R; ++k)
            for (int j = 0; j < 9; ++j) ns[k * 9 + j] = Vt[(R + k) * 9 + j];
        return;
    }
#endif
    int perm[9];
    for (int j = 0; j < 9; ++j) perm[j] = j;
    for (int r = 0; r < R; ++r) {
        int pi = r, pj = r;
        double pv = -1.0;
        for (int i = r; i < R; ++i)
            for (int j = r; j < 9; ++j) {
                const double v = std::fabs(a[i * 9 + j]);
                if (v > pv) { pv = v; pi = i; pj = j; }
            }
        if (pi != r)
            for (int j = 0; j < 9; ++j) std::swap(a[r * 9 + j], a[pi * 9 + j]);
        if (pj != r) {
            for (int i = 0; i < R; ++i) std::swap(a[i * 9 + r], a[i * 9 + pj]);
            std::swap(perm[r], perm[pj]);
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

double poly_eval(const double* c, int deg, double x) {  // c[0] + c[1] x + ...
    double v = c[deg];
    for (int i = deg - 1; i >= 0; --i) v = v * x + c[i];
    return v;
}

// The root inside a sign-change bracket [lo, hi] (flo = p(lo), non-zero; p(hi) has the other sign): bisection until
// the bracket is narrower than 2^-26 of where it sits (or cannot be split any more, or 200 steps), then three Newton
// steps from its midpoint, each accepted only if it lands strictly inside the bracket.  (Rounds 1-2 bisected down to
// adjacent doubles, ~75 polynomial evaluations per root; the bracket pins the root and the monotone stretch, Newton
// finishes in the 3 evaluations of p and p' it is good at.  dc = the coefficients of p', degree deg - 1.)
constexpr double kRootRelWidth = 1.4901161193847656e-08;  // 2^-26
#ifdef ORACLE_ROOT_STATS
long g_root_iter_hist[201];
#endif
double bracket_root(const double* c, const double* dc, int deg, double lo, double hi, double flo) {
#ifdef ORACLE_ROOT_STATS
    struct Tally { int n = 0; ~Tally() { ++g_root_iter_hist[n]; } } tally;
#define ORACLE_ROOT_TICK ++tally.n
#else
#define ORACLE_ROOT_TICK (void)0
#endif
    for (int it = 0; it < 200; ++it) {
        ORACLE_ROOT_TICK;
        const double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        const double fm = poly_eval(c, deg, mid);
        if (fm == 0.0) return mid;
        if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else { hi = mid; }
        if (hi - lo <= kRootRelWidth * (std::fabs(lo) + std::fabs(hi))) break;
    }
    double r = 0.5 * (lo + hi);
    for (int n = 0; n < 3; ++n) {
        const double f = poly_eval(c, deg, r), d = poly_eval(dc, deg - 1, r);
        const double rn = r - f / d;
        if (rn > lo && rn < hi) r = rn;  // (a NaN or infinite step fails both comparisons)
    }
    return r;
}

// real roots of one polynomial whose derivative's real roots (`crit`, ascending) are known:
// between consecutive critical points the polynomial is monotone, so each sign change brackets
// exactly one root (bracket_root).  dc: the derivative's coefficients.
int roots_between(const double* c, const double* dc, int deg, const double* crit, int nc, double* roots) {
    if (deg == 1) {
        roots[0] = -c[0] / c[1];
        return 1;
    }
    double bound = 0.0;  // Cauchy bound on the root magnitudes
    for (int i = 0; i < deg; ++i) bound = std::max(bound, std::fabs(c[i] / c[deg]));
    bound = 1.0 + bound;
    double edges[12];
    int ne = 0;
    edges[ne++] = -bound;
    for (int i = 0; i < nc; ++i)
        if (crit[i] > -bound && crit[i] < bound) edges[ne++] = crit[i];
    edges[ne++] = bound;
    int nr = 0;
    for (int i = 0; i + 1 < ne; ++i) {
        const double lo = edges[i], hi = edges[i + 1];
        const double flo = poly_eval(c, deg, lo);
        const double fhi = poly_eval(c, deg, hi);
        if (flo == 0.0) {
            if (nr == 0 || roots[nr - 1] != lo) roots[nr++] = lo;
            continue;
        }
        if (fhi == 0.0) continue;  // picked up as lo of the next interval (or below)
        if ((flo < 0.0) == (fhi < 0.0)) continue;
        roots[nr++] = bracket_root(c, dc, deg, lo, hi, flo);
    }
    if (poly_eval(c, deg, edges[ne - 1]) == 0.0 && (nr == 0 || roots[nr - 1] != edges[ne - 1]))
        roots[nr++] = edges[ne - 1];
    return nr;
}

// all real roots of a polynomial of degree <= 10, ascending (D2): bottom-up over the chain of
// derivatives (degree 1 first), each level bracketed by the roots of the level below.
int real_roots(const double* c_in, int deg_in, double* roots) {
#ifdef ORACLE_COMPANION_ROOTS
    {   // D2 off: FindPolynomialRootsCompanionMatrix (colmap/math/polynomial.cc); c_in is low -> high here
        int hi = deg_in;
        while (hi > 0 && c_in[hi] == 0.0) --hi;  // RemoveLeadingZeros
        const int degree = hi;
        if (degree <= 0) return 0;
        if (degree == 1) {  // FindLinearPolynomialRoots
            roots[0] = -c_in[0] / c_in[1];
            return 1;
        }
        if (degree == 2) {  // FindQuadraticPolynomialRoots
            const double a = c_in[2], b = c_in[1], c = c_in[0];
            const double b2 = b * b, d = b2 - 4 * a * c;
            if (d >= 0) {
                const double sqrt_d = std::sqrt(d);
                const double a2 = 2 * a;
                if (b >= 0) { roots[0] = (-b - sqrt_d) / a2; roots[1] = (2 * c) / (-b - sqrt_d); }
                else { roots[0] = (2 * c) / (-b + sqrt_d); roots[1] = (-b + sqrt_d) / a2; }
                return 2;
            }
            return 0;  // a complex pair
        }
        int lo = 0;
        while (lo < hi && c_in[lo] == 0.0) ++lo;  // RemoveTrailingZeros: zero is a root
        const int n = hi - lo;                     // size of the companion matrix
        int nr = 0;
        if (n >= 1) {
            std::vector<double> C(static_cast<size_t>(n) * n, 0.0), wr(n), wi(n), dummy(1);
            for (int i = 1; i < n; ++i) C[static_cast<size_t>(i) * n + (i - 1)] = 1.0;
            for (int j = 0; j < n; ++j) C[j] = -c_in[hi - 1 - j] / c_in[hi];  // row 0 = -coeffs.tail / coeffs(0)
            const int info = lapack().dgeev(101, 'N', 'N', n, C.data(), n, wr.data(), wi.data(), dummy.data(), 1, dummy.data(), 1);
            if (info != 0) return 0;
            for (int i = 0; i < n; ++i)
                if (std::fabs(wi[i]) <= 1e-10) roots[nr++] = wr[i];  // kMaxRootImag at the call sites
        }
        if (lo > 0) roots[nr++] = 0.0;
        return nr;
    }
#endif
    int deg = deg_in;
    while (deg > 0 && c_in[deg] == 0.0) --deg;
    if (deg == 0) return 0;
    double chain[11][11];  // chain[j] = j-th derivative, degree deg - j
    for (int i = 0; i <= deg; ++i) chain[0][i] = c_in[i];
    for (int j = 1; j < deg; ++j)
        for (int i = 1; i <= deg - j + 1; ++i) chain[j][i - 1] = chain[j - 1][i] * i;
    double crit[10], cur[10];
    int nc = 0;
    for (int j = deg - 1; j >= 0; --j) {
        const int n = roots_between(chain[j], j + 1 < deg ? chain[j + 1] : nullptr, deg - j, crit, nc, cur);
        nc = n;
        for (int i = 0; i < n; ++i) crit[i] = cur[i];
    }
    for (int i = 0; i < nc; ++i) roots[i] = crit[i];
    return nc;
}

// ----------------------------------------------------------------------------------------------
// estimators/utils.cc
// ----------------------------------------------------------------------------------------------
void center_and_normalize(const std::vector<Pt>& pts, std::vector<Pt>* normed, Mat3* T) {
    const size_t n = pts.size();
    const double cx = det_sum64(n, [&](size_t k) { return pts[k].x; }) / n;
    const double cy = det_sum64(n, [&](size_t k) { return pts[k].y; }) / n;
    double rms = det_sum64(n, [&](size_t k) {
        const double dx = pts[k].x - cx, dy = pts[k].y - cy;
        return dx * dx + dy * dy;
    });
    rms = std::sqrt(rms / n);
    const double nf = std::sqrt(2.0) / rms;
    T->m[0] = nf; T->m[1] = 0; T->m[2] = -nf * cx;
    T->m[3] = 0; T->m[4] = nf; T->m[5] = -nf * cy;
    T->m[6] = 0; T->m[7] = 0; T->m[8] = 1;
    normed->resize(n);
    const double* M = T->m;
    for (size_t i = 0; i < n; ++i) {
        const double p0 = pts[i].x, p1 = pts[i].y;
        const double np0 = M[0] * p0 + M[1] * p1 + M[2];
        const double np1 = M[3] * p0 + M[4] * p1 + M[5];
        const double np2 = M[6] * p0 + M[7] * p1 + M[8];
        const double inv = 1.0 / np2;
        (*normed)[i].x = np0 * inv;
        (*normed)[i].y = np1 * inv;
    }
}

// ComputeSquaredSampsonError, exact operation order of SURVEY.md A.3
double sampson(const Mat3& E, const Pt& p1, const Pt& p2) {
    const double* e = E.m;
    const double x1_0 = p1.x, x1_1 = p1.y, x2_0 = p2.x, x2_1 = p2.y;
    const double Ex1_0 = e[0] * x1_0 + e[1] * x1_1 + e[2];
    const double Ex1_1 = e[3] * x1_0 + e[4] * x1_1 + e[5];
    const double Ex1_2 = e[6] * x1_0 + e[7] * x1_1 + e[8];
    const double Etx2_0 = e[0] * x2_0 + e[3] * x2_1 + e[6];
    const double Etx2_1 = e[1] * x2_0 + e[4] * x2_1 + e[7];
    const double x2tEx1 = x2_0 * Ex1_0 + x2_1 * Ex1_1 + Ex1_2;
    return x2tEx1 * x2tEx1 /
           (Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1 + Etx2_0 * Etx2_0 + Etx2_1 * Etx2_1);
}

// HomographyMatrixEstimator::Residuals (forward transfer error in image 2)
double h_residual(const Mat3& Hm, const Pt& p1, const Pt& p2) {
    const double* H = Hm.m;
    const double s_0 = p1.x, s_1 = p1.y, d_0 = p2.x, d_1 = p2.y;
    const double pd_0 = H[0] * s_0 + H[1] * s_1 + H[2];
    const double pd_1 = H[3] * s_0 + H[4] * s_1 + H[5];
    const double pd_2 = H[6] * s_0 + H[7] * s_1 + H[8];
    const double inv_pd_2 = 1.0 / pd_2;
    const double dd_0 = d_0 - pd_0 * inv_pd_2;
    const double dd_1 = d_1 - pd_1 * inv_pd_2;
    return dd_0 * dd_0 + dd_1 * dd_1;
}

// ----------------------------------------------------------------------------------------------
// estimators
// ----------------------------------------------------------------------------------------------
enum EstKind { EST_F7 = 0, EST_F8 = 1, EST_H = 2, EST_T = 3, EST_E5 = 4 };
int est_min_samples(EstKind k) {
    switch (k) {
        case EST_F7: return 7;
        case EST_F8: return 8;
        case EST_H: return 4;
        case EST_T: return 1;
        case EST_E5: return 5;
    }
    return 0;
}

// FundamentalMatrixSevenPointEstimator::Estimate
std::vector<Mat3> estimate_f7(const std::vector<Pt>& p1, const std::vector<Pt>& p2) {
    double A[7 * 9];
    for (int i = 0; i < 7; ++i) {
        const double x0 = p1[i].x, y0 = p1[i].y, x1 = p2[i].x, y1 = p2[i].y;
        double* r = A + i * 9;
        r[0] = x1 * x0; r[1] = x1 * y0; r[2] = x1;
        r[3] = y1 * x0; r[4] = y1 * y0; r[5] = y1;
        r[6] = x0; r[7] = y0; r[8] = 1;
    }
    double ns[2 * 9];
    nullspace9(7, A, ns);
    double f1[9], f2[9];
    for (int i = 0; i < 9; ++i) { f2[i] = ns[9 + i]; f1[i] = ns[i] - f2[i]; }
    // det(lambda * f1 + f2) = c3 l^3 + c2 l^2 + c1 l + c0, entries e_ij = f1_ij l + f2_ij
    auto mul11 = [](const double* a, const double* b, double* o) {  // (a0 + a1 l)(b0 + b1 l)
        o[0] = a[0] * b[0];
        o[1] = a[0] * b[1] + a[1] * b[0];
        o[2] = a[1] * b[1];
    };
    auto minor2 = [&](int i, int j, int k, int l, double* o) {  // e_i e_j - e_k e_l (degree 2)
        const double a[2] = {f2[i], f1[i]}, b[2] = {f2[j], f1[j]};
        const double c[2] = {f2[k], f1[k]}, d[2] = {f2[l], f1[l]};
        double u[3], w[3];
        mul11(a, b, u);
        mul11(c, d, w);
        for (int t = 0; t < 3; ++t) o[t] = u[t] - w[t];
    };
    double m0[3], m1[3], m2[3];
    minor2(4, 8, 5, 7, m0);
    minor2(3, 8, 5, 6, m1);
    minor2(3, 7, 4, 6, m2);
    double c[4] = {0, 0, 0, 0};
    auto acc = [&](int idx, const double* m, double sign) {  // c += sign * e_idx * m
        const double e0 = f2[idx], e1 = f1[idx];
        c[0] += sign * (e0 * m[0]);
        c[1] += sign * (e0 * m[1] + e1 * m[0]);
        c[2] += sign * (e0 * m[2] + e1 * m[1]);
        c[3] += sign * (e1 * m[2]);
    };
    acc(0, m0, 1.0);
    acc(1, m1, -1.0);
    acc(2, m2, 1.0);
    double roots[3];
    const int nr = real_roots(c, 3, roots);
    std::vector<Mat3> models;
    for (int i = 0; i < nr; ++i) {
        const double lambda = roots[i];
        Mat3 F;
        for (int k = 0; k < 9; ++k) F.m[k] = lambda * f1[k] + f2[k];
        const double kEps = 1e-10;
        if (std::fabs(F.m[8]) < kEps) continue;
        const double inv = F.m[8];
        for (int k = 0; k < 9; ++k) F.m[k] = F.m[k] / inv;
        models.push_back(F);
    }
    return models;
}

// A^T A of the K x 9 design matrix whose k-th row is row(k, out9), summed in det_sum64 order
template <typename RowFn>
void accumulate_ata(size_t K, RowFn row, double* ata /* 81 */) {
    std::vector<double> rows(K * 9);
    for (size_t k = 0; k < K; ++k) row(k, rows.data() + k * 9);
    for (int i = 0; i < 9; ++i)
        for (int j = i; j < 9; ++j) {
            const double s = det_sum64(K, [&](size_t k) { return rows[k * 9 + i] * rows[k * 9 + j]; });
            ata[i * 9 + j] = s;
            ata[j * 9 + i] = s;
        }
}

#ifdef ORACLE_LAPACK_SVD
// D1 off: the `count` right singular vectors of the smallest singular values of the K x 9 design matrix,
// out row k = V.col(9 - count + k) (Eigen's / LAPACK's descending order)
template <typename RowFn>
void lapack_null_vectors(size_t K, RowFn row, int count, double* out) {
    std::vector<double> A(std::max<size_t>(K, 9) * 9, 0.0);  // padded with zero rows up to 9 x 9: full V either way
    for (size_t k = 0; k < K; ++k) row(k, A.data() + k * 9);
    const int m = static_cast<int>(std::max<size_t>(K, 9));
    std::vector<double> S(9);
    double Vt[81];
    lapack_svd(m, 9, A.data(), S.data(), nullptr, Vt);
    for (int k = 0; k < count; ++k)
        for (int j = 0; j < 9; ++j) out[k * 9 + j] = Vt[(9 - count + k) * 9 + j];
}
#endif

// FundamentalMatrixEightPointEstimator::Estimate
std::vector<Mat3> estimate_f8(const std::vector<Pt>& p1, const std::vector<Pt>& p2) {
    std::vector<Pt> n1, n2;
    Mat3 T1, T2;
    center_and_normalize(p1, &n1, &T1);
    center_and_normalize(p2, &n2, &T2);
    auto f8_row = [&](size_t k, double* r) {
        r[0] = n1[k].x * n2[k].x; r[1] = n1[k].y * n2[k].x; r[2] = n2[k].x;
        r[3] = n1[k].x * n2[k].y; r[4] = n1[k].y * n2[k].y; r[5] = n2[k].y;
        r[6] = n1[k].x; r[7] = n1[k].y; r[8] = 1.0;
    };
    double f[9];
#ifdef ORACLE_LAPACK_SVD
    lapack_null_vectors(p1.size(), f8_row, 1, f);
#else
    double ata[81];
    accumulate_ata(p1.size(), f8_row, ata);
    smallest_eigvec9(ata, f);
#endif
    Mat3 Fh;
    for (int k = 0; k < 9; ++k) Fh.m[k] = f[k];
#ifdef ORACLE_LAPACK_SVD
    {   // rank 2 as upstream: F = U diag(s0, s1, 0) V^T of the 3 x 3 SVD
        double S3[3], U3[9], Vt3[9];
        lapack_svd(3, 3, Fh.m, S3, U3, Vt3);
        Mat3 Fr;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Fr.m[3 * i + j] = U3[3 * i] * S3[0] * Vt3[j] + U3[3 * i + 1] * S3[1] * Vt3[3 + j];
        return {mat3_mul(mat3_mul(mat3_t(T2), Fr), T1)};
    }
#endif
    // rank 2: remove the component along the smallest right singular vector v3 of Fh
    // (= smallest eigenvector of Fh^T Fh):  F' = Fh - (Fh v3) v3^T
    double ftf[9], v[9];
    const Mat3 FtF = mat3_mul(mat3_t(Fh), Fh);
    for (int k = 0; k < 9; ++k) ftf[k] = FtF.m[k];
    jacobi_eigen(3, ftf, v);
    int b = 0;
    for (int i = 1; i < 3; ++i)
        if (ftf[i * 3 + i] < ftf[b * 3 + b]) b = i;
    const double v3[3] = {v[0 * 3 + b], v[1 * 3 + b], v[2 * 3 + b]};
    Mat3 Fr;
    for (int i = 0; i < 3; ++i) {
        const double fv = Fh.m[3 * i] * v3[0] + Fh.m[3 * i + 1] * v3[1] + Fh.m[3 * i + 2] * v3[2];
        for (int j = 0; j < 3; ++j) Fr.m[3 * i + j] = Fh.m[3 * i + j] - fv * v3[j];
    }
    return {mat3_mul(mat3_mul(mat3_t(T2), Fr), T1)};
}

// The homography through exactly four correspondences, in closed form (part of D1: COLMAP takes the
// null vector of the 8 x 9 DLT matrix from a Jacobi SVD; the matrix has an exact one-dimensional null
// space for four points in general position, and this is that vector written out).  With
// S = [s0 s1 s2] and D = [d0 d1 d2] (homogeneous columns), adj(S) has rows s1 x s2, s2 x s0, s0 x s1 and
//   Hhat = D * diag(mu_k / lam_k) * adj(S),   lam = adj(S) s3,   mu = adj(D) d3
// maps s_k -> d_k for k = 0..3 (a change of projective basis).  Scaled to unit Frobenius norm like the
// singular vector it replaces.  Three collinear (or repeated) points give 0/0 -> NaN -> no inliers.
void h4_closed_form(const Pt* s, const Pt* d, double* h) {
    double a[3][3], b[3][3], c[3][3];
    for (int k = 0; k < 3; ++k) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        a[k][0] = s[p].y - s[q].y; a[k][1] = s[q].x - s[p].x; a[k][2] = s[p].x * s[q].y - s[p].y * s[q].x;
        b[k][0] = d[p].y - d[q].y; b[k][1] = d[q].x - d[p].x; b[k][2] = d[p].x * d[q].y - d[p].y * d[q].x;
    }
    for (int k = 0; k < 3; ++k) {
        const double lam = (a[k][0] * s[3].x + a[k][1] * s[3].y) + a[k][2];
        const double mu = (b[k][0] * d[3].x + b[k][1] * d[3].y) + b[k][2];
        const double r = mu / lam;
        c[k][0] = r * a[k][0]; c[k][1] = r * a[k][1]; c[k][2] = r * a[k][2];
    }
    for (int j = 0; j < 3; ++j) {
        h[j] = (d[0].x * c[0][j] + d[1].x * c[1][j]) + d[2].x * c[2][j];
        h[3 + j] = (d[0].y * c[0][j] + d[1].y * c[1][j]) + d[2].y * c[2][j];
        h[6 + j] = (c[0][j] + c[1][j]) + c[2][j];
    }
    double n2 = 0.0;
    for (int j = 0; j < 9; ++j) n2 = n2 + h[j] * h[j];
    const double inv = 1.0 / std::sqrt(n2);
    for (int j = 0; j < 9; ++j) h[j] = h[j] * inv;
}

// HomographyMatrixEstimator::Estimate (normalised DLT); minimal case (N == 4) in closed form,
// over-determined through A^T A (D1)
std::vector<Mat3> estimate_h(const std::vector<Pt>& p1, const std::vector<Pt>& p2) {
    const size_t N = p1.size();
    std::vector<Pt> n1, n2;
    Mat3 T1, T2;
    center_and_normalize(p1, &n1, &T1);
    center_and_normalize(p2, &n2, &T2);
    auto row_a = [&](size_t i, double* r) {
        const double s_0 = n1[i].x, s_1 = n1[i].y, d_0 = n2[i].x;
        r[0] = -s_0; r[1] = -s_1; r[2] = -1; r[3] = 0; r[4] = 0; r[5] = 0;
        r[6] = s_0 * d_0; r[7] = s_1 * d_0; r[8] = d_0;
    };
    auto row_b = [&](size_t i, double* r) {
        const double s_0 = n1[i].x, s_1 = n1[i].y, d_1 = n2[i].y;
        r[0] = 0; r[1] = 0; r[2] = 0; r[3] = -s_0; r[4] = -s_1; r[5] = -1;
        r[6] = s_0 * d_1; r[7] = s_1 * d_1; r[8] = d_1;
    };
    double h[9];
    if (N == 4) {
#ifdef ORACLE_LAPACK_SVD
        double A[8 * 9];
        for (size_t i = 0; i < 4; ++i) { row_a(i, A + i * 9); row_b(i, A + (4 + i) * 9); }
        nullspace9(8, A, h);
#else
        h4_closed_form(n1.data(), n2.data(), h);
#endif
    } else {
        // rows 0..N-1 are the "a" rows, N..2N-1 the "b" rows (COLMAP's i / j = N + i layout)
        auto h_row = [&](size_t k, double* r) { if (k < N) row_a(k, r); else row_b(k - N, r); };
#ifdef ORACLE_LAPACK_SVD
        lapack_null_vectors(2 * N, h_row, 1, h);
#else
        double ata[81];
        accumulate_ata(2 * N, h_row, ata);
        smallest_eigvec9(ata, h);
#endif
    }
    Mat3 Hh;
    for (int k = 0; k < 9; ++k) Hh.m[k] = h[k];
    // T2^-1 in closed form for the similarity [nf 0 -nf cx; 0 nf -nf cy; 0 0 1]
    Mat3 T2i;
    const double inv_nf = 1.0 / T2.m[0];
    T2i.m[0] = inv_nf; T2i.m[1] = 0; T2i.m[2] = -T2.m[2] * inv_nf;
    T2i.m[3] = 0; T2i.m[4] = inv_nf; T2i.m[5] = -T2.m[5] * inv_nf;
    T2i.m[6] = 0; T2i.m[7] = 0; T2i.m[8] = 1;
    return {mat3_mul(mat3_mul(T2i, Hh), T1)};
}

// TranslationTransformEstimator<2>::Estimate: model = mean(dst) - mean(src), stored in m[0..1]
std::vector<Mat3> estimate_t(const std::vector<Pt>& p1, const std::vector<Pt>& p2) {
    const size_t n = p1.size();
    const double sx = det_sum64(n, [&](size_t k) { return p1[k].x; }) / n;
    const double sy = det_sum64(n, [&](size_t k) { return p1[k].y; }) / n;
    const double dx = det_sum64(n, [&](size_t k) { return p2[k].x; }) / n;
    const double dy = det_sum64(n, [&](size_t k) { return p2[k].y; }) / n;
    Mat3 t{};
    t.m[0] = dx - sx;
    t.m[1] = dy - sy;
    return {t};
}
double t_residual(const Mat3& t, const Pt& p1, const Pt& p2) {
    const double d0 = p2.x - p1.x - t.m[0];
    const double d1 = p2.y - p1.y - t.m[1];
    return d0 * d0 + d1 * d1;
}

std::vector<Mat3> estimate_e5(const std::vector<Pt>& p1, const std::vector<Pt>& p2);  // below

std::vector<Mat3> estimate(EstKind k, const std::vector<Pt>& a, const std::vector<Pt>& b) {
    switch (k) {
        case EST_F7: return estimate_f7(a, b);
        case EST_F8: return estimate_f8(a, b);
        case EST_H: return estimate_h(a, b);
        case EST_T: return estimate_t(a, b);
        case EST_E5: return estimate_e5(a, b);
    }
    return {};
}
void residuals(EstKind k, const std::vector<Pt>& X, const std::vector<Pt>& Y, const Mat3& M,
               std::vector<double>* r) {
    r->resize(X.size());
    for (size_t i = 0; i < X.size(); ++i) {
        switch (k) {
            case EST_F7: case EST_F8: case EST_E5: (*r)[i] = sampson(M, X[i], Y[i]); break;
            case EST_H: (*r)[i] = h_residual(M, X[i], Y[i]); break;
            case EST_T: (*r)[i] = t_residual(M, X[i], Y[i]); break;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// LORANSAC<Estimator, LocalEstimator>::Estimate (colmap/optim/loransac.h), literal control flow
// ----------------------------------------------------------------------------------------------
Report lo_ransac(EstKind est, EstKind local_est, const RansacOptions& opt_in, Prng& prng,
                 const std::vector<Pt>& X, const std::vector<Pt>& Y) {
    const int kMin = est_min_samples(est), kLocalMin = est_min_samples(local_est);
    RansacOptions options = opt_in;
    options.max_num_trials = static_cast<int64_t>(ransac_max_trials(opt_in, kMin));

    Report report;
    report.success = false;
    report.num_trials = 0;
    const size_t num_samples = X.size();
    if (num_samples < static_cast<size_t>(kMin)) return report;

    Support best_support;
    Mat3 best_model{};
    bool best_model_is_local = false;
    bool abort = false;
    const double max_residual = options.max_error * options.max_error;

    std::vector<double> res, best_local_res;
    std::vector<Pt> X_inlier, Y_inlier, X_rand(kMin), Y_rand(kMin);
    std::vector<size_t> sidx(kMin);

    Sampler sampler(kMin);
    sampler.initialize(num_samples);
    size_t max_num_trials = static_cast<size_t>(options.max_num_trials);
    size_t dyn_max_num_trials = max_num_trials;

    for (report.num_trials = 0; report.num_trials < max_num_trials; ++report.num_trials) {
        if (abort) {
            report.num_trials += 1;
            break;
        }
        sampler.sample(prng, sidx.data());
        for (int i = 0; i < kMin; ++i) { X_rand[i] = X[sidx[i]]; Y_rand[i] = Y[sidx[i]]; }
        const std::vector<Mat3> sample_models = estimate(est, X_rand, Y_rand);
        for (const Mat3& sample_model : sample_models) {
            residuals(est, X, Y, sample_model, &res);
            const Support support = evaluate_support(res, max_residual);
            if (better(support, best_support)) {
                best_support = support;
                best_model = sample_model;
                best_model_is_local = false;
                if (support.num_inliers > static_cast<size_t>(kMin) &&
                    support.num_inliers >= static_cast<size_t>(kLocalMin)) {
                    const size_t kMaxNumLocalTrials = 10;
                    for (size_t lt = 0; lt < kMaxNumLocalTrials; ++lt) {
                        X_inlier.clear();
                        Y_inlier.clear();
                        for (size_t i = 0; i < res.size(); ++i)
                            if (res[i] <= max_residual) { X_inlier.push_back(X[i]); Y_inlier.push_back(Y[i]); }
                        const std::vector<Mat3> local_models = estimate(local_est, X_inlier, Y_inlier);
                        const size_t prev_best_num_inliers = best_support.num_inliers;
                        for (const Mat3& local_model : local_models) {
                            residuals(local_est, X, Y, local_model, &res);
                            const Support local_support = evaluate_support(res, max_residual);
                            if (better(local_support, best_support)) {
                                best_support = local_support;
                                best_model = local_model;
                                best_model_is_local = true;
                                std::swap(res, best_local_res);
                            }
                        }
                        if (best_support.num_inliers <= prev_best_num_inliers) break;
                        std::swap(res, best_local_res);
                    }
                }
                dyn_max_num_trials = compute_num_trials(best_support.num_inliers, num_samples,
                                                        options.confidence,
                                                        options.dyn_num_trials_multiplier, kMin);
            }
            if (report.num_trials >= dyn_max_num_trials &&
                report.num_trials >= static_cast<size_t>(options.min_num_trials)) {
                abort = true;
                break;
            }
        }
    }
    report.support = best_support;
    report.model = best_model;
    if (report.support.num_inliers < static_cast<size_t>(kMin)) return report;
    report.success = true;
    residuals(best_model_is_local ? local_est : est, X, Y, report.model, &res);
    report.inlier_mask.resize(num_samples);
    for (size_t i = 0; i < res.size(); ++i) report.inlier_mask[i] = res[i] <= max_residual;
    return report;
}

// ----------------------------------------------------------------------------------------------
// cameras: colmap/sensor/models.h (3.9.1), the eleven models, as Camera::CamFromImg /
// CamFromImgThreshold / CalibrationMatrix use them.  Called per matched point by
// EstimateCalibratedTwoViewGeometry (reference side: /root/reference/pycolmap/estimators/
// essential_matrix.h:33-46, /root/reference/pycolmap/scene/camera.h:136-165).  Restated from the
// upstream templates with T = double (confidence: high for the pinhole / radial / OpenCV models and
// IterativeUndistortion, medium for the FOV closed form and the thin-prism fisheye tail).
// ----------------------------------------------------------------------------------------------
struct ModelInfo { int num_params; int num_focal; };  // focal idxs 0..num_focal-1, principal point next
const ModelInfo kModels[11] = {
    {3, 1},   // 0 SIMPLE_PINHOLE        f, cx, cy
    {4, 2},   // 1 PINHOLE               fx, fy, cx, cy
    {4, 1},   // 2 SIMPLE_RADIAL         f, cx, cy, k
    {5, 1},   // 3 RADIAL                f, cx, cy, k1, k2
    {8, 2},   // 4 OPENCV                fx, fy, cx, cy, k1, k2, p1, p2
    {8, 2},   // 5 OPENCV_FISHEYE        fx, fy, cx, cy, k1, k2, k3, k4
    {12, 2},  // 6 FULL_OPENCV           fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6
    {5, 2},   // 7 FOV                   fx, fy, cx, cy, omega
    {4, 1},   // 8 SIMPLE_RADIAL_FISHEYE f, cx, cy, k
    {5, 1},   // 9 RADIAL_FISHEYE        f, cx, cy, k1, k2
    {12, 2},  // 10 THIN_PRISM_FISHEYE   fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1
};
bool camera_supported(const Camera& c) { return c.model_id >= 0 && c.model_id <= 10; }

// <Model>::Distortion(extra_params, u, v, &du, &dv)
void model_distortion(int model, const double* extra, double u, double v, double* du, double* dv) {
    const double kEps = std::numeric_limits<double>::epsilon();
    switch (model) {
        case 2: {  // SimpleRadialCameraModel
            const double k = extra[0];
            const double u2 = u * u;
            const double v2 = v * v;
            const double r2 = u2 + v2;
            const double radial = k * r2;
            *du = u * radial;
            *dv = v * radial;
            return;
        }
        case 3: {  // RadialCameraModel
            const double k1 = extra[0];
            const double k2 = extra[1];
            const double u2 = u * u;
            const double v2 = v * v;
            const double r2 = u2 + v2;
            const double radial = k1 * r2 + k2 * r2 * r2;
            *du = u * radial;
            *dv = v * radial;
            return;
        }
        case 4: {  // OpenCVCameraModel
            const double k1 = extra[0];
            const double k2 = extra[1];
            const double p1 = extra[2];
            const double p2 = extra[3];
            const double u2 = u * u;
            const double uv = u * v;
            const double v2 = v * v;
            const double r2 = u2 + v2;
            const double radial = k1 * r2 + k2 * r2 * r2;
            *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
            *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
            return;
        }
        case 6: {  // FullOpenCVCameraModel
            const double k1 = extra[0], k2 = extra[1], p1 = extra[2], p2 = extra[3];
            const double k3 = extra[4], k4 = extra[5], k5 = extra[6], k6 = extra[7];
            const double u2 = u * u;
            const double uv = u * v;
            const double v2 = v * v;
            const double r2 = u2 + v2;
            const double r4 = r2 * r2;
            const double r6 = r4 * r2;
            const double radial = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6);
            *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) - u;
            *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) - v;
            return;
        }
        case 5: case 8: case 9: {  // OpenCVFisheye / SimpleRadialFisheye / RadialFisheye
            const double r = std::sqrt(u * u + v * v);
            if (r > kEps) {
                const double theta = std::atan(r);
                const double theta2 = theta * theta;
                double thetad;
                if (model == 8) {
                    thetad = theta * (1.0 + extra[0] * theta2);
                } else if (model == 9) {
                    const double theta4 = theta2 * theta2;
                    thetad = theta * (1.0 + extra[0] * theta2 + extra[1] * theta4);
                } else {
                    const double theta4 = theta2 * theta2;
                    const double theta6 = theta4 * theta2;
                    const double theta8 = theta4 * theta4;
                    thetad = theta * (1.0 + extra[0] * theta2 + extra[1] * theta4 + extra[2] * theta6 +
                                      extra[3] * theta8);
                }
                *du = u * thetad / r - u;
                *dv = v * thetad / r - v;
            } else {
                *du = 0.0;
                *dv = 0.0;
            }
            return;
        }
        case 10: {  // ThinPrismFisheyeCameraModel
            const double k1 = extra[0], k2 = extra[1], p1 = extra[2], p2 = extra[3];
            const double k3 = extra[4], k4 = extra[5], sx1 = extra[6], sy1 = extra[7];
            const double u2 = u * u;
            const double uv = u * v;
            const double v2 = v * v;
            const double r2 = u2 + v2;
            const double r4 = r2 * r2;
            const double r6 = r4 * r2;
            const double r8 = r6 * r2;
            const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
            *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) + sx1 * r2;
            *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) + sy1 * r2;
            return;
        }
    }
    *du = 0.0;
    *dv = 0.0;
}

// BaseCameraModel<CameraModel>::IterativeUndistortion: Newton iteration with a numerical
// (central difference) Jacobian; Eigen's fixed-size 2 x 2 inverse is adjugate / determinant.
void iterative_undistortion(int model, const double* extra, double* u, double* v) {
    const size_t kNumIterations = 100;
    const double kMaxStepNorm = 1e-10;
    const double kRelStepSize = 1e-6;
    const double x0[2] = {*u, *v};
    double x[2] = {*u, *v};
    for (size_t i = 0; i < kNumIterations; ++i) {
        const double step0 = std::max(std::numeric_limits<double>::epsilon(), std::abs(kRelStepSize * x[0]));
        const double step1 = std::max(std::numeric_limits<double>::epsilon(), std::abs(kRelStepSize * x[1]));
        double dx[2], dx_0b[2], dx_0f[2], dx_1b[2], dx_1f[2];
        model_distortion(model, extra, x[0], x[1], &dx[0], &dx[1]);
        model_distortion(model, extra, x[0] - step0, x[1], &dx_0b[0], &dx_0b[1]);
        model_distortion(model, extra, x[0] + step0, x[1], &dx_0f[0], &dx_0f[1]);
        model_distortion(model, extra, x[0], x[1] - step1, &dx_1b[0], &dx_1b[1]);
        model_distortion(model, extra, x[0], x[1] + step1, &dx_1f[0], &dx_1f[1]);
        double J[2][2];
        J[0][0] = 1 + (dx_0f[0] - dx_0b[0]) / (2 * step0);
        J[0][1] = (dx_1f[0] - dx_1b[0]) / (2 * step1);
        J[1][0] = (dx_0f[1] - dx_0b[1]) / (2 * step0);
        J[1][1] = 1 + (dx_1f[1] - dx_1b[1]) / (2 * step1);
        const double invdet = 1.0 / (J[0][0] * J[1][1] - J[1][0] * J[0][1]);
        const double Ji[2][2] = {{J[1][1] * invdet, -J[0][1] * invdet}, {-J[1][0] * invdet, J[0][0] * invdet}};
        const double rhs[2] = {x[0] + dx[0] - x0[0], x[1] + dx[1] - x0[1]};
        const double step_x[2] = {Ji[0][0] * rhs[0] + Ji[0][1] * rhs[1], Ji[1][0] * rhs[0] + Ji[1][1] * rhs[1]};
        x[0] -= step_x[0];
        x[1] -= step_x[1];
        if (step_x[0] * step_x[0] + step_x[1] * step_x[1] < kMaxStepNorm) break;
    }
    *u = x[0];
    *v = x[1];
}

Pt cam_from_img(const Camera& c, const Pt& p) {
    const ModelInfo& mi = kModels[c.model_id];
    const double f1 = c.params[0], f2 = c.params[mi.num_focal - 1];
    const double c1 = c.params[mi.num_focal], c2 = c.params[mi.num_focal + 1];
    const double* extra = c.params + mi.num_focal + 2;
    double u = (p.x - c1) / f1;
    double v = (p.y - c2) / f2;
    if (c.model_id <= 1) return Pt{u, v};
    if (c.model_id == 7) {  // FOVCameraModel::Undistortion, closed form
        const double omega = extra[0];
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
        return Pt{u * factor, v * factor};
    }
    iterative_undistortion(c.model_id, extra, &u, &v);
    if (c.model_id == 10) {  // thin-prism fisheye: back from the equidistant angle to the plane
        const double theta = std::sqrt(u * u + v * v);
        // upstream: theta * ceres::cos(theta), ceres::sin(theta); GCC at -O2 turns the pair into one sincos
        // call (cexpi) - made explicit so that it does not depend on the optimiser
        double sin_theta, cos_theta;
        ::sincos(theta, &sin_theta, &cos_theta);
        const double theta_cos_theta = theta * cos_theta;
        if (theta_cos_theta > std::numeric_limits<double>::epsilon()) {
            const double scale = sin_theta / theta_cos_theta;
            u *= scale;
            v *= scale;
        }
    }
    return Pt{u, v};
}
// Camera::ImgFromCam (colmap/sensor/models.h): normalised image plane -> pixels.  Every model: distort, then
// x = f1 * (u + du) + c1, y = f2 * (v + dv) + c2; FOV's Distortion returns the distorted point itself; the thin-prism
// fisheye first maps the plane to the equidistant angle (theta = atan r).
Pt img_from_cam(const Camera& c, const Pt& p) {
    const ModelInfo& mi = kModels[c.model_id];
    const double f1 = c.params[0], f2 = c.params[mi.num_focal - 1];
    const double c1 = c.params[mi.num_focal], c2 = c.params[mi.num_focal + 1];
    const double* extra = c.params + mi.num_focal + 2;
    double u = p.x, v = p.y;
    if (c.model_id <= 1) return Pt{f1 * u + c1, f2 * v + c2};
    if (c.model_id == 7) {  // FOVCameraModel::Distortion
        const double omega = extra[0];
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
        return Pt{f1 * (u * factor) + c1, f2 * (v * factor) + c2};
    }
    if (c.model_id == 10) {
        const double r = std::sqrt(u * u + v * v);
        if (r > std::numeric_limits<double>::epsilon()) {
            const double theta = std::atan(r);
            u = theta * u / r;
            v = theta * v / r;
        }
    }
    double du, dv;
    model_distortion(c.model_id, extra, u, v, &du, &dv);
    return Pt{f1 * (u + du) + c1, f2 * (v + dv) + c2};
}
double cam_from_img_threshold(const Camera& c, double threshold) {
    const ModelInfo& mi = kModels[c.model_id];
    double mean_focal_length = 0;
    for (int i = 0; i < mi.num_focal; ++i) mean_focal_length += c.params[i];
    mean_focal_length /= mi.num_focal;
    return threshold / mean_focal_length;
}

// ----------------------------------------------------------------------------------------------
// two_view_geometry.cc
// ----------------------------------------------------------------------------------------------
struct V3 { double v[3]; };
// cam2_from_cam1 of EstimateTwoViewGeometryPose (filled when TwoViewGeometryOptions.compute_relative_pose)
struct RelPose {
    bool ok = false;
    int config = UNDEFINED;       // PLANAR_OR_PANORAMIC resolved
    Mat3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    V3 t{{0, 0, 0}};
    double qvec[4] = {1, 0, 0, 0};
    double tri_angle = 0.0;
    size_t num_points3D = 0;
};
struct Tvg {
    int config = UNDEFINED;
    Mat3 E{}, F{}, H{};
    std::vector<char> inlier_mask;  // over the input matches
    size_t num_inliers = 0;
    size_t trials[4] = {0, 0, 0, 0};   // E, F, H, watermark
    size_t inl[3] = {0, 0, 0};         // E, F, H support
    RelPose pose;
};
RelPose estimate_relative_pose(const Camera& c1, const std::vector<Pt>& pts1, const Camera& c2,
                               const std::vector<Pt>& pts2, const uint32_t* matches, size_t M, const Tvg& g);

bool in_bbox(const Pt& p, double minx, double maxx, double miny, double maxy) {
    return p.x >= minx && p.x <= maxx && p.y >= miny && p.y <= maxy;
}

bool detect_watermark(const Camera& c1, const std::vector<Pt>& p1, const Camera& c2,
                      const std::vector<Pt>& p2, size_t num_inliers,
                      const std::vector<char>& mask, const TvgOptions& o, Prng& prng, size_t* trials) {
    if (!o.detect_watermark) return false;
    const double diagonal1 = std::sqrt(static_cast<double>(c1.width * c1.width + c1.height * c1.height));
    const double diagonal2 = std::sqrt(static_cast<double>(c2.width * c2.width + c2.height * c2.height));
    const double minx1 = o.watermark_border_size * diagonal1, miny1 = minx1;
    const double maxx1 = c1.width - minx1, maxy1 = c1.height - miny1;
    const double minx2 = o.watermark_border_size * diagonal2, miny2 = minx2;
    const double maxx2 = c2.width - minx2, maxy2 = c2.height - miny2;
    std::vector<Pt> ip1(num_inliers), ip2(num_inliers);
    size_t num_in_border = 0, j = 0;
    for (size_t i = 0; i < mask.size(); ++i) {
        if (mask[i]) {
            ip1[j] = p1[i];
            ip2[j] = p2[i];
            j += 1;
            if (!in_bbox(p1[i], minx1, maxx1, miny1, maxy1) && !in_bbox(p2[i], minx2, maxx2, miny2, maxy2))
                num_in_border += 1;
        }
    }
    const double ratio = static_cast<double>(num_in_border) / num_inliers;
    if (ratio < o.watermark_min_inlier_ratio) return false;
    RansacOptions ro = o.ransac;
    ro.min_inlier_ratio = o.watermark_min_inlier_ratio;
    const Report rep = lo_ransac(EST_T, EST_T, ro, prng, ip1, ip2);
    *trials = rep.num_trials;
    const double inlier_ratio = static_cast<double>(rep.support.num_inliers) / num_inliers;
    return inlier_ratio >= o.watermark_min_inlier_ratio;
}

Tvg estimate_two_view_geometry(const Camera& c1, const std::vector<Pt>& pts1, const Camera& c2,
                               const std::vector<Pt>& pts2, const uint32_t* matches, size_t M,
                               const TvgOptions& o, uint32_t seed) {
    Tvg g;
    Prng prng(seed);  // D4: per-pair reseed
    const size_t min_num_inliers = static_cast<size_t>(o.min_num_inliers);
    if (M < min_num_inliers) {
        g.config = DEGENERATE;
        return g;
    }
    std::vector<Pt> mp1(M), mp2(M);
    for (size_t i = 0; i < M; ++i) { mp1[i] = pts1[matches[2 * i]]; mp2[i] = pts2[matches[2 * i + 1]]; }

    const bool calibrated = !o.force_H_use && c1.has_prior_focal_length && c2.has_prior_focal_length;
    Report E_rep, F_rep, H_rep;
    if (calibrated) {
        std::vector<Pt> n1(M), n2(M);
        for (size_t i = 0; i < M; ++i) { n1[i] = cam_from_img(c1, mp1[i]); n2[i] = cam_from_img(c2, mp2[i]); }
        RansacOptions eo = o.ransac;
        eo.max_error = (cam_from_img_threshold(c1, o.ransac.max_error) +
                        cam_from_img_threshold(c2, o.ransac.max_error)) / 2;
        E_rep = lo_ransac(EST_E5, EST_E5, eo, prng, n1, n2);
        g.E = E_rep.model;
        g.trials[0] = E_rep.num_trials;
        g.inl[0] = E_rep.support.num_inliers;
    }
    if (!o.force_H_use) {
        F_rep = lo_ransac(EST_F7, EST_F8, o.ransac, prng, mp1, mp2);
        g.F = F_rep.model;
        g.trials[1] = F_rep.num_trials;
        g.inl[1] = F_rep.support.num_inliers;
    }
    H_rep = lo_ransac(EST_H, EST_H, o.ransac, prng, mp1, mp2);
    g.H = H_rep.model;
    g.trials[2] = H_rep.num_trials;
    g.inl[2] = H_rep.support.num_inliers;

    const std::vector<char>* best_mask = nullptr;
    size_t num_inliers = 0;
    if (o.force_H_use) {
        // EstimateCalibratedHomography
        if (!H_rep.success || H_rep.support.num_inliers < min_num_inliers) { g.config = DEGENERATE; return g; }
        g.config = PLANAR_OR_PANORAMIC;
        best_mask = &H_rep.inlier_mask;
        num_inliers = H_rep.support.num_inliers;
    } else if (calibrated) {
        if ((!E_rep.success && !F_rep.success && !H_rep.success) ||
            (E_rep.support.num_inliers < min_num_inliers && F_rep.support.num_inliers < min_num_inliers &&
             H_rep.support.num_inliers < min_num_inliers)) {
            g.config = DEGENERATE;
            return g;
        }
        const double E_F = static_cast<double>(E_rep.support.num_inliers) / F_rep.support.num_inliers;
        const double H_F = static_cast<double>(H_rep.support.num_inliers) / F_rep.support.num_inliers;
        const double H_E = static_cast<double>(H_rep.support.num_inliers) / E_rep.support.num_inliers;
        if (E_rep.success && E_F > o.min_E_F_inlier_ratio && E_rep.support.num_inliers >= min_num_inliers) {
            if (E_rep.support.num_inliers >= F_rep.support.num_inliers) {
                num_inliers = E_rep.support.num_inliers;
                best_mask = &E_rep.inlier_mask;
            } else {
                num_inliers = F_rep.support.num_inliers;
                best_mask = &F_rep.inlier_mask;
            }
            if (H_E > o.max_H_inlier_ratio) {
                g.config = PLANAR_OR_PANORAMIC;
                if (H_rep.support.num_inliers > num_inliers) {
                    num_inliers = H_rep.support.num_inliers;
                    best_mask = &H_rep.inlier_mask;
                }
            } else {
                g.config = CALIBRATED;
            }
        } else if (F_rep.success && F_rep.support.num_inliers >= min_num_inliers) {
            num_inliers = F_rep.support.num_inliers;
            best_mask = &F_rep.inlier_mask;
            if (H_F > o.max_H_inlier_ratio) {
                g.config = PLANAR_OR_PANORAMIC;
                if (H_rep.support.num_inliers > num_inliers) {
                    num_inliers = H_rep.support.num_inliers;
                    best_mask = &H_rep.inlier_mask;
                }
            } else {
                g.config = UNCALIBRATED;
            }
        } else if (H_rep.success && H_rep.support.num_inliers >= min_num_inliers) {
            num_inliers = H_rep.support.num_inliers;
            best_mask = &H_rep.inlier_mask;
            g.config = PLANAR_OR_PANORAMIC;
        } else {
            g.config = DEGENERATE;
            return g;
        }
    } else {
        if ((!F_rep.success && !H_rep.success) ||
            (F_rep.support.num_inliers < min_num_inliers && H_rep.support.num_inliers < min_num_inliers)) {
            g.config = DEGENERATE;
            return g;
        }
        const double H_F = static_cast<double>(H_rep.support.num_inliers) / F_rep.support.num_inliers;
        best_mask = &F_rep.inlier_mask;
        num_inliers = F_rep.support.num_inliers;
        if (H_F > o.max_H_inlier_ratio) {
            g.config = PLANAR_OR_PANORAMIC;
            if (H_rep.support.num_inliers >= F_rep.support.num_inliers) {
                num_inliers = H_rep.support.num_inliers;
                best_mask = &H_rep.inlier_mask;
            }
        } else {
            g.config = UNCALIBRATED;
        }
    }
    g.num_inliers = num_inliers;
    g.inlier_mask.assign(M, 0);
    if (best_mask != nullptr && !best_mask->empty())
        for (size_t i = 0; i < M; ++i) g.inlier_mask[i] = (*best_mask)[i];
    else
        g.num_inliers = 0;
    if (o.detect_watermark && best_mask != nullptr && !best_mask->empty() &&
        detect_watermark(c1, mp1, c2, mp2, num_inliers, *best_mask, o, prng, &g.trials[3]))
        g.config = WATERMARK;
    if (o.compute_relative_pose) {
        g.pose = estimate_relative_pose(c1, pts1, c2, pts2, matches, M, g);
        if (g.pose.ok) g.config = g.pose.config;
    }
    return g;
}

// ----------------------------------------------------------------------------------------------
// 5-point essential matrix (Nister / Stewenius formulation, re-derived; D1, D2).
// E = x E1 + y E2 + z E3 + E4 over the 4-D null space of the 5 x 9 epipolar constraints;
// det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0 give 10 cubics in (x, y, z).  Monomials are ordered
// so that Gauss-Jordan on the 10 x 20 coefficient matrix leaves every equation with only
// z-polynomial multiples of {x, y, 1}; three combinations make a 3 x 3 polynomial matrix B(z)
// whose determinant is the degree-10 polynomial in z.
// ----------------------------------------------------------------------------------------------
// Multivariate polynomials in (x, y, z) by degree class, fixed monomial orders:
//  P1 (4):  x  y  z  1
//  P2 (10): x^2  y^2  xy  xz  x  yz  y  z^2  z  1
//  P3 (20): 0 x^3   1 y^3   2 x^2 y  3 x y^2  4 x^2 z  5 x^2   6 y^2 z  7 y^2   8 x y z  9 x y
//          10 x z^2 11 x z  12 x    13 y z^2 14 y z   15 y    16 z^3   17 z^2  18 z    19 1
// Products accumulate in the fixed (i, j) loop order of the tables below (no data-dependent
// skipping), so a device implementation with the same tables is bit-identical.
constexpr int kE1[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
constexpr int kE2[10][3] = {{2, 0, 0}, {0, 2, 0}, {1, 1, 0}, {1, 0, 1}, {1, 0, 0},
                            {0, 1, 1}, {0, 1, 0}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
constexpr int kE3[20][3] = {
    {3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0},
    {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0},
    {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
constexpr int idx2(int ex, int ey, int ez) {
    for (int i = 0; i < 10; ++i)
        if (kE2[i][0] == ex && kE2[i][1] == ey && kE2[i][2] == ez) return i;
    return -1;
}
constexpr int idx3(int ex, int ey, int ez) {
    for (int i = 0; i < 20; ++i)
        if (kE3[i][0] == ex && kE3[i][1] == ey && kE3[i][2] == ez) return i;
    return -1;
}
struct P1 { double c[4]; };
struct P2 { double c[10]; };
struct Poly3 { double c[20]; };
P2 mul11(const P1& a, const P1& b) {
    P2 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.c[idx2(kE1[i][0] + kE1[j][0], kE1[i][1] + kE1[j][1], kE1[i][2] + kE1[j][2])] += a.c[i] * b.c[j];
    return r;
}
Poly3 mul21(const P2& a, const P1& b) {
    Poly3 r{};
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j)
            r.c[idx3(kE2[i][0] + kE1[j][0], kE2[i][1] + kE1[j][1], kE2[i][2] + kE1[j][2])] += a.c[i] * b.c[j];
    return r;
}
P2 add2(const P2& a, const P2& b) { P2 r; for (int i = 0; i < 10; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
P2 sub2(const P2& a, const P2& b) { P2 r; for (int i = 0; i < 10; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
Poly3 poly_add(const Poly3& a, const Poly3& b) { Poly3 r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
Poly3 poly_sub(const Poly3& a, const Poly3& b) { Poly3 r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
Poly3 poly_scale(const Poly3& a, double s) { Poly3 r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] * s; return r; }

// univariate polynomial in z (degree <= 10)
struct PolyZ { double c[11]; int deg; };
PolyZ pz(int deg) { PolyZ p; for (int i = 0; i < 11; ++i) p.c[i] = 0.0; p.deg = deg; return p; }
PolyZ pz_mul(const PolyZ& a, const PolyZ& b) {
    PolyZ r = pz(a.deg + b.deg);
    for (int i = 0; i <= a.deg; ++i)
        for (int j = 0; j <= b.deg; ++j) r.c[i + j] += a.c[i] * b.c[j];
    return r;
}
PolyZ pz_sub(const PolyZ& a, const PolyZ& b) {
    PolyZ r = pz(std::max(a.deg, b.deg));
    for (int i = 0; i <= r.deg; ++i) r.c[i] = (i <= a.deg ? a.c[i] : 0.0) - (i <= b.deg ? b.c[i] : 0.0);
    return r;
}
PolyZ pz_add(const PolyZ& a, const PolyZ& b) {
    PolyZ r = pz(std::max(a.deg, b.deg));
    for (int i = 0; i <= r.deg; ++i) r.c[i] = (i <= a.deg ? a.c[i] : 0.0) + (i <= b.deg ? b.c[i] : 0.0);
    return r;
}

std::vector<Mat3> estimate_e5(const std::vector<Pt>& p1, const std::vector<Pt>& p2) {
    // 5-point is also COLMAP's local estimator: with more than 5 points it uses the 4 smallest
    // right singular vectors of the N x 9 system; we take the 4 smallest eigenvectors of A^T A.
    const size_t N = p1.size();
    double nsp[4 * 9];
    auto fill_row = [&](size_t i, double* r) {
        const double x1 = p1[i].x, y1 = p1[i].y, x2 = p2[i].x, y2 = p2[i].y;
        r[0] = x2 * x1; r[1] = x2 * y1; r[2] = x2;
        r[3] = y2 * x1; r[4] = y2 * y1; r[5] = y2;
        r[6] = x1; r[7] = y1; r[8] = 1;
    };
    if (N == 5) {
        double A[5 * 9];
        for (size_t i = 0; i < 5; ++i) fill_row(i, A + i * 9);
        nullspace9(5, A, nsp);
    } else {
#ifdef ORACLE_LAPACK_SVD
        lapack_null_vectors(N, fill_row, 4, nsp);  // E = svd.matrixV().block<9, 4>(0, 5)
#else
        double ata[81], v[81];
        accumulate_ata(N, fill_row, ata);
        jacobi_eigen(9, ata, v);
        int order[9];
        for (int i = 0; i < 9; ++i) order[i] = i;
        for (int i = 0; i < 9; ++i)  // selection sort by eigenvalue, stable
            for (int j = i + 1; j < 9; ++j)
                if (ata[order[j] * 9 + order[j]] < ata[order[i] * 9 + order[i]]) std::swap(order[i], order[j]);
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 9; ++i) nsp[k * 9 + i] = v[i * 9 + order[3 - k]];
#endif
    }
    // E(x,y,z) entries as linear forms: x*nsp[0] + y*nsp[1] + z*nsp[2] + nsp[3]
    P1 e[9];
    for (int k = 0; k < 9; ++k) e[k] = P1{{nsp[0 * 9 + k], nsp[1 * 9 + k], nsp[2 * 9 + k], nsp[3 * 9 + k]}};
    // det(E)
    Poly3 eq[10];
    eq[0] = poly_add(poly_sub(mul21(sub2(mul11(e[4], e[8]), mul11(e[5], e[7])), e[0]),
                              mul21(sub2(mul11(e[3], e[8]), mul11(e[5], e[6])), e[1])),
                     mul21(sub2(mul11(e[3], e[7]), mul11(e[4], e[6])), e[2]));
    // EEt = E E^T (degree 2), tr = trace(EEt)
    P2 eet[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            eet[3 * i + j] = add2(add2(mul11(e[3 * i], e[3 * j]), mul11(e[3 * i + 1], e[3 * j + 1])),
                                  mul11(e[3 * i + 2], e[3 * j + 2]));
    const P2 tr = add2(add2(eet[0], eet[4]), eet[8]);
    // 2 EEt E - tr E
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const Poly3 sacc = poly_add(poly_add(mul21(eet[3 * i], e[j]), mul21(eet[3 * i + 1], e[3 + j])),
                                        mul21(eet[3 * i + 2], e[6 + j]));
            eq[1 + 3 * i + j] = poly_sub(poly_scale(sacc, 2.0), mul21(tr, e[3 * i + j]));
        }
    // Gauss-Jordan on the 10 x 20 matrix, pivoting within the first 10 columns (rows only)
    double G[10][20];
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 20; ++c) G[r][c] = eq[r].c[c];
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        double pv = std::fabs(G[col][col]);
        for (int r = col + 1; r < 10; ++r)
            if (std::fabs(G[r][col]) > pv) { pv = std::fabs(G[r][col]); piv = r; }
        if (piv != col)
            for (int c = 0; c < 20; ++c) std::swap(G[col][c], G[piv][c]);
        const double inv = 1.0 / G[col][col];
        for (int c = 0; c < 20; ++c) G[col][c] = G[col][c] * inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = G[r][col];
            for (int c = 0; c < 20; ++c) G[r][c] = G[r][c] - f * G[col][c];
        }
    }
    // After elimination row r reads  mono_r + sum_{c>=10} G[r][c] mono_c = 0  with
    // mono_4 = x^2 z, mono_5 = x^2, mono_6 = y^2 z, mono_7 = y^2, mono_8 = xyz, mono_9 = xy.
    // <k> := row(2k+4) - z * row(2k+5), k = 0,1,2 cancels the leading monomials and leaves
    //   x * a(z) + y * b(z) + c(z) = 0  with deg a,b <= 3, deg c <= 4.
    PolyZ B[3][3];
    for (int k = 0; k < 3; ++k) {
        const double* hi = G[4 + 2 * k];  // leading x^2 z / y^2 z / xyz
        const double* lo = G[5 + 2 * k];  // leading x^2 / y^2 / xy
        PolyZ a = pz(3), b = pz(3), c = pz(4);
        // hi: x z^2 (10), x z (11), x (12), y z^2 (13), y z (14), y (15), z^3 (16), z^2 (17), z (18), 1 (19)
        a.c[2] += hi[10]; a.c[1] += hi[11]; a.c[0] += hi[12];
        b.c[2] += hi[13]; b.c[1] += hi[14]; b.c[0] += hi[15];
        c.c[3] += hi[16]; c.c[2] += hi[17]; c.c[1] += hi[18]; c.c[0] += hi[19];
        // minus z * lo
        a.c[3] -= lo[10]; a.c[2] -= lo[11]; a.c[1] -= lo[12];
        b.c[3] -= lo[13]; b.c[2] -= lo[14]; b.c[1] -= lo[15];
        c.c[4] -= lo[16]; c.c[3] -= lo[17]; c.c[2] -= lo[18]; c.c[1] -= lo[19];
        B[k][0] = a; B[k][1] = b; B[k][2] = c;
    }
    // det B(z): degree 10
    const PolyZ m0 = pz_sub(pz_mul(B[1][1], B[2][2]), pz_mul(B[1][2], B[2][1]));
    const PolyZ m1 = pz_sub(pz_mul(B[1][0], B[2][2]), pz_mul(B[1][2], B[2][0]));
    const PolyZ m2 = pz_sub(pz_mul(B[1][0], B[2][1]), pz_mul(B[1][1], B[2][0]));
    const PolyZ det = pz_add(pz_sub(pz_mul(B[0][0], m0), pz_mul(B[0][1], m1)), pz_mul(B[0][2], m2));
    double roots[10];
    const int nr = real_roots(det.c, 10, roots);
    std::vector<Mat3> models;
    for (int i = 0; i < nr; ++i) {
        const double z = roots[i];
        // (x, y, 1) spans the null space of B(z).  Upstream takes the last right singular vector X of the 3 x 3
        // matrix (JacobiSVD), skips the root when |X(2)| < 1e-10 and uses X(0) / X(2), X(1) / X(2).  Here (D1) the
        // null vector is the cross product of the two rows that give the longest one - the same direction, from
        // all three rows like the SVD, without the SVD.
        double Bz[3][3];
        for (int k = 0; k < 3; ++k) {
            Bz[k][0] = poly_eval(B[k][0].c, 3, z);
            Bz[k][1] = poly_eval(B[k][1].c, 3, z);
            Bz[k][2] = poly_eval(B[k][2].c, 4, z);
        }
        double X[3] = {0, 0, 0}, best_n2 = -1.0;
        const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (const auto& pr : pairs) {
            const double* u = Bz[pr[0]];
            const double* w = Bz[pr[1]];
            const double c0 = u[1] * w[2] - u[2] * w[1], c1 = u[2] * w[0] - u[0] * w[2], c2 = u[0] * w[1] - u[1] * w[0];
            const double n2 = c0 * c0 + c1 * c1 + c2 * c2;
            if (n2 > best_n2) { best_n2 = n2; X[0] = c0; X[1] = c1; X[2] = c2; }
        }
        const double nn = std::sqrt(best_n2);
        for (double& v : X) v = v / nn;
        if (!(std::fabs(X[2]) >= 1e-10)) continue;  // also drops NaN
        const double x = X[0] / X[2];
        const double y = X[1] / X[2];
        Mat3 E;
        for (int k = 0; k < 9; ++k) E.m[k] = x * nsp[k] + y * nsp[9 + k] + z * nsp[18 + k] + nsp[27 + k];
        // essential_vec /= essential_vec.norm()
        double n2 = 0.0;
        for (int k = 0; k < 9; ++k) n2 += E.m[k] * E.m[k];
        const double nrm = std::sqrt(n2);
        for (int k = 0; k < 9; ++k) E.m[k] = E.m[k] / nrm;
        models.push_back(E);
    }
    return models;
}

// ----------------------------------------------------------------------------------------------
// Relative pose of a verified pair (colmap/estimators/two_view_geometry.cc
// EstimateTwoViewGeometryPose; colmap/geometry/{essential_matrix,homography_matrix,triangulation,
// pose}.cc), restated without Eigen (D1: the 3x3 / 4x4 SVDs are taken from the Jacobi
// eigen-decomposition of A^T A; the rotations and translations they lead to do not depend on the
// SVD's sign and ordering conventions, only the order in which the four candidate poses are tried
// does, and that only matters when two candidates tie in the cheirality count).
//   CALIBRATED / UNCALIBRATED:        PoseFromEssentialMatrix(E, ...)
//   PLANAR / PANORAMIC / PLANAR_OR_..: PoseFromHomographyMatrix(H, K1, K2, ...)
// then tri_angle = median triangulation angle of the points in front of both cameras, and
// PLANAR_OR_PANORAMIC resolves to PANORAMIC (zero translation) or PLANAR.
// ----------------------------------------------------------------------------------------------
V3 v3_cross(const V3& a, const V3& b) {
    return V3{{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
double v3_dot(const V3& a, const V3& b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
double v3_norm(const V3& a) { return std::sqrt(v3_dot(a, a)); }
V3 v3_scale(const V3& a, double s) { return V3{{a.v[0] * s, a.v[1] * s, a.v[2] * s}}; }
V3 v3_normalized(const V3& a) {
    const double n = v3_norm(a);
    return V3{{a.v[0] / n, a.v[1] / n, a.v[2] / n}};
}
V3 mat3_vec(const Mat3& a, const V3& x) {
    V3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = a.m[3 * i] * x.v[0] + a.m[3 * i + 1] * x.v[1] + a.m[3 * i + 2] * x.v[2];
    return r;
}
double mat3_det(const Mat3& a) {
    const double* m = a.m;
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
Mat3 mat3_inv(const Mat3& a) {
    const double* m = a.m;
    const double d = mat3_det(a);
    Mat3 r;
    r.m[0] = (m[4] * m[8] - m[5] * m[7]) / d; r.m[1] = (m[2] * m[7] - m[1] * m[8]) / d; r.m[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    r.m[3] = (m[5] * m[6] - m[3] * m[8]) / d; r.m[4] = (m[0] * m[8] - m[2] * m[6]) / d; r.m[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    r.m[6] = (m[3] * m[7] - m[4] * m[6]) / d; r.m[7] = (m[1] * m[6] - m[0] * m[7]) / d; r.m[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return r;
}
Mat3 calibration_matrix(const Camera& c) {  // Camera::CalibrationMatrix
    Mat3 k{};
    const int nf = kModels[c.model_id].num_focal;
    k.m[0] = c.params[0];
    k.m[4] = c.params[nf - 1];
    k.m[2] = c.params[nf];
    k.m[5] = c.params[nf + 1];
    k.m[8] = 1.0;
    return k;
}

// A = U diag(S) V^T, S descending, from the eigen-decomposition of A^T A.  u_k = A v_k / s_k for the two
// largest singular values, u_2 = u_0 x u_1 (so det U = +1 and a vanishing third singular value is fine).
void svd3(const Mat3& A, Mat3* U, double S[3], Mat3* V) {
#ifdef ORACLE_LAPACK_SVD
    {
        double Vt[9];
        lapack_svd(3, 3, A.m, S, U->m, Vt);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) V->m[3 * i + j] = Vt[3 * j + i];
        return;
    }
#endif
    double ata[9], ev[9];
    const Mat3 At = mat3_t(A);
    const Mat3 P = mat3_mul(At, A);
    std::memcpy(ata, P.m, sizeof ata);
    jacobi_eigen(3, ata, ev);
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (ata[4 * ord[j]] > ata[4 * ord[i]]) std::swap(ord[i], ord[j]);
    V3 vcol[3], ucol[3];
    for (int k = 0; k < 3; ++k) {
        const double lam = ata[4 * ord[k]];
        S[k] = std::sqrt(lam < 0.0 ? 0.0 : lam);
        vcol[k] = V3{{ev[0 * 3 + ord[k]], ev[1 * 3 + ord[k]], ev[2 * 3 + ord[k]]}};
    }
    for (int k = 0; k < 2; ++k) {
        if (S[k] == 0.0) {  // rank-deficient input (the all-zero E): fall back to the unit vector, U = I
            ucol[k] = V3{{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, 0.0}};
            continue;
        }
        const double inv = 1.0 / S[k];
        ucol[k] = v3_scale(mat3_vec(A, vcol[k]), inv);
    }
    ucol[2] = v3_cross(ucol[0], ucol[1]);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) { U->m[3 * i + k] = ucol[k].v[i]; V->m[3 * i + k] = vcol[k].v[i]; }
}

// DecomposeEssentialMatrix
void decompose_essential(const Mat3& E, Mat3* R1, Mat3* R2, V3* t) {
    Mat3 U, V;
    double S[3];
    svd3(E, &U, S, &V);
    Mat3 Vt = mat3_t(V);
    if (mat3_det(U) < 0) for (double& x : U.m) x = -x;
    if (mat3_det(Vt) < 0) for (double& x : Vt.m) x = -x;
    const Mat3 W{{0, 1, 0, -1, 0, 0, 0, 0, 1}};
    *R1 = mat3_mul(mat3_mul(U, W), Vt);
    *R2 = mat3_mul(mat3_mul(U, mat3_t(W)), Vt);
    *t = v3_normalized(V3{{U.m[2], U.m[5], U.m[8]}});
}

// TriangulatePoint: DLT, the right singular vector of the smallest singular value of the 4 x 4 system,
// dehomogenised.  P1 = [I | 0], P2 = [R | t].
V3 triangulate_point(const Mat3& R, const V3& t, const Pt& x1, const Pt& x2) {
    double A[4][4];
    const double P1[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    double P2[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P2[i][j] = R.m[3 * i + j]; P2[i][3] = t.v[i]; }
    for (int j = 0; j < 4; ++j) {
        A[0][j] = x1.x * P1[2][j] - P1[0][j];
        A[1][j] = x1.y * P1[2][j] - P1[1][j];
        A[2][j] = x2.x * P2[2][j] - P2[0][j];
        A[3][j] = x2.y * P2[2][j] - P2[1][j];
    }
#ifdef ORACLE_LAPACK_SVD
    {
        double S4[4], Vt4[16], Af[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Af[4 * i + j] = A[i][j];
        lapack_svd(4, 4, Af, S4, nullptr, Vt4);
        const double w4 = Vt4[12 + 3];
        return V3{{Vt4[12] / w4, Vt4[13] / w4, Vt4[14] / w4}};
    }
#endif
    double ata[16], ev[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
            for (int k = 0; k < 4; ++k) sum += A[k][i] * A[k][j];
            ata[4 * i + j] = sum;
        }
    jacobi_eigen(4, ata, ev);
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (ata[5 * i] < ata[5 * best]) best = i;
    const double w = ev[3 * 4 + best];
    return V3{{ev[0 * 4 + best] / w, ev[1 * 4 + best] / w, ev[2 * 4 + best] / w}};
}

// CheckCheirality: the triangulated points in front of both cameras (and not absurdly far)
void check_cheirality(const Mat3& R, const V3& t, const std::vector<Pt>& p1, const std::vector<Pt>& p2,
                      std::vector<V3>* points3D) {
    const double kMinDepth = std::numeric_limits<double>::epsilon();
    const double max_depth = 1000.0 * v3_norm(mat3_vec(mat3_t(R), t));
    const double n2 = std::sqrt(R.m[2] * R.m[2] + R.m[5] * R.m[5] + R.m[8] * R.m[8]);  // |P2.col(2)|
    points3D->clear();
    for (size_t i = 0; i < p1.size(); ++i) {
        const V3 X = triangulate_point(R, t, p1[i], p2[i]);
        const double depth1 = (0.0 * X.v[0] + 0.0 * X.v[1] + 1.0 * X.v[2] + 0.0 * 1.0) * 1.0;
        if (depth1 > kMinDepth && depth1 < max_depth) {
            const double depth2 = (R.m[6] * X.v[0] + R.m[7] * X.v[1] + R.m[8] * X.v[2] + t.v[2] * 1.0) * n2;
            if (depth2 > kMinDepth && depth2 < max_depth) points3D->push_back(X);
        }
    }
}

struct PoseCandidates { std::vector<Mat3> R; std::vector<V3> t; std::vector<V3> n; };  // n: plane normals (H only)

// the candidate (later ones win ties) with the most points in front of both cameras
void best_candidate(const PoseCandidates& c, const std::vector<Pt>& p1, const std::vector<Pt>& p2, Mat3* R, V3* t,
                    std::vector<V3>* points3D) {
    points3D->clear();
    for (size_t i = 0; i < c.R.size(); ++i) {
        std::vector<V3> cand;
        check_cheirality(c.R[i], c.t[i], p1, p2, &cand);
        if (cand.size() >= points3D->size()) { *R = c.R[i]; *t = c.t[i]; *points3D = cand; }
    }
}

// PoseFromEssentialMatrix
void pose_from_essential(const Mat3& E, const std::vector<Pt>& p1, const std::vector<Pt>& p2, Mat3* R, V3* t,
                         std::vector<V3>* points3D) {
    Mat3 R1, R2;
    V3 tt;
    decompose_essential(E, &R1, &R2, &tt);
    PoseCandidates c;
    c.R = {R1, R2, R1, R2};
    c.t = {tt, tt, v3_scale(tt, -1.0), v3_scale(tt, -1.0)};
    best_candidate(c, p1, p2, R, t, points3D);
}

int sign_of(double x) { return (0.0 < x) - (x < 0.0); }
double opposite_of_minor(const Mat3& S, int row, int col) {
    const int col1 = col == 0 ? 1 : 0, col2 = col == 2 ? 1 : 2;
    const int row1 = row == 0 ? 1 : 0, row2 = row == 2 ? 1 : 2;
    return S.m[3 * row1 + col2] * S.m[3 * row2 + col1] - S.m[3 * row1 + col1] * S.m[3 * row2 + col2];
}
Mat3 homography_rotation(const Mat3& Hn, const V3& tstar, const V3& n, double v) {
    Mat3 M;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M.m[3 * i + j] = (i == j ? 1.0 : 0.0) - (2.0 / v) * tstar.v[i] * n.v[j];
    return mat3_mul(Hn, M);
}
// DecomposeHomographyMatrix (Malis & Vargas, "Deeper understanding of the homography decomposition
// for vision-based control", analytical method)
void decompose_homography(const Mat3& H, const Mat3& K1, const Mat3& K2, PoseCandidates* out) {
    Mat3 Hn = mat3_mul(mat3_mul(mat3_inv(K2), H), K1);
    {
        Mat3 U, V;
        double S[3];
        svd3(Hn, &U, S, &V);
        for (double& x : Hn.m) x /= S[1];
    }
    if (mat3_det(Hn) < 0) for (double& x : Hn.m) x *= -1.0;
    Mat3 S = mat3_mul(mat3_t(Hn), Hn);
    S.m[0] -= 1.0; S.m[4] -= 1.0; S.m[8] -= 1.0;
    double inf_norm = 0.0;  // lpNorm<Infinity> of a matrix expression: largest absolute coefficient
    for (double x : S.m) inf_norm = std::max(inf_norm, std::fabs(x));
    out->R.clear(); out->t.clear(); out->n.clear();
    if (inf_norm < 1e-3) {  // H is a rotation
        out->R = {Hn};
        out->t = {V3{{0, 0, 0}}};
        out->n = {V3{{0, 0, 0}}};
        return;
    }
    const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    const double rtM00 = std::sqrt(M00), rtM11 = std::sqrt(M11), rtM22 = std::sqrt(M22);
    const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    const int e12 = sign_of(M12), e02 = sign_of(M02), e01 = sign_of(M01);
    const double nS[3] = {std::fabs(S.m[0]), std::fabs(S.m[4]), std::fabs(S.m[8])};
    int idx = 0;  // std::max_element: first maximum
    for (int i = 1; i < 3; ++i)
        if (nS[i] > nS[idx]) idx = i;
    V3 np1, np2;
    if (idx == 0) {
        np1 = V3{{S.m[0], S.m[1] + rtM22, S.m[2] + e12 * rtM11}};
        np2 = V3{{S.m[0], S.m[1] - rtM22, S.m[2] - e12 * rtM11}};
    } else if (idx == 1) {
        np1 = V3{{S.m[1] + rtM22, S.m[4], S.m[5] - e02 * rtM00}};
        np2 = V3{{S.m[1] - rtM22, S.m[4], S.m[5] + e02 * rtM00}};
    } else {
        np1 = V3{{S.m[2] + e01 * rtM11, S.m[5] + rtM00, S.m[8]}};
        np2 = V3{{S.m[2] - e01 * rtM11, S.m[5] - rtM00, S.m[8]}};
    }
    const double traceS = S.m[0] + S.m[4] + S.m[8];
    const double v = 2.0 * std::sqrt(1.0 + traceS - M00 - M11 - M22);
    const double ESii = sign_of(S.m[4 * idx]);
    const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
    const double r = std::sqrt(r_2), n_t = std::sqrt(nt_2);
    const V3 n1 = v3_normalized(np1), n2 = v3_normalized(np2);
    const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
    V3 t1s, t2s;
    for (int i = 0; i < 3; ++i) {
        t1s.v[i] = half_nt * (esii_t_r * n2.v[i] - n_t * n1.v[i]);
        t2s.v[i] = half_nt * (esii_t_r * n1.v[i] - n_t * n2.v[i]);
    }
    const Mat3 R1 = homography_rotation(Hn, t1s, n1, v);
    const V3 t1 = mat3_vec(R1, t1s);
    const Mat3 R2 = homography_rotation(Hn, t2s, n2, v);
    const V3 t2 = mat3_vec(R2, t2s);
    out->R = {R1, R1, R2, R2};
    out->t = {t1, v3_scale(t1, -1.0), t2, v3_scale(t2, -1.0)};
    out->n = {v3_scale(n1, -1.0), n1, v3_scale(n2, -1.0), n2};
}

// Eigen::Quaterniond(rotation matrix) -> (w, x, y, z)
void rotation_to_quaternion(const Mat3& R, double q[4]) {
    const double* m = R.m;
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
}

// CalculateTriangulationAngles + Median
double median_tri_angle(const V3& c1, const V3& c2, const std::vector<V3>& X) {
    const double baseline2 = (c1.v[0] - c2.v[0]) * (c1.v[0] - c2.v[0]) + (c1.v[1] - c2.v[1]) * (c1.v[1] - c2.v[1]) +
                             (c1.v[2] - c2.v[2]) * (c1.v[2] - c2.v[2]);
    std::vector<double> ang(X.size());
    for (size_t i = 0; i < X.size(); ++i) {
        double r1 = 0.0, r2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            r1 += (X[i].v[k] - c1.v[k]) * (X[i].v[k] - c1.v[k]);
            r2 += (X[i].v[k] - c2.v[k]) * (X[i].v[k] - c2.v[k]);
        }
        const double den = 2.0 * std::sqrt(r1 * r2);
        if (den == 0.0) { ang[i] = 0.0; continue; }
        const double nom = r1 + r2 - baseline2;
        const double a = std::fabs(std::acos(nom / den));
        ang[i] = std::min(a, M_PI - a);
    }
    std::sort(ang.begin(), ang.end());
    const size_t mid = ang.size() / 2;
    if (ang.size() % 2 == 0) return 0.5 * ang[mid] + 0.5 * ang[mid - 1];
    return ang[mid];
}

// EstimateTwoViewGeometryPose.  pts: keypoints; matches/inlier_mask as produced by the estimation.
RelPose estimate_relative_pose(const Camera& c1, const std::vector<Pt>& pts1, const Camera& c2,
                               const std::vector<Pt>& pts2, const uint32_t* matches, size_t M, const Tvg& g) {
    RelPose out;
    out.config = g.config;
    if (g.config != CALIBRATED && g.config != UNCALIBRATED && g.config != PLANAR && g.config != PANORAMIC &&
        g.config != PLANAR_OR_PANORAMIC)
        return out;
    std::vector<Pt> n1, n2;
    for (size_t i = 0; i < M; ++i)
        if (i < g.inlier_mask.size() && g.inlier_mask[i]) {
            n1.push_back(cam_from_img(c1, pts1[matches[2 * i]]));
            n2.push_back(cam_from_img(c2, pts2[matches[2 * i + 1]]));
        }
    std::vector<V3> X;
    if (g.config == CALIBRATED || g.config == UNCALIBRATED) {
        // g.E also for UNCALIBRATED ("most likely leads to an ill-defined reconstruction", upstream): the E
        // model of the calibrated path, or the default all-zero E of the uncalibrated path
        pose_from_essential(g.E, n1, n2, &out.R, &out.t, &X);
    } else {
        PoseCandidates c;
        decompose_homography(g.H, calibration_matrix(c1), calibration_matrix(c2), &c);
        best_candidate(c, n1, n2, &out.R, &out.t, &X);
    }
    out.ok = true;
    rotation_to_quaternion(out.R, out.qvec);
    out.num_points3D = X.size();
    if (X.empty()) {
        out.tri_angle = 0.0;
    } else {
        const V3 c2c = v3_scale(mat3_vec(mat3_t(out.R), out.t), -1.0);
        out.tri_angle = median_tri_angle(V3{{0, 0, 0}}, c2c, X);
    }
    if (g.config == PLANAR_OR_PANORAMIC) {
        if (v3_norm(out.t) == 0.0) { out.config = PANORAMIC; out.tri_angle = 0.0; }
        else out.config = PLANAR;
    }
    return out;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// C API (ctypes) for tests and the CPU baseline
// ----------------------------------------------------------------------------------------------
// EstimateMultipleTwoViewGeometries (colmap/estimators/two_view_geometry.cc), the multiple_models
// path: estimate, set the inliers aside, estimate again on what remains, until a round comes back
// DEGENERATE.  WATERMARK rounds are dropped (their inliers still leave the pool) when
// multiple_ignore_watermark is set.  No geometry -> DEGENERATE; one -> that geometry; several ->
// config MULTIPLE whose inlier matches are the geometries' inlier matches one after the other (the
// models of a MULTIPLE geometry stay default-constructed).  Every round reseeds like any other
// EstimateTwoViewGeometry call here (D4).  inlier_mask[i] = 1 + index of the geometry match i belongs
// to (0: none), which also encodes COLMAP's concatenation order.
Tvg estimate_multiple_two_view_geometries(const Camera& c1, const std::vector<Pt>& pts1, const Camera& c2,
                                          const std::vector<Pt>& pts2, const uint32_t* matches, size_t M,
                                          const TvgOptions& o, uint32_t seed) {
    TvgOptions single = o;
    single.multiple_models = 0;
    std::vector<size_t> remaining(M);
    for (size_t i = 0; i < M; ++i) remaining[i] = i;
    std::vector<Tvg> geometries;
    std::vector<char> label(M, 0);
    for (int round = 0; round < 254; ++round) {
        std::vector<uint32_t> rm(2 * remaining.size());
        for (size_t k = 0; k < remaining.size(); ++k) {
            rm[2 * k] = matches[2 * remaining[k]];
            rm[2 * k + 1] = matches[2 * remaining[k] + 1];
        }
        Tvg g = estimate_two_view_geometry(c1, pts1, c2, pts2, rm.data(), remaining.size(), single, seed);
        if (g.config == DEGENERATE) break;
        const bool keep = !(o.multiple_ignore_watermark && g.config == WATERMARK);
        if (keep) geometries.push_back(g);
        std::vector<size_t> next;
        for (size_t k = 0; k < remaining.size(); ++k) {
            const bool in = k < g.inlier_mask.size() && g.inlier_mask[k];
            if (in) {
                if (keep) label[remaining[k]] = static_cast<char>(geometries.size());
            } else {
                next.push_back(remaining[k]);
            }
        }
        if (next.size() == remaining.size()) break;  // nothing left the pool: COLMAP would spin here
        remaining.swap(next);
    }
    Tvg out;
    out.inlier_mask = label;
    if (geometries.empty()) {
        out.config = DEGENERATE;
        std::fill(out.inlier_mask.begin(), out.inlier_mask.end(), 0);
    } else if (geometries.size() == 1) {
        const Tvg& g = geometries[0];
        out.config = g.config;
        out.E = g.E; out.F = g.F; out.H = g.H;
        out.num_inliers = g.num_inliers;
        for (int i = 0; i < 4; ++i) out.trials[i] = g.trials[i];
        for (int i = 0; i < 3; ++i) out.inl[i] = g.inl[i];
        out.pose = g.pose;
    } else {
        out.config = MULTIPLE;
        for (const Tvg& g : geometries) out.num_inliers += g.num_inliers;
    }
    return out;
}

#define ORACLE_TVG_VERSION "tvg-r4: D1 jacobi-AtA/gauss-jordan, D2 bisect-2^-26+3-newton, D3 det_sum64, D4 reseed-per-pair"
extern "C" {
const char* oracle_tvg_version() { return ORACLE_TVG_VERSION; }

struct oracle_tvg_options {
    int32_t min_num_inliers;
    int32_t detect_watermark;
    int32_t multiple_ignore_watermark;
    int32_t force_H_use;
    int32_t compute_relative_pose;
    int32_t multiple_models;
    double min_E_F_inlier_ratio;
    double max_H_inlier_ratio;
    double watermark_min_inlier_ratio;
    double watermark_border_size;
    double max_error;
    double min_inlier_ratio;
    double confidence;
    double dyn_num_trials_multiplier;
    int64_t min_num_trials;
    int64_t max_num_trials;
};

struct oracle_camera {
    int32_t model_id;
    int32_t has_prior_focal_length;
    uint64_t width, height;
    double params[12];
};

struct oracle_tvg_result {
    int32_t config;
    int32_t num_inliers;
    double E[9], F[9], H[9];
    int64_t trials[4];
    int64_t inl[3];
    // EstimateTwoViewGeometryPose (compute_relative_pose); pose_ok = 0 leaves the defaults (identity, 0)
    int32_t pose_ok;
    int32_t num_points3D;
    double qvec[4], tvec[3], R[9];
    double tri_angle;
};

static void put_pose(const RelPose& p, oracle_tvg_result* out) {
    out->pose_ok = p.ok ? 1 : 0;
    out->num_points3D = static_cast<int32_t>(p.num_points3D);
    std::memcpy(out->qvec, p.qvec, sizeof out->qvec);
    std::memcpy(out->tvec, p.t.v, sizeof out->tvec);
    std::memcpy(out->R, p.R.m, sizeof out->R);
    out->tri_angle = p.tri_angle;
}

static TvgOptions to_opts(const oracle_tvg_options* o) {
    TvgOptions t;
    t.min_num_inliers = o->min_num_inliers;
    t.min_E_F_inlier_ratio = o->min_E_F_inlier_ratio;
    t.max_H_inlier_ratio = o->max_H_inlier_ratio;
    t.watermark_min_inlier_ratio = o->watermark_min_inlier_ratio;
    t.watermark_border_size = o->watermark_border_size;
    t.detect_watermark = o->detect_watermark;
    t.multiple_ignore_watermark = o->multiple_ignore_watermark;
    t.force_H_use = o->force_H_use;
    t.compute_relative_pose = o->compute_relative_pose;
    t.multiple_models = o->multiple_models;
    t.ransac = RansacOptions{o->max_error, o->min_inlier_ratio, o->confidence,
                             o->dyn_num_trials_multiplier, o->min_num_trials, o->max_num_trials};
    return t;
}
static Camera to_cam(const oracle_camera* c) {
    Camera r;
    r.model_id = c->model_id;
    r.has_prior_focal_length = c->has_prior_focal_length;
    r.width = c->width;
    r.height = c->height;
    std::memcpy(r.params, c->params, sizeof r.params);
    return r;
}

// C++ defaults of TwoViewGeometryOptions
void oracle_tvg_options_default(oracle_tvg_options* o) {
    o->min_num_inliers = 15;
    o->detect_watermark = 1;
    o->multiple_ignore_watermark = 1;
    o->force_H_use = 0;
    o->compute_relative_pose = 0;
    o->multiple_models = 0;
    o->min_E_F_inlier_ratio = 0.95;
    o->max_H_inlier_ratio = 0.8;
    o->watermark_min_inlier_ratio = 0.7;
    o->watermark_border_size = 0.1;
    o->max_error = 4.0;
    o->min_inlier_ratio = 0.25;
    o->confidence = 0.999;
    o->dyn_num_trials_multiplier = 3.0;
    o->min_num_trials = 100;
    o->max_num_trials = 10000;
}

// EstimateTwoViewGeometry for one pair. pts: all keypoints (x,y) of each image as doubles;
// matches: M x 2 uint32.  inlier_mask: M chars out.  Returns 0, or -1 for unsupported input.
int oracle_estimate_two_view_geometry(const oracle_camera* cam1, const double* pts1, size_t n1,
                                      const oracle_camera* cam2, const double* pts2, size_t n2,
                                      const uint32_t* matches, size_t M,
                                      const oracle_tvg_options* opts, uint32_t seed,
                                      oracle_tvg_result* out, char* inlier_mask) {
    const Camera c1 = to_cam(cam1), c2 = to_cam(cam2);
    if (!camera_supported(c1) || !camera_supported(c2)) return -1;
    for (size_t i = 0; i < M; ++i)
        if (matches[2 * i] >= n1 || matches[2 * i + 1] >= n2) return -1;
    std::vector<Pt> a(n1), b(n2);
    for (size_t i = 0; i < n1; ++i) a[i] = Pt{pts1[2 * i], pts1[2 * i + 1]};
    for (size_t i = 0; i < n2; ++i) b[i] = Pt{pts2[2 * i], pts2[2 * i + 1]};
    const Tvg g = opts->multiple_models
                      ? estimate_multiple_two_view_geometries(c1, a, c2, b, matches, M, to_opts(opts), seed)
                      : estimate_two_view_geometry(c1, a, c2, b, matches, M, to_opts(opts), seed);
    out->config = g.config;
    out->num_inliers = static_cast<int32_t>(g.num_inliers);
    std::memcpy(out->E, g.E.m, sizeof out->E);
    std::memcpy(out->F, g.F.m, sizeof out->F);
    std::memcpy(out->H, g.H.m, sizeof out->H);
    for (int i = 0; i < 4; ++i) out->trials[i] = static_cast<int64_t>(g.trials[i]);
    for (int i = 0; i < 3; ++i) out->inl[i] = static_cast<int64_t>(g.inl[i]);
    for (size_t i = 0; i < M; ++i) inlier_mask[i] = i < g.inlier_mask.size() ? g.inlier_mask[i] : 0;
    put_pose(g.pose, out);
    return 0;
}

// EstimateTwoViewGeometryPose on a given geometry: config, E, H and the inlier matches (Mi x 2).
// out->config is the config after the call; out->pose_ok its return value.
int oracle_estimate_two_view_geometry_pose(const oracle_camera* cam1, const double* pts1, size_t n1,
                                           const oracle_camera* cam2, const double* pts2, size_t n2,
                                           const uint32_t* inlier_matches, size_t Mi, int32_t config,
                                           const double* E9, const double* H9, oracle_tvg_result* out) {
    const Camera c1 = to_cam(cam1), c2 = to_cam(cam2);
    if (!camera_supported(c1) || !camera_supported(c2)) return -1;
    for (size_t i = 0; i < Mi; ++i)
        if (inlier_matches[2 * i] >= n1 || inlier_matches[2 * i + 1] >= n2) return -1;
    std::vector<Pt> a(n1), b(n2);
    for (size_t i = 0; i < n1; ++i) a[i] = Pt{pts1[2 * i], pts1[2 * i + 1]};
    for (size_t i = 0; i < n2; ++i) b[i] = Pt{pts2[2 * i], pts2[2 * i + 1]};
    Tvg g;
    g.config = config;
    std::memcpy(g.E.m, E9, sizeof g.E.m);
    std::memcpy(g.H.m, H9, sizeof g.H.m);
    g.inlier_mask.assign(Mi, 1);
    g.num_inliers = Mi;
    const RelPose p = estimate_relative_pose(c1, a, c2, b, inlier_matches, Mi, g);
    std::memset(out, 0, sizeof *out);
    out->config = p.ok ? p.config : config;
    out->num_inliers = static_cast<int32_t>(Mi);
    put_pose(p, out);
    return 0;
}

// PoseFromHomographyMatrix (colmap/geometry/homography_matrix.cc; pycolmap.homography_decomposition,
// /root/reference/pycolmap/geometry/homography_matrix.h:13-31): the candidate of DecomposeHomographyMatrix with the most
// points in front of both cameras (later candidates win ties), its plane normal and those points.  points: n x 2 in
// camera coordinates; points3D: room for n x 3.  Returns the number of points3D.
int oracle_pose_from_homography(const double* H9, const double* K1_9, const double* K2_9, const double* points1,
                                const double* points2, size_t n, double* R9, double* t3, double* normal3,
                                double* points3D) {
    Mat3 H, K1, K2;
    std::memcpy(H.m, H9, sizeof H.m);
    std::memcpy(K1.m, K1_9, sizeof K1.m);
    std::memcpy(K2.m, K2_9, sizeof K2.m);
    std::vector<Pt> a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
        a[i] = Pt{points1[2 * i], points1[2 * i + 1]};
        b[i] = Pt{points2[2 * i], points2[2 * i + 1]};
    }
    PoseCandidates c;
    decompose_homography(H, K1, K2, &c);
    Mat3 R{};
    V3 t{}, nrm{};
    std::vector<V3> X;
    for (size_t i = 0; i < c.R.size(); ++i) {
        std::vector<V3> cand;
        check_cheirality(c.R[i], c.t[i], a, b, &cand);
        if (cand.size() >= X.size()) { R = c.R[i]; t = c.t[i]; nrm = c.n[i]; X = cand; }
    }
    std::memcpy(R9, R.m, sizeof R.m);
    std::memcpy(t3, t.v, sizeof t.v);
    std::memcpy(normal3, nrm.v, sizeof nrm.v);
    for (size_t i = 0; i < X.size(); ++i) std::memcpy(points3D + 3 * i, X[i].v, sizeof X[i].v);
    return static_cast<int>(X.size());
}

// The same for a batch of pairs, one pair per OpenMP thread at a time (used to time the oracle on
// all host cores).  Pair p: cameras cams1[p] / cams2[p], keypoints pts1[p] (n1[p] x 2) / pts2[p],
// matches[p] (M[p] x 2), mask written at inlier_masks + mask_off[p].  Returns the number of pairs
// with unsupported input.
int oracle_estimate_two_view_geometry_batch(size_t npairs, const oracle_camera* cams1, const double* const* pts1,
                                            const size_t* n1, const oracle_camera* cams2,
                                            const double* const* pts2, const size_t* n2,
                                            const uint32_t* const* matches, const size_t* M,
                                            const size_t* mask_off, const oracle_tvg_options* opts,
                                            uint32_t seed, oracle_tvg_result* out, char* inlier_masks,
                                            int num_threads) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1) reduction(+ : bad)
    for (long p = 0; p < static_cast<long>(npairs); ++p)
        if (oracle_estimate_two_view_geometry(&cams1[p], pts1[p], n1[p], &cams2[p], pts2[p], n2[p], matches[p],
                                              M[p], opts, seed, &out[p], inlier_masks + mask_off[p]) != 0)
            ++bad;
    return bad;
}

// Single-model LO-RANSAC with a fresh PRNG(seed): the semantics of pycolmap's
// fundamental_matrix_estimation / homography_matrix_estimation / essential_matrix_estimation
// (/root/reference/pycolmap/estimators/fundamental_matrix.h:17-39).  kind: 0 = F (7pt/8pt),
// 1 = H, 2 = E (points already in camera coordinates).  Returns success (0/1).
int oracle_ransac_estimate(int kind, const double* p1, const double* p2, size_t n,
                           const oracle_tvg_options* ro, uint32_t seed, double* model9,
                           int64_t* num_inliers, int64_t* num_trials, char* inlier_mask) {
    std::vector<Pt> a(n), b(n);
    for (size_t i = 0; i < n; ++i) { a[i] = Pt{p1[2 * i], p1[2 * i + 1]}; b[i] = Pt{p2[2 * i], p2[2 * i + 1]}; }
    Prng prng(seed);
    const RansacOptions o{ro->max_error, ro->min_inlier_ratio, ro->confidence,
                          ro->dyn_num_trials_multiplier, ro->min_num_trials, ro->max_num_trials};
    const EstKind e = kind == 0 ? EST_F7 : kind == 1 ? EST_H : EST_E5;
    const EstKind l = kind == 0 ? EST_F8 : kind == 1 ? EST_H : EST_E5;
    const Report r = lo_ransac(e, l, o, prng, a, b);
    std::memcpy(model9, r.model.m, 9 * sizeof(double));
    *num_inliers = static_cast<int64_t>(r.support.num_inliers);
    *num_trials = static_cast<int64_t>(r.num_trials);
    for (size_t i = 0; i < n; ++i) inlier_mask[i] = (r.success && i < r.inlier_mask.size()) ? r.inlier_mask[i] : 0;
    return r.success ? 1 : 0;
}

// unit-test hooks ------------------------------------------------------------------------------
// Camera::CamFromImg of n image points; returns -1 for an unknown model
int oracle_cam_from_img(const oracle_camera* cam, const double* xy, size_t n, double* uv) {
    const Camera c = to_cam(cam);
    if (!camera_supported(c)) return -1;
    for (size_t i = 0; i < n; ++i) {
        const Pt q = cam_from_img(c, Pt{xy[2 * i], xy[2 * i + 1]});
        uv[2 * i] = q.x;
        uv[2 * i + 1] = q.y;
    }
    return 0;
}

int oracle_img_from_cam(const oracle_camera* cam, const double* uv, size_t n, double* xy) {
    const Camera c = to_cam(cam);
    if (!camera_supported(c)) return -1;
    for (size_t i = 0; i < n; ++i) {
        const Pt q = img_from_cam(c, Pt{uv[2 * i], uv[2 * i + 1]});
        xy[2 * i] = q.x;
        xy[2 * i + 1] = q.y;
    }
    return 0;
}
double oracle_cam_from_img_threshold(const oracle_camera* cam, double threshold) {
    return cam_from_img_threshold(to_cam(cam), threshold);
}
void oracle_calibration_matrix(const oracle_camera* cam, double* K9) {
    const Mat3 k = calibration_matrix(to_cam(cam));
    std::memcpy(K9, k.m, sizeof k.m);
}
void oracle_sampson_error(const double* p1, const double* p2, size_t n, const double* E9, double* out) {
    Mat3 E;
    std::memcpy(E.m, E9, sizeof E.m);
    for (size_t i = 0; i < n; ++i) out[i] = sampson(E, Pt{p1[2 * i], p1[2 * i + 1]}, Pt{p2[2 * i], p2[2 * i + 1]});
}
void oracle_h_residuals(const double* p1, const double* p2, size_t n, const double* H9, double* out) {
    Mat3 H;
    std::memcpy(H.m, H9, sizeof H.m);
    for (size_t i = 0; i < n; ++i) out[i] = h_residual(H, Pt{p1[2 * i], p1[2 * i + 1]}, Pt{p2[2 * i], p2[2 * i + 1]});
}
// kind: 0 F7 (n==7), 1 F8, 2 H, 4 E5.  Writes up to 10 models; returns the count.
int oracle_estimate_models(int kind, const double* p1, const double* p2, size_t n, double* models) {
    std::vector<Pt> a(n), b(n);
    for (size_t i = 0; i < n; ++i) { a[i] = Pt{p1[2 * i], p1[2 * i + 1]}; b[i] = Pt{p2[2 * i], p2[2 * i + 1]}; }
    const std::vector<Mat3> ms = estimate(static_cast<EstKind>(kind), a, b);
    for (size_t k = 0; k < ms.size() && k < 10; ++k) std::memcpy(models + 9 * k, ms[k].m, 9 * sizeof(double));
    return static_cast<int>(std::min<size_t>(ms.size(), 10));
}
// the sample stream of RandomSampler(k) over `total` items after SetPRNGSeed(seed): trials x k
void oracle_sample_stream(uint32_t seed, uint32_t total, uint32_t k, uint32_t trials, uint32_t* out) {
    Prng prng(seed);
    Sampler s(k);
    s.initialize(total);
    std::vector<size_t> idx(k);
    for (uint32_t t = 0; t < trials; ++t) {
        s.sample(prng, idx.data());
        for (uint32_t i = 0; i < k; ++i) out[t * k + i] = static_cast<uint32_t>(idx[i]);
    }
}
// raw std::uniform_int_distribution<uint32_t>(lo, hi) draws from mt19937(seed)
void oracle_uniform_draws(uint32_t seed, const uint32_t* lo, const uint32_t* hi, uint32_t n, uint32_t* out) {
    Prng prng(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = prng.uniform(lo[i], hi[i]);
}
int64_t oracle_compute_num_trials(int64_t num_inliers, int64_t num_samples, double confidence,
                                  double multiplier, int min_num_samples) {
    const size_t v = compute_num_trials(static_cast<size_t>(num_inliers), static_cast<size_t>(num_samples),
                                        confidence, multiplier, min_num_samples);
    return v > static_cast<size_t>(std::numeric_limits<int64_t>::max()) ? std::numeric_limits<int64_t>::max()
                                                                         : static_cast<int64_t>(v);
}
int oracle_real_roots(const double* coeffs, int deg, double* roots) { return real_roots(coeffs, deg, roots); }
#ifdef ORACLE_ROOT_STATS
void oracle_root_iter_hist(long* out) { for (int i = 0; i <= 200; ++i) { out[i] = g_root_iter_hist[i]; g_root_iter_hist[i] = 0; } }
#endif
void oracle_jacobi_eigen(int n, double* a, double* v) { jacobi_eigen(n, a, v); }
double oracle_det_sum64(const double* x, size_t n) { return det_sum64(n, [&](size_t k) { return x[k]; }); }

}  // extern "C"
