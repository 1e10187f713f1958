// tvg.hip — batched two-view geometric verification (COLMAP EstimateTwoViewGeometry,
// SURVEY.md A.3) on gfx950: LO-RANSAC over E (5-pt), F (7-pt / 8-pt), H (DLT), model selection
// and the watermark test.
//
// Mapping.  A persistent grid of wavefronts pulls image pairs from a queue; ONE WAVE owns one
// pair at a time (no workgroup barriers anywhere), and inside the wave the 64 lanes are
//   * 64 RANSAC trials for the minimal solvers (one trial per lane, tvg_math.h),
//   * 64 strided correspondences for residual scoring, support counting, normalisation sums and
//     A^T A accumulation (the fixed 64-way strided + butterfly order of the oracle's det_sum64,
//     which is exactly what a wave computes with __shfl_xor),
//   * one redundant copy of the small dense solves of the local-optimisation step (all lanes
//     run the same Jacobi on the same wave-uniform A^T A: no divergence, no broadcast).
// RANSAC is sequential by definition (the adaptive trial count depends on the best model so
// far); what is data-INdependent is the sample stream, so per 64-trial chunk the wave first draws
// the 64 samples (mt19937 + libstdc++'s Lemire uniform_int + the persistent partial Fisher-Yates
// permutation, replayed exactly), solves the 64 minimal problems in parallel, then replays
// acceptance / local optimisation / early exit in trial order.  If a RANSAC stops inside a
// chunk, the PRNG is rolled back to the position the sequential algorithm would have left it in
// (snapshot + recorded draw counts), because the next RANSAC of the pair continues the stream.
//
// FP64 everywhere, -ffp-contract=off, IEEE divide/sqrt: results are bit-identical to
// oracle/tvg_oracle.cc (inlier masks, configs, model bit patterns).
#include "amc_internal.h"
#include "tvg_math.h"

namespace amc {
using namespace tvg;

enum : int { K_F7 = 0, K_F8 = 1, K_H = 2, K_T = 3, K_E5 = 4 };
__device__ __forceinline__ int kmin_of(int kind) {
    return kind == K_F7 ? 7 : kind == K_F8 ? 8 : kind == K_H ? 4 : kind == K_T ? 1 : 5;
}

// per-wave global workspace (doubles), arrays of length Mcap each
enum : int { W_X1 = 0, W_Y1, W_X2, W_Y2, W_NX1, W_NY1, W_NX2, W_NY2, W_IX1, W_IY1, W_IX2, W_IY2,
             W_JX1, W_JY1, W_JX2, W_JY2, W_NUM_ARRAYS };
constexpr int kMaxModels = 10;
constexpr int kModelDoubles = 64 * kMaxModels * 9;

__host__ __device__ inline size_t tvg_ws_doubles(uint32_t mcap) {
    return (size_t)W_NUM_ARRAYS * mcap + kModelDoubles;
}
__host__ __device__ inline size_t tvg_ws_bytes_extra(uint32_t mcap) { return (size_t)4 * mcap; }  // 3 masks + pad

struct Wave {
    int lane;
    unsigned long long prof[5];
    // LDS
    uint32_t* mt;      // 624
    uint32_t* snap;    // 624
    uint16_t* sidx;    // 64 x 8
    uint32_t* rawcnt;  // 64
    uint16_t* perm;    // mcap
    double* lpts;      // 4 x pts_cap: the active RANSAC's correspondences, when they fit (LDS)
    uint32_t pts_cap;
    double* jacA;      // 81: A^T A / eigenvalues (LDS)
    double* jacV;      // 81: eigenvectors (LDS)
    int mti;           // uniform
    // global workspace
    double* ws;
    uint8_t* masks;    // 3 x mcap
    uint32_t mcap;
    __device__ double* arr(int a) const { return ws + (size_t)a * mcap; }
    __device__ double* models() const { return ws + (size_t)W_NUM_ARRAYS * mcap; }
};

// Global-memory hand-off between lanes of ONE wave (a lane reads what another lane of the same
// wave stored): drain this wave's stores, then keep the compiler from moving accesses across.
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// broadcast lane `src`'s double to the whole wave through the scalar unit (src is wave-uniform)
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// ---- wave reductions in the oracle's det_sum64 order ---------------------------------------------
__device__ __forceinline__ double butterfly(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- mt19937 (wave-cooperative twist, uniform extraction) ---------------------------------------
__device__ void mt_twist(uint32_t* mt, int lane) {
    for (int base = 0; base < 624; base += 64) {
        const int i = base + lane;
        uint32_t y = 0, m397 = 0;
        if (i < 624) {
            const uint32_t a = mt[i], b = mt[(i + 1) % 624];
            y = (a & 0x80000000u) | (b & 0x7fffffffu);
            m397 = mt[(i + 397) % 624];
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read before any lane writes
        if (i < 624) mt[i] = m397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}
__device__ __forceinline__ uint32_t rng_raw(Wave& w, uint32_t& nraw) {
    if (w.mti >= 624) {
        mt_twist(w.mt, w.lane);
        w.mti = 0;
    }
    ++nraw;
    return mt_temper(w.mt[w.mti++]);
}
// std::uniform_int_distribution<uint32_t>(lo, hi) on mt19937, libstdc++ >= 11 (Lemire)
__device__ uint32_t rng_uniform(Wave& w, uint32_t lo, uint32_t hi, uint32_t& nraw) {
    const uint32_t urange = hi - lo;
    if (urange == 0xFFFFFFFFu) return rng_raw(w, nraw) + lo;
    const uint32_t range = urange + 1u;
    uint64_t product = (uint64_t)rng_raw(w, nraw) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)rng_raw(w, nraw) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32) + lo;
}

// ---- residual of correspondence k under a wave-uniform model ------------------------------------
__device__ __forceinline__ double residual_k(int kind, const double* m, const double* x1,
                                             const double* y1, const double* x2, const double* y2,
                                             int k) {
    if (kind == K_H) return h_residual(m, x1[k], y1[k], x2[k], y2[k]);
    if (kind == K_T) return t_residual(m, x1[k], y1[k], x2[k], y2[k]);
    return sampson(m, x1[k], y1[k], x2[k], y2[k]);
}

struct Support {
    int cnt;
    double sum;
};
__device__ __forceinline__ bool better(const Support a, const Support b) {
    if (a.cnt > b.cnt) return true;
    return a.cnt == b.cnt && a.sum < b.sum;
}
// InlierSupportMeasurer::Evaluate.  The count comes from ballots (wave-uniform by construction);
// the residual sum is only ever consulted when the count ties or beats the best so far
// (Compare()), so its 64-way butterfly is skipped otherwise (`need_sum_from` = that count).
__device__ Support score(int kind, const double* m, const double* x1, const double* y1,
                         const double* x2, const double* y2, int M, double max_res, int lane,
                         int need_sum_from) {
    double acc = 0.0;
    int cnt = 0;
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        bool in = false;
        if (k < M) {
            const double r = residual_k(kind, m, x1, y1, x2, y2, k);
            in = r <= max_res;
            if (in) acc += r;
        }
        cnt += __popcll(__ballot(in));
    }
    Support s;
    s.cnt = cnt;
    s.sum = cnt >= need_sum_from ? butterfly(acc) : 1.7976931348623157e308;
    return s;
}

// CenterAndNormalizeImagePoints over K points (src -> dst), wave-cooperative
__device__ void center_and_normalize(const double* sx, const double* sy, int K, double* dx,
                                     double* dy, double* T, int lane) {
    double ax = 0.0, ay = 0.0;
    for (int k = lane; k < K; k += 64) { ax += sx[k]; ay += sy[k]; }
    const double cx = butterfly(ax) / (double)K;
    const double cy = butterfly(ay) / (double)K;
    double ar = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double ddx = sx[k] - cx, ddy = sy[k] - cy;
        ar += ddx * ddx + ddy * ddy;
    }
    double rms = butterfly(ar);
    rms = dsqrt(rms / (double)K);
    const double nf = dsqrt(2.0) / rms;
    T[0] = nf; T[1] = 0; T[2] = -nf * cx;
    T[3] = 0; T[4] = nf; T[5] = -nf * cy;
    T[6] = 0; T[7] = 0; T[8] = 1;
    for (int k = lane; k < K; k += 64) {
        const double p0 = sx[k], p1 = sy[k];
        const double np0 = T[0] * p0 + T[1] * p1 + T[2];
        const double np1 = T[3] * p0 + T[4] * p1 + T[5];
        const double np2 = T[6] * p0 + T[7] * p1 + T[8];
        const double inv = 1.0 / np2;
        dx[k] = np0 * inv;
        dy[k] = np1 * inv;
    }
    wave_mem_sync();
}

// design-matrix row of correspondence k for the local estimators
//   mode 0: epipolar row [x1 x2, y1 x2, x2, x1 y2, y1 y2, y2, x1, y1, 1]      (F8 / E5)
//   mode 1: homography rows; k < K -> "a" row, k >= K -> "b" row of point k-K   (H)
__device__ __forceinline__ void design_row(int mode, const double* x1, const double* y1,
                                           const double* x2, const double* y2, int K, int k,
                                           double* r) {
    if (mode == 0) {
        r[0] = x1[k] * x2[k]; r[1] = y1[k] * x2[k]; r[2] = x2[k];
        r[3] = x1[k] * y2[k]; r[4] = y1[k] * y2[k]; r[5] = y2[k];
        r[6] = x1[k]; r[7] = y1[k]; r[8] = 1.0;
    } else if (k < K) {
        const double s_0 = x1[k], s_1 = y1[k], d_0 = x2[k];
        r[0] = -s_0; r[1] = -s_1; r[2] = -1; r[3] = 0; r[4] = 0; r[5] = 0;
        r[6] = s_0 * d_0; r[7] = s_1 * d_0; r[8] = d_0;
    } else {
        const int i = k - K;
        const double s_0 = x1[i], s_1 = y1[i], d_1 = y2[i];
        r[0] = 0; r[1] = 0; r[2] = 0; r[3] = -s_0; r[4] = -s_1; r[5] = -1;
        r[6] = s_0 * d_1; r[7] = s_1 * d_1; r[8] = d_1;
    }
}
// A^T A (9 x 9, symmetric) over `rows` design rows, every entry in det_sum64 order: each lane keeps
// the 45 partial sums of its strided rows (one pass over the data), then 45 butterflies
__device__ void accumulate_ata(int mode, const double* x1, const double* y1, const double* x2,
                               const double* y2, int K, int rows, double* ata, int lane) {
    double acc[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) acc[e] = 0.0;
    for (int k = lane; k < rows; k += 64) {
        double r[9];
        design_row(mode, x1, y1, x2, y2, K, k, r);
        int e = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = i; j < 9; ++j) acc[e++] += r[i] * r[j];
    }
    int e = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = i; j < 9; ++j) {
            const double sres = butterfly(acc[e++]);
            ata[i * 9 + j] = sres;
            ata[j * 9 + i] = sres;
        }
}

// Cyclic Jacobi on a symmetric n x n matrix held in LDS, the whole wave cooperating: the
// rotation parameters are wave-uniform; lanes 0..n-1 update the n entries of the two columns,
// then of the two rows, lanes 32..32+n-1 the eigenvector columns.  Every element sees exactly the
// arithmetic of the scalar jacobi_eigen (tvg_math.h) in the same order, so results are
// bit-identical; only the memory (LDS instead of scratch) and the parallelism differ.
__device__ void jacobi_eigen_wave(int n, double* A, double* V, int lane) {
    for (int i = lane; i < n * n; i += 64) V[i] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    double total = 0.0;
    for (int i = 0; i < n * n; ++i) total += A[i] * A[i];
    const double tol = total * 1e-32;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (!(off > tol)) break;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (dabs(theta) + dsqrt(theta * theta + 1.0));
                const double c = 1.0 / dsqrt(t * t + 1.0);
                const double s = t * c;
                const int k = lane & 31;
                const bool colA = lane < n, colV = lane >= 32 && k < n;
                double* Mx = colV ? V : A;
                double xp = 0.0, xq = 0.0;
                if (colA || colV) { xp = Mx[k * n + p]; xq = Mx[k * n + q]; }
                __builtin_amdgcn_wave_barrier();
                if (colA || colV) {
                    Mx[k * n + p] = c * xp - s * xq;
                    Mx[k * n + q] = s * xp + c * xq;
                }
                __builtin_amdgcn_wave_barrier();
                if (colA) { xp = A[p * n + k]; xq = A[q * n + k]; }
                __builtin_amdgcn_wave_barrier();
                if (colA) {
                    A[p * n + k] = c * xp - s * xq;
                    A[q * n + k] = s * xp + c * xq;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}
// eigenvector of the smallest eigenvalue after jacobi_eigen_wave (first minimum, like the oracle)
__device__ void smallest_eigvec9_wave(const double* A, const double* V, double* x) {
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i * 9 + i] < A[best * 9 + best]) best = i;
    for (int i = 0; i < 9; ++i) x[i] = V[i * 9 + best];
}

// local estimator on the K inlier correspondences in the I arrays -> models (uniform), count
__device__ int local_estimate(Wave& w, int kind, int K, double* models) {
    const int lane = w.lane;
    const double *ix1 = w.arr(W_IX1), *iy1 = w.arr(W_IY1), *ix2 = w.arr(W_IX2), *iy2 = w.arr(W_IY2);
    if (kind == K_T) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int k = lane; k < K; k += 64) { a += ix1[k]; b += iy1[k]; c += ix2[k]; d += iy2[k]; }
        const double sx = butterfly(a) / (double)K, sy = butterfly(b) / (double)K;
        const double dx = butterfly(c) / (double)K, dy = butterfly(d) / (double)K;
        for (int i = 0; i < 9; ++i) models[i] = 0.0;
        models[0] = dx - sx;
        models[1] = dy - sy;
        return 1;
    }
    if (kind == K_E5) {
        if (K == 5) {
            double a[5], b[5], c[5], d[5];
            for (int i = 0; i < 5; ++i) { a[i] = ix1[i]; b[i] = iy1[i]; c[i] = ix2[i]; d[i] = iy2[i]; }
            return estimate_e5_minimal(a, b, c, d, models);
        }
        accumulate_ata(0, ix1, iy1, ix2, iy2, K, K, w.jacA, lane);
        __builtin_amdgcn_wave_barrier();
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double nsp[4 * 9];
        e5_nullspace_from_eig(w.jacA, w.jacV, nsp);
        return e5_from_nullspace(nsp, models);
    }
    if (kind == K_H && K == 4) {
        double a[4], b[4], c[4], d[4];
        for (int i = 0; i < 4; ++i) { a[i] = ix1[i]; b[i] = iy1[i]; c[i] = ix2[i]; d[i] = iy2[i]; }
        estimate_h4(a, b, c, d, models);
        return 1;
    }
    double T1[9], T2[9];
    double* ata = w.jacA;
    double *jx1 = w.arr(W_JX1), *jy1 = w.arr(W_JY1), *jx2 = w.arr(W_JX2), *jy2 = w.arr(W_JY2);
    center_and_normalize(ix1, iy1, K, jx1, jy1, T1, lane);
    center_and_normalize(ix2, iy2, K, jx2, jy2, T2, lane);
    if (kind == K_F8) {
        accumulate_ata(0, jx1, jy1, jx2, jy2, K, K, ata, lane);
        __builtin_amdgcn_wave_barrier();
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double f[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, f);
        f8_from_vec(f, T1, T2, models);
    } else {
        accumulate_ata(1, jx1, jy1, jx2, jy2, K, 2 * K, ata, lane);
        __builtin_amdgcn_wave_barrier();
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double h[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, h);
        h_denormalize(h, T1, T2, models);
    }
    return 1;
}

struct Report {
    bool success;
    int num_trials;
    Support support;
    double model[9];
};

struct RansacCfg {
    int est, local_est;
    double max_res;          // max_error^2
    int max_trials;          // already clamped as the RANSAC constructor does
    int min_trials;
    const uint32_t* dyn_tab; // dyn_max_num_trials by num_inliers (host libm), or nullptr
};

// ordered compaction of the inliers of `model` (kind) into the I arrays; returns K
__device__ int extract_inliers(Wave& w, int kind, const double* model, const double* x1,
                               const double* y1, const double* x2, const double* y2, int M,
                               double max_res) {
    double *ix1 = w.arr(W_IX1), *iy1 = w.arr(W_IY1), *ix2 = w.arr(W_IX2), *iy2 = w.arr(W_IY2);
    int base = 0;
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + w.lane;
        bool in = false;
        if (k < M) in = residual_k(kind, model, x1, y1, x2, y2, k) <= max_res;
        const unsigned long long bal = __ballot(in);
        if (in) {
            const int pos = base + __popcll(bal & ((1ull << w.lane) - 1ull));
            ix1[pos] = x1[k]; iy1[pos] = y1[k]; ix2[pos] = x2[k]; iy2[pos] = y2[k];
        }
        base += __popcll(bal);
    }
    wave_mem_sync();
    return base;
}

// LORANSAC<est, local_est>::Estimate over the M correspondences (x1,y1)->(x2,y2); mask: M bytes
__device__ Report lo_ransac(Wave& w, const RansacCfg& cfg, const double* x1, const double* y1,
                            const double* x2, const double* y2, int M, uint8_t* mask) {
    const int lane = w.lane;
    const int kMin = kmin_of(cfg.est), kLocalMin = kmin_of(cfg.local_est);
    Report rep;
    rep.success = false;
    rep.num_trials = 0;
    rep.support.cnt = 0;
    rep.support.sum = 1.7976931348623157e308;  // numeric_limits<double>::max()
    for (int i = 0; i < 9; ++i) rep.model[i] = 0.0;
    if (M < kMin) return rep;

    Support best = rep.support;
    double best_model[9];
    for (int i = 0; i < 9; ++i) best_model[i] = 0.0;
    bool best_is_local = false;
    uint32_t dyn_max = (uint32_t)cfg.max_trials;

    // correspondences into LDS when they fit: every scoring pass and every sample gather reads them
    if ((uint32_t)M <= w.pts_cap) {
        double *l0 = w.lpts, *l1 = l0 + w.pts_cap, *l2 = l1 + w.pts_cap, *l3 = l2 + w.pts_cap;
        for (int k = lane; k < M; k += 64) { l0[k] = x1[k]; l1[k] = y1[k]; l2[k] = x2[k]; l3[k] = y2[k]; }
        __builtin_amdgcn_wave_barrier();
        x1 = l0; y1 = l1; x2 = l2; y2 = l3;
    }

    // sampler.Initialize(M).  The first kMin entries of the persistent permutation are touched by
    // every draw: they live in (wave-uniform) registers, the rest in LDS.
    for (int k = lane; k < M; k += 64) w.perm[k] = (uint16_t)k;
    uint32_t pr[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pr[i] = (uint32_t)i;
    __builtin_amdgcn_wave_barrier();

    double* models = w.models();
    bool aborted = false;
    int abort_trial = -1;
    for (int chunk = 0; chunk < cfg.max_trials && !aborted; chunk += 64) {
        const int nT = min(64, cfg.max_trials - chunk);
        // ---- snapshot the generator, draw the chunk's samples (wave-uniform, sequential) ----
        for (int i = lane; i < 624; i += 64) w.snap[i] = w.mt[i];
        const int snap_mti = w.mti;
        __builtin_amdgcn_wave_barrier();
        uint32_t nraw = 0;
        const uint32_t last = (uint32_t)(M - 1);
        unsigned long long tp0 = __builtin_readcyclecounter();
        for (int t = 0; t < nT; ++t) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (i < kMin) {
                    const uint32_t j = rng_uniform(w, (uint32_t)i, last, nraw);
                    // swap(perm[i], perm[j])
                    if (j < (uint32_t)kMin) {
                        uint32_t vj = pr[0];
#pragma unroll
                        for (int q = 1; q < 7; ++q) vj = (j == (uint32_t)q) ? pr[q] : vj;
                        const uint32_t vi = pr[i];
#pragma unroll
                        for (int q = 0; q < 7; ++q) pr[q] = (j == (uint32_t)q) ? vi : pr[q];
                        pr[i] = vj;
                    } else {
                        const uint32_t vj = w.perm[j];
                        w.perm[j] = (uint16_t)pr[i];
                        pr[i] = vj;
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i) w.sidx[t * 8 + i] = (uint16_t)pr[i];
                w.rawcnt[t] = nraw;
            }
        }
        __builtin_amdgcn_wave_barrier();
        { const unsigned long long tp1 = __builtin_readcyclecounter(); w.prof[0] += tp1 - tp0; tp0 = tp1; }
        // ---- 64 minimal problems, one per lane.  F / H / T models stay in the solving lane's
        //      registers (slot i = i-th root; `valid` marks the ones the estimator returned) and are
        //      broadcast with __shfl during the replay; E models (up to 10) go through global memory.
        int nmod = 0;
        unsigned valid = 0;
        double mym[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) mym[i] = 0.0;
        if (lane < nT) {
            if (cfg.est == K_F7) {
                double sx1[7], sy1[7], sx2[7], sy2[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int sI = w.sidx[lane * 8 + i];
                    sx1[i] = x1[sI]; sy1[i] = y1[sI]; sx2[i] = x2[sI]; sy2[i] = y2[sI];
                }
                nmod = estimate_f7(sx1, sy1, sx2, sy2, mym);
                valid = (1u << nmod) - 1u;
            } else if (cfg.est == K_H) {
                double sx1[4], sy1[4], sx2[4], sy2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sI = w.sidx[lane * 8 + i];
                    sx1[i] = x1[sI]; sy1[i] = y1[sI]; sx2[i] = x2[sI]; sy2[i] = y2[sI];
                }
                estimate_h4(sx1, sy1, sx2, sy2, mym);
                nmod = 1;
                valid = 1u;
            } else if (cfg.est == K_E5) {
                double sx1[5], sy1[5], sx2[5], sy2[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int sI = w.sidx[lane * 8 + i];
                    sx1[i] = x1[sI]; sy1[i] = y1[sI]; sx2[i] = x2[sI]; sy2[i] = y2[sI];
                }
                nmod = estimate_e5_minimal(sx1, sy1, sx2, sy2, models + (size_t)lane * kMaxModels * 9);
            } else {  // K_T: model = dst - src of the single sample
                const int sI = w.sidx[lane * 8];
                mym[0] = x2[sI] - x1[sI];
                mym[1] = y2[sI] - y1[sI];
                nmod = 1;
                valid = 1u;
            }
        }
        wave_mem_sync();
        { const unsigned long long tp1 = __builtin_readcyclecounter(); w.prof[1] += tp1 - tp0; tp0 = tp1; }
        // ---- replay in trial order ------------------------------------------------------------
        for (int t = 0; t < nT && !aborted; ++t) {
            const int trial = chunk + t;
            const int n = __shfl(nmod, t);
            for (int m = 0; m < n; ++m) {
                double sm[9];
                if (cfg.est == K_E5) {
                    const double* src = models + ((size_t)t * kMaxModels + m) * 9;
                    for (int i = 0; i < 9; ++i) sm[i] = src[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const double mine = m == 0 ? mym[i] : (m == 1 ? mym[9 + i] : mym[18 + i]);
                        sm[i] = readlane_f64(mine, t);
                    }
                }
                const Support sup = score(cfg.est, sm, x1, y1, x2, y2, M, cfg.max_res, lane, best.cnt);
                if (better(sup, best)) {
                    const unsigned long long tl0 = __builtin_readcyclecounter();
                    best = sup;
                    for (int i = 0; i < 9; ++i) best_model[i] = sm[i];
                    best_is_local = false;
                    if (sup.cnt > kMin && sup.cnt >= kLocalMin) {
                        // recursive local optimisation: inliers of the sample model first, then of
                        // the improved local model (COLMAP swaps residual vectors to the same effect)
                        int cur_kind = cfg.est;
                        double cur[9];
                        for (int i = 0; i < 9; ++i) cur[i] = sm[i];
                        for (int lt = 0; lt < 10; ++lt) {
                            const int K = extract_inliers(w, cur_kind, cur, x1, y1, x2, y2, M, cfg.max_res);
                            double lm[kMaxModels * 9];
                            const int nl = local_estimate(w, cfg.local_est, K, lm);
                            const int prev = best.cnt;
                            for (int q = 0; q < nl; ++q) {
                                const Support ls = score(cfg.local_est, lm + 9 * q, x1, y1, x2, y2, M,
                                                         cfg.max_res, lane, best.cnt);
                                if (better(ls, best)) {
                                    best = ls;
                                    for (int i = 0; i < 9; ++i) best_model[i] = lm[9 * q + i];
                                    best_is_local = true;
                                }
                            }
                            if (best.cnt <= prev) break;
                            cur_kind = cfg.local_est;
                            for (int i = 0; i < 9; ++i) cur[i] = best_model[i];
                        }
                    }
                    dyn_max = cfg.dyn_tab ? cfg.dyn_tab[best.cnt] : 0xFFFFFFFFu;
                    w.prof[3] += __builtin_readcyclecounter() - tl0;
                }
                if ((uint32_t)trial >= dyn_max && trial >= cfg.min_trials) {
                    aborted = true;
                    abort_trial = trial;
                    break;
                }
            }
        }
        { const unsigned long long tp1 = __builtin_readcyclecounter(); w.prof[2] += tp1 - tp0; }
        if (aborted) {
            // roll the generator back to where the sequential algorithm stopped drawing
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < 624; i += 64) w.mt[i] = w.snap[i];
            w.mti = snap_mti;
            __builtin_amdgcn_wave_barrier();
            const uint32_t consumed = w.rawcnt[abort_trial - chunk];
            uint32_t dummy = 0;
            for (uint32_t i = 0; i < consumed; ++i) (void)rng_raw(w, dummy);
        }
    }
    // report.num_trials exactly as the for/abort dance of loransac.h leaves it
    rep.num_trials = aborted ? ((abort_trial + 1 < cfg.max_trials) ? abort_trial + 2 : abort_trial + 1)
                             : cfg.max_trials;
    rep.support = best;
    for (int i = 0; i < 9; ++i) rep.model[i] = best_model[i];
    if (best.cnt < kMin) return rep;
    rep.success = true;
    const int fk = best_is_local ? cfg.local_est : cfg.est;
    for (int k = lane; k < M; k += 64)
        mask[k] = residual_k(fk, rep.model, x1, y1, x2, y2, k) <= cfg.max_res ? 1 : 0;
    wave_mem_sync();
    return rep;
}

__device__ __forceinline__ bool in_bbox(double x, double y, double minx, double maxx, double miny,
                                        double maxy) {
    return x >= minx && x <= maxx && y >= miny && y <= maxy;
}

// EstimateTwoViewGeometry for pair q, by one wave
__device__ __noinline__ void process_pair(Wave& w, uint32_t q, const TvgImage* __restrict__ imgs,
                                          const TvgPair* __restrict__ pairs,
                                          const uint32_t* __restrict__ matches,
                                          const uint32_t* __restrict__ trial_tabs,
                                          const uint32_t* __restrict__ mt_init, const TvgParams& P,
                                          TvgOut* __restrict__ out, uint8_t* __restrict__ out_mask) {
    const int lane = w.lane;
    const uint32_t mcap = w.mcap;
    for (int i = 0; i < 5; ++i) w.prof[i] = 0;
    const unsigned long long tstart = __builtin_readcyclecounter();
    const TvgPair pr = pairs[q];
    const TvgImage im1 = imgs[pr.slot1], im2 = imgs[pr.slot2];
    const int M = (int)pr.M;
    amc_tvg g;
    g.config = AMC_TVG_UNDEFINED;
    g.num_inliers = 0;
    for (int i = 0; i < 9; ++i) { g.E[i] = 0; g.F[i] = 0; g.H[i] = 0; }
    for (int i = 0; i < 4; ++i) g.num_trials[i] = 0;
    for (int i = 0; i < 3; ++i) g.model_inliers[i] = 0;
    uint8_t* omask = out_mask + pr.mask_off;
    for (int k = lane; k < M; k += 64) omask[k] = 0;

    if (M < P.min_num_inliers) {
        g.config = AMC_TVG_DEGENERATE;
        if (lane == 0) out[q].g = g;
        return;
    }
    // ---- matched points (FeatureKeypointsToPointsVector: float -> double) ------------------
    double *X1 = w.arr(W_X1), *Y1 = w.arr(W_Y1), *X2 = w.arr(W_X2), *Y2 = w.arr(W_Y2);
    const uint32_t* mm = matches + 2 * pr.match_off;
    for (int k = lane; k < M; k += 64) {
        const uint32_t i1 = mm[2 * k], i2 = mm[2 * k + 1];
        X1[k] = (double)im1.kp[2 * (size_t)i1];
        Y1[k] = (double)im1.kp[2 * (size_t)i1 + 1];
        X2[k] = (double)im2.kp[2 * (size_t)i2];
        Y2[k] = (double)im2.kp[2 * (size_t)i2 + 1];
    }
    // ---- SetPRNGSeed(seed): generator state as std::mt19937(seed) leaves it ---------------
    for (int i = lane; i < 624; i += 64) w.mt[i] = mt_init[i];
    w.mti = 624;
    wave_mem_sync();

    const bool calibrated = !P.force_H_use && im1.cam.has_prior && im2.cam.has_prior;
    uint8_t *maskE = w.masks, *maskF = w.masks + mcap, *maskH = w.masks + 2 * (size_t)mcap;
    Report E_rep, F_rep, H_rep;
    E_rep.success = F_rep.success = H_rep.success = false;
    E_rep.support.cnt = F_rep.support.cnt = H_rep.support.cnt = 0;
    E_rep.num_trials = F_rep.num_trials = 0;
    for (int i = 0; i < 9; ++i) { E_rep.model[i] = 0; F_rep.model[i] = 0; }

    RansacCfg cfg;
    cfg.min_trials = P.min_num_trials;
    if (calibrated) {
        double *N1x = w.arr(W_NX1), *N1y = w.arr(W_NY1), *N2x = w.arr(W_NX2), *N2y = w.arr(W_NY2);
        const CameraDev c1 = im1.cam, c2 = im2.cam;
        for (int k = lane; k < M; k += 64) {
            if (c1.model_id == AMC_CAM_SIMPLE_PINHOLE) {
                N1x[k] = (X1[k] - c1.params[1]) / c1.params[0];
                N1y[k] = (Y1[k] - c1.params[2]) / c1.params[0];
            } else {
                N1x[k] = (X1[k] - c1.params[2]) / c1.params[0];
                N1y[k] = (Y1[k] - c1.params[3]) / c1.params[1];
            }
            if (c2.model_id == AMC_CAM_SIMPLE_PINHOLE) {
                N2x[k] = (X2[k] - c2.params[1]) / c2.params[0];
                N2y[k] = (Y2[k] - c2.params[2]) / c2.params[0];
            } else {
                N2x[k] = (X2[k] - c2.params[2]) / c2.params[0];
                N2y[k] = (Y2[k] - c2.params[3]) / c2.params[1];
            }
        }
        wave_mem_sync();
        const double f1 = c1.model_id == AMC_CAM_SIMPLE_PINHOLE ? c1.params[0] : (c1.params[0] + c1.params[1]) / 2.0;
        const double f2 = c2.model_id == AMC_CAM_SIMPLE_PINHOLE ? c2.params[0] : (c2.params[0] + c2.params[1]) / 2.0;
        const double e_err = (P.max_error / f1 + P.max_error / f2) / 2;
        cfg.est = K_E5; cfg.local_est = K_E5;
        cfg.max_res = e_err * e_err;
        cfg.max_trials = P.max_trials[0];
        cfg.dyn_tab = trial_tabs + pr.tab_off[0];
        E_rep = lo_ransac(w, cfg, N1x, N1y, N2x, N2y, M, maskE);
        for (int i = 0; i < 9; ++i) g.E[i] = E_rep.model[i];
        g.num_trials[0] = E_rep.num_trials;
        g.model_inliers[0] = E_rep.support.cnt;
    }
    if (!P.force_H_use) {
        cfg.est = K_F7; cfg.local_est = K_F8;
        cfg.max_res = P.max_error * P.max_error;
        cfg.max_trials = P.max_trials[1];
        cfg.dyn_tab = trial_tabs + pr.tab_off[1];
        F_rep = lo_ransac(w, cfg, X1, Y1, X2, Y2, M, maskF);
        for (int i = 0; i < 9; ++i) g.F[i] = F_rep.model[i];
        g.num_trials[1] = F_rep.num_trials;
        g.model_inliers[1] = F_rep.support.cnt;
    }
    cfg.est = K_H; cfg.local_est = K_H;
    cfg.max_res = P.max_error * P.max_error;
    cfg.max_trials = P.max_trials[2];
    cfg.dyn_tab = trial_tabs + pr.tab_off[2];
    H_rep = lo_ransac(w, cfg, X1, Y1, X2, Y2, M, maskH);
    for (int i = 0; i < 9; ++i) g.H[i] = H_rep.model[i];
    g.num_trials[2] = H_rep.num_trials;
    g.model_inliers[2] = H_rep.support.cnt;

    // ---- model selection (two_view_geometry.cc), wave-uniform --------------------------------
    const int minI = P.min_num_inliers;
    const int Ei = E_rep.support.cnt, Fi = F_rep.support.cnt, Hi = H_rep.support.cnt;
    const uint8_t* best_mask = nullptr;
    bool best_ok = false;  // best_mask non-null and non-empty (its RANSAC succeeded)
    int num_inliers = 0;
    bool done = false;
    if (P.force_H_use) {
        if (!H_rep.success || Hi < minI) { g.config = AMC_TVG_DEGENERATE; done = true; }
        else { g.config = AMC_TVG_PLANAR_OR_PANORAMIC; best_mask = maskH; best_ok = true; num_inliers = Hi; }
    } else if (calibrated) {
        if ((!E_rep.success && !F_rep.success && !H_rep.success) || (Ei < minI && Fi < minI && Hi < minI)) {
            g.config = AMC_TVG_DEGENERATE; done = true;
        } else {
            const double E_F = (double)Ei / (double)Fi, H_F = (double)Hi / (double)Fi, H_E = (double)Hi / (double)Ei;
            if (E_rep.success && E_F > P.min_E_F_inlier_ratio && Ei >= minI) {
                if (Ei >= Fi) { num_inliers = Ei; best_mask = maskE; best_ok = E_rep.success; }
                else { num_inliers = Fi; best_mask = maskF; best_ok = F_rep.success; }
                if (H_E > P.max_H_inlier_ratio) {
                    g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                    if (Hi > num_inliers) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
                } else g.config = AMC_TVG_CALIBRATED;
            } else if (F_rep.success && Fi >= minI) {
                num_inliers = Fi; best_mask = maskF; best_ok = true;
                if (H_F > P.max_H_inlier_ratio) {
                    g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                    if (Hi > num_inliers) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
                } else g.config = AMC_TVG_UNCALIBRATED;
            } else if (H_rep.success && Hi >= minI) {
                num_inliers = Hi; best_mask = maskH; best_ok = true; g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
            } else { g.config = AMC_TVG_DEGENERATE; done = true; }
        }
    } else {
        if ((!F_rep.success && !H_rep.success) || (Fi < minI && Hi < minI)) {
            g.config = AMC_TVG_DEGENERATE; done = true;
        } else {
            const double H_F = (double)Hi / (double)Fi;
            best_mask = maskF; best_ok = F_rep.success; num_inliers = Fi;
            if (H_F > P.max_H_inlier_ratio) {
                g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                if (Hi >= Fi) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
            } else g.config = AMC_TVG_UNCALIBRATED;
        }
    }
    if (!done) {
        if (best_ok) {
            g.num_inliers = num_inliers;
            for (int k = lane; k < M; k += 64) omask[k] = best_mask[k];
        } else {
            g.num_inliers = 0;
        }
        // ---- DetectWatermark -----------------------------------------------------------------
        if (P.detect_watermark && best_ok) {
            const CameraDev c1 = im1.cam, c2 = im2.cam;
            const double diagonal1 = dsqrt((double)(c1.width * c1.width + c1.height * c1.height));
            const double diagonal2 = dsqrt((double)(c2.width * c2.width + c2.height * c2.height));
            const double minx1 = P.watermark_border_size * diagonal1, miny1 = minx1;
            const double maxx1 = (double)c1.width - minx1, maxy1 = (double)c1.height - miny1;
            const double minx2 = P.watermark_border_size * diagonal2, miny2 = minx2;
            const double maxx2 = (double)c2.width - minx2, maxy2 = (double)c2.height - miny2;
            double *ix1 = w.arr(W_NX1), *iy1 = w.arr(W_NY1), *ix2 = w.arr(W_NX2), *iy2 = w.arr(W_NY2);
            int basep = 0, border = 0;
            for (int k0 = 0; k0 < M; k0 += 64) {
                const int k = k0 + lane;
                const bool in = k < M && best_mask[k];
                const unsigned long long bal = __ballot(in);
                if (in) {
                    const int pos = basep + __popcll(bal & ((1ull << lane) - 1ull));
                    ix1[pos] = X1[k]; iy1[pos] = Y1[k]; ix2[pos] = X2[k]; iy2[pos] = Y2[k];
                    if (!in_bbox(X1[k], Y1[k], minx1, maxx1, miny1, maxy1) &&
                        !in_bbox(X2[k], Y2[k], minx2, maxx2, miny2, maxy2))
                        ++border;
                }
                basep += __popcll(bal);
            }
            wave_mem_sync();
            border = wave_sum_int(border);
            const double ratio = (double)border / (double)num_inliers;
            if (!(ratio < P.watermark_min_inlier_ratio)) {
                cfg.est = K_T; cfg.local_est = K_T;
                cfg.max_res = P.max_error * P.max_error;
                cfg.max_trials = P.max_trials[3];
                cfg.dyn_tab = nullptr;  // never consulted: max_trials[3] <= min_num_trials (host check)
                const Report T_rep = lo_ransac(w, cfg, ix1, iy1, ix2, iy2, num_inliers, w.masks + 3 * (size_t)mcap);
                g.num_trials[3] = T_rep.num_trials;
                const double inlier_ratio = (double)T_rep.support.cnt / (double)num_inliers;
                if (inlier_ratio >= P.watermark_min_inlier_ratio) g.config = AMC_TVG_WATERMARK;
            }
        }
    }
    if (lane == 0) {
        out[q].g = g;
        w.prof[4] = __builtin_readcyclecounter() - tstart;
        for (int i = 0; i < 5; ++i) out[q].prof[i] = w.prof[i];
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kTvgWavesPerSimd, kTvgWavesPerSimd))) void tvg_kernel(
    const TvgImage* __restrict__ imgs, const TvgPair* __restrict__ pairs, uint32_t npairs,
    const uint32_t* __restrict__ matches, const uint32_t* __restrict__ trial_tabs,
    const uint32_t* __restrict__ mt_init, TvgParams P, double* __restrict__ ws_all,
    uint8_t* __restrict__ mask_ws_all, uint32_t mcap, uint32_t pts_cap, uint32_t* __restrict__ queue_head,
    TvgOut* __restrict__ out, uint8_t* __restrict__ out_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const size_t lds_per_wave = (size_t)162 * 8 + (size_t)4 * pts_cap * 8 + (size_t)(624 + 624 + 64) * 4 +
                                64 * 8 * 2 + (size_t)((mcap + 7) / 8 * 8) * 2;
    char* base = smem + (size_t)wid * ((lds_per_wave + 15) / 16 * 16);
    Wave w;
    w.lane = lane;
    w.jacA = reinterpret_cast<double*>(base);
    w.jacV = w.jacA + 81;
    w.lpts = w.jacA + 162;
    w.pts_cap = pts_cap;
    w.mt = reinterpret_cast<uint32_t*>(base + (162 + (size_t)4 * pts_cap) * 8);
    w.snap = w.mt + 624;
    w.rawcnt = w.snap + 624;
    w.sidx = reinterpret_cast<uint16_t*>(w.rawcnt + 64);
    w.perm = w.sidx + 64 * 8;
    w.mcap = mcap;
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wid;
    w.ws = ws_all + gw * tvg_ws_doubles(mcap);
    w.masks = mask_ws_all + gw * tvg_ws_bytes_extra(mcap);

    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(queue_head, 1u);
        q = __shfl(q, 0);
        if (q >= npairs) break;
        process_pair(w, q, imgs, pairs, matches, trial_tabs, mt_init, P, out, out_mask);
    }
}

size_t tvg_ws_doubles_host(uint32_t mcap) { return tvg_ws_doubles(mcap); }
size_t tvg_ws_mask_bytes_host(uint32_t mcap) { return tvg_ws_bytes_extra(mcap); }

size_t tvg_lds_bytes(uint32_t mcap, uint32_t pts_cap, int waves) {
    const size_t per = (size_t)162 * 8 + (size_t)4 * pts_cap * 8 + (size_t)(624 + 624 + 64) * 4 + 64 * 8 * 2 +
                       (size_t)((mcap + 7) / 8 * 8) * 2;
    return (size_t)waves * ((per + 15) / 16 * 16);
}
// how many correspondences of the active RANSAC fit in LDS next to everything else (4 waves/block)
uint32_t tvg_pts_cap(uint32_t mcap) {
    const size_t budget = 160 * 1024 / (4 * kTvgWavesPerSimd);
    const size_t other = tvg_lds_bytes(mcap, 0, 1) + 64;
    if (other >= budget) return 0;
    const size_t cap = (budget - other) / 32 / 64 * 64;
    return (uint32_t)(cap < mcap ? cap : mcap);
}

hipError_t launch_tvg(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs,
                      const uint32_t* matches, const uint32_t* trial_tabs, const uint32_t* mt_init,
                      const TvgParams& P, double* ws, uint8_t* mask_ws, uint32_t mcap,
                      uint32_t num_waves, uint32_t* queue_head, TvgOut* out, uint8_t* out_mask,
                      hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    const int waves_per_block = 4;
    const uint32_t blocks = (num_waves + waves_per_block - 1) / waves_per_block;
    const uint32_t pts_cap = tvg_pts_cap(mcap);
    const size_t lds = tvg_lds_bytes(mcap, pts_cap, waves_per_block);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tvg_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(queue_head, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tvg_kernel, dim3(blocks), dim3(64 * waves_per_block), lds, s, imgs, pairs,
                       npairs, matches, trial_tabs, mt_init, P, ws, mask_ws, mcap, pts_cap, queue_head, out,
                       out_mask);
    return hipGetLastError();
}

}  // namespace amc
