// tvg.hip — batched two-view geometric verification (COLMAP EstimateTwoViewGeometry,
// SURVEY.md A.3) on gfx950: LO-RANSAC over E (5-pt), F (7-pt / 8-pt), H (DLT), model selection
// and the watermark test.
//
// Mapping.  A persistent grid of wavefronts pulls image pairs from a queue; ONE WAVE owns one
// pair at a time (no workgroup barriers anywhere; 2 waves per SIMD), and inside the wave the 64
// lanes are
//   * 64 RANSAC trials for the minimal solvers (one trial per lane, tvg_math.h),
//   * 64 strided correspondences for inlier counting, residual scoring, normalisation sums and
//     A^T A accumulation (the fixed 64-way strided + butterfly order of the oracle's det_sum64,
//     which is exactly what a wave computes with __shfl_xor),
//   * cooperating workers on the single dense problems of the local-optimisation step: the
//     disjoint rotations of a Jacobi round, the sign-change brackets of a root-finding level.
// RANSAC is sequential by definition (the adaptive trial count depends on the best model so
// far); what is data-INdependent is the sample stream.  So per 64-trial chunk the wave draws the
// 64 samples (mt19937 + libstdc++'s Lemire uniform_int + the persistent partial Fisher-Yates
// permutation, reproduced exactly; sample_chunk), solves the 64 minimal problems in parallel
// (solve_chunk), counts the inliers of every resulting model (count_chunk), and then replays only
// the trials that can matter - a model whose count reaches the best so far, or the first trial at
// the adaptive limit - in trial order, re-scoring them in full (lo_ransac).  If a RANSAC stops
// inside a chunk, the PRNG is rolled back to the position the sequential algorithm would have left
// it in (snapshot + recorded draw counts), because the next RANSAC of the pair continues the stream.
// Every phase is its own __noinline__ function with by-value, scalarised arguments: they are
// register-allocated separately, so the 128-VGPR budget spills only inside the lane-local solvers.
//
// FP64 everywhere, -ffp-contract=off, IEEE divide/sqrt: results are bit-identical to
// oracle/tvg_oracle.cc (inlier masks, configs, model bit patterns).
#include <algorithm>
#include <cstdio>

#include "amc_internal.h"
#include "camera_math.h"
#include "tvg_math.h"

namespace amc {
using namespace tvg;

enum : int { K_F7 = 0, K_F8 = 1, K_H = 2, K_T = 3, K_E5 = 4 };
__device__ __forceinline__ int kmin_of(int kind) {
    return kind == K_F7 ? 7 : kind == K_F8 ? 8 : kind == K_H ? 4 : kind == K_T ? 1 : 5;
}

// per-wave global workspace (doubles), arrays of length Mcap each
enum : int { W_X1 = 0, W_Y1, W_X2, W_Y2, W_NX1, W_NY1, W_NX2, W_NY2, W_NUM_ARRAYS };
constexpr int kMaxModels = 10;
constexpr int kModelDoubles = 64 * kMaxModels * 9;

__host__ __device__ inline size_t tvg_ws_doubles(uint32_t mcap) {
    return (size_t)W_NUM_ARRAYS * mcap + kModelDoubles + 312;  // + 624-word generator snapshot
}
// LDS bytes of one wave: jacA + jacV | points | mt, rawcnt | sidx | perm | inl
// (the generator snapshot of a chunk, read back only on an abort, lives in the global workspace)
__host__ __device__ inline size_t tvg_lds_per_wave(uint32_t mcap, uint32_t pts_cap) {
    const size_t per = (size_t)162 * 8 + (size_t)4 * pts_cap * 8 + (size_t)(624 + 64) * 4 + 64 * 8 * 2 +
                       (size_t)((mcap + 7) / 8 * 8) * 2 * 2;
    return (per + 15) / 16 * 16;
}
__host__ __device__ inline size_t tvg_ws_bytes_extra(uint32_t mcap) { return (size_t)4 * mcap; }  // 3 masks + pad

// LDS objects are addressed through address-space-3 pointers so that every access is a ds_*
// instruction (a generic pointer makes the compiler emit flat_* loads, which take the
// vector-memory path and cost several hundred cycles each at this occupancy).
#define AMC_LDS __attribute__((address_space(3)))
typedef AMC_LDS double lds_f64;
typedef AMC_LDS uint32_t lds_u32;
typedef AMC_LDS uint16_t lds_u16;

// Algorithmic work of a pair, counted as the sequential algorithm does it (TvgOut::work): what COLMAP's loops
// evaluate - every model of every trial up to the stopping trial against all M correspondences, every local model
// against all M, one final residual pass per successful RANSAC - not what this kernel skips by early exit.
enum : int { WK_SAMPSON = 0, WK_HRES, WK_TRES, WK_E5MIN, WK_F7MIN, WK_H4MIN, WK_LO_E5, WK_LO_F8, WK_LO_H, WK_LO_POINTS,
              WK_TRIALS, WK_COUNT };
__device__ __forceinline__ int wk_residual_slot(int kind) { return kind == K_H ? WK_HRES : (kind == K_T ? WK_TRES : WK_SAMPSON); }

struct Wave {
    int lane;
    unsigned long long prof[8];
    unsigned long long* work;  // the pair's TvgOut::work (global memory, lane 0 adds to it: a few times per 64 trials)
    // LDS
    lds_u32* mt;      // 624
    uint32_t* snap;   // 624, global workspace: written once per chunk, read on abort / sampler fallback
    lds_u16* sidx;    // 64 x 8
    lds_u32* rawcnt;  // 64
    lds_u16* perm;    // mcap
    lds_u16* inl;     // mcap: ordered inlier index list of the local-optimisation step
    lds_f64* lpts;    // 4 x pts_cap: the active RANSAC's correspondences, when they fit
    uint32_t pts_cap;
    lds_f64* jacA;    // 81: A^T A / eigenvalues
    lds_f64* jacV;    // 81: eigenvectors
    int mti;          // uniform
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
// LDS hand-off inside the wave: LDS operations of a wave complete in order, the barrier only
// stops the compiler from reordering across it
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
__device__ __noinline__ void mt_twist(lds_u32* mt, int lane) {
    for (int base = 0; base < 624; base += 64) {
        const int i = base + lane;
        uint32_t y = 0, m397 = 0;
        if (i < 624) {
            const uint32_t a = mt[i], b = mt[(i + 1) % 624];
            y = (a & 0x80000000u) | (b & 0x7fffffffu);
            m397 = mt[(i + 397) % 624];
        }
        wave_lds_sync();  // every lane has read before any lane writes
        if (i < 624) mt[i] = m397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        wave_lds_sync();
    }
}
// ---- RandomSampler::Sample for a chunk of nT consecutive trials -----------------------------------
// The sample stream does not depend on the data, so a chunk's draws are produced ahead of the
// trials that use them.  Per draw the sequential algorithm does j = uniform_int(i, M-1) and
// swap(perm[i], perm[j]).  Fast path: all nT*kMin raw words are tempered and turned into j by
// the lanes in parallel (Lemire's multiply-shift; the rejection branch `low < range` has
// probability range / 2^32 per draw); only the swaps stay sequential, as wave-uniform scalar code
// with the trial's LDS reads and writes in flight together.  If any draw of the chunk needs the
// rejection branch the generator is restored and the chunk is redone draw by draw, exactly as
// libstdc++ does it.  perm[0..kMin) lives in registers (pr), the rest in LDS.
struct SamplerState {
    int mti;
    uint32_t pr[7];
};
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int kMin>
__device__ __forceinline__ SamplerState sample_chunk_t(lds_u32* mt, const uint32_t* snap, lds_u16* perm, lds_u16* sidx,
                                                       lds_u32* rawcnt, SamplerState st, int M, int nT, int lane,
                                                       int force_slow) {
    const int snap_mti = st.mti;
    const int need = nT * kMin;
    // ---- parallel: tempered raw word -> j, for the whole chunk ----
    bool slowflag = force_slow != 0;
    {
        int mti = st.mti, done = 0;
        while (done < need) {
            if (mti >= 624) {
                mt_twist(mt, lane);
                mti = 0;
            }
            const int take = min(624 - mti, need - done);
            for (int n0 = 0; n0 < take; n0 += 64) {
                const int n = n0 + lane;
                if (n < take) {
                    const int gdraw = done + n;
                    const int t = gdraw / kMin, i = gdraw - t * kMin;
                    const uint32_t range = (uint32_t)(M - i);
                    const uint64_t product = (uint64_t)mt_temper(mt[mti + n]) * (uint64_t)range;
                    if ((uint32_t)product < range) slowflag = true;
                    sidx[t * 8 + i] = (uint16_t)((uint32_t)i + (uint32_t)(product >> 32));
                }
            }
            mti += take;
            done += take;
        }
        st.mti = mti;
    }
    wave_lds_sync();
    if (__ballot(slowflag) == 0ull) {
        // ---- sequential swaps ----
        // Lane i < kMin owns slot i of the permutation's head (prv); one trial is, per slot, v = perm[j],
        // perm[j] = prv, prv = v - independent across slots as long as the trial's j are distinct and
        // none falls into the head.  Those trials (classified up front, lane t looks at trial t) take
        // one LDS read + two writes per lane, and the only dependence from trial to trial is the read
        // of t feeding the write of t + 1, so consecutive trials overlap in the in-order LDS queue.
        // The others (a few per cent) run the scalar, slot-by-slot code on the gathered head.
        // Lane t keeps trial t's draws in registers (jj); the loop below fetches them with readlane, so the
        // LDS queue only carries the permutation traffic and the wait before a trial's write is for the read
        // issued one trial earlier (sidx of trial t is stored during trial t + 1 for the same reason).
        unsigned long long slowmask;
        uint32_t jj[7];
        {
            bool odd = false;
#pragma unroll
            for (int i = 0; i < 7; ++i)
                jj[i] = (i < kMin && lane < nT) ? (uint32_t)sidx[lane * 8 + i] : 0xFFFF0000u + (uint32_t)i;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                odd |= jj[i] < (uint32_t)kMin;
#pragma unroll
                for (int q = i + 1; q < 7; ++q) odd |= jj[i] == jj[q];
            }
            if (lane < nT) rawcnt[lane] = (uint32_t)((lane + 1) * kMin);
            slowmask = __ballot(odd && lane < nT);
        }
        wave_lds_sync();  // every lane has its draws before the rows are overwritten with the samples
        uint32_t prv = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) prv = lane == i ? st.pr[i] : prv;
        const bool slot = lane < kMin;
        int t = 0;
        while (t < nT) {
            const unsigned long long rest = slowmask >> t;
            int run = rest ? (int)__builtin_ctzll(rest) : 64;
            run = min(run, nT - t);
            // two trials per round on alternating registers: the value read by one trial is stored by the
            // next, and nothing in between needs it (no copy, so no wait on the read just issued).  `pend` is
            // the trial whose samples (the value about to be stored) are not in sidx yet; at the start of a
            // run that store repeats what trial t - 1 already wrote (row 0 when there is none: rewritten below).
            int pend = max(t - 1, 0);
            uint32_t a = prv;
            const int e = t + run;
            while (t + 1 < e) {
                uint32_t j0 = 0, j1 = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (i < kMin) {
                        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)jj[i], t);
                        const uint32_t x1 = (uint32_t)__builtin_amdgcn_readlane((int)jj[i], t + 1);
                        j0 = lane == i ? x0 : j0;
                        j1 = lane == i ? x1 : j1;
                    }
                if (slot) {
                    const uint32_t b = perm[j0];
                    perm[j0] = (uint16_t)a;
                    sidx[pend * 8 + lane] = (uint16_t)a;
                    a = perm[j1];
                    perm[j1] = (uint16_t)b;
                    sidx[t * 8 + lane] = (uint16_t)b;
                }
                pend = t + 1;
                t += 2;
            }
            if (t < e) {
                uint32_t j0 = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (i < kMin) {
                        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)jj[i], t);
                        j0 = lane == i ? x0 : j0;
                    }
                if (slot) {
                    const uint32_t b = perm[j0];
                    perm[j0] = (uint16_t)a;
                    sidx[pend * 8 + lane] = (uint16_t)a;
                    a = b;
                }
                pend = t;
                ++t;
            }
            if (run > 0 && slot) sidx[pend * 8 + lane] = (uint16_t)a;
            prv = a;
            if (t >= nT) break;
            // trial t touches the head or draws an index twice: slot by slot on the gathered head
            uint32_t j[7], pr[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                j[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)jj[i], t) : 0xFFFFu;
                pr[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)prv, i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (i < kMin) {
                    if (j[i] < (uint32_t)kMin) {
                        uint32_t vj = pr[0];
#pragma unroll
                        for (int q = 1; q < 7; ++q) vj = (j[i] == (uint32_t)q) ? pr[q] : vj;
                        const uint32_t vi = pr[i];
#pragma unroll
                        for (int q = 0; q < 7; ++q) pr[q] = (j[i] == (uint32_t)q) ? vi : pr[q];
                        pr[i] = vj;
                    } else {
                        const uint32_t vj = sgpr(perm[j[i]]);
                        perm[j[i]] = (uint16_t)pr[i];
                        pr[i] = vj;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) prv = (i < kMin && lane == i) ? pr[i] : prv;
            if (slot) sidx[t * 8 + lane] = (uint16_t)prv;
            ++t;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) st.pr[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)prv, i) : st.pr[i];
        wave_lds_sync();
        return st;
    }
    // ---- a draw hit the rejection branch: restore the generator, redo the chunk draw by draw ----
    for (int i = lane; i < 624; i += 64) mt[i] = snap[i];
    wave_lds_sync();
    int mti = snap_mti;
    uint32_t nraw = 0;
    for (int t = 0; t < nT; ++t) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i < kMin) {
                const uint32_t range = (uint32_t)(M - i);
                uint64_t product;
                uint32_t low;
                {
                    if (mti >= 624) { mt_twist(mt, lane); mti = 0; }
                    ++nraw;
                    product = (uint64_t)mt_temper(sgpr(mt[mti++])) * (uint64_t)range;
                    low = (uint32_t)product;
                }
                if (low < range) {
                    const uint32_t threshold = (0u - range) % range;
                    while (low < threshold) {
                        if (mti >= 624) { mt_twist(mt, lane); mti = 0; }
                        ++nraw;
                        product = (uint64_t)mt_temper(sgpr(mt[mti++])) * (uint64_t)range;
                        low = (uint32_t)product;
                    }
                }
                const uint32_t jj = (uint32_t)(product >> 32) + (uint32_t)i;
                if (jj < (uint32_t)kMin) {
                    uint32_t vj = st.pr[0];
#pragma unroll
                    for (int q = 1; q < 7; ++q) vj = (jj == (uint32_t)q) ? st.pr[q] : vj;
                    const uint32_t vi = st.pr[i];
#pragma unroll
                    for (int q = 0; q < 7; ++q) st.pr[q] = (jj == (uint32_t)q) ? vi : st.pr[q];
                    st.pr[i] = vj;
                } else {
                    const uint32_t vj = sgpr(perm[jj]);
                    perm[jj] = (uint16_t)st.pr[i];
                    st.pr[i] = vj;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) sidx[t * 8 + i] = (uint16_t)st.pr[i];
            rawcnt[t] = nraw;
        }
    }
    st.mti = mti;
    wave_lds_sync();
    return st;
}

__device__ __noinline__ SamplerState sample_chunk(lds_u32* mt_, const uint32_t* snap_, lds_u16* perm_, lds_u16* sidx_,
                                                  lds_u32* rawcnt_, SamplerState st, int M_, int kMin_, int nT_,
                                                  int lane, int force_slow_) {
    // everything but `lane` is wave-uniform: move it to scalar registers
    lds_u32* mt = (lds_u32*)(uintptr_t)sgpr((uint32_t)(uintptr_t)mt_);
    const uint32_t* snap = snap_;
    lds_u16* perm = (lds_u16*)(uintptr_t)sgpr((uint32_t)(uintptr_t)perm_);
    lds_u16* sidx = (lds_u16*)(uintptr_t)sgpr((uint32_t)(uintptr_t)sidx_);
    lds_u32* rawcnt = (lds_u32*)(uintptr_t)sgpr((uint32_t)(uintptr_t)rawcnt_);
    const int M = (int)sgpr((uint32_t)M_), kMin = (int)sgpr((uint32_t)kMin_), nT = (int)sgpr((uint32_t)nT_);
    const int force_slow = (int)sgpr((uint32_t)force_slow_);
    st.mti = (int)sgpr((uint32_t)st.mti);
#pragma unroll
    for (int i = 0; i < 7; ++i) st.pr[i] = sgpr(st.pr[i]);
    if (kMin == 7) return sample_chunk_t<7>(mt, snap, perm, sidx, rawcnt, st, M, nT, lane, force_slow);
    if (kMin == 4) return sample_chunk_t<4>(mt, snap, perm, sidx, rawcnt, st, M, nT, lane, force_slow);
    if (kMin == 5) return sample_chunk_t<5>(mt, snap, perm, sidx, rawcnt, st, M, nT, lane, force_slow);
    return sample_chunk_t<1>(mt, snap, perm, sidx, rawcnt, st, M, nT, lane, force_slow);
}

// ---- the active RANSAC's correspondences: LDS copy when it fits, global arrays otherwise ---------
struct Pts {
    const lds_f64* l;  // x1 | y1 | x2 | y2, each `ls` long
    const double* g;   // x1 | y1 | x2 | y2, each `gs` long
    uint32_t ls, gs;
    bool lds;          // wave-uniform
};
// wave-uniform values that arrive in vector registers (function arguments) -> scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double uni(double v) { return readlane_f64(v, 0); }
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = uni((uint32_t)u), hi = uni((uint32_t)(u >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ Pts uni(Pts P) {
    Pts Q;
    Q.l = (const lds_f64*)(uintptr_t)uni((uint32_t)(uintptr_t)P.l);
    Q.g = uni_ptr(P.g);
    Q.ls = uni(P.ls);
    Q.gs = uni(P.gs);
    Q.lds = uni((int)P.lds) != 0;
    return Q;
}

template <bool L>
__device__ __forceinline__ void load_pt(const Pts& P, int k, double& a, double& b, double& c, double& d) {
    if (L) {
        a = P.l[k]; b = P.l[P.ls + k]; c = P.l[2 * P.ls + k]; d = P.l[3 * P.ls + k];
    } else {
        a = P.g[k]; b = P.g[P.gs + k]; c = P.g[2 * (size_t)P.gs + k]; d = P.g[3 * (size_t)P.gs + k];
    }
}
__device__ __forceinline__ void load_pt_any(const Pts& P, int k, double& a, double& b, double& c, double& d) {
    if (P.lds) load_pt<true>(P, k, a, b, c, d);
    else load_pt<false>(P, k, a, b, c, d);
}

__device__ __forceinline__ double residual_of(int kind, const double* m, double a, double b, double c, double d) {
    if (kind == K_H) return h_residual(m, a, b, c, d);
    if (kind == K_T) return t_residual(m, a, b, c, d);
    return sampson(m, a, b, c, d);
}

struct Model9 {
    double v[9];
};
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
template <bool L, int KIND>
__device__ __forceinline__ Support score_impl(const double* m, const Pts& P, int M, double max_res, int lane,
                                              int need_sum_from) {
    double acc = 0.0;
    int cnt = 0;
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        bool in = false;
        if (k < M) {
            double a, b, c, d;
            load_pt<L>(P, k, a, b, c, d);
            const double r = KIND == K_H ? h_residual(m, a, b, c, d)
                                         : (KIND == K_T ? t_residual(m, a, b, c, d) : sampson(m, a, b, c, d));
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
__device__ __noinline__ Support score(int kind, const Model9 mv, const Pts P_, int M_, double max_res_, int lane,
                                      int need_sum_from) {
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = uni(mv.v[i]);
    const Pts P = uni(P_);
    const int M = uni(M_);
    const double max_res = uni(max_res_);
    if (P.lds) {
        if (kind == K_H) return score_impl<true, K_H>(m, P, M, max_res, lane, need_sum_from);
        if (kind == K_T) return score_impl<true, K_T>(m, P, M, max_res, lane, need_sum_from);
        return score_impl<true, K_F7>(m, P, M, max_res, lane, need_sum_from);
    }
    if (kind == K_H) return score_impl<false, K_H>(m, P, M, max_res, lane, need_sum_from);
    if (kind == K_T) return score_impl<false, K_T>(m, P, M, max_res, lane, need_sum_from);
    return score_impl<false, K_F7>(m, P, M, max_res, lane, need_sum_from);
}

// ---- local optimisation over the ordered inlier list w.inl[0..K) ---------------------------------
// ordered compaction of the inlier indices of `model` (kind); returns K
template <bool L>
__device__ __forceinline__ int extract_impl(lds_u16* inl, int lane, int kind, const double* model, const Pts& P,
                                            int M, double max_res) {
    int base = 0;
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        bool in = false;
        if (k < M) {
            double a, b, c, d;
            load_pt<L>(P, k, a, b, c, d);
            in = residual_of(kind, model, a, b, c, d) <= max_res;
        }
        const unsigned long long bal = __ballot(in);
        if (in) inl[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)k;
        base += __popcll(bal);
    }
    wave_lds_sync();
    return base;
}
__device__ __noinline__ int extract_inliers(lds_u16* inl, int lane, int kind, const Model9 mv, const Pts P, int M,
                                            double max_res) {
    return P.lds ? extract_impl<true>(inl, lane, kind, mv.v, P, M, max_res)
                 : extract_impl<false>(inl, lane, kind, mv.v, P, M, max_res);
}

// CenterAndNormalizeImagePoints over the K listed points of image `img` (0: x1,y1; 1: x2,y2):
// only the transform T is produced; the normalised coordinates are recomputed where they are
// consumed (apply_T), with the operations of the reference loop, instead of being stored.
struct LoCtx {  // what the local estimators need of the wave, passed by value (registers)
    lds_u16* inl;
    lds_f64* jacA;
    lds_f64* jacV;
    int lane;
};
template <bool L>
__device__ __forceinline__ void center_T_impl(const LoCtx& w, const Pts& P, int img, int K, double* T) {
    const int lane = w.lane;
    double ax = 0.0, ay = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt<L>(P, w.inl[k], p[0], p[1], p[2], p[3]);
        ax += p[2 * img]; ay += p[2 * img + 1];
    }
    const double cx = butterfly(ax) / (double)K;
    const double cy = butterfly(ay) / (double)K;
    double ar = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt<L>(P, w.inl[k], p[0], p[1], p[2], p[3]);
        const double ddx = p[2 * img] - cx, ddy = p[2 * img + 1] - cy;
        ar += ddx * ddx + ddy * ddy;
    }
    double rms = butterfly(ar);
    rms = dsqrt(rms / (double)K);
    const double nf = dsqrt(2.0) / rms;
    T[0] = nf; T[1] = 0; T[2] = -nf * cx;
    T[3] = 0; T[4] = nf; T[5] = -nf * cy;
    T[6] = 0; T[7] = 0; T[8] = 1;
}
__device__ __forceinline__ void apply_T(const double* T, double p0, double p1, double& o0, double& o1) {
    const double np0 = T[0] * p0 + T[1] * p1 + T[2];
    const double np1 = T[3] * p0 + T[4] * p1 + T[5];
    const double np2 = T[6] * p0 + T[7] * p1 + T[8];
    const double inv = 1.0 / np2;
    o0 = np0 * inv;
    o1 = np1 * inv;
}

// A^T A (9 x 9, symmetric) over the design rows of the listed correspondences, every entry in
// det_sum64 order: each lane keeps the 45 partial sums of its strided rows, then 45 butterflies.
//   MODE 0: epipolar row [x1 x2, y1 x2, x2, x1 y2, y1 y2, y2, x1, y1, 1]      (F8 / E5), K rows
//   MODE 1: homography rows; r < K -> "a" row of point r, r >= K -> "b" row of point r-K, 2K rows
// NORM: correspondences are normalised by T1 / T2 first.
template <bool L, int MODE, bool NORM>
__device__ __forceinline__ void ata_impl(const LoCtx& w, const Pts& P, int K, const double* T1, const double* T2) {
    const int lane = w.lane;
    const int rows = MODE == 1 ? 2 * K : K;
    double acc[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) acc[e] = 0.0;
    for (int r0 = lane; r0 < rows; r0 += 64) {
        const int k = (MODE == 1 && r0 >= K) ? r0 - K : r0;
        double x1, y1, x2, y2;
        load_pt<L>(P, w.inl[k], x1, y1, x2, y2);
        if (NORM) {
            apply_T(T1, x1, y1, x1, y1);
            apply_T(T2, x2, y2, x2, y2);
        }
        double r[9];
        if (MODE == 0) {
            r[0] = x1 * x2; r[1] = y1 * x2; r[2] = x2;
            r[3] = x1 * y2; r[4] = y1 * y2; r[5] = y2;
            r[6] = x1; r[7] = y1; r[8] = 1.0;
        } else if (r0 < K) {
            r[0] = -x1; r[1] = -y1; r[2] = -1; r[3] = 0; r[4] = 0; r[5] = 0;
            r[6] = x1 * x2; r[7] = y1 * x2; r[8] = x2;
        } else {
            r[0] = 0; r[1] = 0; r[2] = 0; r[3] = -x1; r[4] = -y1; r[5] = -1;
            r[6] = x1 * y2; r[7] = y1 * y2; r[8] = y2;
        }
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
            if (lane == 0) {
                w.jacA[i * 9 + j] = sres;
                w.jacA[j * 9 + i] = sres;
            }
        }
    wave_lds_sync();
}

// Round-robin Jacobi (tvg_math.h jacobi_eigen) on a symmetric n x n matrix held in LDS, the whole
// wave cooperating.  Per round: lane e < n/2 computes the rotation of the round's e-th pair; lane
// (e, k) = e * n + k then updates entry k of columns p_e, q_e of A and V, and after a barrier entry
// k of rows p_e, q_e of A.  Every element sees exactly the arithmetic of the scalar version, so the
// results are bit-identical; the pairs of a round being disjoint, no two lanes touch one entry
// within a phase.
// jacobi_pair(9, r, e, p, q) without the run-time modulo (m = 9 rounds, 4 pairs per round)
__device__ __forceinline__ void jacobi_pair9(int r, int e, int& p, int& q) {
    int x = r + e + 1, y = r - (e + 1);
    x = x >= 9 ? x - 9 : x;
    y = y < 0 ? y + 9 : y;
    p = x < y ? x : y;
    q = x < y ? y : x;
}
__device__ __noinline__ void jacobi_eigen_wave(int n_, lds_f64* A, lds_f64* V, int lane) {
    constexpr int n = 9, rounds = 9, np = 4;  // the only size the kernel decomposes as a wave
    (void)n_;
    for (int i = lane; i < n * n; i += 64) V[i] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    wave_lds_sync();
    double total = 0.0;
    for (int i = 0; i < n * n; ++i) total += A[i] * A[i];
    const double tol = total * 1e-32;
    const int e_of = lane / n, k_of = lane - e_of * n;  // this lane's (pair, entry) in the update phases
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < n - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (!(off > tol)) break;
        for (int r = 0; r < rounds; ++r) {
            // rotation parameters: lane e computes pair e
            double c = 1.0, s = 0.0;
            bool act = false;
            if (lane < np) {
                int p, q;
                jacobi_pair9(r, lane, p, q);
                act = jacobi_rotation(A[p * n + p], A[q * n + q], A[p * n + q], c, s);
            }
            const double ce = __shfl(c, e_of), se = __shfl(s, e_of);
            const bool acte = __shfl((int)act, e_of) != 0 && e_of < np;
            int p = 0, q = 0;
            if (e_of < np) jacobi_pair9(r, e_of, p, q);
            wave_lds_sync();
            if (acte) {  // columns p, q of A and V, entry k
                const double akp = A[k_of * n + p], akq = A[k_of * n + q];
                const double vkp = V[k_of * n + p], vkq = V[k_of * n + q];
                A[k_of * n + p] = ce * akp - se * akq;
                A[k_of * n + q] = se * akp + ce * akq;
                V[k_of * n + p] = ce * vkp - se * vkq;
                V[k_of * n + q] = se * vkp + ce * vkq;
            }
            wave_lds_sync();
            if (acte) {  // rows p, q of A, entry k
                const double apk = A[p * n + k_of], aqk = A[q * n + k_of];
                A[p * n + k_of] = ce * apk - se * aqk;
                A[q * n + k_of] = se * apk + ce * aqk;
            }
            wave_lds_sync();
        }
    }
}
// eigenvector of the smallest eigenvalue after jacobi_eigen_wave (first minimum, like the oracle)
__device__ void smallest_eigvec9_wave(const lds_f64* A, const lds_f64* V, double* x) {
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i * 9 + i] < A[best * 9 + best]) best = i;
    for (int i = 0; i < 9; ++i) x[i] = V[i * 9 + best];
}

// ---- real roots of ONE polynomial by the whole wave (local optimisation's 5-point solve) ----------
// tvg_math.h's RootChain walks the chain of derivatives and, per level, bisects the sign-change
// brackets one after the other.  The brackets of a level are independent, so here lane i takes
// bracket i (same arithmetic per bracket, hence the same roots bit for bit) and the level costs one
// bisection instead of up to R of them; the ordered, de-duplicated root list is then assembled
// exactly as roots_between_t does it.
template <int DEG, int R>
struct WaveRootChain {
    static __device__ __forceinline__ int run(const double (&c)[DEG + 1], double* roots, lds_f64* tmp, int lane) {
        double crit[R];
        const int nc = WaveRootChain<DEG, R - 1>::run(c, crit, tmp, lane);
        double d[R + 1];
        poly_derivative_t<DEG, DEG - R>(c, d);
        double bound = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) bound = dmax(bound, dabs(d[i] / d[R]));
        bound = 1.0 + bound;
        int ne = 0;
        tmp[ne++] = -bound;
        for (int i = 0; i < nc; ++i)
            if (crit[i] > -bound && crit[i] < bound) tmp[ne++] = crit[i];
        tmp[ne++] = bound;
        wave_lds_sync();
        // one bracket per lane: 0 nothing, 1 exact root at the lower edge, 2 bracketed root
        int kind = 0;
        double val = 0.0;
        if (lane + 1 < ne) {
            double lo = tmp[lane], hi = tmp[lane + 1];
            double flo = poly_eval_t<R>(d, lo);
            const double fhi = poly_eval_t<R>(d, hi);
            if (flo == 0.0) {
                kind = 1;
                val = lo;
            } else if (fhi != 0.0 && (flo < 0.0) != (fhi < 0.0)) {
                for (int it = 0; it < 200; ++it) {
                    const double mid = 0.5 * (lo + hi);
                    if (mid == lo || mid == hi) break;
                    const double fm = poly_eval_t<R>(d, mid);
                    if (fm == 0.0) { lo = mid; hi = mid; break; }
                    if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else { hi = mid; }
                }
                kind = 2;
                val = 0.5 * (lo + hi);
            }
        }
        const double last = tmp[ne - 1];
        wave_lds_sync();  // tmp is rewritten by the next level
        int nr = 0;
        for (int i = 0; i + 1 < ne; ++i) {
            const int k = __builtin_amdgcn_readlane(kind, i);
            const double r = readlane_f64(val, i);
            if (k == 1) {
                if (nr == 0 || roots[nr - 1] != r) roots[nr++] = r;
            } else if (k == 2) {
                roots[nr++] = r;
            }
        }
        if (poly_eval_t<R>(d, last) == 0.0 && (nr == 0 || roots[nr - 1] != last)) roots[nr++] = last;
        return nr;
    }
};
template <int DEG>
struct WaveRootChain<DEG, 1> {
    static __device__ __forceinline__ int run(const double (&c)[DEG + 1], double* roots, lds_f64*, int) {
        double d[2];
        poly_derivative_t<DEG, DEG - 1>(c, d);
        roots[0] = -d[0] / d[1];
        return 1;
    }
};
// all real roots of a degree-10 polynomial (wave-uniform input), ascending; = real_roots_t<10>
__device__ __noinline__ int real_roots10_wave(const double* c_in, double* roots, lds_f64* tmp, int lane) {
    double c[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) c[i] = c_in[i];
    if (c[10] == 0.0) return real_roots_t<10>(c, roots);  // degenerate leading coefficient: plain path
    return WaveRootChain<10, 10>::run(c, roots, tmp, lane);
}

// ---- the 5-point solve of ONE problem by the whole wave (local optimisation) -----------------------
// e5_build keeps the 10 x 20 constraint matrix of a solve in one lane: 200 live doubles, most of them in scratch
// memory, and with one problem per wave every lane did the same elimination.  Here the rows are still computed by
// every lane (same expressions, row by row, so that only one row is live), but lane c < 20 keeps just column c,
// and the Gauss-Jordan elimination runs on those 20 columns in parallel: per pivot the pivot column is broadcast
// (ten readlanes), every lane searches the pivot and applies the row swap to its own column, and one multiply and
// nine multiply-subtracts finish the step.  Every element goes through the operations e5_build applies to it, in
// the same order: the same bits.  E E^T and its trace sit in `sc` (>= 100 doubles of LDS) between the two passes.
__device__ __noinline__ void e5_build_wave(const double* nsp, E5Polys& P, lds_f64* sc, int lane) {
    double e[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int d = 0; d < 4; ++d) e[k][d] = nsp[d * 9 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double a[10], b[10], c[10];
            e5_mul11(e[3 * i], e[3 * j], a);
            e5_mul11(e[3 * i + 1], e[3 * j + 1], b);
            e5_mul11(e[3 * i + 2], e[3 * j + 2], c);
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < 10; ++t) sc[(3 * i + j) * 10 + t] = (a[t] + b[t]) + c[t];
            }
        }
    wave_lds_sync();
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 10; ++t) sc[90 + t] = (sc[t] + sc[40 + t]) + sc[80 + t];
    }
    wave_lds_sync();
    double g[10];  // this lane's column of G
    auto keep = [&](double& dst, const double (&row)[20]) {
        double x = row[19];
#pragma unroll
        for (int c = 18; c >= 0; --c) x = lane == c ? row[c] : x;
        dst = x;
    };
    {   // det(E) -> row 0
        double a[10], b[10], d[10], t0[20], t1[20], t2[20], row[20];
        e5_mul11(e[4], e[8], a); e5_mul11(e[5], e[7], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[0], t0);
        e5_mul11(e[3], e[8], a); e5_mul11(e[5], e[6], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[1], t1);
        e5_mul11(e[3], e[7], a); e5_mul11(e[4], e[6], b);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = a[i] - b[i];
        e5_mul21(d, e[2], t2);
#pragma unroll
        for (int i = 0; i < 20; ++i) row[i] = (t0[i] - t1[i]) + t2[i];
        keep(g[0], row);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double q[10], acc[20], tmp[20], row[20];
#pragma unroll
            for (int t = 0; t < 10; ++t) q[t] = sc[(3 * i) * 10 + t];
            e5_mul21(q, e[j], acc);
#pragma unroll
            for (int t = 0; t < 10; ++t) q[t] = sc[(3 * i + 1) * 10 + t];
            e5_mul21(q, e[3 + j], tmp);
#pragma unroll
            for (int t = 0; t < 20; ++t) acc[t] = acc[t] + tmp[t];
#pragma unroll
            for (int t = 0; t < 10; ++t) q[t] = sc[(3 * i + 2) * 10 + t];
            e5_mul21(q, e[6 + j], tmp);
#pragma unroll
            for (int t = 0; t < 20; ++t) acc[t] = acc[t] + tmp[t];
#pragma unroll
            for (int t = 0; t < 10; ++t) q[t] = sc[90 + t];
            e5_mul21(q, e[3 * i + j], tmp);
#pragma unroll
            for (int t = 0; t < 20; ++t) row[t] = acc[t] * 2.0 - tmp[t];
            keep(g[1 + 3 * i + j], row);
        }
    // Gauss-Jordan with partial pivoting on the left 10 x 10 block, one column per lane
#pragma unroll
    for (int col = 0; col < 10; ++col) {
        double bc[10];  // column `col` as it stands, wave-uniform
#pragma unroll
        for (int r = 0; r < 10; ++r) bc[r] = readlane_f64(g[r], col);
        int piv = col;
        double pv = dabs(bc[col]);
#pragma unroll
        for (int r = col + 1; r < 10; ++r)
            if (dabs(bc[r]) > pv) { pv = dabs(bc[r]); piv = r; }
#pragma unroll
        for (int r = col + 1; r < 10; ++r) {
            const bool sw = piv == r;
            const double t = g[col], u = bc[col];
            g[col] = sw ? g[r] : g[col];
            g[r] = sw ? t : g[r];
            bc[col] = sw ? bc[r] : bc[col];
            bc[r] = sw ? u : bc[r];
        }
        const double inv = 1.0 / bc[col];
        g[col] = g[col] * inv;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            g[r] = g[r] - bc[r] * g[col];
        }
    }
    // rows 4..9 of columns 10..19 -> every lane
    wave_lds_sync();
    if (lane >= 10 && lane < 20) {
#pragma unroll
        for (int r = 0; r < 6; ++r) sc[r * 10 + (lane - 10)] = g[4 + r];
    }
    wave_lds_sync();
    double hl[6][10];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 10; ++c) hl[r][c] = sc[r * 10 + c];
    wave_lds_sync();  // sc is the root finder's scratch next
    e5_finish(hl, P);
}
// e5_models with root i on lane i; the models come back wave-uniform, in root order
__device__ __noinline__ int e5_models_wave(const double* nsp, const E5Polys& P, const double* roots, int nr, double* models,
                                           int lane) {
    double z = roots[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) z = lane == i ? roots[i] : z;
    double E[9];
    const bool ok = e5_model_from_root(nsp, P, z, E) && lane < nr;
    unsigned long long mask = __ballot(ok);
    int nm = 0;
    while (mask) {
        const int src = (int)__builtin_ctzll(mask);
        mask &= mask - 1;
#pragma unroll
        for (int i = 0; i < 9; ++i) models[9 * nm + i] = readlane_f64(E[i], src);
        ++nm;
    }
    return nm;
}

// local estimator on the K listed inlier correspondences -> models (uniform), count
template <bool L>
__device__ __forceinline__ int local_estimate_impl(const LoCtx& w, int kind, const Pts& P, int K, double* models) {
    const int lane = w.lane;
    if (kind == K_T) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int k = lane; k < K; k += 64) {
            double p0, p1, p2, p3;
            load_pt<L>(P, w.inl[k], p0, p1, p2, p3);
            a += p0; b += p1; c += p2; d += p3;
        }
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
            for (int i = 0; i < 5; ++i) load_pt<L>(P, w.inl[i], a[i], b[i], c[i], d[i]);
            return estimate_e5_minimal(a, b, c, d, models);
        }
        ata_impl<L, 0, false>(w, P, K, nullptr, nullptr);
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double nsp[4 * 9];
        e5_nullspace_from_eig(w.jacA, w.jacV, nsp);
        wave_lds_sync();  // jacA doubles as the root finder's scratch from here on
        E5Polys polys;
        e5_build_wave(nsp, polys, w.jacA, lane);
        double roots[10];
        const int nr = real_roots10_wave(polys.det, roots, w.jacA, lane);
        return e5_models_wave(nsp, polys, roots, nr, models, lane);
    }
    if (kind == K_H && K == 4) {
        double a[4], b[4], c[4], d[4];
        for (int i = 0; i < 4; ++i) load_pt<L>(P, w.inl[i], a[i], b[i], c[i], d[i]);
        estimate_h4(a, b, c, d, models);
        return 1;
    }
    double T1[9], T2[9];
    center_T_impl<L>(w, P, 0, K, T1);
    center_T_impl<L>(w, P, 1, K, T2);
    if (kind == K_F8) {
        ata_impl<L, 0, true>(w, P, K, T1, T2);
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double f[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, f);
        f8_from_vec(f, T1, T2, models);
    } else {
        ata_impl<L, 1, true>(w, P, K, T1, T2);
        jacobi_eigen_wave(9, w.jacA, w.jacV, lane);
        double h[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, h);
        h_denormalize(h, T1, T2, models);
    }
    return 1;
}
__device__ __noinline__ int local_estimate(const LoCtx w, int kind, const Pts P, int K, double* models) {
    return P.lds ? local_estimate_impl<true>(w, kind, P, K, models)
                 : local_estimate_impl<false>(w, kind, P, K, models);
}

// ---- a chunk's minimal problems and the inlier count of every resulting model --------------------
// One minimal problem per lane (F / H / T models stay with the solving lane, slot i = i-th root; E
// models, up to 10 per trial, go to global memory).  Then every model of the chunk is broadcast in
// turn (v_readlane -> scalar registers) and the whole wave counts its inliers over the
// correspondences (ballot + popcount).  A model can only change the course of the sequential
// algorithm if its count reaches the best count so far, so all the replay needs per trial is the
// largest count among its models; the few trials that qualify are re-scored in full there.
struct ChunkModels {
    double mym[27];
    int nmod;    // models of this lane's trial
    int maxcnt;  // max inlier count over them (-1: none)
    unsigned long long cyc_solve, cyc_count;
};
template <int KIND>
__device__ __forceinline__ double residual_t(const double (&m)[9], double a, double b, double c, double d) {
    return KIND == K_H ? h_residual(m, a, b, c, d) : (KIND == K_T ? t_residual(m, a, b, c, d) : sampson(m, a, b, c, d));
}
// ---- division-free inlier tests for the counting loop ----------------------------------------------
// Counting only needs the DECISION residual <= max_res, and for both residuals that is a polynomial inequality:
//   homography  (d0 - pd0/pd2)^2 + (d1 - pd1/pd2)^2 <= T   <=>   (d0 pd2 - pd0)^2 + (d1 pd2 - pd1)^2 <= T pd2^2
//   Sampson     c^2 / den <= T                              <=>   c^2 <= T den               (den > 0)
// evaluated here with fused multiply-adds (no division: a third of the instructions of the reference expression
// and no rcp -> Newton -> fixup dependency chain).  The two sides are NOT the reference's roundings, so the test
// is only trusted away from the boundary: with L and R the two sides, `in` when L <= R (1 - 1e-8), `out` when
// L >= R (1 + 1e-8), and the (practically never taken) band in between - or an R that is not a normal positive
// number - sends the whole 64-point batch through the exact residual.  Why the band suffices: both this
// expression and the reference one are backward-stable evaluations of the same real quantity whose relative error
// at the boundary is <= ~10 eps x (largest coordinate / max_error) - the cancellation in d - p and in x2^T E x1;
// lo_ransac switches the fast test off unless that ratio is below 1e5 (RansacCfg::fast_count), which bounds both
// errors by ~1e-10, a hundredth of the band.  The counts are therefore exactly the reference's counts.
constexpr double kFastLo = 1.0 - 1e-8, kFastHi = 1.0 + 1e-8;
#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 2)
// diagnostic build: shader-clock cycles of the parts of the counting loop, summed over all waves (lane 0 adds)
__device__ unsigned long long g_tvg_diag[8];
#define DIAG_T0() const unsigned long long dt0_ = __builtin_readcyclecounter()
#define DIAG_ADD(slot) do { if (lane == 0) atomicAdd(&g_tvg_diag[slot], __builtin_readcyclecounter() - dt0_); } while (0)
#else
#define DIAG_T0() do {} while (0)
#define DIAG_ADD(slot) do {} while (0)
#endif
template <int KIND>
__device__ __forceinline__ void fast_inlier(const double (&m)[9], double a, double b, double c, double d, double T,
                                            bool& in, bool& amb) {
    double Lq, R;
    if (KIND == K_H) {
        const double pd0 = __fma_rn(m[0], a, __fma_rn(m[1], b, m[2]));
        const double pd1 = __fma_rn(m[3], a, __fma_rn(m[4], b, m[5]));
        const double pd2 = __fma_rn(m[6], a, __fma_rn(m[7], b, m[8]));
        const double u = __fma_rn(c, pd2, -pd0), v = __fma_rn(d, pd2, -pd1);
        Lq = __fma_rn(u, u, v * v);
        R = T * (pd2 * pd2);
    } else {
        const double Ex1_0 = __fma_rn(m[0], a, __fma_rn(m[1], b, m[2]));
        const double Ex1_1 = __fma_rn(m[3], a, __fma_rn(m[4], b, m[5]));
        const double Ex1_2 = __fma_rn(m[6], a, __fma_rn(m[7], b, m[8]));
        const double Etx2_0 = __fma_rn(m[0], c, __fma_rn(m[3], d, m[6]));
        const double Etx2_1 = __fma_rn(m[1], c, __fma_rn(m[4], d, m[7]));
        const double x2tEx1 = __fma_rn(c, Ex1_0, __fma_rn(d, Ex1_1, Ex1_2));
        Lq = x2tEx1 * x2tEx1;
        R = T * __fma_rn(Ex1_0, Ex1_0, __fma_rn(Ex1_1, Ex1_1, __fma_rn(Etx2_0, Etx2_0, Etx2_1 * Etx2_1)));
    }
    const bool sane = R > 1e-200 && R < 1e200;  // false for 0, denormal-ish, huge, inf and NaN
    in = sane && Lq <= R * kFastLo;
    amb = !sane || (Lq > R * kFastLo && Lq < R * kFastHi);
}
// inliers of U consecutive full 64-point batches starting at k0: U independent residual chains in
// flight (at two waves per SIMD little else hides the FP64 and LDS latencies)
template <bool L, int KIND, int U, bool FAST>
__device__ __forceinline__ int count_batches(const double (&m)[9], const Pts& P, int k0, double max_res, int lane) {
    double a[U], b[U], c[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) load_pt<L>(P, k0 + 64 * u + lane, a[u], b[u], c[u], d[u]);
    int cnt = 0;
    if (FAST && KIND != K_T) {
        bool in[U], amb = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool am;
            fast_inlier<KIND>(m, a[u], b[u], c[u], d[u], max_res, in[u], am);
            amb |= am;
        }
        if (__builtin_expect(__ballot(amb) == 0ull, 1)) {
#pragma unroll
            for (int u = 0; u < U; ++u) cnt += __popcll(__ballot(in[u]));
            return cnt;
        }
    }
#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 1)
    if (FAST && KIND != K_T) return 0;  // diagnostic build: a taken fallback corrupts the counts (parity tests then fail)
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) cnt += __popcll(__ballot(residual_t<KIND>(m, a[u], b[u], c[u], d[u]) <= max_res));
    return cnt;
}
// Inlier count of one model, or - as soon as even counting every remaining correspondence as an
// inlier could not reach `thr` - an upper bound below `thr`.  thr is the best count when the chunk
// started: the best only grows, so such a model can never become a candidate and its exact count
// is irrelevant (see the replay in lo_ransac).
template <bool L, int KIND, bool FAST>
__device__ __forceinline__ int count_model(const double (&m)[9], const Pts& P, int M, double max_res, int lane,
                                           int thr) {
    int cnt = 0;
    int k0 = 0;
    for (; k0 + 256 <= M; k0 += 256) {
        DIAG_T0();
        cnt += count_batches<L, KIND, 4, FAST>(m, P, k0, max_res, lane);
        DIAG_ADD(0);
        if (lane == 0) { DIAG_T0(); DIAG_ADD(3); }  // slot 3: the cost of one timer pair itself
        if (cnt + (M - (k0 + 256)) < thr) return cnt + (M - (k0 + 256));
    }
    DIAG_T0();
    if (k0 + 128 <= M) {
        cnt += count_batches<L, KIND, 2, FAST>(m, P, k0, max_res, lane);
        k0 += 128;
        if (cnt + (M - k0) < thr) return cnt + (M - k0);
    }
    if (k0 + 64 <= M) {
        cnt += count_batches<L, KIND, 1, FAST>(m, P, k0, max_res, lane);
        k0 += 64;
    }
    if (k0 < M) {  // ragged tail
        const int k = k0 + lane;
        bool in = false;
        if (k < M) {
            double a, b, c, d;
            load_pt<L>(P, k, a, b, c, d);
            in = residual_t<KIND>(m, a, b, c, d) <= max_res;
        }
        cnt += __popcll(__ballot(in));
    }
    DIAG_ADD(1);
    return cnt;
}
template <bool L, int KIND, int NM, bool FAST>
__device__ __forceinline__ int count_lane_models(const double (&mym)[27], int nmod, const Pts& P, int M,
                                                 double max_res, int nT, int lane, int thr) {
    int maxcnt = -1;
    for (int t = 0; t < nT; ++t) {
        const int n = __builtin_amdgcn_readlane(nmod, t);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m < n) {
                DIAG_T0();
                double sm[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mym[9 * m + i], t);
                DIAG_ADD(2);
                const int c = count_model<L, KIND, FAST>(sm, P, M, max_res, lane, thr);
                if (lane == t) maxcnt = max(maxcnt, c);
#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 2)
                if (lane == 0) atomicAdd(&g_tvg_diag[4], 1ull);
#endif
            }
        }
    }
    return maxcnt;
}
template <bool L, bool FAST>
__device__ __forceinline__ int count_global_models(const double* models, int nmod, const Pts& P, int M,
                                                   double max_res, int nT, int lane, int thr) {
    int maxcnt = -1;
    for (int t = 0; t < nT; ++t) {
        const int n = __builtin_amdgcn_readlane(nmod, t);
        for (int m = 0; m < n; ++m) {
            const double* src = models + ((size_t)t * kMaxModels + m) * 9;
            double sm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(src[i], 0);
            const int c = count_model<L, K_F7, FAST>(sm, P, M, max_res, lane, thr);
            if (lane == t) maxcnt = max(maxcnt, c);
        }
    }
    return maxcnt;
}

// ---- counting with the correspondences held in registers --------------------------------------------
// count_model above reads the points of every batch from LDS again for every model: per model a broadcast, an LDS
// round trip, the arithmetic, then the ragged tail with another round trip - at two waves per SIMD those latencies
// are what the loop costs (measured: ~2,100 cycles per model of 300 points whether the residual takes 34 or 20
// FP64 instructions).  Here the loop is turned inside out: the wave loads all M <= 64 * NB correspondences once
// per chunk (NB batches of 64, 8 VGPRs each), and every model of the chunk's 64 trials is then counted against the
// registers - the inner loop is a broadcast plus straight FP64 arithmetic on independent batches.  Lanes past M in
// the last batch hold a copy of point 0 and are masked out of the ballots.  The division-free test (fast_inlier)
// decides; a model with an ambiguous point is recounted with the exact residual, from the same registers.
template <int KIND, int NB>
__device__ __forceinline__ int count_regs(const double (&sm)[9], const double (&a)[NB], const double (&b)[NB],
                                          const double (&c)[NB], const double (&d)[NB], unsigned long long last_valid,
                                          int M, double max_res, int thr) {
    int cnt = 0;
    bool amb = false;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        bool in, am;
        fast_inlier<KIND>(sm, a[u], b[u], c[u], d[u], max_res, in, am);
        amb |= am;
        unsigned long long bal = __ballot(in);
        if (u == NB - 1) bal &= last_valid;
        cnt += __popcll(bal);
        // as in count_model: a model that cannot reach the best count even if every remaining point were an inlier
        if ((u & 1) == 1 && u + 1 < NB) {
            const int rest = M - 64 * (u + 1);
            if (cnt + rest < thr && __ballot(amb) == 0ull) return cnt + rest;
        }
    }
    if (__builtin_expect(__ballot(amb) == 0ull, 1)) return cnt;
    cnt = 0;  // some point sat in the band around the threshold: the reference residual decides, for the whole model
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        unsigned long long bal = __ballot(residual_t<KIND>(sm, a[u], b[u], c[u], d[u]) <= max_res);
        if (u == NB - 1) bal &= last_valid;
        cnt += __popcll(bal);
    }
    return cnt;
}
// ---- homography counting: packed-FP32 pre-filter ------------------------------------------------------
// The homography RANSAC of a non-planar pair runs to its trial cap and nearly all of its models count nearly all
// matches as outliers by a wide margin; the decision u'^2 + v'^2 <= w^2 (u' = c' w - p0', v' = d' w - p1', the
// threshold folded into rows 0 / 1 of the model and the image-2 coordinates by s = 1 / sqrt(T)) does not need 53
// bits there.  It is first evaluated in FP32 on two 64-point batches at a time (v_pk_fma_f32: two batches per
// instruction), TOGETHER WITH A BOUND ON ITS OWN ERROR; a point is decided in FP32 only if |t32| exceeds that bound,
// and a model with any undecided point is recounted by the FP64 path (count_regs: division-free test, then the
// reference residual inside its band).
//
// The bound (u = 2^-24, every FP32 operation below is a single rounding; C = largest |coordinate| of the pair,
// Cs = C s; m = the scaled model as the lane computed it in FP64; A0 = (|m0| + |m1|) C + |m2|, A1, Aw likewise):
//   p0, p1, w:   |p32 - p| <= 5u A            (coefficient and coordinate conversions + two FMAs)
//   u', v':      |u32 - u'| <= E0 + u |u32|,  E0 = 5u max(A0, A1) + 6u Cs Aw
//   L = u'^2 + v'^2, R = w^2, t = L - R:
//   |t32 - t| <= 6u |t32| + 6u R32 + 2 E0 (|u32| + |v32|) + 2 Ew |w32| + 2 E0^2 + Ew^2,   Ew = 5u Aw
// and the FP64 test is itself only trusted outside |t| <= 2e-8 R, so a point is decided here iff
//   |t32| > 4.2e-7 R32 + kE (|u32| + |v32|) + kW |w32| + K0,   kE = 2.05 E0, kW = 2.05 Ew, K0 = 2.05 (2 E0^2 + Ew^2)
// (the constants carry > 2 % of slack for the FP32 rounding of the bound itself and the 6u |t32| term; NaN / inf
// compare false = undecided).  For 1600 x 1200 images and a 4 px threshold the band is ~1e-3 of R: a point is
// undecided only within ~1e-3 px of the threshold circle.  (Replacing |u| + |v| and |w| by AM-GM bounds in L and R
// saves the absolute values but leaves 29 % of the models undecided instead of 4 %: the sampled homographies of a
// non-planar scene are wild, many points have |w| far below the model's typical value.)
// Checked by a diagnostic build that counts every model both ways (tools/diag_build_tvg.sh 8): 80 million models of
// the bench's verify leg, 3.9 % undecided, no decided model with a count different from the FP64 path's.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef H32Model H32Lane;  // tvg_math.h: the scaled float model and the constants of its error bound (h32_prepare)
__device__ __forceinline__ float readlane_f32(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
struct H32PkOps {  // h32_eval on two points per instruction
    static __device__ __forceinline__ v2f fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ v2f abs(v2f a) { return __builtin_elementwise_abs(a); }
    static __device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
};
// count of one model over NP pairs of batches held as packed floats
template <int NP>
__device__ __forceinline__ int count_h32(const H32Model& hm, const v2f (&A)[NP],
                                         const v2f (&B)[NP], const v2f (&Cs)[NP], const v2f (&Ds)[NP],
                                         unsigned long long valid_lo_last, unsigned long long valid_hi_last, int M, int thr,
                                         bool& undecided) {
    int cnt = 0;
    bool und = false;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        v2f t, band;
        h32_eval<v2f, H32PkOps>(hm, A[q], B[q], Cs[q], Ds[q], t, band);
        const bool dec0 = fabsf(t.x) > band.x, dec1 = fabsf(t.y) > band.y;
        unsigned long long in0 = __ballot(t.x < 0.0f), in1 = __ballot(t.y < 0.0f);
        unsigned long long un0 = __ballot(!dec0), un1 = __ballot(!dec1);
        if (q == NP - 1) {
            in0 &= valid_lo_last; un0 &= valid_lo_last;
            in1 &= valid_hi_last; un1 &= valid_hi_last;
        }
        cnt += __popcll(in0) + __popcll(in1);
        und |= (un0 | un1) != 0ull;
        if (q + 1 < NP) {  // a model that cannot reach the best count even if every remaining point were an inlier
            const int rest = M - 128 * (q + 1);
            if (!und && cnt + rest < thr) {
                undecided = false;
                return cnt + rest;
            }
        }
    }
    undecided = und;
    return cnt;
}
#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 8)
__device__ unsigned long long g_h32_diag[4];  // models, undecided models, decided models whose count differs from FP64
#endif

// KIND K_F7 / K_H: the models are in the solving lanes' registers (mym, NM per trial); KIND K_E5: in global memory
// One block of up to 768 correspondences [off, off + Mb) against every model of the chunk; `acc` (lane t: the running
// counts of trial t's models) carries the counts from block to block - pairs with more matches than the registers
// hold are counted 768 at a time.  A model that cannot reach `thr` even if every remaining match (of this block and
// of the `rest_after` ones behind it) were an inlier is not evaluated: its count is only ever compared with `thr`.
template <int KIND, int NM, int NB>
__device__ __forceinline__ void count_block_regs(const double (&mym)[27], const double* models, int nmod, const Pts& P,
                                                 int off, int Mb, int rest_after, double max_res, int nT, int lane, int thr,
                                                 double cmax, int (&acc)[(KIND == K_E5 ? kMaxModels : NM)]) {
    constexpr int NMX = KIND == K_E5 ? kMaxModels : NM;
    // one load per chunk of 64 trials: from LDS, or - pairs whose points do not fit the wave's LDS share - straight
    // from the global arrays (the latency is paid once per chunk, not once per model)
    double a[NB], b[NB], c[NB], d[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = 64 * u + lane;
        load_pt_any(P, off + (k < Mb ? k : 0), a[u], b[u], c[u], d[u]);
    }
    const int tail = Mb - 64 * (NB - 1);  // 1 .. 64 valid lanes in the last batch
    const unsigned long long last_valid = tail >= 64 ? ~0ull : ((1ull << tail) - 1ull);
    if (KIND == K_H) {
        // packed-FP32 pre-filter (above): pairs of batches as float2, the lanes' models scaled and rounded once
        constexpr int NP = (NB + 1) / 2;
        const double s = 1.0 / dsqrt(max_res);
        v2f A2[NP], B2[NP], C2[NP], D2[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int u0 = 2 * q, u1 = (2 * q + 1 < NB) ? 2 * q + 1 : 2 * q;
            A2[q] = (v2f){(float)a[u0], (float)a[u1]};
            B2[q] = (v2f){(float)b[u0], (float)b[u1]};
            C2[q] = (v2f){(float)(c[u0] * s), (float)(c[u1] * s)};
            D2[q] = (v2f){(float)(d[u0] * s), (float)(d[u1] * s)};
        }
        // the last pair's halves: batch 2 (NP - 1) is full unless it is the last batch; batch 2 NP - 1 may not exist
        // (odd NB: that half holds a copy of the other batch and counts nothing)
        const unsigned long long v_lo = (2 * (NP - 1) == NB - 1) ? last_valid : ~0ull;
        const unsigned long long v_hi = (2 * NP - 1 <= NB - 1) ? ((2 * NP - 1 == NB - 1) ? last_valid : ~0ull) : 0ull;
        const H32Lane hl = h32_prepare(&mym[0], s, cmax);
        for (int t = 0; t < nT; ++t) {
            if (__builtin_amdgcn_readlane(nmod, t) < 1) continue;
            const int prev = __builtin_amdgcn_readlane(acc[0], t);
            int cc = Mb;
            if (prev + Mb + rest_after >= thr) {
                const int thr_b = thr - prev - rest_after;  // what this block has to contribute for the model to matter
                H32Model hm;  // trial t's scaled model and bound constants, wave-uniform
#pragma unroll
                for (int i = 0; i < 9; ++i) hm.m[i] = readlane_f32(hl.m[i], t);
                hm.kE = readlane_f32(hl.kE, t);
                hm.kW = readlane_f32(hl.kW, t);
                hm.K0 = readlane_f32(hl.K0, t);
                bool und;
                cc = count_h32<NP>(hm, A2, B2, C2, D2, v_lo, v_hi, Mb, thr_b, und);
#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 8)
                {
                    double sm[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mym[i], t);
                    const int c64 = count_regs<K_H, NB>(sm, a, b, c, d, last_valid, Mb, max_res, 0);
                    bool und2;
                    const int c32 = count_h32<NP>(hm, A2, B2, C2, D2, v_lo, v_hi, Mb, 0, und2);
                    if (lane == 0) {
                        atomicAdd(&g_h32_diag[0], 1ull);
                        if (und2) atomicAdd(&g_h32_diag[1], 1ull);
                        else if (c32 != c64) atomicAdd(&g_h32_diag[2], 1ull);
                    }
                }
#endif
                if (und) {
                    double sm[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mym[i], t);
                    cc = count_regs<K_H, NB>(sm, a, b, c, d, last_valid, Mb, max_res, thr_b);
                }
            }
            if (lane == t) acc[0] += cc;
        }
        return;
    }
    for (int t = 0; t < nT; ++t) {
        const int n = __builtin_amdgcn_readlane(nmod, t);
#pragma unroll
        for (int m = 0; m < NMX; ++m) {
            if (m >= n) continue;  // (uniform; `break` would keep the loop from unrolling and acc[m] from registers)
            const int prev = __builtin_amdgcn_readlane(acc[m], t);
            int cc = Mb;
            if (prev + Mb + rest_after >= thr) {
                double sm[9];
                if (KIND == K_E5) {
                    const double* src = models + ((size_t)t * kMaxModels + m) * 9;
#pragma unroll
                    for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(src[i], 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mym[9 * (NM == 1 ? 0 : m) + i], t);
                }
                cc = count_regs<(KIND == K_E5 ? K_F7 : KIND), NB>(sm, a, b, c, d, last_valid, Mb, max_res, thr - prev - rest_after);
            }
            if (lane == t) acc[m] += cc;
        }
    }
}
template <int KIND, int NM>
__device__ __forceinline__ void count_block_regs_nb(const double (&mym)[27], const double* models, int nmod, const Pts& P,
                                                    int off, int Mb, int rest_after, double max_res, int nT, int lane, int thr,
                                                    double cmax, int (&acc)[(KIND == K_E5 ? kMaxModels : NM)]) {
#define AMC_CB(NB_) case NB_: count_block_regs<KIND, NM, NB_>(mym, models, nmod, P, off, Mb, rest_after, max_res, nT, lane, thr, cmax, acc); break;
    switch ((Mb + 63) / 64) {
        AMC_CB(1) AMC_CB(2) AMC_CB(3) AMC_CB(4) AMC_CB(5) AMC_CB(6) AMC_CB(7) AMC_CB(8) AMC_CB(9) AMC_CB(10) AMC_CB(11)
        default: count_block_regs<KIND, NM, 12>(mym, models, nmod, P, off, Mb, rest_after, max_res, nT, lane, thr, cmax, acc); break;
    }
#undef AMC_CB
}
constexpr int kRegBlock = 768;  // 12 batches of 64: 96 VGPRs of points (the phase has 256 to itself)
// all M correspondences, kRegBlock at a time; returns (lane t) the largest count among trial t's models, each either
// exact or an upper bound below `thr`
template <int KIND, int NM>
__device__ __forceinline__ int count_chunk_regs_nb(const double (&mym)[27], const double* models, int nmod, const Pts& P,
                                                   int M, double max_res, int nT, int lane, int thr, double cmax) {
    constexpr int NMX = KIND == K_E5 ? kMaxModels : NM;
    int acc[NMX];
#pragma unroll
    for (int m = 0; m < NMX; ++m) acc[m] = 0;
    for (int off = 0; off < M; off += kRegBlock) {
        const int Mb = min(kRegBlock, M - off);
        count_block_regs_nb<KIND, NM>(mym, models, nmod, P, off, Mb, M - off - Mb, max_res, nT, lane, thr, cmax, acc);
    }
    int maxcnt = -1;
#pragma unroll
    for (int m = 0; m < NMX; ++m)
        if (m < nmod) maxcnt = max(maxcnt, acc[m]);
    return maxcnt;
}

__device__ __noinline__ void solve_chunk(ChunkModels* out, int est, const Pts P, const lds_u16* sidx, int nT,
                                         int lane, double* models) {
    const unsigned long long c0 = __builtin_readcyclecounter();
    int nmod = 0;
    double mym[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) mym[i] = 0.0;
    if (lane < nT) {
        if (est == K_F7) {
            double sx1[7], sy1[7], sx2[7], sy2[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) load_pt_any(P, sidx[lane * 8 + i], sx1[i], sy1[i], sx2[i], sy2[i]);
            nmod = estimate_f7(sx1, sy1, sx2, sy2, mym);
        } else if (est == K_H) {
            double sx1[4], sy1[4], sx2[4], sy2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) load_pt_any(P, sidx[lane * 8 + i], sx1[i], sy1[i], sx2[i], sy2[i]);
            estimate_h4(sx1, sy1, sx2, sy2, mym);
            nmod = 1;
        } else if (est == K_E5) {
            double sx1[5], sy1[5], sx2[5], sy2[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) load_pt_any(P, sidx[lane * 8 + i], sx1[i], sy1[i], sx2[i], sy2[i]);
            nmod = estimate_e5_minimal(sx1, sy1, sx2, sy2, models + (size_t)lane * kMaxModels * 9);
        } else {  // K_T: model = dst - src of the single sample
            double a, b, c, d;
            load_pt_any(P, sidx[lane * 8], a, b, c, d);
            mym[0] = c - a;
            mym[1] = d - b;
            nmod = 1;
        }
    }
    wave_mem_sync();
#pragma unroll
    for (int i = 0; i < 27; ++i) out->mym[i] = mym[i];
    out->nmod = nmod;
    out->cyc_solve = __builtin_readcyclecounter() - c0;
}

__device__ __noinline__ void count_chunk(ChunkModels* io, int est_, const Pts P_, int M_, double max_res_, int nT_,
                                         int lane, const double* models_, int thr_, int fast_, double cmax_) {
    const unsigned long long c1 = __builtin_readcyclecounter();
    const int est = uni(est_), M = uni(M_), nT = uni(nT_), thr = uni(thr_);
    const bool fast = uni(fast_) != 0;
    const double max_res = uni(max_res_), cmax = uni(cmax_);
    const Pts P = uni(P_);
    const double* models = uni_ptr(models_);
    double mym[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) mym[i] = io->mym[i];
    const int nmod = io->nmod;
    int maxcnt;
    // the points of nearly every pair fit the LDS share; the global-memory path keeps the exact test only
    if (fast && M >= 1 && est != K_T) {
        if (est == K_F7) maxcnt = count_chunk_regs_nb<K_F7, 3>(mym, models, nmod, P, M, max_res, nT, lane, thr, cmax);
        else if (est == K_H) maxcnt = count_chunk_regs_nb<K_H, 1>(mym, models, nmod, P, M, max_res, nT, lane, thr, cmax);
        else maxcnt = count_chunk_regs_nb<K_E5, 1>(mym, models, nmod, P, M, max_res, nT, lane, thr, cmax);
    } else if (P.lds && fast) {
        if (est == K_F7) maxcnt = count_lane_models<true, K_F7, 3, true>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_H) maxcnt = count_lane_models<true, K_H, 1, true>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_T) maxcnt = count_lane_models<true, K_T, 1, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else maxcnt = count_global_models<true, true>(models, nmod, P, M, max_res, nT, lane, thr);
    } else if (P.lds) {
        if (est == K_F7) maxcnt = count_lane_models<true, K_F7, 3, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_H) maxcnt = count_lane_models<true, K_H, 1, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_T) maxcnt = count_lane_models<true, K_T, 1, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else maxcnt = count_global_models<true, false>(models, nmod, P, M, max_res, nT, lane, thr);
    } else {
        if (est == K_F7) maxcnt = count_lane_models<false, K_F7, 3, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_H) maxcnt = count_lane_models<false, K_H, 1, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else if (est == K_T) maxcnt = count_lane_models<false, K_T, 1, false>(mym, nmod, P, M, max_res, nT, lane, thr);
        else maxcnt = count_global_models<false, false>(models, nmod, P, M, max_res, nT, lane, thr);
    }
    io->maxcnt = maxcnt;
    io->cyc_count = __builtin_readcyclecounter() - c1;
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
    const double* wm_cut;    // K_T only: inlier-ratio cut-offs by trial count (TvgParams::wm_cut), max_trials + 1 entries
    int force_slow_sampler;  // test hook: always take the draw-by-draw sampler path
    int no_fast_count;       // test hook (AMC_TVG_EXACT_COUNT=1): the counting loop evaluates the reference residual only
};

// LORANSAC<est, local_est>::Estimate over the M correspondences in the four arrays at gx (x1 | y1
// | x2 | y2, each gstride long); mask: M bytes
__device__ Report lo_ransac(Wave& w_io, const RansacCfg& cfg, const double* gx, uint32_t gstride, int M,
                            uint8_t* mask) {
    Wave w = w_io;  // by-value copy: the fields live in registers, not behind a pointer
    const int lane = w.lane;
    const int kMin = kmin_of(cfg.est), kLocalMin = kmin_of(cfg.local_est);
    LoCtx lo;
    lo.inl = w.inl; lo.jacA = w.jacA; lo.jacV = w.jacV; lo.lane = lane;
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
    Pts P;
    P.g = gx; P.gs = gstride; P.l = w.lpts; P.ls = w.pts_cap;
    P.lds = (uint32_t)M <= w.pts_cap;
    int fast_count = 0;
    double cmax = 0.0;  // largest |coordinate| of the pair's correspondences
    {
        double amax = 0.0;
        for (int k = lane; k < M; k += 64) {
            const double p0 = gx[k], p1 = gx[gstride + k], p2 = gx[2 * (size_t)gstride + k], p3 = gx[3 * (size_t)gstride + k];
            if (P.lds) {
                w.lpts[k] = p0;
                w.lpts[w.pts_cap + k] = p1;
                w.lpts[2 * w.pts_cap + k] = p2;
                w.lpts[3 * w.pts_cap + k] = p3;
            }
            amax = dmax(dmax(amax, dmax(dabs(p0), dabs(p1))), dmax(dabs(p2), dabs(p3)));
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) amax = dmax(amax, __shfl_xor(amax, sh));
        // the division-free counting test is trusted only while (largest coordinate / max_error) <= 1e5 (see
        // fast_inlier); a NaN coordinate leaves amax as it was or NaN - either way the comparison below decides
        fast_count = (cfg.no_fast_count == 0 && amax * amax <= 1e10 * cfg.max_res) ? 1 : 0;
        cmax = amax;
    }

    // sampler.Initialize(M).  The first kMin entries of the persistent permutation are touched by
    // every draw: they live in (wave-uniform) registers, the rest in LDS.
    for (int k = lane; k < M; k += 64) w.perm[k] = (uint16_t)k;
    SamplerState ss;
#pragma unroll
    for (int i = 0; i < 7; ++i) ss.pr[i] = (uint32_t)i;
    wave_lds_sync();

    double* models = w.models();
    bool aborted = false;
    int abort_trial = -1;
    for (int chunk = 0; chunk < cfg.max_trials && !aborted; chunk += 64) {
        const int nT = min(64, cfg.max_trials - chunk);
        // ---- snapshot the generator, draw the chunk's samples (wave-uniform, sequential) ----
        for (int i = lane; i < 624; i += 64) w.snap[i] = w.mt[i];
        const int snap_mti = w.mti;
        wave_mem_sync();  // the snapshot is read back by other lanes (fallback sampler, rollback)
        wave_lds_sync();
        unsigned long long tp0 = __builtin_readcyclecounter();
        ss.mti = w.mti;
        ss = sample_chunk(w.mt, w.snap, w.perm, w.sidx, w.rawcnt, ss, M, kMin, nT, lane, cfg.force_slow_sampler);
        w.mti = ss.mti;
        { const unsigned long long tp1 = __builtin_readcyclecounter(); w.prof[0] += tp1 - tp0; tp0 = tp1; }
        // ---- 64 minimal problems + the inlier count of every model (solve_count_chunk) ---------
        ChunkModels cm;
        solve_chunk(&cm, cfg.est, P, w.sidx, nT, lane, models);
        count_chunk(&cm, cfg.est, P, M, cfg.max_res, nT, lane, models, best.cnt, fast_count, cmax);
        w.prof[1] += cm.cyc_solve;
        w.prof[5] += cm.cyc_count;  // the counting loop alone (also part of prof[2])
        tp0 = __builtin_readcyclecounter();
        // ---- replay in trial order.  Only two kinds of trial can change anything: one holding a
        //      model whose count reaches the best so far (candidate: re-scored in full, exactly as
        //      the sequential loop would), and the first trial with a model at or beyond the
        //      adaptive trial limit (abort).  Everything in between is skipped.
        const unsigned long long live = nT == 64 ? ~0ull : ((1ull << nT) - 1ull);
        int t = 0;
        while (!aborted) {
            const long long lim = (long long)(dyn_max > (uint32_t)cfg.min_trials ? dyn_max : (uint32_t)cfg.min_trials) - chunk;
            const unsigned long long cand = __ballot(cm.nmod > 0 && cm.maxcnt >= best.cnt);
            const unsigned long long stop = __ballot(cm.nmod > 0 && (long long)lane >= lim);
            const unsigned long long ev = (cand | stop) & live & (t >= 64 ? 0ull : (~0ull << t));
            if (ev == 0ull) break;
            t = (int)__builtin_ctzll(ev);
            const int trial = chunk + t;
            const int n = __builtin_amdgcn_readlane(cm.nmod, t);
            for (int m = 0; m < n; ++m) {
                Model9 smv;
                double* sm = smv.v;
                if (cfg.est == K_E5) {
                    const double* src = models + ((size_t)t * kMaxModels + m) * 9;
                    for (int i = 0; i < 9; ++i) sm[i] = src[i];
                } else {
                    for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(cm.mym[9 * m + i], t);
                }
                const Support sup = score(cfg.est, smv, P, M, cfg.max_res, lane, best.cnt);
                if (better(sup, best)) {
                    const unsigned long long tl0 = __builtin_readcyclecounter();
                    best = sup;
                    for (int i = 0; i < 9; ++i) best_model[i] = sm[i];
                    best_is_local = false;
                    if (sup.cnt > kMin && sup.cnt >= kLocalMin) {
                        // recursive local optimisation: inliers of the sample model first, then of
                        // the improved local model (COLMAP swaps residual vectors to the same effect)
                        int cur_kind = cfg.est;
                        Model9 cur;
                        for (int i = 0; i < 9; ++i) cur.v[i] = sm[i];
                        for (int lt = 0; lt < 10; ++lt) {
                            const int K = extract_inliers(w.inl, lane, cur_kind, cur, P, M, cfg.max_res);
                            double lm[kMaxModels * 9];
                            const unsigned long long tle = __builtin_readcyclecounter();
                            const int nl = local_estimate(lo, cfg.local_est, P, K, lm);
                            if (lane == 0) {
                                w.work[wk_residual_slot(cfg.local_est)] += (unsigned long long)nl * (unsigned long long)M;
                                if (cfg.local_est == K_E5) w.work[WK_LO_E5] += 1;
                                else if (cfg.local_est == K_F8) w.work[WK_LO_F8] += 1;
                                else if (cfg.local_est == K_H) w.work[WK_LO_H] += 1;
                                w.work[WK_LO_POINTS] += (unsigned long long)K;
                            }
                            if (cfg.local_est == K_E5) w.prof[6] += __builtin_readcyclecounter() - tle;
                            else if (cfg.local_est == K_F8) w.prof[7] += __builtin_readcyclecounter() - tle;
                            const int prev = best.cnt;
                            for (int q = 0; q < nl; ++q) {
                                Model9 lmv;
                                for (int i = 0; i < 9; ++i) lmv.v[i] = lm[9 * q + i];
                                const Support ls = score(cfg.local_est, lmv, P, M, cfg.max_res, lane, best.cnt);
                                if (better(ls, best)) {
                                    best = ls;
                                    for (int i = 0; i < 9; ++i) best_model[i] = lm[9 * q + i];
                                    best_is_local = true;
                                }
                            }
                            if (best.cnt <= prev) break;
                            cur_kind = cfg.local_est;
                            for (int i = 0; i < 9; ++i) cur.v[i] = best_model[i];
                        }
                    }
                    if (cfg.dyn_tab) {
                        dyn_max = cfg.dyn_tab[best.cnt];
                    } else if (cfg.wm_cut) {
                        // first T in [0, max_trials] with r >= wm_cut[T] (the cut-offs do not increase with T)
                        const double r = (double)best.cnt / (double)M;
                        int lo_t = 0, hi_t = cfg.max_trials + 1;  // answer in [lo_t, hi_t]; hi_t = none
                        while (lo_t < hi_t) {
                            const int mid = (lo_t + hi_t) >> 1;
                            if (r >= cfg.wm_cut[mid]) hi_t = mid; else lo_t = mid + 1;
                        }
                        dyn_max = lo_t <= cfg.max_trials ? (uint32_t)lo_t : 0xFFFFFFFFu;
                    } else {
                        dyn_max = 0xFFFFFFFFu;
                    }
                    w.prof[3] += __builtin_readcyclecounter() - tl0;
                }
                if ((uint32_t)trial >= dyn_max && trial >= cfg.min_trials) {
                    aborted = true;
                    abort_trial = trial;
                    break;
                }
            }
            ++t;
        }
        w.prof[2] += cm.cyc_count;
        { const unsigned long long tp1 = __builtin_readcyclecounter(); w.prof[2] += tp1 - tp0; }
        {   // algorithmic work of the chunk: the trials the sequential loop ran, their models x M residuals
            const int upto = aborted ? abort_trial - chunk : nT - 1;
            const int nmodels = wave_sum_int(lane <= upto ? cm.nmod : 0);
            if (lane == 0) {
                w.work[wk_residual_slot(cfg.est)] += (unsigned long long)nmodels * (unsigned long long)M;
                w.work[cfg.est == K_E5 ? WK_E5MIN : (cfg.est == K_F7 ? WK_F7MIN : (cfg.est == K_H ? WK_H4MIN : WK_TRIALS))] +=
                    (unsigned long long)(upto + 1);
            }
        }
        if (aborted) {
            // roll the generator back to where the sequential algorithm stopped drawing
            wave_lds_sync();
            for (int i = lane; i < 624; i += 64) w.mt[i] = w.snap[i];
            w.mti = snap_mti;
            wave_lds_sync();
            int consumed = (int)sgpr(w.rawcnt[abort_trial - chunk]);
            while (consumed > 0) {  // discard `consumed` raw words
                if (w.mti >= 624) { mt_twist(w.mt, lane); w.mti = 0; }
                const int step = min(624 - w.mti, consumed);
                w.mti += step;
                consumed -= step;
            }
        }
    }
    // report.num_trials exactly as the for/abort dance of loransac.h leaves it
    rep.num_trials = aborted ? ((abort_trial + 1 < cfg.max_trials) ? abort_trial + 2 : abort_trial + 1)
                             : cfg.max_trials;
    rep.support = best;
    for (int i = 0; i < 9; ++i) rep.model[i] = best_model[i];
    w_io.mti = w.mti;
    for (int i = 0; i < 8; ++i) w_io.prof[i] = w.prof[i];
    if (best.cnt >= kMin && lane == 0)
        w.work[wk_residual_slot(best_is_local ? cfg.local_est : cfg.est)] += (unsigned long long)M;
    if (best.cnt < kMin) return rep;
    rep.success = true;
    const int fk = best_is_local ? cfg.local_est : cfg.est;
    for (int k = lane; k < M; k += 64) {
        double a, b, c, d;
        load_pt_any(P, k, a, b, c, d);
        mask[k] = residual_of(fk, rep.model, a, b, c, d) <= cfg.max_res ? 1 : 0;
    }
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
    for (int i = 0; i < 8; ++i) w.prof[i] = 0;
    const TvgPair pr = pairs[q];
    const uint32_t oq = pr.orig;  // results are stored by the caller's pair index, whatever the queue order
    w.work = out[oq].work;
    if (lane == 0)
        for (int i = 0; i < 12; ++i) w.work[i] = 0;
    const unsigned long long tstart = __builtin_readcyclecounter();
    // the image records are read field by field where they are needed (wave-uniform scalar loads): a by-value
    // copy of both would hold 2 x 38 dwords of camera parameters in scalar registers for the whole pair
    const TvgImage* __restrict__ pim1 = imgs + pr.slot1;
    const TvgImage* __restrict__ pim2 = imgs + pr.slot2;
    const int M = (int)pr.M;
    amc_tvg g;
    g.config = AMC_TVG_UNDEFINED;
    g.num_inliers = 0;
    for (int i = 0; i < 9; ++i) { g.E[i] = 0; g.F[i] = 0; g.H[i] = 0; }
    for (int i = 0; i < 4; ++i) g.num_trials[i] = 0;
    for (int i = 0; i < 3; ++i) g.model_inliers[i] = 0;
    uint8_t* omask = out_mask + pr.mask_off;
    for (int k = lane; k < M; k += 64) omask[k] = 0;

    if (P.mode == 0 && M < P.min_num_inliers) {
        g.config = AMC_TVG_DEGENERATE;
        if (lane == 0) out[oq].g = g;
        return;
    }
    // ---- matched points (FeatureKeypointsToPointsVector: float -> double) ------------------
    double *X1 = w.arr(W_X1), *Y1 = w.arr(W_Y1), *X2 = w.arr(W_X2), *Y2 = w.arr(W_Y2);
    const uint32_t* mm = matches + 2 * pr.match_off;
    // The indices are checked here, where they are read anyway (the host only checks the pairs that
    // return above): a pair with a match past an image's keypoints is counted and not estimated - the
    // host then fails the whole call with AMC_E_INVALID, naming the match.
    bool bad = false;
    {
        const float* __restrict__ kp1 = pim1->kp;
        const float* __restrict__ kp2 = pim2->kp;
        const double* __restrict__ kd1 = pim1->kp64;
        const double* __restrict__ kd2 = pim2->kp64;
        const uint32_t rows1 = pim1->rows, rows2 = pim2->rows;
        for (int k = lane; k < M; k += 64) {
            const uint32_t i1 = mm[2 * k], i2 = mm[2 * k + 1];
            if (i1 >= rows1 || i2 >= rows2) {
                bad = true;
                continue;
            }
            X1[k] = kd1 ? kd1[2 * (size_t)i1] : (double)kp1[2 * (size_t)i1];
            Y1[k] = kd1 ? kd1[2 * (size_t)i1 + 1] : (double)kp1[2 * (size_t)i1 + 1];
            X2[k] = kd2 ? kd2[2 * (size_t)i2] : (double)kp2[2 * (size_t)i2];
            Y2[k] = kd2 ? kd2[2 * (size_t)i2 + 1] : (double)kp2[2 * (size_t)i2 + 1];
        }
    }
    if (__any(bad)) {
        g.config = AMC_TVG_UNDEFINED;
        if (lane == 0) {
            atomicAdd(P.bad_index_count, 1u);
            out[oq].g = g;
        }
        return;
    }
    // ---- SetPRNGSeed(seed): generator state as std::mt19937(seed) leaves it ---------------
    for (int i = lane; i < 624; i += 64) w.mt[i] = mt_init[i];
    w.mti = 624;
    wave_mem_sync();

    // mode 0: the EstimateTwoViewGeometry dispatch; modes 1 / 2 / 3: exactly one of F / H / E
    const bool calibrated = P.mode == 0 ? (!P.force_H_use && pim1->cam.has_prior && pim2->cam.has_prior) : P.mode == 3;
    const bool run_F = P.mode == 0 ? !P.force_H_use : P.mode == 1;
    const bool run_H = P.mode == 0 || P.mode == 2;
    uint8_t *maskE = w.masks, *maskF = w.masks + mcap, *maskH = w.masks + 2 * (size_t)mcap;
    Report E_rep, F_rep, H_rep;
    E_rep.success = F_rep.success = H_rep.success = false;
    E_rep.support.cnt = F_rep.support.cnt = H_rep.support.cnt = 0;
    E_rep.num_trials = F_rep.num_trials = 0;
    for (int i = 0; i < 9; ++i) { E_rep.model[i] = 0; F_rep.model[i] = 0; }

    RansacCfg cfg;
    cfg.wm_cut = nullptr;
    cfg.min_trials = P.min_num_trials;
    cfg.force_slow_sampler = P.force_slow_sampler;
    cfg.no_fast_count = P.no_fast_count;
    if (calibrated) {
        double *N1x = w.arr(W_NX1), *N1y = w.arr(W_NY1), *N2x = w.arr(W_NX2), *N2y = w.arr(W_NY2);
        // Camera::CamFromImg of the matched points.  SIMPLE_PINHOLE / PINHOLE: (x - c) / f in place.  Cameras
        // with distortion parameters: the keypoints were lifted once per image (kpn), gather from there.
        {
            const int model1 = pim1->cam.model_id, model2 = pim2->cam.model_id;
            const double* __restrict__ kn1 = pim1->kpn;
            const double* __restrict__ kn2 = pim2->kpn;
            const int nf1 = cam::num_focal(model1), nf2 = cam::num_focal(model2);
            const double f1x = pim1->cam.params[0], f1y = pim1->cam.params[nf1 - 1];
            const double c1x = pim1->cam.params[nf1], c1y = pim1->cam.params[nf1 + 1];
            const double f2x = pim2->cam.params[0], f2y = pim2->cam.params[nf2 - 1];
            const double c2x = pim2->cam.params[nf2], c2y = pim2->cam.params[nf2 + 1];
            for (int k = lane; k < M; k += 64) {
                if (kn1) {
                    const uint32_t i1 = mm[2 * k];
                    N1x[k] = kn1[2 * (size_t)i1];
                    N1y[k] = kn1[2 * (size_t)i1 + 1];
                } else {
                    N1x[k] = (X1[k] - c1x) / f1x;
                    N1y[k] = (Y1[k] - c1y) / f1y;
                }
                if (kn2) {
                    const uint32_t i2 = mm[2 * k + 1];
                    N2x[k] = kn2[2 * (size_t)i2];
                    N2y[k] = kn2[2 * (size_t)i2 + 1];
                } else {
                    N2x[k] = (X2[k] - c2x) / f2x;
                    N2y[k] = (Y2[k] - c2y) / f2y;
                }
            }
        }
        wave_mem_sync();
        // E threshold: (cam1.CamFromImgThreshold(e) + cam2.CamFromImgThreshold(e)) / 2
        const double e_err = (cam::cam_from_img_threshold(pim1->cam.model_id, pim1->cam.params, P.max_error) +
                              cam::cam_from_img_threshold(pim2->cam.model_id, pim2->cam.params, P.max_error)) / 2;
        cfg.est = K_E5; cfg.local_est = K_E5;
        cfg.max_res = e_err * e_err;
        cfg.max_trials = P.max_trials[0];
        cfg.dyn_tab = trial_tabs + pr.tab_off[0];
        E_rep = lo_ransac(w, cfg, N1x, mcap, M, maskE);
        for (int i = 0; i < 9; ++i) g.E[i] = E_rep.model[i];
        g.num_trials[0] = E_rep.num_trials;
        g.model_inliers[0] = E_rep.support.cnt;
    }
    if (run_F) {
        cfg.est = K_F7; cfg.local_est = K_F8;
        cfg.max_res = P.max_error * P.max_error;
        cfg.max_trials = P.max_trials[1];
        cfg.dyn_tab = trial_tabs + pr.tab_off[1];
        F_rep = lo_ransac(w, cfg, X1, mcap, M, maskF);
        for (int i = 0; i < 9; ++i) g.F[i] = F_rep.model[i];
        g.num_trials[1] = F_rep.num_trials;
        g.model_inliers[1] = F_rep.support.cnt;
    }
    H_rep.num_trials = 0;
    for (int i = 0; i < 9; ++i) H_rep.model[i] = 0;
    if (run_H) {
        cfg.est = K_H; cfg.local_est = K_H;
        cfg.max_res = P.max_error * P.max_error;
        cfg.max_trials = P.max_trials[2];
        cfg.dyn_tab = trial_tabs + pr.tab_off[2];
        H_rep = lo_ransac(w, cfg, X1, mcap, M, maskH);
        for (int i = 0; i < 9; ++i) g.H[i] = H_rep.model[i];
        g.num_trials[2] = H_rep.num_trials;
        g.model_inliers[2] = H_rep.support.cnt;
    }
    if (P.mode != 0) {
        // single-RANSAC report: config carries report.success, the mask is report.inlier_mask
        const Report& r = P.mode == 1 ? F_rep : (P.mode == 2 ? H_rep : E_rep);
        const uint8_t* rm = P.mode == 1 ? maskF : (P.mode == 2 ? maskH : maskE);
        g.config = r.success ? 1 : 0;
        g.num_inliers = r.support.cnt;
        if (r.success)
            for (int k = lane; k < M; k += 64) omask[k] = rm[k];
        if (lane == 0) {
            out[oq].g = g;
            w.prof[4] = __builtin_readcyclecounter() - tstart;
            for (int i = 0; i < 8; ++i) out[oq].prof[i] = w.prof[i];
        }
        return;
    }

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
            const uint64_t w1 = pim1->cam.width, h1 = pim1->cam.height, w2 = pim2->cam.width, h2 = pim2->cam.height;
            const double diagonal1 = dsqrt((double)(w1 * w1 + h1 * h1));
            const double diagonal2 = dsqrt((double)(w2 * w2 + h2 * h2));
            const double minx1 = P.watermark_border_size * diagonal1, miny1 = minx1;
            const double maxx1 = (double)w1 - minx1, maxy1 = (double)h1 - miny1;
            const double minx2 = P.watermark_border_size * diagonal2, miny2 = minx2;
            const double maxx2 = (double)w2 - minx2, maxy2 = (double)h2 - miny2;
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
                cfg.dyn_tab = nullptr;
                cfg.wm_cut = P.wm_cut;
                const Report T_rep = lo_ransac(w, cfg, ix1, mcap, num_inliers, w.masks + 3 * (size_t)mcap);
                g.num_trials[3] = T_rep.num_trials;
                const double inlier_ratio = (double)T_rep.support.cnt / (double)num_inliers;
                if (inlier_ratio >= P.watermark_min_inlier_ratio) g.config = AMC_TVG_WATERMARK;
            }
        }
    }
    if (lane == 0) {
        out[oq].g = g;
        w.prof[4] = __builtin_readcyclecounter() - tstart;
        for (int i = 0; i < 8; ++i) out[oq].prof[i] = w.prof[i];
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
    const size_t lds_per_wave = tvg_lds_per_wave(mcap, pts_cap);
    AMC_LDS char* base = (AMC_LDS char*)smem + (size_t)wid * lds_per_wave;
    Wave w;
    w.lane = lane;
    w.jacA = reinterpret_cast<lds_f64*>(base);
    w.jacV = w.jacA + 81;
    w.lpts = w.jacA + 162;
    w.pts_cap = pts_cap;
    w.mt = reinterpret_cast<lds_u32*>(base + (162 + (size_t)4 * pts_cap) * 8);
    w.rawcnt = w.mt + 624;
    w.sidx = reinterpret_cast<lds_u16*>(w.rawcnt + 64);
    w.perm = w.sidx + 64 * 8;
    w.inl = w.perm + (mcap + 7) / 8 * 8;
    w.mcap = mcap;
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wid;
    w.ws = ws_all + gw * tvg_ws_doubles(mcap);
    w.snap = reinterpret_cast<uint32_t*>(w.ws + (size_t)W_NUM_ARRAYS * mcap + kModelDoubles);
    w.masks = mask_ws_all + gw * tvg_ws_bytes_extra(mcap);

    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(queue_head, 1u);
        q = __shfl(q, 0);
        if (q >= npairs) break;
        process_pair(w, q, imgs, pairs, matches, trial_tabs, mt_init, P, out, out_mask);
    }
}

#if defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 2)
void tvg_diag_report() {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tvg_diag), sizeof h) != hipSuccess) return;
    std::fprintf(stderr, "[amc tvg diag] models counted %llu; cycles per model: 256-point groups %.0f, rest of count_model %.0f, "
                 "model broadcast %.0f, (timer pair %.0f)\n", h[4], (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4],
                 (double)h[3] / (h[4] ? h[4] : 1));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tvg_diag), z, sizeof z);
}
#elif defined(AMC_TVG_DIAG) && (AMC_TVG_DIAG & 8)
// diagnostic build 8: every homography model is counted by the FP32 pre-filter AND by the FP64 path
void tvg_diag_report() {
    unsigned long long h[4];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_h32_diag), sizeof h) != hipSuccess) return;
    std::fprintf(stderr, "[amc tvg diag] homography models through the FP32 pre-filter %llu: undecided %llu (%.4f %%), decided "
                 "with a count different from the FP64 path: %llu\n", h[0], h[1], h[0] ? 100.0 * (double)h[1] / (double)h[0] : 0.0, h[2]);
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_h32_diag), z, sizeof z);
}
#else
void tvg_diag_report() {}
#endif
size_t tvg_ws_doubles_host(uint32_t mcap) { return tvg_ws_doubles(mcap); }
size_t tvg_ws_mask_bytes_host(uint32_t mcap) { return tvg_ws_bytes_extra(mcap); }

size_t tvg_lds_bytes(uint32_t mcap, uint32_t pts_cap, int waves) {
    return (size_t)waves * tvg_lds_per_wave(mcap, pts_cap);
}
// how many correspondences of the active RANSAC fit in LDS next to everything else
uint32_t tvg_pts_cap(uint32_t mcap, int waves_per_block) {
    const size_t budget = 160 * 1024 / ((size_t)waves_per_block * kTvgWavesPerSimd);
    const size_t other = tvg_lds_bytes(mcap, 0, 1) + 64;
    if (other >= budget) return 0;
    const size_t cap = (budget - other) / 32 / 2 * 2;  // 4 arrays of doubles, kept 16-byte aligned
    return (uint32_t)(cap < mcap ? cap : mcap);
}

// ComputeSquaredSampsonError over n correspondences (points n x 2, E row-major)
__global__ __launch_bounds__(256) void sampson_kernel(const double* __restrict__ p1, const double* __restrict__ p2,
                                                      size_t n, const double* __restrict__ E9,
                                                      double* __restrict__ out) {
    double e[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = E9[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = sampson(e, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
}
hipError_t launch_sampson(const double* p1, const double* p2, size_t n, const double* E9, double* out,
                          hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(sampson_kernel, dim3(blocks), dim3(256), 0, s, p1, p2, n, E9, out);
    return hipGetLastError();
}

hipError_t launch_tvg(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs,
                      const uint32_t* matches, const uint32_t* trial_tabs, const uint32_t* mt_init,
                      const TvgParams& P, double* ws, uint8_t* mask_ws, uint32_t mcap,
                      uint32_t num_waves, int waves_per_block, uint32_t* queue_head, TvgOut* out,
                      uint8_t* out_mask, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    const uint32_t blocks = (num_waves + waves_per_block - 1) / waves_per_block;
    const uint32_t pts_cap = tvg_pts_cap(mcap, waves_per_block);
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
