// tvg_core.h — device code of the batched two-view geometric verification (COLMAP EstimateTwoViewGeometry,
// SURVEY.md A.3) on gfx950: one LO-RANSAC (LORANSAC<Est, LocalEst>::Estimate) by one wavefront.  Included by the two
// kernels that run it, each with its own register budget and occupancy:
//   tvg_e.hip   the essential-matrix RANSAC (5-point solver and its degree-10 root finder: 256 VGPRs, 2 waves per SIMD)
//   tvg_fh.hip  the fundamental-matrix and homography RANSACs, model selection, watermark test (AMC_FH_WAVES = 3 per SIMD)
// (one translation unit each: a device function shared by kernels with different occupancy attributes is compiled
// for the loosest of them).
//
// Mapping.  A persistent grid of wavefronts pulls image pairs from a queue; ONE WAVE owns one pair at a time (no
// workgroup barriers anywhere), and inside the wave the 64 lanes are
//   * 64 RANSAC trials for the minimal solvers (one trial per lane, tvg_math.h; the 5-point solve's 10 x 20 elimination
//     four lanes per trial, its root finding dealt out by sign-change bracket),
//   * 64 MODELS for inlier counting: every lane keeps one model of the chunk in registers and the correspondences
//     stream past through the scalar unit (s_load_dwordx16 batches from a per-wave table -> SGPR operands of packed-FP32
//     FMAs): no ballots, no broadcasts, no cross-lane traffic in the inner loop.  What a lane counts is an UPPER BOUND
//     of its model's inliers (the correspondences that are not outliers beyond the evaluation's own error bound),
//   * 64 strided correspondences for exact residual scoring, normalisation sums and A^T A accumulation (the fixed
//     64-way strided + butterfly order of the oracle's det_sum64, which is exactly what a wave computes with
//     __shfl_xor),
//   * cooperating workers on the single dense problems of the local-optimisation step: the disjoint rotations of a
//     Jacobi round, the sign-change brackets of a root-finding level.
// RANSAC is sequential by definition (the adaptive trial count depends on the best model so far); what is
// data-INdependent is the sample stream.  std::mt19937(seed) is the same sequence for every pair, so the host lays
// out its tempered words once (TvgParams::stream) and a RANSAC's generator state is just a position in that table:
// per 64-trial chunk the wave turns 64 x kMin words into the 64 samples (libstdc++'s Lemire uniform_int + the
// persistent partial Fisher-Yates permutation, reproduced exactly; sample_chunk), solves the 64 minimal problems in
// parallel (solve_chunk), counts the inliers of every resulting model (count_chunk), and then replays only the trials
// that can matter - a model whose count reaches the best so far, or the first trial at the adaptive limit - in trial
// order, re-scoring them in full (lo_ransac).  If a RANSAC stops inside a chunk, the position is set back to where
// the sequential algorithm would have stopped drawing, because the next RANSAC of the pair continues the stream.
//
// FP64 everywhere a decision is final (every model whose bound reaches the best count is re-scored with the reference
// residual), -ffp-contract=off, IEEE divide/sqrt: results are bit-identical to oracle/tvg_oracle.cc (inlier masks,
// configs, model bit patterns).
#pragma once
#include <algorithm>
#include <cstdio>
#include <vector>

#include "amc_internal.h"
#include "camera_math.h"
#include "tvg_math.h"

namespace amc {
namespace {
using namespace tvg;

enum : int { K_F7 = 0, K_F8 = 1, K_H = 2, K_T = 3, K_E5 = 4 };
__device__ __forceinline__ constexpr int kmin_of(int kind) {
    return kind == K_F7 ? 7 : kind == K_F8 ? 8 : kind == K_H ? 4 : kind == K_T ? 1 : 5;
}

constexpr int kMaxModels = 10;
constexpr int kModelDoubles = 64 * kMaxModels * 9;
// The chunk's model table (E / F: up to kMaxModels per trial) is element-major: entry i of model m of trial t at
// (m * 9 + i) * 64 + t.  Round 6 - it was (t * kMaxModels + m) * 9 + i, every lane writing its trial's models into
// 720 bytes of its own: a store instruction of the solving lanes touched 64 lines (5,760 partial-line writes per
// 5-point chunk, and the wave waited for them before the counting loop could read the table); element-major a store
// is 512 contiguous bytes.
#if defined(AMC_MODELS_BY_TRIAL)   // (A/B hook: the round-5 layout)
__device__ __forceinline__ constexpr size_t model_at(int m, int i, int t) { return ((size_t)t * kMaxModels + m) * 9 + i; }
#else
__device__ __forceinline__ constexpr size_t model_at(int m, int i, int t) { return (size_t)(m * 9 + i) * 64 + (size_t)t; }
#endif

// LDS objects are addressed through address-space-3 pointers so that every access is a ds_* instruction (a generic
// pointer makes the compiler emit flat_* loads, which take the vector-memory path and cost several hundred cycles).
// The per-pair cycle counters (TvgOut::prof, printed by the host with AMC_TVG_PROFILE=1) are compiled in only with
// -DAMC_TVG_PROF or in the -DAMC_TVG_LODIAG build (round 6): eight 64-bit counters live through the whole pair and a
// dozen clock reads per 64-trial chunk - each one a wait for every outstanding LDS and scalar access - cost registers
// (spills around the chunk's calls) in the build that ships.
#if defined(AMC_TVG_LODIAG) || defined(AMC_TVG_PROF)
#define AMC_TVG_PROF_ON 1
#else
#define AMC_TVG_PROF_ON 0
#endif
__device__ __forceinline__ unsigned long long prof_clock() {
#if AMC_TVG_PROF_ON
    return __builtin_readcyclecounter();
#else
    return 0ull;
#endif
}
#define AMC_LDS __attribute__((address_space(3)))
typedef AMC_LDS double lds_f64;
typedef AMC_LDS uint32_t lds_u32;
typedef AMC_LDS int32_t lds_i32;
typedef AMC_LDS uint16_t lds_u16;
// The two per-pair index arrays of mcap entries (the sampler's permutation, the inlier list).  They are what grows
// with a pair's match count: in LDS (2 x 2 bytes per match) a workgroup's 160 KB end at ~38,000 matches.  The "big"
// builds of the two kernels (tvg_e_big.hip / tvg_fh_big.hip: -DAMC_TVG_BIG) keep them in the wave's global workspace
// instead - slower (flat accesses), for the few pairs beyond that, up to the 65,535 the 16-bit indices can name.
#if defined(AMC_TVG_BIG)
typedef uint16_t idx_u16;
#else
typedef lds_u16 idx_u16;
#endif
// Wave-uniform reads of global tables through the scalar data cache (s_load_*): the table is either never written by
// the kernel (the sample stream) or written by this wave, drained and followed by s_dcache_inv (scalar_table_sync).
#define AMC_CONST __attribute__((address_space(4)))
// The waves' workspaces are global memory, but the pointers to them travel through structs and calls, where the compiler
// loses the address space and emits flat_* accesses (64-bit address arithmetic on the vector unit, and a flat access
// counts on the LDS counter too: every LDS wait then also waits for the outstanding loads from memory).  gptr() names
// the address space where a workspace is touched: global_load / global_store with a scalar base (AMC_TVG_FLAT=1: the
// round-6 code, for A/B runs).
#if defined(AMC_TVG_FLAT)
#define AMC_GLOBAL
#else
#define AMC_GLOBAL __attribute__((address_space(1)))
#endif
template <class T>
__device__ __forceinline__ AMC_GLOBAL T* gptr(T* p) { return (AMC_GLOBAL T*)p; }

// Algorithmic work of a pair, counted as the sequential algorithm does it (TvgOut::work): what COLMAP's loops
// evaluate - every model of every trial up to the stopping trial against all M correspondences, every local model
// against all M, one final residual pass per successful RANSAC - not what this kernel skips.
enum : int { WK_SAMPSON = 0, WK_HRES, WK_TRES, WK_E5MIN, WK_F7MIN, WK_H4MIN, WK_LO_E5, WK_LO_F8, WK_LO_H, WK_LO_POINTS,
              WK_TRIALS, WK_EXACT_FLOP, WK_COUNT };
__device__ __forceinline__ constexpr int wk_residual_slot(int kind) { return kind == K_H ? WK_HRES : (kind == K_T ? WK_TRES : WK_SAMPSON); }

// LDS scratch of the essential-matrix kernel's wave-balanced root finder (real_roots10_lanes)
struct RootScratch {
    lds_f64* coef;  // coefficient k of lane l's polynomial at coef[k * 64 + l]
    lds_f64* lo;    // kRootCap; a bracket's root comes back here
    lds_f64* hi;
    lds_f64* flo;
    lds_u16* src;   // the lane whose polynomial the bracket belongs to
};

struct Wave {
    int lane;
    unsigned long long prof[8];
    unsigned long long* work;  // the pair's TvgOut::work (global memory, lane 0 adds to it: a few times per 64 trials)
    // LDS
    lds_u16* sidx;    // 64 x 8: the chunk's samples
    lds_u32* rawcnt;  // 64: raw words consumed up to and including trial t of the chunk
    lds_i32* tmax;    // 64: largest count among trial t's models (count_models)
    lds_u16* mlist;   // 64 x kMaxModels: the chunk's models in (trial, root) order (count_models)
    idx_u16* perm;    // mcap: the sampler's persistent permutation
    idx_u16* inl;     // mcap: ordered inlier index list of the local-optimisation step
    lds_f64* jacA;    // 81: A^T A / eigenvalues (and scratch of the wave-wide 5-point solve)
    lds_f64* jacV;    // 81: eigenvectors
    // the sample stream: tempered words of std::mt19937(seed), soff = words consumed so far (wave-uniform)
    const uint32_t* stream;
    uint32_t stream_len;
    uint32_t soff;
    uint32_t* err;    // [0] += 1 when a RANSAC ran past the end of the stream table (the host retries with a longer one)
    RootScratch rootscr;  // essential-matrix kernel only
    // global workspace
    double* ws;        // W_NUM_ARRAYS x mcap point arrays | AoS table | models
    uint8_t* masks;    // 4 x mcap
    uint32_t mcap;
};
// per-wave global workspace: two point regions of mcap correspondences - the pair's matched points and the watermark
// test's inlier subset - as ARRAYS OF RECORDS, (x1, y1, x2, y2) in 32 bytes (round 6; four arrays of mcap doubles
// before).  The minimal solvers gather their samples from here, every lane another correspondence: a record is one
// 64-byte line per lane where the four arrays were four, and the regions of a CU's waves together (13.5 KB each at 420
// matches, 384 F/H waves per XCD) are far beyond the XCD's 4 MB of L2 - these gathers were most of the kernels' reads
// from the memory side (profiles/r06/pmc_tvg_r06.json) ...
enum : int { W_X1 = 0, W_Y1, W_X2, W_Y2, W_AX1, W_AY1, W_AX2, W_AY2, W_NUM_ARRAYS };
struct alignas(32) PtRec {
    double x1, y1, x2, y2;
};
typedef double pt4 __attribute__((ext_vector_type(4)));  // a record as one 32-byte value (loads through gptr())
// ... then the tables the counting loops read through the scalar cache: (x1, y1, x2, y2) as doubles, one 32-byte
// record per correspondence, and the packed-FP32 table of the homography pre-filter (16 bytes per correspondence,
// two correspondences interleaved: a0 a1 b0 b1 c0' c1' d0' d1'), then the chunk's models (E / F: kMaxModels per trial)
__host__ __device__ inline size_t tvg_ws_doubles(uint32_t mcap) {
    return (size_t)W_NUM_ARRAYS * mcap + (size_t)4 * mcap + (size_t)2 * mcap + kModelDoubles;
}
__device__ __forceinline__ double* ws_arr(const Wave& w, int a) { return w.ws + (size_t)a * w.mcap; }
// the two point regions (W_X1 .. W_Y2 and W_AX1 .. W_AY2 as blocks of 4 x mcap doubles: mcap records)
__device__ __forceinline__ PtRec* ws_pts(const Wave& w, int region) { return reinterpret_cast<PtRec*>(w.ws + (size_t)region * 4 * w.mcap); }
// the essential-matrix kernel's waves: the model region doubles as the staging area of the minimal solver's
// constraint matrices (64 problems x (10 x 20 matrix + 6 x 10 result), element-major: e5_eliminate_quads)
constexpr int kE5StageG = 200, kE5StageHl = 60;
constexpr int kE5StageDoubles = 64 * (kE5StageG + kE5StageHl);
__host__ __device__ inline size_t tvg_ws_doubles_e(uint32_t mcap) {
    return tvg_ws_doubles(mcap) - kModelDoubles + (kE5StageDoubles > kModelDoubles ? kE5StageDoubles : kModelDoubles);
}
__device__ __forceinline__ double* ws_p64(const Wave& w) { return w.ws + (size_t)W_NUM_ARRAYS * w.mcap; }
__device__ __forceinline__ float* ws_p32(const Wave& w) { return reinterpret_cast<float*>(w.ws + (size_t)(W_NUM_ARRAYS + 4) * w.mcap); }
__device__ __forceinline__ double* ws_models(const Wave& w) { return w.ws + (size_t)(W_NUM_ARRAYS + 6) * w.mcap; }
// LDS bytes of one wave: jacA + jacV | sidx | rawcnt | tmax | mlist | perm | inl
__host__ __device__ inline size_t tvg_idx_bytes(uint32_t mcap) { return (size_t)((mcap + 7) / 8 * 8) * 2 * 2; }  // perm + inl
__host__ __device__ inline size_t tvg_lds_per_wave_fixed() {
    const size_t per = (size_t)162 * 8 + 64 * 8 * 2 + 64 * 4 + 64 * 4 + 64 * kMaxModels * 2;
    return (per + 15) / 16 * 16;
}
__host__ __device__ inline size_t tvg_lds_per_wave(uint32_t mcap) {
#if defined(AMC_TVG_BIG)
    (void)mcap;
    return tvg_lds_per_wave_fixed();
#else
    return (tvg_lds_per_wave_fixed() + tvg_idx_bytes(mcap) + 15) / 16 * 16;
#endif
}
__host__ __device__ inline size_t tvg_ws_bytes_extra(uint32_t mcap) { return (size_t)4 * mcap; }  // 4 masks
// big builds: the index arrays, one region per wave behind all waves' point workspaces (8-byte aligned)
__host__ __device__ inline size_t tvg_idx_doubles(uint32_t mcap) { return (tvg_idx_bytes(mcap) + 7) / 8; }
// The essential-matrix kernel's minimal solves spread the root finder's brackets over the wave (real_roots10_lanes):
// per wave the level's coefficients of every lane (11 x 64 doubles) and one pass of bracket records
// (kRootCap x (lo, hi, f(lo)) + the owning lane).  Only the coefficients get LDS of their own, behind the common
// layout; the bracket records lie OVER parts of the common layout that hold nothing while the root finder runs
// (round 5: 8,960 -> 5,632 bytes per wave, so that a wave's share of a CU's 160 KB at three waves per SIMD - 13.3 KB -
// holds pairs of up to ~970 matches instead of none):
//   lo, hi, src over jacA | jacV | sidx   (the Jacobi scratch is the local optimisation's; the chunk's sample indices
//                                          were consumed by the constraint rows before the root finder starts)
//   flo         over tmax | mlist         (written by count_chunk, after solve_chunk)
// rawcnt, between them, stays: the sampler's draw counts are read again when a trial aborts the chunk.
constexpr int kRootCap = 128;  // = one pair of brackets per lane and pass
constexpr size_t kRootScratchBytes = (size_t)11 * 64 * 8;
static_assert((size_t)102 * 8 <= kRootScratchBytes, "the local 5-point solve's uniform data (kE5UniDoubles) fits the coefficient area");
__host__ __device__ inline size_t tvg_lds_per_wave_e(uint32_t mcap) { return tvg_lds_per_wave(mcap) + kRootScratchBytes; }
__device__ __forceinline__ RootScratch root_scratch_carve(AMC_LDS char* wave_base, AMC_LDS char* behind) {
    constexpr size_t kJac = (size_t)162 * 8, kSidx = (size_t)64 * 8 * 2, kRaw = 64 * 4, kTmax = 64 * 4, kMlist = (size_t)64 * kMaxModels * 2;
    static_assert((size_t)2 * kRootCap * 8 + kRootCap * 2 <= kJac + kSidx, "lo | hi | src fit in front of rawcnt");
    static_assert((size_t)kRootCap * 8 <= kTmax + kMlist, "flo fits behind rawcnt");
    RootScratch S;
    S.coef = reinterpret_cast<lds_f64*>(behind);
    S.lo = reinterpret_cast<lds_f64*>(wave_base);
    S.hi = S.lo + kRootCap;
    S.src = reinterpret_cast<lds_u16*>(S.hi + kRootCap);
    S.flo = reinterpret_cast<lds_f64*>(wave_base + kJac + kSidx + kRaw);
    return S;
}

__device__ __forceinline__ void wave_carve(Wave& w, AMC_LDS char* base, uint32_t mcap) {
    w.jacA = reinterpret_cast<lds_f64*>(base);
    w.jacV = w.jacA + 81;
    w.sidx = reinterpret_cast<lds_u16*>(base + 162 * 8);
    w.rawcnt = reinterpret_cast<lds_u32*>(w.sidx + 64 * 8);
    w.tmax = reinterpret_cast<lds_i32*>(w.rawcnt + 64);
    w.mlist = reinterpret_cast<lds_u16*>(w.tmax + 64);
#if !defined(AMC_TVG_BIG)
    w.perm = w.mlist + 64 * kMaxModels;
    w.inl = w.perm + (mcap + 7) / 8 * 8;
#endif
    w.mcap = mcap;
}
#if defined(AMC_TVG_BIG)
// the wave's index arrays in global memory (idx_ws: tvg_idx_doubles(mcap) doubles per wave)
__device__ __forceinline__ void wave_carve_idx(Wave& w, double* idx_ws, uint32_t mcap) {
    w.perm = reinterpret_cast<idx_u16*>(idx_ws);
    w.inl = w.perm + (mcap + 7) / 8 * 8;
}
#endif

// Global-memory hand-off between lanes of ONE wave (a lane reads what another lane of the same
// wave stored): drain this wave's stores, then keep the compiler from moving accesses across.
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// ... and when the reader is the scalar unit (AMC_CONST loads): the stores are drained to L2 (the vector L1 is
// write-through), then the scalar data cache - which does not snoop vector stores, and may hold lines of the previous
// pair's table at the same addresses - is invalidated
__device__ __forceinline__ void scalar_table_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the invalidation has completed before the next s_load issues
    __builtin_amdgcn_wave_barrier();
}
// LDS hand-off inside the wave: LDS operations of a wave complete in order, the barrier only
// stops the compiler from reordering across it
__device__ __forceinline__ void wave_lds_sync() {
#if defined(AMC_TVG_BIG)
    // the index arrays are global memory here: a lane reads what another lane of the wave stored
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
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
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
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
template <class T>
__device__ __forceinline__ AMC_LDS T* uni_lds(AMC_LDS T* p) {
    return (AMC_LDS T*)(uintptr_t)uni((uint32_t)(uintptr_t)p);
}
__device__ __forceinline__ idx_u16* uni_idx(idx_u16* p) {
#if defined(AMC_TVG_BIG)
    return uni_ptr(p);
#else
    return uni_lds(p);
#endif
}
template <class T>
__device__ __forceinline__ const AMC_CONST T* as_const_table(const T* p) {  // p wave-uniform
    return (const AMC_CONST T*)(unsigned long long)uni_ptr(p);
}

// ---- RandomSampler::Sample for a chunk of nT consecutive trials -----------------------------------
// The sample stream does not depend on the data, so a chunk's draws are produced ahead of the trials that use them.
// Per draw the sequential algorithm does j = uniform_int(i, M-1) and swap(perm[i], perm[j]).  Fast path: all
// nT*kMin words are turned into j by the lanes in parallel (Lemire's multiply-shift on the tempered word; the
// rejection branch `low < range` has probability range / 2^32 per draw); only the swaps stay sequential.  If any
// draw of the chunk needs the rejection branch the chunk is redone draw by draw from the same position, exactly as
// libstdc++ does it.  perm[0..kMin) lives in registers (pr), the rest in LDS.
struct SamplerState {
    uint32_t off;   // stream position
    uint32_t pr[7];
};
__device__ __forceinline__ uint32_t stream_word(const AMC_CONST uint32_t* stream, uint32_t len, uint32_t pos, bool& over) {
    over |= pos >= len;
    return stream[pos < len ? pos : len - 1];
}

// N consecutive plain trials of one slot of the permutation's head (sample_chunk_t): trial k reads perm[jk[k]] and
// stores there what trial k - 1 read (the head's value `a` for the first), which is also trial k - 1's sample - the read
// of a trial is in flight while the previous one's value is stored.  Returns the head's value after the run.
template <int N>
__device__ __forceinline__ uint32_t swap_run(idx_u16* perm, lds_u16* srow, const uint32_t (&jk)[8], uint32_t a) {
    uint32_t v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        v[k] = perm[jk[k]];
        perm[jk[k]] = (uint16_t)(k == 0 ? a : v[k - 1]);
        if (k > 0) srow[(k - 1) * 8] = (uint16_t)v[k - 1];
    }
    srow[(N - 1) * 8] = (uint16_t)v[N - 1];
    return v[N - 1];
}

template <int kMin>
__device__ __forceinline__ SamplerState sample_chunk_t(const uint32_t* stream, uint32_t slen, idx_u16* perm, lds_u16* sidx,
                                                       lds_u32* rawcnt, lds_u16* jt, SamplerState st, int M, int nT, int lane,
                                                       int force_slow, uint32_t* err) {
    const int need = nT * kMin;
    static_assert((size_t)7 * 64 * 2 <= (size_t)162 * 8, "the transposed draws fit over jacA | jacV");
    // ---- parallel: tempered word -> j, for the whole chunk ----
    bool slowflag = force_slow != 0;
    const bool fits = st.off + (uint32_t)need <= slen;
    if (fits) {
        for (int n0 = 0; n0 < need; n0 += 64) {
            const int n = n0 + lane;
            if (n < need) {
                const int t = n / kMin, i = n - t * kMin;
                const uint32_t range = (uint32_t)(M - i);
                const uint64_t product = (uint64_t)stream[st.off + (uint32_t)n] * (uint64_t)range;
                if ((uint32_t)product < range) slowflag = true;
                const uint16_t jv = (uint16_t)((uint32_t)i + (uint32_t)(product >> 32));
                sidx[t * 8 + i] = jv;
                jt[i * 64 + t] = jv;
            }
        }
    } else {
        slowflag = true;  // the table ends inside this chunk: the draw-by-draw path checks every word
    }
    wave_lds_sync();
    if (__ballot(slowflag) == 0ull) {
        // ---- sequential swaps ----
        // Lane i < kMin owns slot i of the permutation's head (its value: `a`); one trial is, per slot, v = perm[j],
        // perm[j] = a, a = v - independent across slots as long as the trial's j are distinct and none falls into
        // the head.  Those trials (classified up front, lane t looks at trial t) take one LDS read and two writes per
        // slot lane, and the only dependence from trial to trial is the read of t feeding the write of t + 1.
        // The others (a few per cent) run the scalar, slot-by-slot code on the gathered head.
        // Round 6: the slot lanes take their draws from the transposed copy jt[i * 64 + t] - eight trials in one
        // 16-byte read, every draw then a bit field of a register - instead of one v_readlane + select per draw and
        // trial from the registers of lane t: ~6 instructions per trial instead of ~30 (the loop is bound by
        // instruction issue beside the SIMD's other waves, not by the LDS: ~475 cycles per trial before).
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
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) a = lane == i ? st.pr[i] : a;
        const bool slot = lane < kMin;
        const lds_u16* jrow = jt + (slot ? lane : 0) * 64;
        int t_ = 0;
        while (t_ < nT) {
            // a run of plain trials: up to the next odd trial, the end of the chunk or of the 8-trial block of draws
            // (t, run: wave-uniform - named so, or the loop's control flow is compiled as vector branches)
            const int t = uni(t_);
            const unsigned long long rest = slowmask >> t;
            const int k0 = t & 7;
            int run_ = rest ? (int)__builtin_ctzll(rest) : 64;
            run_ = min(min(run_, nT - t), 8 - k0);
            const int run = uni(run_);
            t_ = t + (run > 0 ? run : 1);
            if (run > 0) {
                if (slot) {
                    // this slot's draws of trials [t - k0, t - k0 + 8), two per register; the run's first one moved to bit 0
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 qv = *reinterpret_cast<const AMC_LDS u32x4*>(jrow + (t - k0));
                    unsigned long long lo = ((unsigned long long)qv[1] << 32) | qv[0], hi = ((unsigned long long)qv[3] << 32) | qv[2];
                    const int sh = 16 * k0;
                    if (sh >= 64) { lo = hi >> (sh - 64); hi = 0ull; }
                    else if (sh > 0) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
                    const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = (uint32_t)hi, w3 = (uint32_t)(hi >> 32);
                    const uint32_t jk[8] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16, w2 & 0xFFFFu, w2 >> 16, w3 & 0xFFFFu, w3 >> 16};
                    lds_u16* srow = sidx + t * 8 + lane;
                    switch (run) {   // (one straight-line piece of code per length: the waits are then placed per access)
                        case 1: a = swap_run<1>(perm, srow, jk, a); break;
                        case 2: a = swap_run<2>(perm, srow, jk, a); break;
                        case 3: a = swap_run<3>(perm, srow, jk, a); break;
                        case 4: a = swap_run<4>(perm, srow, jk, a); break;
                        case 5: a = swap_run<5>(perm, srow, jk, a); break;
                        case 6: a = swap_run<6>(perm, srow, jk, a); break;
                        case 7: a = swap_run<7>(perm, srow, jk, a); break;
                        default: a = swap_run<8>(perm, srow, jk, a); break;
                    }
                }
                continue;
            }
            // trial t touches the head or draws an index twice: slot by slot on the gathered head
            uint32_t j[7], pr[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                j[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)jj[i], t) : 0xFFFFu;
                pr[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)a, i) : 0u;
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
            for (int i = 0; i < 7; ++i) a = (i < kMin && lane == i) ? pr[i] : a;
            if (slot) sidx[t * 8 + lane] = (uint16_t)a;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) st.pr[i] = i < kMin ? (uint32_t)__builtin_amdgcn_readlane((int)a, i) : st.pr[i];
        st.off += (uint32_t)need;
        wave_lds_sync();
        return st;
    }
    // ---- a draw hit the rejection branch (or the table is about to end): the chunk draw by draw ----
    const AMC_CONST uint32_t* cs = as_const_table(stream);
    uint32_t pos = st.off;
    uint32_t nraw = 0;
    bool over = false;
    for (int t = 0; t < nT; ++t) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i < kMin) {
                const uint32_t range = (uint32_t)(M - i);
                uint64_t product;
                uint32_t low;
                {
                    ++nraw;
                    product = (uint64_t)stream_word(cs, slen, pos++, over) * (uint64_t)range;
                    low = (uint32_t)product;
                }
                if (low < range) {
                    const uint32_t threshold = (0u - range) % range;
                    while (low < threshold && !over) {
                        ++nraw;
                        product = (uint64_t)stream_word(cs, slen, pos++, over) * (uint64_t)range;
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
    if (over && lane == 0) atomicAdd(err, 1u);
    st.off = pos;
    wave_lds_sync();
    return st;
}

#ifndef AMC_SAMPLE_INLINE
#define AMC_SAMPLE_INLINE __noinline__
#endif
template <int kMin>
__device__ AMC_SAMPLE_INLINE SamplerState sample_chunk(const uint32_t* stream_, uint32_t slen_, idx_u16* perm_, lds_u16* sidx_,
                                                  lds_u32* rawcnt_, lds_u16* jt_, SamplerState st, int M_, int nT_, int lane,
                                                  int force_slow_, uint32_t* err_) {
    // everything but `lane` is wave-uniform: move it to scalar registers
    const uint32_t* stream = uni_ptr(stream_);
    uint32_t* err = uni_ptr(err_);
    idx_u16* perm = uni_idx(perm_);
    lds_u16* sidx = uni_lds(sidx_);
    lds_u32* rawcnt = uni_lds(rawcnt_);
    lds_u16* jt = uni_lds(jt_);
    const int M = uni(M_), nT = uni(nT_), force_slow = uni(force_slow_);
    const uint32_t slen = uni(slen_);
    st.off = uni(st.off);
#pragma unroll
    for (int i = 0; i < 7; ++i) st.pr[i] = sgpr(st.pr[i]);
    return sample_chunk_t<kMin>(stream, slen, perm, sidx, rawcnt, jt, st, M, nT, lane, force_slow, err);
}

// ---- the active RANSAC's correspondences: four arrays in the wave's global workspace -----------------
struct Pts {
    const double* g;   // records of (x1, y1, x2, y2), 32-byte aligned (PtRec)
    uint32_t gs;       // (capacity of the region: unused by the record layout)
};
__device__ __forceinline__ Pts uni(Pts P) {
    Pts Q;
    Q.g = uni_ptr(P.g);
    Q.gs = uni(P.gs);
    return Q;
}
__device__ __forceinline__ void load_pt(const Pts& P, int k, double& a, double& b, double& c, double& d) {
    const pt4 r = gptr(reinterpret_cast<const pt4*>(P.g))[k];  // two 16-byte loads from one line
    a = r[0]; b = r[1]; c = r[2]; d = r[3];
}

template <int KIND>
__device__ __forceinline__ double residual_t(const double* m, double a, double b, double c, double d) {
    return KIND == K_H ? h_residual(m, a, b, c, d) : (KIND == K_T ? t_residual(m, a, b, c, d) : sampson(m, a, b, c, d));
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
// (A/B hooks, round 6, same box, kernels per 124,750 pairs: score() inlined into the replay loop 421.1 -> 417.1 / 419.8 ->
// 415.5 ms - the caller's live registers are not spilled around ~50 calls per RANSAC; extract_inliers inlined 450 ms: no)
#ifndef AMC_SCORE_INLINE
#define AMC_SCORE_INLINE __forceinline__
#endif
#ifndef AMC_EXTRACT_INLINE
#define AMC_EXTRACT_INLINE __noinline__
#endif
template <int KIND>
__device__ AMC_SCORE_INLINE Support score(const Model9 mv, const Pts P_, int M_, double max_res_, int lane, int need_sum_from) {
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = uni(mv.v[i]);
    const Pts P = uni(P_);
    const int M = uni(M_);
    const double max_res = uni(max_res_);
    double acc = 0.0;
    int cnt = 0;
    // four batches of 64 at a time: their loads are issued together (the residual chain of a batch would otherwise
    // start only when its own loads have come back, one round trip to L2 per batch); the lane's partial sum still
    // takes its residuals in batch order
    for (int k0 = 0; k0 < M; k0 += 256) {
        double a[4], b[4], c[4], d[4];
        bool val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            val[u] = k < M;
            load_pt(P, val[u] ? k : 0, a[u], b[u], c[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double r = residual_t<KIND>(m, a[u], b[u], c[u], d[u]);
            const bool in = val[u] && r <= max_res;
            if (in) acc += r;
            cnt += __popcll(__ballot(in));
        }
    }
    Support s;
    s.cnt = cnt;
    s.sum = cnt >= need_sum_from ? butterfly(acc) : 1.7976931348623157e308;
    return s;
}

// ---- local optimisation over the ordered inlier list w.inl[0..K) ---------------------------------
// ordered compaction of the inlier indices of `model` (kind); returns K
__device__ AMC_EXTRACT_INLINE int extract_inliers(idx_u16* inl, int lane, int kind, const Model9 mv, const Pts P, int M,
                                            double max_res) {
    const double* model = mv.v;
    int base = 0;
    for (int k0 = 0; k0 < M; k0 += 256) {
        double a[4], b[4], c[4], d[4];
        bool val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            val[u] = k < M;
            load_pt(P, val[u] ? k : 0, a[u], b[u], c[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            const bool in = val[u] && residual_of(kind, model, a[u], b[u], c[u], d[u]) <= max_res;
            const unsigned long long bal = __ballot(in);
            if (in) inl[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)k;
            base += __popcll(bal);
        }
    }
    wave_lds_sync();
    return base;
}

// CenterAndNormalizeImagePoints over the K listed points of image `img` (0: x1,y1; 1: x2,y2):
// only the transform T is produced; the normalised coordinates are recomputed where they are
// consumed (apply_T), with the operations of the reference loop, instead of being stored.
// the steps of a local solve as calls (default) or inlined into local_estimate (-DAMC_LO_SUB=__forceinline__: A/B hook)
#ifndef AMC_LO_SUB
#define AMC_LO_SUB __noinline__
#endif
struct LoCtx {  // what the local estimators need of the wave, passed by value (registers)
    idx_u16* inl;
    lds_f64* jacA;
    lds_f64* jacV;
    lds_f64* uni;   // essential-matrix kernel: >= kE5UniDoubles of LDS for the local 5-point solve's wave-uniform data
    int lane;
};
// The local 5-point solve's wave-uniform working set - null-space basis (36), constraint polynomials B (45) and det (11),
// roots (10) - lives in LDS (round 6; over the minimal solver's coefficient area, idle during the local optimisation).
// As arrays of the calling function they were private memory: every lane kept its copy in scratch, 512 bytes per
// double and store, written and read back across the solve's four calls - most of what the essential-matrix kernel
// moved through the memory side.
constexpr int kE5UniNsp = 0, kE5UniB = 36, kE5UniDet = 81, kE5UniRoots = 92, kE5UniDoubles = 102;
__device__ __forceinline__ void center_T(const LoCtx& w, const Pts& P, int img, int K, double* T) {
    const int lane = w.lane;
    double ax = 0.0, ay = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt(P, w.inl[k], p[0], p[1], p[2], p[3]);
        ax += p[2 * img]; ay += p[2 * img + 1];
    }
    const double cx = butterfly(ax) / (double)K;
    const double cy = butterfly(ay) / (double)K;
    double ar = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt(P, w.inl[k], p[0], p[1], p[2], p[3]);
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
// both images' transforms in two passes over the listed records instead of four (round 6): a record holds both
// images' coordinates, and each image's sums see the addends center_T gives them, in the same order
__device__ __forceinline__ void center_T_both(const LoCtx& w, const Pts& P, int K, double* T1, double* T2) {
    const int lane = w.lane;
    double ax1 = 0.0, ay1 = 0.0, ax2 = 0.0, ay2 = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt(P, w.inl[k], p[0], p[1], p[2], p[3]);
        ax1 += p[0]; ay1 += p[1]; ax2 += p[2]; ay2 += p[3];
    }
    const double cx1 = butterfly(ax1) / (double)K, cy1 = butterfly(ay1) / (double)K;
    const double cx2 = butterfly(ax2) / (double)K, cy2 = butterfly(ay2) / (double)K;
    double ar1 = 0.0, ar2 = 0.0;
    for (int k = lane; k < K; k += 64) {
        double p[4];
        load_pt(P, w.inl[k], p[0], p[1], p[2], p[3]);
        const double dx1 = p[0] - cx1, dy1 = p[1] - cy1, dx2 = p[2] - cx2, dy2 = p[3] - cy2;
        ar1 += dx1 * dx1 + dy1 * dy1;
        ar2 += dx2 * dx2 + dy2 * dy2;
    }
    const double rms1 = dsqrt(butterfly(ar1) / (double)K), rms2 = dsqrt(butterfly(ar2) / (double)K);
    const double nf1 = dsqrt(2.0) / rms1, nf2 = dsqrt(2.0) / rms2;
    T1[0] = nf1; T1[1] = 0; T1[2] = -nf1 * cx1;
    T1[3] = 0; T1[4] = nf1; T1[5] = -nf1 * cy1;
    T1[6] = 0; T1[7] = 0; T1[8] = 1;
    T2[0] = nf2; T2[1] = 0; T2[2] = -nf2 * cx2;
    T2[3] = 0; T2[4] = nf2; T2[5] = -nf2 * cy2;
    T2[6] = 0; T2[7] = 0; T2[8] = 1;
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
// det_sum64 order: each lane keeps the partial sums of its strided rows, then one butterfly per entry.
//   MODE 0: epipolar row [x1 x2, y1 x2, x2, x1 y2, y1 y2, y2, x1, y1, 1]      (F8 / E5), K rows
//   MODE 1: homography rows; r < K -> "a" row of point r, r >= K -> "b" row of point r-K, 2K rows
// NORM: correspondences are normalised by T1 / T2 first.
// The 45 entries are accumulated in two passes over the rows (rows 0..3 of the triangle: 30 entries, rows 4..8: 15)
// so that the accumulators of a pass fit the register budget of a 4-waves-per-SIMD kernel; an entry's sum sees the
// same addends in the same order either way.
template <int MODE, bool NORM, int I0, int I1>
__device__ __forceinline__ void ata_pass(const LoCtx& w, const Pts& P, int K, const double* T1, const double* T2) {
    constexpr int NE = (9 - I0) * (9 - I0 + 1) / 2 - (9 - I1) * (9 - I1 + 1) / 2;  // entries (i, j >= i) with I0 <= i < I1
    const int lane = w.lane;
    const int rows = MODE == 1 ? 2 * K : K;
    double acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = 0.0;
    for (int r0 = lane; r0 < rows; r0 += 64) {
        const int k = (MODE == 1 && r0 >= K) ? r0 - K : r0;
        double x1, y1, x2, y2;
        load_pt(P, w.inl[k], x1, y1, x2, y2);
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
        for (int i = I0; i < I1; ++i)
#pragma unroll
            for (int j = i; j < 9; ++j) acc[e++] += r[i] * r[j];
    }
    int e = 0;
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = i; j < 9; ++j) {
            const double sres = butterfly(acc[e++]);
            if (lane == 0) {
                w.jacA[i * 9 + j] = sres;
                w.jacA[j * 9 + i] = sres;
            }
        }
}
template <int MODE, bool NORM>
__device__ __forceinline__ void ata_impl(const LoCtx& w, const Pts& P, int K, const double* T1, const double* T2) {
    ata_pass<MODE, NORM, 0, 4>(w, P, K, T1, T2);
    ata_pass<MODE, NORM, 4, 9>(w, P, K, T1, T2);
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
__device__ AMC_LO_SUB void jacobi_eigen_wave(lds_f64* A, lds_f64* V, int lane) {
    constexpr int n = 9, rounds = 9, np = 4;  // the only size the kernel decomposes as a wave
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
__device__ __forceinline__ void smallest_eigvec9_wave(const lds_f64* A, const lds_f64* V, double* x) {
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i * 9 + i] < A[best * 9 + best]) best = i;
    for (int i = 0; i < 9; ++i) x[i] = V[i * 9 + best];
}
// value of lane OWN of every quad, in all four lanes of the quad (DPP quad_perm: no LDS, no scalar unit)
template <int OWN>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_mov_dpp(v, OWN * 0x55, 0xf, 0xf, true);
}
template <int OWN>
__device__ __forceinline__ double quad_bcast(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)quad_bcast<OWN>((int)(unsigned)u), hi = (unsigned)quad_bcast<OWN>((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// ---- real roots of ONE polynomial by the whole wave (local optimisation's 5-point solve) ----------
// tvg_math.h's RootChain walks the chain of derivatives and, per level, solves the sign-change
// brackets two at a time.  The brackets of a level are independent, so here lane i takes
// bracket i (same arithmetic per bracket, hence the same roots bit for bit) and the level costs one
// bracket solve instead of up to R of them; the ordered, de-duplicated root list is then assembled
// exactly as roots_between_t does it.
template <int DEG, int R>
struct WaveRootChain {
    static __device__ __forceinline__ int run(const double (&c)[DEG + 1], double* roots, lds_f64* tmp, int lane) {
        double crit[R];
        const int nc = WaveRootChain<DEG, R - 1>::run(c, crit, tmp, lane);
        double d[R + 1], dd[R];
        poly_derivative_t<DEG, DEG - R>(c, d);
        poly_derivative_t<DEG, DEG - R + 1>(c, dd);  // the derivative of d
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
        // one bracket per QUAD of lanes (round 6; one per lane before): 0 nothing, 1 exact root at the lower edge, 2
        // bracketed root.  At most ten of the wave's lanes had work, and a bracket's bisection is one dependent chain -
        // Horner, then the step's comparisons - so the level cost the latency of ~40 such steps.  The quad takes two steps
        // per round: lane 0 evaluates the midpoint and takes the step; lanes 1 and 2 at the same time take the step
        // that follows if the upper / the lower end moves - its midpoint is known before the first step is decided -
        // and the outcome that applies is picked afterwards (quad broadcasts).  Every step that counts sees the plain
        // loop's operands in the plain loop's order: the same brackets, the same roots, bit for bit.
        int kind = 0;
        double val = 0.0;
        const int qb = lane >> 2, role = lane & 3;
        if (qb + 1 < ne) {
            double lo = tmp[qb], hi = tmp[qb + 1];
            const double flo = poly_eval_t<R>(d, lo);
            const double fhi = poly_eval_t<R>(d, hi);
            if (flo == 0.0) {
                kind = 1;
                val = lo;
            } else if (fhi != 0.0 && (flo < 0.0) != (fhi < 0.0)) {
                // bracket_root: bisection to 2^-26 of the bracket's position, then three bracketed Newton steps
                bool zero = false, act = true;
                const bool neg_lo = flo < 0.0;
                for (int it = 0; it < 200 && act; it += 2) {
                    const double mid0 = 0.5 * (lo + hi);
                    // this lane's step: the first one (roles 0, 3), or the second one after hi <- mid0 (1) / lo <- mid0 (2)
                    double blo = role == 2 ? mid0 : lo, bhi = role == 1 ? mid0 : hi;
                    const double bx = 0.5 * (blo + bhi);
                    bool bact = !(bx == blo || bx == bhi), bzero = zero;
                    const double fm = poly_eval_t<R>(d, bx);
                    bracket_step(bx, fm, blo, bhi, neg_lo, bact, bzero);  // (tvg_math.h: the plain loop's step, branch-free)
                    const int bflags = (bact ? 1 : 0) | (bzero ? 2 : 0);
                    const double lo1 = quad_bcast<0>(blo), hi1 = quad_bcast<0>(bhi);
                    const int fl1 = quad_bcast<0>(bflags);
                    const double loL = quad_bcast<1>(blo), hiL = quad_bcast<1>(bhi), loR = quad_bcast<2>(blo), hiR = quad_bcast<2>(bhi);
                    const int flL = quad_bcast<1>(bflags), flR = quad_bcast<2>(bflags);
                    // still active after the first step: exactly one end moved to mid0, and the second step is the one
                    // taken on that assumption (a first step that is the 200th is never followed: 200 is even)
                    const bool second = (fl1 & 1) != 0, moved_lo = lo1 == mid0;
                    lo = second ? (moved_lo ? loR : loL) : lo1;
                    hi = second ? (moved_lo ? hiR : hiL) : hi1;
                    const int fl = second ? (moved_lo ? flR : flL) : fl1;
                    act = (fl & 1) != 0;
                    zero = (fl & 2) != 0;
                }
                double r = 0.5 * (lo + hi);
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const double fr = poly_eval_t<R>(d, r), dr = poly_eval_t<R - 1>(dd, r);
                    const double rn = r - fr / dr;
                    r = (!zero && rn > lo && rn < hi) ? rn : r;
                }
                kind = 2;
                val = r;
            }
        }
        const double last = tmp[ne - 1];
        wave_lds_sync();  // tmp is rewritten by the next level
        int nr = 0;
        for (int i = 0; i + 1 < ne; ++i) {
            const int k = __builtin_amdgcn_readlane(kind, 4 * i);
            const double r = readlane_f64(val, 4 * i);
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
__device__ AMC_LO_SUB int real_roots10_wave(const lds_f64* c_in, lds_f64* roots_out, lds_f64* tmp, int lane) {
    double c[11], roots[10];
#pragma unroll
    for (int i = 0; i <= 10; ++i) c[i] = c_in[i];
#pragma unroll
    for (int i = 0; i < 10; ++i) roots[i] = 0.0;
    const int nr = c[10] == 0.0 ? real_roots_t<10>(c, roots)  // degenerate leading coefficient: plain path
                                : WaveRootChain<10, 10>::run(c, roots, tmp, lane);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) roots_out[i] = roots[i];
    }
    wave_lds_sync();
    return nr;
}

// ---- the 5-point solve of ONE problem by the whole wave (local optimisation) -----------------------
// e5_build keeps the 10 x 20 constraint matrix of a solve in one lane: 200 live doubles, most of them in scratch
// memory, and with one problem per wave every lane did the same elimination.  Here (round 4) the rows themselves are
// dealt out: the nine entries of E E^T and then the nine rows of 2 (E E^T) E - tr(E E^T) E are one (i, j) per lane -
// the same expressions on every lane, the operands read from LDS at an address that depends on the lane -, the
// determinant row (another expression) is computed by every lane as before, and the rows reach the lanes that keep
// the columns (lane c < 20 keeps column c) through LDS.  The Gauss-Jordan elimination runs on those 20 columns in
// parallel: per pivot the pivot column is broadcast (ten readlanes), every lane searches the pivot and applies the
// row swap to its own column, and one multiply and nine multiply-subtracts finish the step.  Every element goes
// through the operations e5_build applies to it, in the same order: the same bits.
// sc: >= 162 doubles of LDS (jacA + jacV).  Layout while the rows are built: [0, 90) E E^T, [100, 136) E's basis.
#if !defined(AMC_TVG_E5_ROWS_ALL_LANES)
__device__ AMC_LO_SUB void e5_build_wave(const lds_f64* nsp, lds_f64* PB, lds_f64* Pdet, lds_f64* sc, int lane) {
    lds_f64* el = sc + 100;  // el[k * 4 + d] = e[k][d]
    if (lane < 36) {
        const int k = lane >> 2, d = lane & 3;
        el[lane] = nsp[d * 9 + k];
    }
    wave_lds_sync();
    const int l9 = lane < 9 ? lane : 0;  // lanes 9 .. 63 redo entry / row (0, 0) and drop it
    const int ri = l9 / 3, rj = l9 - 3 * ri;
    auto lde = [&](int k, double (&v)[4]) {
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = el[k * 4 + d];
    };
    {   // E E^T: entry (ri, rj)
        double x0[4], x1[4], x2[4], y0[4], y1[4], y2[4], a[10], b[10], c[10];
        lde(3 * ri, x0); lde(3 * ri + 1, x1); lde(3 * ri + 2, x2);
        lde(3 * rj, y0); lde(3 * rj + 1, y1); lde(3 * rj + 2, y2);
        e5_mul11(x0, y0, a);
        e5_mul11(x1, y1, b);
        e5_mul11(x2, y2, c);
        if (lane < 9) {
#pragma unroll
            for (int t = 0; t < 10; ++t) sc[l9 * 10 + t] = (a[t] + b[t]) + c[t];
        }
    }
    wave_lds_sync();
    double g[10];  // this lane's column of G
    {   // det(E) -> row 0, on every lane (wave-uniform operands)
        double e[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int d = 0; d < 4; ++d) e[k][d] = nsp[d * 9 + k];
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
        double x = row[19];
#pragma unroll
        for (int c = 18; c >= 0; --c) x = lane == c ? row[c] : x;
        g[0] = x;
    }
    double row[20];
    {   // row 1 + 3 ri + rj
        double q[10], ev[4], acc[20], tmp[20], tr[10];
#pragma unroll
        for (int t = 0; t < 10; ++t) tr[t] = (sc[t] + sc[40 + t]) + sc[80 + t];
#pragma unroll
        for (int t = 0; t < 10; ++t) q[t] = sc[(3 * ri) * 10 + t];
        lde(rj, ev);
        e5_mul21(q, ev, acc);
#pragma unroll
        for (int t = 0; t < 10; ++t) q[t] = sc[(3 * ri + 1) * 10 + t];
        lde(3 + rj, ev);
        e5_mul21(q, ev, tmp);
#pragma unroll
        for (int t = 0; t < 20; ++t) acc[t] = acc[t] + tmp[t];
#pragma unroll
        for (int t = 0; t < 10; ++t) q[t] = sc[(3 * ri + 2) * 10 + t];
        lde(6 + rj, ev);
        e5_mul21(q, ev, tmp);
#pragma unroll
        for (int t = 0; t < 20; ++t) acc[t] = acc[t] + tmp[t];
        lde(3 * ri + rj, ev);
        e5_mul21(tr, ev, tmp);
#pragma unroll
        for (int t = 0; t < 20; ++t) row[t] = acc[t] * 2.0 - tmp[t];
    }
    wave_lds_sync();  // every read of E E^T and of the basis is done: sc carries the rows to the column lanes
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (lane < 9) {
#pragma unroll
            for (int c = 0; c < 10; ++c) sc[lane * 10 + c] = row[10 * half + c];
        }
        wave_lds_sync();
        const int c = lane - 10 * half;
        if (c >= 0 && c < 10) {
#pragma unroll
            for (int r = 0; r < 9; ++r) g[1 + r] = sc[r * 10 + c];
        }
        wave_lds_sync();
    }
    if (lane >= 20) {  // no column: defined values, never read
#pragma unroll
        for (int r = 1; r < 10; ++r) g[r] = 0.0;
    }
#else
__device__ AMC_LO_SUB void e5_build_wave(const lds_f64* nsp, lds_f64* PB, lds_f64* Pdet, lds_f64* sc, int lane) {
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
#endif
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
    E5Polys P;
    e5_finish(hl, P);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int d = 0; d < 5; ++d) PB[(k * 3 + t) * 5 + d] = P.B[k][t][d];
#pragma unroll
        for (int i = 0; i <= 10; ++i) Pdet[i] = P.det[i];
    }
    wave_lds_sync();
}
// e5_models with root i on lane i; the models come back wave-uniform, in root order
__device__ AMC_LO_SUB int e5_models_wave(const lds_f64* nsp_, const lds_f64* PB, const lds_f64* roots, int nr, lds_f64* models,
                                           int lane) {
    const double z = roots[lane < 10 ? lane : 0];
    double nsp[36];
    E5Polys P;
#pragma unroll
    for (int i = 0; i < 36; ++i) nsp[i] = nsp_[i];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int d = 0; d < 5; ++d) P.B[k][t][d] = PB[(k * 3 + t) * 5 + d];
#pragma unroll
    for (int i = 0; i <= 10; ++i) P.det[i] = 0.0;  // (not read by e5_model_from_root)
    double E[9];
    const bool ok = e5_model_from_root(nsp, P, z, E) && lane < nr;
    unsigned long long mask = __ballot(ok);
    int nm = 0;
    while (mask) {
        const int src = (int)__builtin_ctzll(mask);
        mask &= mask - 1;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double e = readlane_f64(E[i], src);
            if (lane == 0) models[9 * nm + i] = e;
        }
        ++nm;
    }
    return nm;
}

// Diagnostic build (-DAMC_TVG_LODIAG, tools/variant_build_tvg.sh): shader-clock cycles of the stages of the local
// estimators, summed over all waves (lane 0 adds); printed by the host with AMC_TVG_PROFILE=1.
#if defined(AMC_TVG_LODIAG)
__device__ unsigned long long g_lo_diag[64];   // 48 .. 55: stages of the minimal 5-point chunk
#define LODIAG_T0() unsigned long long lodiag_t_ = __builtin_readcyclecounter()
#define LODIAG_LAP(slot) do { const unsigned long long n_ = __builtin_readcyclecounter(); \
                              if (lane == 0) atomicAdd(&g_lo_diag[slot], n_ - lodiag_t_); lodiag_t_ = n_; } while (0)
#define LODIAG_COUNT(slot) do { if (lane == 0) atomicAdd(&g_lo_diag[slot], 1ull); } while (0)
// when a persistent wave started and when it found the queue empty (100 MHz wall clock): how long the machine drains
// while the last pairs are finished (one launch per kernel: AMC_TVG_SLICES=1, one size class)
__device__ unsigned long long g_wave_span[2][4096];
#define LODIAG_WAVE_START(gw) do { if (lane == 0 && (gw) < 4096) g_wave_span[0][gw] = wall_clock64(); } while (0)
#define LODIAG_WAVE_END(gw) do { if (lane == 0 && (gw) < 4096) g_wave_span[1][gw] = wall_clock64(); } while (0)
inline void lodiag_report_spans(const char* name) {
    static unsigned long long h[2][4096];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wave_span), sizeof h) != hipSuccess) return;
    std::vector<double> st, en;
    for (int i = 0; i < 4096; ++i)
        if (h[0][i] && h[1][i]) { st.push_back((double)h[0][i]); en.push_back((double)h[1][i]); }
    if (st.empty()) return;
    const double t0 = *std::min_element(st.begin(), st.end());
    std::sort(en.begin(), en.end());
    const double last = en.back();
    double idle = 0;
    for (double e : en) idle += last - e;
    const size_t n = en.size();
    std::fprintf(stderr, "[amc tvg lodiag spans] %s: %zu waves, span %.2f ms; waves done at min %.2f p10 %.2f median %.2f p90 %.2f max %.2f ms; "
                 "wave-slots idle before the kernel ends %.3f of the span\n", name, n, (last - t0) * 1e-5, (en[0] - t0) * 1e-5,
                 (en[n / 10] - t0) * 1e-5, (en[n / 2] - t0) * 1e-5, (en[n * 9 / 10] - t0) * 1e-5, (last - t0) * 1e-5,
                 idle / ((last - t0) * (double)n));
    static unsigned long long z[2][4096] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_span), z, sizeof z);
}
#else
#define LODIAG_WAVE_START(gw) do {} while (0)
#define LODIAG_WAVE_END(gw) do {} while (0)
#define LODIAG_T0() do {} while (0)
#define LODIAG_LAP(slot) do {} while (0)
#define LODIAG_COUNT(slot) do {} while (0)
#endif

// local estimator on the K listed inlier correspondences -> models (uniform), count
#ifndef AMC_LOCAL_INLINE
#define AMC_LOCAL_INLINE __noinline__
#endif
// The models come back in LDS, at w.jacA (round 6: they are wave-uniform, and as a private array of the caller every
// lane stored its copy to scratch - 46 KB per local 5-point solve - and read it back).
__device__ __forceinline__ void put_models(lds_f64* dst, const double* m, int n, int lane) {
    wave_lds_sync();  // (every lane has read what the solve kept in jacA | jacV)
    if (lane == 0)
        for (int i = 0; i < n; ++i) dst[i] = m[i];
}
template <int LOCAL>
__device__ AMC_LOCAL_INLINE int local_estimate(const LoCtx w, const Pts P, int K) {
    const int lane = w.lane;
    lds_f64* out = w.jacA;
    double models[9];
    if (LOCAL == K_T) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int k = lane; k < K; k += 64) {
            double p0, p1, p2, p3;
            load_pt(P, w.inl[k], p0, p1, p2, p3);
            a += p0; b += p1; c += p2; d += p3;
        }
        const double sx = butterfly(a) / (double)K, sy = butterfly(b) / (double)K;
        const double dx = butterfly(c) / (double)K, dy = butterfly(d) / (double)K;
        for (int i = 0; i < 9; ++i) models[i] = 0.0;
        models[0] = dx - sx;
        models[1] = dy - sy;
        put_models(out, models, 9, lane);
        return 1;
    }
    if (LOCAL == K_E5) {
        if (K == 5) {
            double a[5], b[5], c[5], d[5], em[kMaxModels * 9];
            for (int i = 0; i < 5; ++i) load_pt(P, w.inl[i], a[i], b[i], c[i], d[i]);
            const int nm = estimate_e5_minimal(a, b, c, d, em);
            put_models(out, em, 9 * nm, lane);
            return nm;
        }
        LODIAG_T0();
        LODIAG_COUNT(0);
        ata_impl<0, false>(w, P, K, nullptr, nullptr);
        LODIAG_LAP(1);
        jacobi_eigen_wave(w.jacA, w.jacV, lane);
        LODIAG_LAP(2);
        lds_f64* nsp = w.uni + kE5UniNsp;
        {   // e5_nullspace_from_eig (tvg_math.h): the four eigenvectors of the smallest eigenvalues, entry `lane` by lane `lane`
            int order[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) order[i] = i;
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = i + 1; j < 9; ++j)
                    if (w.jacA[order[j] * 9 + order[j]] < w.jacA[order[i] * 9 + order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
            if (lane < 36) {
                const int k = lane / 9, i = lane - 9 * k;
                const int col = k == 0 ? order[3] : (k == 1 ? order[2] : (k == 2 ? order[1] : order[0]));
                nsp[lane] = w.jacV[i * 9 + col];
            }
        }
        wave_lds_sync();  // jacA doubles as the root finder's scratch from here on
        e5_build_wave(nsp, w.uni + kE5UniB, w.uni + kE5UniDet, w.jacA, lane);
        LODIAG_LAP(3);
        const int nr = real_roots10_wave(w.uni + kE5UniDet, w.uni + kE5UniRoots, w.jacA, lane);
        LODIAG_LAP(4);
        const int nm = e5_models_wave(nsp, w.uni + kE5UniB, w.uni + kE5UniRoots, nr, out, lane);
        LODIAG_LAP(5);
        return nm;
    }
    if (LOCAL == K_H && K == 4) {
        double a[4], b[4], c[4], d[4];
        for (int i = 0; i < 4; ++i) load_pt(P, w.inl[i], a[i], b[i], c[i], d[i]);
        estimate_h4(a, b, c, d, models);
        put_models(out, models, 9, lane);
        return 1;
    }
    double T1[9], T2[9];
#if defined(AMC_TVG_CENTER_SEPARATE)   // (A/B hook: one image at a time, four passes)
    center_T(w, P, 0, K, T1);
    center_T(w, P, 1, K, T2);
#else
    center_T_both(w, P, K, T1, T2);
#endif
    LODIAG_T0();
    LODIAG_COUNT(LOCAL == K_F8 ? 8 : 12);
    if (LOCAL == K_F8) {
        ata_impl<0, true>(w, P, K, T1, T2);
        LODIAG_LAP(9);
        jacobi_eigen_wave(w.jacA, w.jacV, lane);
        LODIAG_LAP(10);
        double f[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, f);
        f8_from_vec(f, T1, T2, models);
        LODIAG_LAP(11);
    } else {
        ata_impl<1, true>(w, P, K, T1, T2);
        LODIAG_LAP(13);
        jacobi_eigen_wave(w.jacA, w.jacV, lane);
        LODIAG_LAP(14);
        double h[9];
        smallest_eigvec9_wave(w.jacA, w.jacV, h);
        h_denormalize(h, T1, T2, models);
        LODIAG_LAP(15);
    }
    put_models(out, models, 9, lane);
    return 1;
}

// ---- real roots of 64 degree-10 polynomials, one per lane, the brackets of a level spread over the wave ------------
// tvg_math.h's RootChain on a lane solves the lane's own sign-change brackets; a wave then pays, per derivative level,
// for the lane with the most brackets times the slowest bracket, and about a third of the lanes do useful work (a
// minimal 5-point solve has 32 brackets over its nine levels, 0 to 8 per level).  Here a level's brackets of ALL
// lanes go to a list in LDS - (lo, hi, f(lo)) and the owning lane; the level's coefficients of every lane beside it -
// and lane l solves records l and 64 + l of each 128-record pass, whoever they belong to, with bracket_pair_root:
// the arithmetic of a bracket does not depend on which lane runs it, so the roots are the ones RootChain returns,
// bit for bit, in the same order (classification and assembly stay with the owning lane, tvg_math.h's own code).
template <int R>
__device__ __forceinline__ void solve_level_balanced(const double (&d)[R + 1], RootLevel<R>& L, int lane, const RootScratch& S) {
    const int nb = __popc(L.todo);
    int incl = nb;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const int o = __shfl_up(incl, sh);
        if (lane >= sh) incl += o;
    }
    const int T = __builtin_amdgcn_readlane(incl, 63);
    if (T == 0) return;
    const int base = incl - nb;
#pragma unroll
    for (int k = 0; k <= R; ++k) S.coef[k * 64 + lane] = d[k];
    for (int p0 = 0; p0 < T; p0 += kRootCap) {
        {   // this lane's records of the pass
            int j = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bool mine = (L.todo >> i) & 1u;
                const int idx = base + j - p0;
                if (mine && idx >= 0 && idx < kRootCap) {
                    S.lo[idx] = L.edges[i];
                    S.hi[idx] = L.edges[i + 1];
                    S.flo[idx] = L.f[i];
                    S.src[idx] = (uint16_t)lane;
                }
                j += mine ? 1 : 0;
            }
        }
        wave_lds_sync();
        const int n = min(kRootCap, T - p0);
        const bool one = lane < n, two = 64 + lane < n;
        const int q0 = one ? lane : 0, q1 = two ? 64 + lane : q0;
        const double lo0 = S.lo[q0], hi0 = S.hi[q0], flo0 = S.flo[q0];
        const double lo1 = S.lo[q1], hi1 = S.hi[q1], flo1 = S.flo[q1];
        const int s0 = (int)S.src[q0], s1 = (int)S.src[q1];
        double c0[R + 1], c1[R + 1], dc0[R], dc1[R];
#pragma unroll
        for (int k = 0; k <= R; ++k) {
            c0[k] = S.coef[k * 64 + s0];
            c1[k] = S.coef[k * 64 + s1];
        }
        // the derivative's coefficients: d'[m] = d[m + 1] * (m + 1) continues poly_derivative_t's chain of products by
        // its last factor, so these are the bits RootChain gets from poly_derivative_t<DEG, DEG - R + 1>
#pragma unroll
        for (int m = 0; m < R; ++m) {
            dc0[m] = c0[m + 1] * (double)(m + 1);
            dc1[m] = c1[m + 1] * (double)(m + 1);
        }
        wave_lds_sync();  // every lane holds its records before the roots overwrite lo[]
        double r0, r1;
        bracket_pair_root<R>(c0, dc0, c1, dc1, lo0, hi0, flo0, lo1, hi1, flo1, one, two, r0, r1);
        if (one) S.lo[lane] = r0;
        if (two) S.lo[64 + lane] = r1;
        wave_lds_sync();
        {   // back to the owners
            int j = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bool mine = (L.todo >> i) & 1u;
                const int idx = base + j - p0;
                if (mine && idx >= 0 && idx < kRootCap) L.val[i] = S.lo[idx];
                j += mine ? 1 : 0;
            }
        }
        wave_lds_sync();
    }
    L.todo = 0u;
}
template <int DEG, int R>
struct LaneRootChain {
    static __device__ __forceinline__ int run(const double (&c)[DEG + 1], double (&roots)[R], bool act, int lane, const RootScratch& S) {
        double crit[R - 1];
        const int nc = LaneRootChain<DEG, R - 1>::run(c, crit, act, lane, S);
        double d[R + 1];
        poly_derivative_t<DEG, DEG - R>(c, d);
        RootLevel<R> L;
        roots_classify<R>(d, crit, nc, L);
        if (!act) L.todo = 0u;  // (a lane without a problem keeps its brackets out of the list)
        solve_level_balanced<R>(d, L, lane, S);
        return roots_assemble<R>(L, roots);
    }
};
template <int DEG>
struct LaneRootChain<DEG, 1> {
    static __device__ __forceinline__ int run(const double (&c)[DEG + 1], double (&roots)[1], bool, int, const RootScratch&) {
        double d[2];
        poly_derivative_t<DEG, DEG - 1>(c, d);
        roots[0] = -d[0] / d[1];
        return 1;
    }
};
// = real_roots_t<10>(c, roots) on every lane with `act` (c[10] != 0); the others return 0 roots.  Called by the whole wave.
__device__ __noinline__ int real_roots10_lanes(const double* c_in, double* roots_out, bool act, int lane, const RootScratch S_) {
    RootScratch S;
    S.coef = uni_lds(S_.coef); S.lo = uni_lds(S_.lo); S.hi = uni_lds(S_.hi); S.flo = uni_lds(S_.flo); S.src = uni_lds(S_.src);
    double c[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) c[i] = act ? c_in[i] : (i == 10 ? 1.0 : 0.0);  // a harmless polynomial for idle lanes
    double r[10];
    const int nr = LaneRootChain<10, 10>::run(c, r, act, lane, S);
#pragma unroll
    for (int i = 0; i < 10; ++i) roots_out[i] = r[i];
    return act ? nr : 0;
}

// ---- a chunk's minimal problems and the inlier count of every resulting model --------------------
// One minimal problem per lane.  H / T models stay with the solving lane (mym); F and E models - up to 3 / 10 per
// trial - go to the wave's global model table.  Then every model of the chunk is counted (count_chunk).  A model can
// only change the course of the sequential algorithm if its count reaches the best count so far, so all the replay
// needs per trial is an upper bound of the largest count among its models; the few
// trials that qualify are re-scored in full there.
// (A/B hooks, round 6: the chunk's per-lane record crosses these two calls through memory - scratch, 64 lanes x 4 bytes per
// dword and store - when they are not inlined; -DAMC_SOLVE_CHUNK_INLINE=__forceinline__ / -DAMC_COUNT_CHUNK_INLINE=...)
#ifndef AMC_SOLVE_CHUNK_INLINE
#define AMC_SOLVE_CHUNK_INLINE __noinline__
#endif
#ifndef AMC_COUNT_CHUNK_INLINE
#define AMC_COUNT_CHUNK_INLINE __forceinline__   // (same box, kernels per 124,750 pairs: 419.6 / 421.7 ms as a call, 414.3 / 417.3 inlined)
#endif
struct ChunkModels {   // returned by value (registers); the models themselves are in the wave's model table
    int nmod;    // models of this lane's trial
    int maxcnt;  // max inlier count over them (-1: none)
    unsigned long long cyc_solve, cyc_count;
};

// ---- division-free inlier tests for the counting loop ----------------------------------------------
// Counting only needs the DECISION residual <= max_res, and for both residuals that is a polynomial inequality:
//   homography  (d0 - pd0/pd2)^2 + (d1 - pd1/pd2)^2 <= T   <=>   (d0 pd2 - pd0)^2 + (d1 pd2 - pd1)^2 <= T pd2^2
//   Sampson     c^2 / den <= T                              <=>   c^2 <= T den               (den > 0)
// evaluated here with fused multiply-adds (no division: a third of the instructions of the reference expression
// and no rcp -> Newton -> fixup dependency chain).  The two sides are NOT the reference's roundings, so the test
// is only trusted away from the boundary: with L and R the two sides, the point is `out` - an outlier beyond doubt -
// when L >= R (1 + 1e-8); below that, or with an R that is not a normal positive number, it stays in the model's
// upper bound.  Why the band suffices: both this
// expression and the reference one are backward-stable evaluations of the same real quantity whose relative error
// at the boundary is <= ~10 eps x (largest coordinate / max_error) - the cancellation in d - p and in x2^T E x1;
// lo_ransac switches the fast test off unless that ratio is below 1e5 (fast_count), which bounds both
// errors by ~1e-10, a hundredth of the band.  A decided point is therefore decided as the reference decides it.
constexpr double kFastHi = 1.0 + 1e-8;
// false only for an outlier beyond doubt (`out` above), so that the sum over the correspondences is an upper bound of
// the inlier count - all the counting loops need (the `in` side of the test is not evaluated any more)
template <int KIND>
__device__ __forceinline__ bool fast_not_outlier(const double (&m)[9], double a, double b, double c, double d, double T) {
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
    const bool sane = ((uint32_t)((unsigned long long)__double_as_longlong(R) >> 32) - 0x16700000u) < 0x53000000u;
    return !sane || Lq < R * kFastHi;  // (a NaN Lq with a sane R: an outlier, as the reference's NaN <= max_res says)
}

// ---- counting, one model at a time, lanes = correspondences (the exact path) -------------------------
// Inlier count of ONE wave-uniform model over the M correspondences by the reference residual (ballot + popcount),
// or - as soon as even counting every remaining correspondence as an inlier could not reach `thr` - an upper bound
// below `thr`.  This is what every count was before the lanes-as-models loops below; it remains for the models those
// cannot decide (a point inside a fast test's band, for a model near the threshold), for the one-point translation
// RANSAC of the watermark test and for the AMC_TVG_EXACT_COUNT=1 test hook.
template <int KIND>
__device__ __forceinline__ int count_model_exact(const double (&m)[9], const Pts& P, int M, double max_res, int lane, int thr) {
    int cnt = 0;
    for (int k0 = 0; k0 < M; k0 += 256) {
        double a[4], b[4], c[4], d[4];
        bool val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            val[u] = k < M;
            load_pt(P, val[u] ? k : 0, a[u], b[u], c[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            cnt += __popcll(__ballot(val[u] && residual_t<KIND>(m, a[u], b[u], c[u], d[u]) <= max_res));
        const int rest = M - (k0 + 256);
        if (rest > 0 && cnt + rest < thr) return cnt + rest;
    }
    return cnt;
}
// all models of the chunk that way: H / T models from the solving lanes' registers ...
template <int KIND>
__device__ __forceinline__ int count_lane_models_exact(const double (&mym)[9], int nmod, const Pts& P, int M,
                                                       double max_res, int nT, int lane, int thr) {
    int maxcnt = -1;
    for (int t = 0; t < nT; ++t) {
        if (__builtin_amdgcn_readlane(nmod, t) < 1) continue;
        double sm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mym[i], t);
        const int c = count_model_exact<KIND>(sm, P, M, max_res, lane, thr);
        if (lane == t) maxcnt = max(maxcnt, c);
    }
    return maxcnt;
}
// ... F / E models from the wave's global model table
__device__ __forceinline__ int count_global_models_exact(const double* models, int nmod, const Pts& P, int M,
                                                         double max_res, int nT, int lane, int thr) {
    int maxcnt = -1;
    for (int t = 0; t < nT; ++t) {
        const int n = __builtin_amdgcn_readlane(nmod, t);
        for (int m = 0; m < n; ++m) {
            const AMC_GLOBAL double* src = gptr(models);
            double sm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(src[model_at(m, i, t)], 0);
            const int c = count_model_exact<K_F7>(sm, P, M, max_res, lane, thr);
            if (lane == t) maxcnt = max(maxcnt, c);
        }
    }
    return maxcnt;
}

// ---- counting with the lanes as MODELS ------------------------------------------------------------------
// The chunk's models are lane-resident (one per lane) and the correspondences are wave-uniform: they are read from
// the per-wave tables in global memory by the scalar unit (s_load_dwordx8 through the scalar data cache) and enter the
// vector FMAs as SGPR operands.  Per correspondence the loop is straight arithmetic plus two compare / add-with-carry
// pairs - no popcounts, no readlane broadcasts, no dependence on a scalar result - and every lane counts
//   ub  the points that are NOT outliers beyond doubt = inliers + the points its fast test cannot decide (its band, or
//       a degenerate right-hand side): an upper bound of the model's inlier count, and the count itself for all but
//       the few models with a point inside the band.
// That is all the replay compares with: a trial whose bound reaches the best count so far is re-scored in full there
// (score<>, the reference residual), one whose bound stays below cannot have changed anything.

// Sampson test in FP64 (fundamental / essential models): table = (x1, y1, x2, y2) doubles per correspondence.
// ub = correspondences that are not outliers beyond doubt.  The next record is requested before the current one is used, so the
// scalar-cache latency overlaps the arithmetic.
struct F64Rec {
    double a, b, c, d;
};
__device__ __forceinline__ F64Rec f64_rec(const AMC_CONST double* tab, int k) {
    F64Rec r;
    r.a = tab[4 * k]; r.b = tab[4 * k + 1]; r.c = tab[4 * k + 2]; r.d = tab[4 * k + 3];
    return r;
}
__device__ __forceinline__ void f64_count1(const double (&m)[9], const F64Rec& r, double T, int& ub) {
    ub += (int)fast_not_outlier<K_F7>(m, r.a, r.b, r.c, r.d, T);
}
// correspondences [k0, k1) against the lane's model; ub (an upper bound of the inlier count) is carried from segment to segment.  The records
// are requested a few 64-byte lines at a time (see count_lanes_h32 on why: one "all loads back" wait per batch).
typedef double d8v __attribute__((ext_vector_type(8)));
#ifndef AMC_F64_BATCH
#define AMC_F64_BATCH 4   // 64-byte lines (2 correspondences each) requested together
#endif
__device__ __forceinline__ void count_lanes_f64(const double (&m)[9], const AMC_CONST double* tab, int k0, int k1, double T,
                                                int& ub) {
    constexpr int kB = AMC_F64_BATCH;
    int k = k0;
    if ((k0 & 1) == 0) {  // (segments start on even correspondences: whole lines)
        const AMC_CONST d8v* tabq = reinterpret_cast<const AMC_CONST d8v*>(tab);
        for (; k + 2 * kB <= k1; k += 2 * kB) {
            d8v q[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) q[j] = tabq[(k >> 1) + j];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                F64Rec r0, r1;
                r0.a = q[j][0]; r0.b = q[j][1]; r0.c = q[j][2]; r0.d = q[j][3];
                r1.a = q[j][4]; r1.b = q[j][5]; r1.c = q[j][6]; r1.d = q[j][7];
                f64_count1(m, r0, T, ub);
                f64_count1(m, r1, T, ub);
            }
        }
    }
    for (; k < k1; ++k) f64_count1(m, f64_rec(tab, k), T, ub);
}
// exact inlier count of one wave-uniform model over the correspondences [k0, M), lanes = correspondences
template <int KIND>
__device__ __forceinline__ int count_range_exact(const double (&m)[9], const Pts& P, int k0, int M, double max_res, int lane) {
    int cnt = 0;
    for (int kb = k0; kb < M; kb += 128) {
        double a[2], b[2], c[2], d[2];
        bool val[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kb + 64 * u + lane;
            val[u] = k < M;
            load_pt(P, val[u] ? k : 0, a[u], b[u], c[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            cnt += __popcll(__ballot(val[u] && residual_t<KIND>(m, a[u], b[u], c[u], d[u]) <= max_res));
    }
    return cnt;
}

// Homography transfer test in packed FP32 with a bound on its own error (tvg_math.h h32_eval: the threshold folded
// into rows 0 / 1 of the model and into the image-2 coordinates, t = u'^2 + v'^2 - w^2, a point is decided only if
// |t32| exceeds the bound on |t32 - t| plus the band inside which the FP64 test itself defers to the reference
// residual; derivation there).  Two correspondences per instruction (v_pk_fma_f32): the table interleaves them
// (a0 a1 | b0 b1 | c0' c1' | d0' d1').  The homography RANSAC of a non-planar pair runs to its trial cap and nearly
// all of its models count nearly all matches as outliers by a wide margin: this loop is where the time of such a
// pair goes.  A diagnostic build of round 2 counted 500 million models both ways (profiles/r02/h32_diag.txt): no
// decided point disagreed with the FP64 count.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef H32Model H32Lane;  // tvg_math.h: the scaled float model and the constants of its error bound (h32_prepare)
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
struct H32Splat {  // the lane's model with every coefficient in both halves of a register pair
    v2f m[9], qR, qK;
};
__device__ __forceinline__ H32Splat h32_splat(const H32Lane& h) {
    H32Splat s;
#pragma unroll
    for (int i = 0; i < 9; ++i) s.m[i] = (v2f){h.m[i], h.m[i]};
    s.qR = (v2f){h.qR, h.qR};
    s.qK = (v2f){h.qK, h.qK};
    return s;
}
// h32_outlier_q (tvg_math.h) on two correspondences: the same operations in the same order
__device__ __forceinline__ v2f h32_q_pk(const H32Splat& h, v2f a, v2f b, v2f cs, v2f ds) {
    const v2f p0 = pk_fma(h.m[0], a, pk_fma(h.m[1], b, h.m[2]));
    const v2f p1 = pk_fma(h.m[3], a, pk_fma(h.m[4], b, h.m[5]));
    const v2f w = pk_fma(h.m[6], a, pk_fma(h.m[7], b, h.m[8]));
    const v2f u = pk_fma(cs, w, -p0), v = pk_fma(ds, w, -p1);
    const v2f R = w * w;
    return pk_fma(u, u, pk_fma(v, v, pk_fma(h.qR, R, h.qK)));
}
// 1.0 where q > 0 (q >= 2^-100; a fraction below that), 0.0 where q <= 0 or NaN: the product clamped to [0, 1] by the
// instruction's output modifier (DX10 clamp: NaN -> 0).  Summed, it counts the outliers beyond doubt - never too many.
__device__ __forceinline__ v2f pk_step(v2f q, v2f big) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(q), "v"(big));
    return r;
}
struct H32Rec {  // two correspondences of the pre-filter table
    v2f a, b, cs, ds;
};
__device__ __forceinline__ H32Rec h32_rec(const AMC_CONST v2f* tab, int k) {
    H32Rec r;
    r.a = tab[4 * k]; r.b = tab[4 * k + 1]; r.cs = tab[4 * k + 2]; r.ds = tab[4 * k + 3];
    return r;
}
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ H32Rec h32_rec_of(const f16v& q, int half) {  // record `half` (0 / 1) of a 64-byte line
    H32Rec r;
    if (half == 0) { r.a = (v2f){q[0], q[1]}; r.b = (v2f){q[2], q[3]}; r.cs = (v2f){q[4], q[5]}; r.ds = (v2f){q[6], q[7]}; }
    else { r.a = (v2f){q[8], q[9]}; r.b = (v2f){q[10], q[11]}; r.cs = (v2f){q[12], q[13]}; r.ds = (v2f){q[14], q[15]}; }
    return r;
}
#ifndef AMC_H32_BATCH
#define AMC_H32_BATCH 5   // 64-byte lines (4 correspondences each) requested together
#endif
// Upper bound of the lane's model's inlier count: M minus the correspondences that are outliers beyond doubt.
// (Round 6: leaving the loop once no model of the chunk can reach the best count any more - ~80 % into the table of a
// non-planar pair - measured: no gain, the look at the bound every 40 correspondences costs what the skipped tail saves.)
__device__ __forceinline__ int count_lanes_h32(const H32Lane& hl, const AMC_CONST v2f* tab, int M) {
    const H32Splat h = h32_splat(hl);
    const v2f big = (v2f){0x1p100f, 0x1p100f};
    v2f nout = (v2f){0.0f, 0.0f};
    const int np = M >> 1, last = ((M + 1) >> 1) - 1;  // full pairs; index of the table's last record
    // The table streams through the scalar cache once per chunk and misses it nearly always (every wave of the CU
    // walks a table of its own), and scalar loads return out of order - the only wait is "all of them".  So the
    // requests go out in batches of AMC_H32_BATCH whole lines and the arithmetic of a batch follows in one piece:
    // one exposed latency per 4 * AMC_H32_BATCH correspondences, covered by the other waves of the SIMD.
    constexpr int kB = AMC_H32_BATCH;
    const AMC_CONST f16v* tabq = reinterpret_cast<const AMC_CONST f16v*>(tab);
    int k = 0;
    for (; k + 2 * kB <= np; k += 2 * kB) {
        f16v q[kB];
#pragma unroll
        for (int j = 0; j < kB; ++j) q[j] = tabq[(k >> 1) + j];
#pragma unroll
        for (int j = 0; j < kB; ++j) {
            const H32Rec r0 = h32_rec_of(q[j], 0), r1 = h32_rec_of(q[j], 1);
            nout += pk_step(h32_q_pk(h, r0.a, r0.b, r0.cs, r0.ds), big);
            nout += pk_step(h32_q_pk(h, r1.a, r1.b, r1.cs, r1.ds), big);
        }
    }
    for (; k < np; ++k) {
        const H32Rec r = h32_rec(tab, k);
        nout += pk_step(h32_q_pk(h, r.a, r.b, r.cs, r.ds), big);
    }
    float total = nout.x + nout.y;
    if (M & 1) {  // the last, unpaired correspondence (the record's second half is a copy: not counted)
        const H32Rec r = h32_rec(tab, last);
        total += pk_step(h32_q_pk(h, r.a, r.b, r.cs, r.ds), big).x;
    }
    return M - (int)total;  // (sums of 0 / 1 below 2^24 are exact; a fractional step only raises the bound)
}

// ---- Sampson outliers in packed FP32 (tvg_math.h s32_outlier_q), two correspondences per instruction -----------------
struct S32Splat {
    v2f m[9], qR, qK;
};
__device__ __forceinline__ S32Splat s32_splat(const S32Model& h) {
    S32Splat s;
#pragma unroll
    for (int i = 0; i < 9; ++i) s.m[i] = (v2f){h.m[i], h.m[i]};
    s.qR = (v2f){h.qR, h.qR};
    s.qK = (v2f){h.qK, h.qK};
    return s;
}
__device__ __forceinline__ v2f s32_q_pk(const S32Splat& h, v2f a, v2f b, v2f c, v2f d) {
    const v2f e0 = pk_fma(h.m[0], a, pk_fma(h.m[1], b, h.m[2]));
    const v2f e1 = pk_fma(h.m[3], a, pk_fma(h.m[4], b, h.m[5]));
    const v2f e2 = pk_fma(h.m[6], a, pk_fma(h.m[7], b, h.m[8]));
    const v2f t0 = pk_fma(h.m[0], c, pk_fma(h.m[3], d, h.m[6]));
    const v2f t1 = pk_fma(h.m[1], c, pk_fma(h.m[4], d, h.m[7]));
    const v2f cc = pk_fma(c, e0, pk_fma(d, e1, e2));
    const v2f den = pk_fma(e0, e0, pk_fma(e1, e1, pk_fma(t0, t0, t1 * t1)));
    return pk_fma(cc, cc, pk_fma(h.qR, den, h.qK));
}
#ifndef AMC_S32_BATCH
#define AMC_S32_BATCH 4
#endif
// correspondences [k0, k1) (k0 even) against the lane's model: ub += those that are not outliers beyond doubt
__device__ __forceinline__ void count_lanes_s32(const S32Splat& h, const AMC_CONST v2f* tab, int k0, int k1, int& ub) {
    const v2f big = (v2f){0x1p100f, 0x1p100f};
    v2f nout = (v2f){0.0f, 0.0f};
    constexpr int kB = AMC_S32_BATCH;
    const AMC_CONST f16v* tabq = reinterpret_cast<const AMC_CONST f16v*>(tab);
    const int r1 = k1 >> 1;  // records [k0 / 2, r1) hold two counted correspondences each
    int r = k0 >> 1;
    if ((r & 1) == 0) {
        for (; r + 2 * kB <= r1; r += 2 * kB) {
            f16v q[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) q[j] = tabq[(r >> 1) + j];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const H32Rec x0 = h32_rec_of(q[j], 0), x1 = h32_rec_of(q[j], 1);
                nout += pk_step(s32_q_pk(h, x0.a, x0.b, x0.cs, x0.ds), big);
                nout += pk_step(s32_q_pk(h, x1.a, x1.b, x1.cs, x1.ds), big);
            }
        }
    }
    for (; r < r1; ++r) {
        const H32Rec x = h32_rec(tab, r);
        nout += pk_step(s32_q_pk(h, x.a, x.b, x.cs, x.ds), big);
    }
    float total = nout.x + nout.y;
    if (k1 & 1) {  // the last, unpaired correspondence (the record's second half is a copy: not counted)
        const H32Rec x = h32_rec(tab, r1);
        total += pk_step(s32_q_pk(h, x.a, x.b, x.cs, x.ds), big).x;
    }
    ub += (k1 - k0) - (int)total;
}

// the models of an F / E chunk (global table, nmod per trial) by the lanes-as-models loop: the valid models are listed
// in (trial, root) order, 64 of them are counted at a time, and the largest count of a trial's models is collected in
// LDS (tmax).  Returns lane t's maxcnt.
template <bool S32>
__device__ __forceinline__ int count_models_lanes(const double* models, int nmod, const double* p64, const float* p32, const Pts& P,
                                                  int M, double max_res, double cmax, int nT, int lane, int thr, lds_u16* mlist,
                                                  lds_i32* tmax) {
    // exclusive prefix of nmod over the lanes
    int incl = nmod;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(incl, s);
        if (lane >= s) incl += o;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int base = incl - nmod;
    for (int m = 0; m < nmod; ++m) mlist[base + m] = (uint16_t)(lane * 16 + m);
    tmax[lane] = -1;
    wave_lds_sync();
    const AMC_CONST double* tab = as_const_table(p64);
    const AMC_CONST v2f* tab32 = as_const_table(reinterpret_cast<const v2f*>(p32));
    for (int g0 = 0; g0 < total; g0 += 64) {
        const int idx = g0 + lane;
        const bool valid = idx < total;
        const int e = (int)mlist[valid ? idx : 0];
        const int t = e >> 4, mi = e & 15;
        const AMC_GLOBAL double* src = gptr(models);
        double mm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) mm[i] = src[model_at(mi, i, t)];
        // 64 correspondences at a time.  After each segment the models that can still reach thr - even if every
        // correspondence to come were an inlier - are counted: the RANSACs this path serves (E, F) have a high best
        // count, most sampled models are far below it, and once only a handful of the group's models are alive the
        // rest of their correspondences is cheaper to score one model at a time across the wave (exact residual)
        // than to keep 64 lanes streaming for them.  A model that dropped out reports ub + (all it has not seen),
        // an upper bound below thr.
        int ub = 0;
#ifndef AMC_CNT_SEG
#define AMC_CNT_SEG 64
#endif
#ifndef AMC_CNT_FEW
#define AMC_CNT_FEW 6
#endif
        constexpr int kSeg = AMC_CNT_SEG, kFewAlive = AMC_CNT_FEW;
        S32Splat hs;
        if (S32) hs = s32_splat(s32_prepare(mm, max_res, cmax));
        for (int k0 = 0; k0 < M; k0 += kSeg) {
            const int k1 = min(k0 + kSeg, M);
            if (S32) count_lanes_s32(hs, tab32, k0, k1, ub);
            else count_lanes_f64(mm, tab, k0, k1, max_res, ub);
            if (k1 < M) {
                unsigned long long alive = __ballot(valid && ub + (M - k1) >= thr);
                if (__popcll(alive) <= kFewAlive) {
                    const bool mine = (alive >> lane) & 1ull;
                    while (alive) {
                        const int src_lane = (int)__builtin_ctzll(alive);
                        alive &= alive - 1;
                        double sm[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) sm[i] = readlane_f64(mm[i], src_lane);
                        const int c = count_range_exact<K_F7>(sm, P, k1, M, max_res, lane);
                        if (lane == src_lane) ub += c;
                    }
                    if (!mine) ub += M - k1;
                    break;
                }
            }
        }
        if (valid) __hip_atomic_fetch_max(&tmax[t], ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    wave_lds_sync();
    const int r = tmax[lane];
    wave_lds_sync();
    (void)nT;
    return nmod > 0 ? r : -1;
}

// ---- the minimal 5-point solve's 10 x 20 elimination, four lanes per problem -----------------------------------------
// One problem per lane keeps 200 doubles of matrix per lane: twice the register file, and the spill traffic - not the
// arithmetic - was what the stage cost.  Here every lane builds its problem's rows (tvg_math.h e5_constraint_rows) and
// hands them to a staging area in the wave's workspace (element-major, so the stores coalesce); then the wave
// eliminates 16 problems at a time, lane 4 q + p holding columns 5 p .. 5 p + 4 of problem q (50 doubles).  Row
// operations act on columns independently, so each lane performs exactly the operations e5_build performs on its
// columns; the pivot row index, 1 / pivot and the multipliers come from the lane that owns the pivot column, by quad
// broadcast (DPP).  Rows 4 .. 9 of the right half go back to the staging area for e5_finish, again one problem per lane.
struct E5StageSink {
    double* stg;
    int lane;
    __device__ __forceinline__ void operator()(int r, const double (&row)[20]) {
        AMC_GLOBAL double* g = gptr(stg);
#pragma unroll
        for (int c = 0; c < 20; ++c) g[(size_t)(r * 20 + c) * 64 + lane] = row[c];
    }
};
template <int COL>
__device__ __forceinline__ void e5_elim_col(double (&g)[10][5]) {
    constexpr int OWN = COL / 5, CL = COL % 5;
    int piv = COL;
    double pv = dabs(g[COL][CL]);
#pragma unroll
    for (int r = COL + 1; r < 10; ++r)
        if (dabs(g[r][CL]) > pv) { pv = dabs(g[r][CL]); piv = r; }
    piv = quad_bcast<OWN>(piv);
    if (piv != COL) {
#pragma unroll
        for (int r = COL + 1; r < 10; ++r)
            if (r == piv) {
#pragma unroll
                for (int c = 0; c < 5; ++c) { const double t = g[COL][c]; g[COL][c] = g[r][c]; g[r][c] = t; }
            }
    }
    const double inv = quad_bcast<OWN>(1.0 / g[COL][CL]);
#pragma unroll
    for (int c = 0; c < 5; ++c) g[COL][c] = g[COL][c] * inv;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r == COL) continue;
        const double f = quad_bcast<OWN>(g[r][CL]);
#pragma unroll
        for (int c = 0; c < 5; ++c) g[r][c] = g[r][c] - f * g[COL][c];
    }
}
__device__ __forceinline__ void e5_eliminate_quads(double* stg_, int nT, int lane) {
    AMC_GLOBAL double* stg = gptr(stg_);
    const int q = lane >> 2, p = lane & 3;
    for (int pass = 0; pass * 16 < nT; ++pass) {
        const int T = pass * 16 + q;
        double g[10][5];
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) g[r][c] = stg[(size_t)(r * 20 + 5 * p + c) * 64 + T];
        e5_elim_col<0>(g); e5_elim_col<1>(g); e5_elim_col<2>(g); e5_elim_col<3>(g); e5_elim_col<4>(g);
        e5_elim_col<5>(g); e5_elim_col<6>(g); e5_elim_col<7>(g); e5_elim_col<8>(g); e5_elim_col<9>(g);
        if (p >= 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 5; ++c) stg[(size_t)(kE5StageG + r * 10 + 5 * (p - 2) + c) * 64 + T] = g[4 + r][c];
        }
    }
}

template <int EST>
__device__ AMC_SOLVE_CHUNK_INLINE ChunkModels solve_chunk(const Pts P_, const lds_u16* sidx_, int nT_, int lane,
                                                double* models_, const RootScratch rootscr) {
    const unsigned long long c0 = prof_clock();
    const Pts P = uni(P_);
    const lds_u16* sidx = uni_lds(sidx_);
    const int nT = uni(nT_);
    double* models = uni_ptr(models_);
    int nmod = 0;
    double mym[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) mym[i] = 0.0;
    if (lane < nT) {
        if (EST == K_F7) {
            double sx1[7], sy1[7], sx2[7], sy2[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) load_pt(P, sidx[lane * 8 + i], sx1[i], sy1[i], sx2[i], sy2[i]);
            double fm[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) fm[i] = 0.0;
            nmod = estimate_f7(sx1, sy1, sx2, sy2, fm);
            AMC_GLOBAL double* dst = gptr(models);
#pragma unroll
            for (int i = 0; i < 27; ++i) dst[model_at(i / 9, i % 9, lane)] = fm[i];
        } else if (EST == K_H) {
            double sx1[4], sy1[4], sx2[4], sy2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) load_pt(P, sidx[lane * 8 + i], sx1[i], sy1[i], sx2[i], sy2[i]);
            estimate_h4(sx1, sy1, sx2, sy2, mym);
            nmod = 1;
        } else if (EST == K_E5) {
            // (below: the 5-point solve is split around its root finder, which the wave runs as one)
        } else {  // K_T: model = dst - src of the single sample
            double a, b, c, d;
            load_pt(P, sidx[lane * 8], a, b, c, d);
            mym[0] = c - a;
            mym[1] = d - b;
            nmod = 1;
        }
    }
    if (EST == K_E5) {
        // estimate_e5_minimal (tvg_math.h) in its three steps - null space + constraint polynomials per lane, the real
        // roots of the 64 determinant polynomials by the whole wave (real_roots10_lanes), the models per lane; the
        // same functions in the same order, so the same bits as the per-lane call
        const bool have = lane < nT;
        LODIAG_T0();
        LODIAG_COUNT(48);
        double nsp[4 * 9];
        E5Polys polys;
#pragma unroll
        for (int i = 0; i <= 10; ++i) polys.det[i] = 0.0;
        if (have) {
            double A[5][9];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                double x1, y1, x2, y2;
                load_pt(P, sidx[lane * 8 + i], x1, y1, x2, y2);
                A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2;
                A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
                A[i][6] = x1; A[i][7] = y1; A[i][8] = 1;
            }
            double ns[4][9];
            nullspace_reg<5>(A, ns);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 9; ++j) nsp[k * 9 + j] = ns[k][j];
            E5StageSink sink{models, lane};
            e5_constraint_rows(nsp, sink);
        }
        wave_mem_sync();
        LODIAG_LAP(49);
        e5_eliminate_quads(models, nT, lane);
        wave_mem_sync();
        LODIAG_LAP(50);
        if (have) {
            double hl[6][10];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 10; ++c) hl[r][c] = gptr(models)[(size_t)(kE5StageG + r * 10 + c) * 64 + lane];
            e5_finish(hl, polys);
        }
        wave_mem_sync();  // (the model region is rewritten with the models below)
        LODIAG_LAP(51);
        double roots[10];
        const bool full = have && polys.det[10] != 0.0;
        int nr = real_roots10_lanes(polys.det, roots, full, lane, rootscr);
        if (have && !full) nr = real_roots_t<10>(polys.det, roots);  // a vanishing leading coefficient: the plain chain
        LODIAG_LAP(52);
        if (have) {   // e5_models (tvg_math.h) with the element-major table as its output
            AMC_GLOBAL double* dst = gptr(models);
#pragma unroll 1
            for (int i = 0; i < nr; ++i) {
                double z = roots[0];   // roots[i] by selects: the array stays in registers
#pragma unroll
                for (int q = 1; q < 10; ++q) z = i == q ? roots[q] : z;
                double E[9];
                if (e5_model_from_root(nsp, polys, z, E)) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) dst[model_at(nmod, k, lane)] = E[k];
                    ++nmod;
                }
            }
        }
        LODIAG_LAP(53);
    }
    if (EST == K_H || EST == K_T) {   // one model per trial: slot 0 of the table (round 6: it crossed to the caller through
        AMC_GLOBAL double* dst = gptr(models);   // a struct in scratch, was reloaded and spilled again for the replay)
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[model_at(0, i, lane)] = mym[i];
    }
    wave_mem_sync();
    ChunkModels out;
    out.nmod = nmod;
    out.maxcnt = -1;
    out.cyc_solve = prof_clock() - c0;
    out.cyc_count = 0;
    return out;
}

struct CountCtx {  // wave-uniform inputs of count_chunk
    Pts P;
    const double* p64;   // AoS doubles (scalar_table_sync'ed)
    const float* p32;    // packed-FP32 homography table
    const double* models;
    lds_u16* mlist;
    lds_i32* tmax;
    int M, nT, thr, fast;
    double max_res, cmax;
};
template <int EST>
__device__ AMC_COUNT_CHUNK_INLINE void count_chunk(ChunkModels* io, const CountCtx cc_, int lane) {
    const unsigned long long c1 = prof_clock();
    const Pts P = uni(cc_.P);
    const int M = uni(cc_.M), nT = uni(cc_.nT), thr = uni(cc_.thr);
    const bool fast = uni(cc_.fast) != 0;
    const double max_res = uni(cc_.max_res), cmax = uni(cc_.cmax);
    const double* models = uni_ptr(cc_.models);
    double mym[9];
    if (EST == K_H || EST == K_T) {
        const AMC_GLOBAL double* src = gptr(models);
#pragma unroll
        for (int i = 0; i < 9; ++i) mym[i] = src[model_at(0, i, lane)];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) mym[i] = 0.0;
    }
    const int nmod = io->nmod;
    int maxcnt;
    if (EST == K_T) {
        maxcnt = count_lane_models_exact<K_T>(mym, nmod, P, M, max_res, nT, lane, thr);
    } else if (!fast) {
        if (EST == K_H) maxcnt = count_lane_models_exact<K_H>(mym, nmod, P, M, max_res, nT, lane, thr);
        else maxcnt = count_global_models_exact(models, nmod, P, M, max_res, nT, lane, thr);
    } else if (EST == K_H) {
        const double s = 1.0 / dsqrt(max_res);
        const H32Lane hl = h32_prepare(mym, s, cmax);
        const int ub = count_lanes_h32(hl, as_const_table(reinterpret_cast<const v2f*>(uni_ptr(cc_.p32))), M);
        maxcnt = nmod > 0 ? ub : -1;
    } else if (uni(cc_.fast) == 2) {
        maxcnt = count_models_lanes<true>(models, nmod, uni_ptr(cc_.p64), uni_ptr(cc_.p32), P, M, max_res, cmax, nT, lane, thr,
                                          uni_lds(cc_.mlist), uni_lds(cc_.tmax));
    } else {
        maxcnt = count_models_lanes<false>(models, nmod, uni_ptr(cc_.p64), uni_ptr(cc_.p32), P, M, max_res, cmax, nT, lane, thr,
                                           uni_lds(cc_.mlist), uni_lds(cc_.tmax));
    }
    io->maxcnt = maxcnt;
    io->cyc_count = prof_clock() - c1;
}

struct Report {
    bool success;
    int num_trials;
    Support support;
    double model[9];
};

struct RansacCfg {
    double max_res;          // max_error^2
    int max_trials;          // already clamped as the RANSAC constructor does
    int min_trials;
    const uint32_t* dyn_tab; // dyn_max_num_trials by num_inliers (host libm), or nullptr
    const double* wm_cut;    // K_T only: inlier-ratio cut-offs by trial count (TvgParams::wm_cut), max_trials + 1 entries
    int force_slow_sampler;  // test hook: always take the draw-by-draw sampler path
    int no_fast_count;       // test hook (AMC_TVG_EXACT_COUNT=1): the counting loops evaluate the reference residual only
    int no_fast32;           // test hook (AMC_TVG_NO_S32=1): the F / E counting loops stay in FP64
};

// LORANSAC<EST, LOCAL>::Estimate over the M correspondences in the four arrays at gx (x1 | y1 | x2 | y2, each gstride
// long); mask: M bytes.  The generator position w.soff is advanced exactly as the sequential algorithm would.
template <int EST, int LOCAL>
__device__ Report lo_ransac(Wave& w_io, const RansacCfg& cfg, const double* gx, uint32_t gstride, int M, uint8_t* mask) {
    Wave w = w_io;  // by-value copy: the fields live in registers, not behind a pointer
    const int lane = w.lane;
#if !defined(AMC_TVG_NO_UNIFORM_STATE)
    // ... and, round 6, in SCALAR registers: every field is wave-uniform, but it arrives through memory (the Wave of
    // the kernel's frame), so the compiler kept ~40 vector registers of pointers and positions alive across the chunk
    // loop's calls and spilled them to scratch around each one (scratch traffic is most of what these kernels move
    // through the memory side; a scalar that does not fit goes to a lane of a vector register instead).  Same for the
    // best model and support below, and the sampler's state.
    w.sidx = uni_lds(w.sidx); w.rawcnt = uni_lds(w.rawcnt); w.tmax = uni_lds(w.tmax); w.mlist = uni_lds(w.mlist);
    w.perm = uni_idx(w.perm); w.inl = uni_idx(w.inl); w.jacA = uni_lds(w.jacA); w.jacV = uni_lds(w.jacV);
    w.stream = uni_ptr(w.stream); w.stream_len = uni(w.stream_len); w.soff = uni(w.soff); w.err = uni_ptr(w.err);
    w.ws = uni_ptr(w.ws); w.masks = uni_ptr(w.masks); w.mcap = uni(w.mcap); w.work = uni_ptr(w.work);
    w.rootscr.coef = uni_lds(w.rootscr.coef); w.rootscr.lo = uni_lds(w.rootscr.lo); w.rootscr.hi = uni_lds(w.rootscr.hi);
    w.rootscr.flo = uni_lds(w.rootscr.flo); w.rootscr.src = uni_lds(w.rootscr.src);
    gx = uni_ptr(gx); gstride = uni(gstride); M = uni(M); mask = uni_ptr(mask);
#define AMC_UNI(x) uni(x)
#else
#define AMC_UNI(x) (x)
#endif
    // the trial limits as SCALAR values: kept in a vector register, max_trials was spilled and - round 6, when the
    // sampler changed the register allocation of the watermark RANSAC - reloaded by the compiler inside the final mask
    // loop's exit block, where EXEC is still zero: report.num_trials came back as whatever the loop had left in the
    // register.  A scalar register does not depend on EXEC.
    const int max_trials = uni(cfg.max_trials), min_trials = uni(cfg.min_trials);
    constexpr int kMin = kmin_of(EST), kLocalMin = kmin_of(LOCAL);
    LoCtx lo;
    lo.inl = w.inl; lo.jacA = w.jacA; lo.jacV = w.jacV; lo.uni = w.rootscr.coef; lo.lane = lane;
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
    uint32_t dyn_max = (uint32_t)max_trials;
    // residuals this RANSAC evaluates with the reference FP64 expression (candidate re-scores, local-optimisation scores,
    // inlier extraction, the final mask; the whole counting loop where no pre-filter runs): work[WK_EXACT_FLOP]
    unsigned long long exact_evals = 0ull;

    Pts P;
    P.g = gx; P.gs = gstride;
    constexpr int kDiagBase = 16 + 8 * (EST == K_F7 ? 0 : (EST == K_H ? 1 : (EST == K_E5 ? 2 : 3)));
    (void)kDiagBase;
    LODIAG_T0();
    LODIAG_COUNT(kDiagBase);
    // the tables of the counting loops (read back through the scalar cache), and the pair's largest |coordinate|
    int fast_count = 0;
    double cmax = 0.0;
    // (the FP64 table of the counting loops IS the point region since the points are records: no second copy - 13.5 KB
    // less per wave to keep in the L2)
    const double* p64 = gx;
    float* p32 = ws_p32(w);
    {
        const double s = 1.0 / dsqrt(cfg.max_res);
        double amax = 0.0;
        for (int k = lane; k < M + (M & 1); k += 64) {
            const int kk = k < M ? k : M - 1;  // odd M: the pre-filter table's last pair repeats the last point
            const pt4 rk = gptr(reinterpret_cast<const pt4*>(gx))[kk];
            const double p0 = rk[0], p1 = rk[1], p2 = rk[2], p3 = rk[3];
            if (EST == K_H) {
                AMC_GLOBAL float* q = gptr(p32) + 8 * (size_t)(k >> 1) + (k & 1);
                q[0] = (float)p0; q[2] = (float)p1; q[4] = (float)(p2 * s); q[6] = (float)(p3 * s);
            } else if (EST != K_T) {
                AMC_GLOBAL float* q = gptr(p32) + 8 * (size_t)(k >> 1) + (k & 1);   // the Sampson pre-filter's table: plain coordinates
                q[0] = (float)p0; q[2] = (float)p1; q[4] = (float)p2; q[6] = (float)p3;
            }
            amax = dmax(dmax(amax, dmax(dabs(p0), dabs(p1))), dmax(dabs(p2), dabs(p3)));
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) amax = dmax(amax, __shfl_xor(amax, sh));
        // the division-free counting test is trusted only while (largest coordinate / max_error) <= 1e5 (see
        // fast_not_outlier); a NaN coordinate leaves amax as it was or NaN - either way the comparison below decides
        fast_count = (cfg.no_fast_count == 0 && amax * amax <= 1e10 * cfg.max_res) ? 1 : 0;
        // ... and the FP32 Sampson pre-filter (s32_outlier_q) is worth its pass while its band stays within ~1 % of
        // the threshold: (largest coordinate / max_error) <= 1e4.  Beyond that the FP64 loop counts.
        if (EST != K_H && fast_count && amax * amax <= 1e8 * cfg.max_res && cfg.no_fast32 == 0) fast_count = 2;
        cmax = amax;
        LODIAG_LAP(kDiagBase + 1);
        if (EST != K_T) scalar_table_sync();
        LODIAG_LAP(kDiagBase + 2);
    }

    // sampler.Initialize(M).  The first kMin entries of the persistent permutation are touched by
    // every draw: they live in (wave-uniform) registers, the rest in LDS.
    for (int k = lane; k < M; k += 64) w.perm[k] = (uint16_t)k;
    SamplerState ss;
#pragma unroll
    for (int i = 0; i < 7; ++i) ss.pr[i] = (uint32_t)i;
    wave_lds_sync();

    double* models = ws_models(w);
    bool aborted = false;
    int abort_trial = -1;
    for (int chunk = 0, nT = 0; chunk < max_trials && !aborted; chunk += nT) {
        nT = min(64, max_trials - chunk);
        if (EST == K_E5) {
            // The 5-point chunk costs what its trials cost (root finding is shared out by bracket, the elimination runs
            // 16 problems at a time), and the adaptive limit only ever falls: trials far beyond it will not be looked
            // at.  So the chunk ends a few trials after the limit (the first trial there that has a model stops the
            // loop); should none of those have one, the next chunk carries on from where this one ended.
            const long long lim = (long long)(dyn_max > (uint32_t)min_trials ? dyn_max : (uint32_t)min_trials);
            const long long want = lim - (long long)chunk + 4;
            // (only ever shrink: with fewer than 8 trials left - a small user max_num_trials - the floor of 8 must not
            // carry the chunk past max_trials)
            if (want < (long long)nT) nT = min(nT, (int)(want > 8 ? want : 8));
        }
        // ---- draw the chunk's samples ----
        const uint32_t chunk_off = w.soff;
        unsigned long long tp0 = prof_clock();
        ss.off = w.soff;
        // (the transposed draws lie over jacA | jacV: the local optimisation's scratch holds nothing between two chunks)
        ss = sample_chunk<kMin>(w.stream, w.stream_len, w.perm, w.sidx, w.rawcnt, reinterpret_cast<lds_u16*>(w.jacA), ss, M, nT, lane,
                                cfg.force_slow_sampler, w.err);
        ss.off = AMC_UNI(ss.off);
#pragma unroll
        for (int i = 0; i < 7; ++i) ss.pr[i] = AMC_UNI(ss.pr[i]);
        w.soff = ss.off;
        { const unsigned long long tp1 = prof_clock(); w.prof[0] += tp1 - tp0; tp0 = tp1; }
        // ---- 64 minimal problems + the inlier count of every model ---------
        ChunkModels cm = solve_chunk<EST>(P, w.sidx, nT, lane, models, w.rootscr);
        CountCtx cc;
        cc.P = P; cc.p64 = p64; cc.p32 = p32; cc.models = models; cc.mlist = w.mlist; cc.tmax = w.tmax;
        cc.M = M; cc.nT = nT; cc.thr = best.cnt; cc.fast = fast_count; cc.max_res = cfg.max_res; cc.cmax = cmax;
        count_chunk<EST>(&cm, cc, lane);
        w.prof[1] += cm.cyc_solve;
#if defined(AMC_TVG_LODIAG)
        if (lane == 0) { atomicAdd(&g_lo_diag[kDiagBase + 6], (unsigned long long)cm.cyc_solve); atomicAdd(&g_lo_diag[kDiagBase + 7], (unsigned long long)cm.cyc_count); }
#endif
        w.prof[5] += cm.cyc_count;  // the counting loop alone (also part of prof[2])
        tp0 = prof_clock();
        // ---- replay in trial order.  Only two kinds of trial can change anything: one holding a
        //      model whose count reaches the best so far (candidate: re-scored in full, exactly as
        //      the sequential loop would), and the first trial with a model at or beyond the
        //      adaptive trial limit (abort).  Everything in between is skipped.
        const unsigned long long live = nT == 64 ? ~0ull : ((1ull << nT) - 1ull);
        int t = 0;
        while (!aborted) {
            const long long lim = (long long)(dyn_max > (uint32_t)min_trials ? dyn_max : (uint32_t)min_trials) - chunk;
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
                if (EST == K_E5 || EST == K_F7) {
                    const AMC_GLOBAL double* src = gptr(models);
                    for (int i = 0; i < 9; ++i) sm[i] = src[model_at(m, i, t)];
                } else {
                    const AMC_GLOBAL double* src = gptr(models);
                    for (int i = 0; i < 9; ++i) sm[i] = src[model_at(0, i, t)];
                }
#if defined(AMC_TVG_LODIAG)
                const unsigned long long sc0_ = prof_clock();
#endif
                const Support sup = score<(EST == K_E5 ? K_F7 : EST)>(smv, P, M, cfg.max_res, lane, best.cnt);
                exact_evals += (unsigned long long)M;
#if defined(AMC_TVG_LODIAG)
                if (lane == 0) { atomicAdd(&g_lo_diag[kDiagBase + 3], 1ull); atomicAdd(&g_lo_diag[kDiagBase + 4], prof_clock() - sc0_); }
#endif
                if (better(sup, best)) {
                    const unsigned long long tl0 = prof_clock();
                    best.cnt = AMC_UNI(sup.cnt); best.sum = AMC_UNI(sup.sum);
                    for (int i = 0; i < 9; ++i) best_model[i] = AMC_UNI(sm[i]);
                    best_is_local = false;
                    if (sup.cnt > kMin && sup.cnt >= kLocalMin) {
                        // recursive local optimisation: inliers of the sample model first, then of
                        // the improved local model (COLMAP swaps residual vectors to the same effect)
                        int cur_kind = EST;
                        Model9 cur;
                        for (int i = 0; i < 9; ++i) cur.v[i] = sm[i];
                        for (int lt = 0; lt < 10; ++lt) {
                            const int K = extract_inliers(w.inl, lane, cur_kind, cur, P, M, cfg.max_res);
                            exact_evals += (unsigned long long)M;
                            const lds_f64* lm = lo.jacA;   // the local models, wave-uniform, in LDS
                            const unsigned long long tle = prof_clock();
                            const int nl = AMC_UNI(local_estimate<LOCAL>(lo, P, K));
                            wave_lds_sync();
                            if (lane == 0) {
                                w.work[wk_residual_slot(LOCAL)] += (unsigned long long)nl * (unsigned long long)M;
                                if (LOCAL == K_E5) w.work[WK_LO_E5] += 1;
                                else if (LOCAL == K_F8) w.work[WK_LO_F8] += 1;
                                else if (LOCAL == K_H) w.work[WK_LO_H] += 1;
                                w.work[WK_LO_POINTS] += (unsigned long long)K;
                            }
                            if (LOCAL == K_E5) w.prof[6] += prof_clock() - tle;
                            else if (LOCAL == K_F8) w.prof[7] += prof_clock() - tle;
                            const int prev = best.cnt;
                            for (int q = 0; q < nl; ++q) {
                                Model9 lmv;
                                for (int i = 0; i < 9; ++i) lmv.v[i] = lm[9 * q + i];
                                const Support ls = score<(LOCAL == K_E5 || LOCAL == K_F8 ? K_F7 : LOCAL)>(lmv, P, M, cfg.max_res, lane, best.cnt);
                                exact_evals += (unsigned long long)M;
                                if (better(ls, best)) {
                                    best.cnt = AMC_UNI(ls.cnt); best.sum = AMC_UNI(ls.sum);
                                    for (int i = 0; i < 9; ++i) best_model[i] = AMC_UNI(lm[9 * q + i]);
                                    best_is_local = true;
                                }
                            }
                            if (best.cnt <= prev) break;
                            cur_kind = LOCAL;
                            for (int i = 0; i < 9; ++i) cur.v[i] = best_model[i];
                        }
                    }
                    if (cfg.dyn_tab) {
                        dyn_max = cfg.dyn_tab[best.cnt];
                    } else if (cfg.wm_cut) {
                        // first T in [0, max_trials] with r >= wm_cut[T] (the cut-offs do not increase with T)
                        const double r = (double)best.cnt / (double)M;
                        int lo_t = 0, hi_t = max_trials + 1;  // answer in [lo_t, hi_t]; hi_t = none
                        while (lo_t < hi_t) {
                            const int mid = (lo_t + hi_t) >> 1;
                            if (r >= cfg.wm_cut[mid]) hi_t = mid; else lo_t = mid + 1;
                        }
                        dyn_max = lo_t <= max_trials ? (uint32_t)lo_t : 0xFFFFFFFFu;
                    } else {
                        dyn_max = 0xFFFFFFFFu;
                    }
                    w.prof[3] += prof_clock() - tl0;
                }
                if ((uint32_t)trial >= dyn_max && trial >= min_trials) {
                    aborted = true;
                    abort_trial = trial;
                    break;
                }
            }
            ++t;
        }
        w.prof[2] += cm.cyc_count;
        { const unsigned long long tp1 = prof_clock(); w.prof[2] += tp1 - tp0; }
        {   // algorithmic work of the chunk: the trials the sequential loop ran, their models x M residuals
            const int upto = aborted ? abort_trial - chunk : nT - 1;
            const int nmodels = wave_sum_int(lane <= upto ? cm.nmod : 0);
            if (EST == K_T || fast_count == 0) exact_evals += (unsigned long long)nmodels * (unsigned long long)M;
            if (lane == 0) {
                w.work[wk_residual_slot(EST)] += (unsigned long long)nmodels * (unsigned long long)M;
                w.work[EST == K_E5 ? WK_E5MIN : (EST == K_F7 ? WK_F7MIN : (EST == K_H ? WK_H4MIN : WK_TRIALS))] +=
                    (unsigned long long)(upto + 1);
            }
        }
        if (aborted) {
            // back to where the sequential algorithm stopped drawing
            wave_lds_sync();
            w.soff = chunk_off + sgpr(w.rawcnt[abort_trial - chunk]);
        }
    }
    // report.num_trials exactly as the for/abort dance of loransac.h leaves it
    rep.num_trials = aborted ? ((abort_trial + 1 < max_trials) ? abort_trial + 2 : abort_trial + 1)
                             : max_trials;
    rep.support = best;
    for (int i = 0; i < 9; ++i) rep.model[i] = best_model[i];
    w_io.soff = w.soff;
    for (int i = 0; i < 8; ++i) w_io.prof[i] = w.prof[i];
    if (best.cnt >= kMin) exact_evals += (unsigned long long)M;  // the final mask
    if (lane == 0) w.work[WK_EXACT_FLOP] += exact_evals * (unsigned long long)(EST == K_H ? 20 : (EST == K_T ? 7 : 33));
    if (best.cnt >= kMin && lane == 0)
        w.work[wk_residual_slot(best_is_local ? LOCAL : EST)] += (unsigned long long)M;
    if (best.cnt < kMin) return rep;
    rep.success = true;
#if defined(AMC_TVG_LODIAG)
    lodiag_t_ = prof_clock();
#endif
    const int fk = best_is_local ? LOCAL : EST;
    for (int k = lane; k < M; k += 64) {
        double a, b, c, d;
        load_pt(P, k, a, b, c, d);
        gptr(mask)[k] = residual_of(fk, rep.model, a, b, c, d) <= cfg.max_res ? 1 : 0;
    }
    wave_mem_sync();
    LODIAG_LAP(kDiagBase + 5);
    return rep;
}

}  // namespace
}  // namespace amc
