// match_mfma.hip — the hot kernel: one-way top-2 scan of 128-D u8 descriptor dot products on
// the int8 matrix cores (v_mfma_i32_32x32x32_i8).  gfx950 only.
//
// "One way" = COLMAP's FindBestMatchesOneWayBruteForce: for every row of
// image X, the best dot product against all rows of image Y (lowest index among ties) and the
// second-largest value with multiplicity.  The host runs it twice per pair:
//   MODE 0  X = image 1 (all rows)            , Y = image 2  -> row table
//   MODE 1  X = image 2 (candidate rows only) , Y = image 1  -> column table, lazily: only the
//           columns some accepted row points at (select_candidates_kernel); the cross check
//           never looks at any other column, so results equal COLMAP's full transposed scan.
// The kernel reports the best VALUE, the 32-row TILE of Y that holds it, and the largest value
// OUTSIDE the best's 16-output unit (a lower bound of the second); resolve_index_kernel
// (match_common.hip) turns the tile into the exact lowest index and completes the second value
// by recomputing those 32 dot products, only for rows that can still pass COLMAP's acceptance
// tests (a larger second only ever rejects).
//
// Why this shape (measured on MI355X, tools/ubench_ops.hip and tools/ubench_mix.hip):
//   * every 32-bit min/max/med3/max3/shift/shift-add is HALF rate on gfx950 (4 clk / wave64);
//   * on one SIMD an int8 MFMA (32 clk) hides only ~6 VALU instructions; each further one costs
//     ~4.5 clk.  So the epilogue budget is 6 VALU per MFMA = 1.5 per output.
// A values-only top-2 insertion (v_med3_i32 + v_max3_i32 + v_max_i32 per TWO candidates) is
// exactly 1.5/output, i.e. VALU-bound.  The scan therefore does less: it reduces each lane's 16
// outputs of a unit to their maximum (8 x v_max3/v_max) and keeps the top two of those maxima
// plus the tile of the best, 13 VALU per 16 outputs = 0.8/output, which leaves the matrix pipe as
// the limiter.  The row's exact second-largest value is completed by resolve_index_kernel from
// the winning tile (see `phase` below).  Everything else is moved off the VALU:
//   * zero point: the matrix core is signed, the arena holds a' = a - 128 (bytes ^ 0x80) and
//         sum a*b = sum a'*b' + 128*SX_i + 128*SY_j - 2^21        (SX, SY = byte sums, int32 exact)
//     The MFMA's A operand is the streamed Y tile and its B operand the resident X tile, so a
//     lane owns ONE X row and its 16 accumulator registers are 16 different Y rows: the Y term
//     128*SY_j is per register and rides in as the MFMA's C operand (16 ints per Y tile,
//     read from LDS); the X term is constant per lane, dropped during the scan and restored
//     (with the -2^21) when the row is decoded.  acc = v - 128*SX_i + 2^21, full int32 range:
//     ANY u8 data, any size.
//   * argmax: values only in the scan; the tile holding the best is tracked with one compare +
//     select per 16 outputs ("did best change?", strict, so the first tile wins ties).
//
// Shape.  One 512-thread workgroup per work item (dynamic queue; items sorted by Y image so
// co-resident workgroups stream the same image out of L2).  8 waves x 128 X rows (four 32-row
// B-operand tiles resident in registers) = a 1024-row block of X; Y streams through LDS in
// 256-row chunks by direct-to-LDS DMA, double buffered, one barrier per chunk; the prepared
// arena is pre-swizzled so the linear DMA image is bank-conflict-free for ds_read_b128.
// Each wave software-pipelines: the 4 MFMAs of unit u+1 are interleaved with the 13 VALU of unit u
// (unit = 32 Y rows x 32 X rows), two accumulator sets.
#include <climits>

#include "amc_internal.h"

namespace amc {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kBN = 256;                // Y rows per LDS chunk (= kRowPad); 512 measured 2 % slower
constexpr int kYT = kBN / 32;           // Y tiles per chunk
constexpr int kWaves = 8;               // waves per workgroup
constexpr int kXT = 4;                  // resident X tiles (32 rows each) per wave
constexpr int kWM = 32 * kXT;           // X rows per wave
constexpr int kBM = kWaves * kWM;       // 1024 X rows per row block
constexpr int kChunkBytes = kBN * kDim; // 32 KiB

// single LDS object (a second __shared__ object de-pipelines the DMA waits)
constexpr int kOffRs = 0;                        // 2 x rs128 chunks (kBN ints each)
constexpr int kOffQ = kOffRs + 2 * kBN * 4;      // queue slot
constexpr int kOffB = kOffQ + 16;                // 2 x descriptor chunks
constexpr int kLdsBytes = kOffB + 2 * kChunkBytes;
static_assert(kBN == kRowPad, "a chunk is the row padding unit");
static_assert(kBN % 64 == 0 && (kChunkBytes / 1024) % kWaves == 0, "chunk must split into tile pairs and 1 KiB DMA pieces per wave");
static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");

// v_med3_i32 / v_max3_i32 pinned by hand (hipcc pattern-matches them only some of the time).
// They read MFMA results directly and hipcc does not pad hazards for inline asm, so the kernel
// is structured so that an accumulator is only ever read one full phase (>= 4 MFMA issues)
// after the MFMAs that produced it were issued; sched_barriers pin that order.
__device__ __forceinline__ int smed3(int a, int b, int c) {
    int d;
    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int smax3(int a, int b, int c) {
    int d;
    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int smax2(int a, int b) {
    int d;
    asm volatile("v_max_i32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

struct YFrag {
    i32x4 f[4];  // 32 Y rows x four 32-deep k-slices (MFMA A operand)
    i32x16 ci;   // 128*SY_j for this lane's 16 Y rows (MFMA C operand)
};

template <int MODE>
__global__ __launch_bounds__(512) void match_mfma_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const uint32_t* __restrict__ order, uint32_t nitems, uint32_t* __restrict__ queue_head,
    const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ candbuf,
    Top2* __restrict__ outbuf, uint32_t* __restrict__ accmask, const float* __restrict__ lut,
    FinalizeParams fp) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    volatile uint32_t* s_q = reinterpret_cast<volatile uint32_t*>(smem + kOffQ);

    for (;;) {
        if (tid == 0) *s_q = atomicAdd(queue_head, 1u);
        __syncthreads();
        const uint32_t q = *s_q;
        __syncthreads();  // everyone has read the slot before it can be rewritten
        if (q >= nitems) break;
        const uint32_t pi = order[q];
        const PairDev p = pairs[pi];
        const ImageDev X = imgs[MODE == 0 ? p.slot1 : p.slot2];
        const ImageDev Y = imgs[MODE == 0 ? p.slot2 : p.slot1];
        const int nrows = (int)(MODE == 0 ? X.rows : cand_cnt[pi]);
        const uint32_t* list = candbuf + p.col_off;  // MODE 1: ascending candidate rows of X
        Top2* out = outbuf + (MODE == 0 ? p.row_off : p.col_off);
        if (nrows == 0 || Y.rows == 0) continue;  // uniform
        const int nchunks = (int)((Y.rows + kBN - 1) / kBN);
        const int nrb = (nrows + kBM - 1) / kBM;

        auto stage = [&](int c, int buf) {
            // the descriptor chunk in 1 KiB pieces dealt to the waves; dest = wave-uniform base + lane*16
            const char* src = reinterpret_cast<const char*>(Y.prep) + (size_t)c * kChunkBytes;
#pragma unroll
            for (int ps = 0; ps < kChunkBytes / 1024 / kWaves; ++ps) {
                const int piece = ps * kWaves + wid;
                __builtin_amdgcn_global_load_lds(
                    (gvoid_t*)(src + piece * 1024 + lane * 16),
                    (lvoid_t*)(smem + kOffB + buf * kChunkBytes + piece * 1024), 16, 0, 0);
            }
            if (wid * 1024 + lane * 16 < kBN * 4) {  // rs128 of this chunk's rows, up to 1 KiB per wave
                const char* rsrc = reinterpret_cast<const char*>(Y.rs128 + (size_t)c * kBN) + wid * 1024;
                __builtin_amdgcn_global_load_lds((gvoid_t*)(rsrc + lane * 16),
                                                 (lvoid_t*)(smem + kOffRs + buf * kBN * 4 + wid * 1024), 16,
                                                 0, 0);
            }
        };

        // A-operand fragments + C-init block of Y tile `yt` of the chunk in LDS buffer `buf`
        auto load_y = [&](YFrag& y, int buf, int yt) {
            const int row = yt * 32 + l31;  // within chunk
            const int sw = (row >> 1) & 7;
            const char* cp = smem + kOffB + buf * kChunkBytes + row * kDim;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                y.f[s] = *reinterpret_cast<const i32x4*>(cp + (((2 * s + lh) ^ sw) * 16));
            // accumulator register r <-> Y row (r&3) + 8*(r>>2) + 4*lh of the tile
            const int* rsb =
                reinterpret_cast<const int*>(smem + kOffRs + buf * kBN * 4) + yt * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const i32x4 v = *reinterpret_cast<const i32x4*>(rsb + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) y.ci[4 * g + e] = v[e];  // the -2^21 lives in xterm
            }
        };

        stage(0, 0);  // every row block starts from chunk 0; the next block's copy is issued early (below)
        for (int rb = 0; rb < nrb; ++rb) {
            const int rowbase = rb * kBM + wid * kWM;
            const bool active = rowbase < nrows;  // wave-uniform

            // ---- resident X state: B-operand fragments, one (best, second, tile) per lane ----
            i32x4 xf[kXT][4];
            int best[kXT], sec[kXT], btile[kXT], xterm[kXT];
            if (active) {
#pragma unroll
                for (int xt = 0; xt < kXT; ++xt) {
                    const int k = rowbase + xt * 32 + l31;
                    // rows past the end of the block are clamped (their results are never stored)
                    const int row = MODE == 0 ? min(k, (int)X.rows_pad - 1) : (int)list[min(k, nrows - 1)];
                    const char* rp = reinterpret_cast<const char*>(X.prep) + (size_t)row * kDim;
                    const int sw = (row >> 1) & 7;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        xf[xt][s] = *reinterpret_cast<const i32x4*>(rp + (((2 * s + lh) ^ sw) * 16));
                    // acc = sum a'b' + 128*SY_j = v - (128*SX_i - 2^21)
                    xterm[xt] = X.rs128[row] - (1 << 21);
                    // COLMAP's floor best = second = 0  <=>  acc = -xterm
                    best[xt] = -xterm[xt];
                    sec[xt] = -xterm[xt];
                    btile[xt] = -1;
                }
            }

            auto mfma4 = [&](i32x16& a, const YFrag& y, int xt) {
                a = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[0], xf[xt][0], y.ci, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s)
                    a = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[s], xf[xt][s], a, 0, 0, 0);
            };
            // One phase = the 4 MFMAs of the NEXT unit (into `an`) with the 13 VALU of the CURRENT
            // unit (reading `ac`, completed by the previous phase) spread between them, 5/4/4/-.
            // Interleaved, the two waves of a SIMD keep the matrix pipe fed (15.6 ns per MFMA in
            // tools/ubench_mix.hip); "4 MFMAs, then 13 VALU" leaves it idle whenever both waves are
            // in their VALU stretch (18.3 ns).
            //
            // 13 VALU for 16 outputs: the scan only keeps, per lane, the top two of the per-unit
            // MAXIMA (8 v_max3/v_max for the maximum of the unit's 16 outputs, then one insertion)
            // and the tile of the best.  The second-largest VALUE of the whole row is either the
            // maximum of another unit - which the running `sec` then holds, ties included - or
            // sits inside the winning tile, where resolve_index_kernel recomputes the 32 dot
            // products anyway to find the index: it takes the second of those 32 as well and the
            // row's second is the larger of the two.
            // insertion of a unit maximum into a lane's (best, second, tile) state
            auto insert = [&](int xt, int m, int tile) {
                const int b0 = best[xt];
                sec[xt] = smed3(b0, sec[xt], m);  // sec <= best always: the new second of the maxima
                const int b = smax2(b0, m);
                btile[xt] = (b != b0) ? tile : btile[xt];  // strict: first tile wins ties
                best[xt] = b;
            };
            // The insertion of a unit's maximum does not touch accumulators, so it is deferred into
            // the hazard slot of the NEXT phase (between its first two MFMAs): pm / ptile carry the
            // pending maximum (of X tile (xtc + 3) & 3) from one phase to the next.
            int pm = INT_MIN, ptile = 0;
            auto phase = [&](i32x16& an, const YFrag& y, int xtn, const i32x16& ac, int xtc, int tile) {
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[0], xf[xtn][0], y.ci, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#if defined(AMC_DIAG) && (AMC_DIAG & 2)
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[1], xf[xtn][1], an, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[2], xf[xtn][2], an, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                pm = smax2(pm, ac[0]);  // timing diagnostic: one VALU per unit keeps the accumulators live
                ptile = tile;
                (void)insert;
#else
                insert((xtc + 3) & 3, pm, ptile);
                __builtin_amdgcn_sched_barrier(0);
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[1], xf[xtn][1], an, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // first read of `ac`: two MFMAs have issued since the one that completed it
                int m0 = smax3(ac[0], ac[1], ac[2]);
                const int m1 = smax3(ac[3], ac[4], ac[5]);
                const int m2 = smax3(ac[6], ac[7], ac[8]);
                int m3 = smax3(ac[9], ac[10], ac[11]);
                __builtin_amdgcn_sched_barrier(0);
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[2], xf[xtn][2], an, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int m4 = smax3(ac[12], ac[13], ac[14]);
                m0 = smax3(m0, m1, m2);
                m3 = smax3(m3, m4, ac[15]);
                pm = smax2(m0, m3);
                ptile = tile;
#endif
                __builtin_amdgcn_sched_barrier(0);
                an = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[3], xf[xtn][3], an, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };

            // ---- software-pipelined scan over all of Y -----------------------------------
            // units in order (ytile, xt = 0..3); accA holds even xt, accB odd xt.  Each phase
            // issues the MFMAs of the NEXT unit, then runs the VALU of the current one.
            YFrag y0, y1;
            i32x16 accA, accB;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk 0 (and this block's X rows) landed
            __syncthreads();
            if (nchunks > 1) stage(1, 1);
            if (active) {
                load_y(y0, 0, 0);
                mfma4(accA, y0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // The sched_barriers pin the interleaving.  An accumulator is read (by inline asm, which
            // hipcc does not hazard-check) only after TWO further MFMAs have issued behind the one
            // that completed it - the matrix pipe runs MFMAs in order, 32 clk each, so the value
            // has been written back for well over the 11 wait states an 8-pass MFMA needs.
#define AMC_PHASE(accn, yfr, xtn, accc, xtc) phase(accn, yfr, xtn, accc, xtc, tile);
#if defined(AMC_DIAG) && (AMC_DIAG & 4)
#define AMC_DIAG_SYNC
#else
#define AMC_DIAG_SYNC asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
#endif
#if defined(AMC_DIAG) && (AMC_DIAG & 8)
#define AMC_DIAG_STAGE(c__)
#else
#define AMC_DIAG_STAGE(c__) if ((c__) + 2 < nchunks) stage((c__) + 2, (c__) & 1);
#endif
            // one step = one Y tile held in `yc`; prefetches the next tile into `yn`
#define AMC_STEP(yc, yn, c_, yt_)                                                              \
    {                                                                                          \
        const int c__ = (c_), yt__ = (yt_);                                                    \
        const bool lastt = (yt__ == kYT - 1);                                                  \
        const bool cross = lastt && (c__ + 1 < nchunks);                                       \
        if (cross) {                                                                           \
            AMC_DIAG_SYNC                                                                      \
            AMC_DIAG_STAGE(c__)                                                                \
        }                                                                                      \
        if (active) {                                                                          \
            /* very last tile: re-read itself (result unused) to stay branch-free */          \
            const int nbuf = cross ? ((c__ + 1) & 1) : (c__ & 1);                              \
            const int nyt = lastt ? (cross ? 0 : yt__) : yt__ + 1;                             \
            const int tile = c__ * kYT + yt__;                                                 \
            load_y(yn, nbuf, nyt);                                                             \
            AMC_PHASE(accB, yc, 1, accA, 0)                                                    \
            AMC_PHASE(accA, yc, 2, accB, 1)                                                    \
            AMC_PHASE(accB, yc, 3, accA, 2)                                                    \
            AMC_PHASE(accA, yn, 0, accB, 3)                                                    \
        }                                                                                      \
    }
            for (int c = 0; c < nchunks; ++c) {
#pragma unroll 1
                for (int yt = 0; yt < kYT; yt += 2) {
                    AMC_STEP(y0, y1, c, yt)
                    AMC_STEP(y1, y0, c, yt + 1)
                }
            }
#undef AMC_STEP
#undef AMC_PHASE
#undef AMC_DIAG_SYNC
#undef AMC_DIAG_STAGE
            if (active) insert(3, pm, ptile);  // the last unit's maximum is still pending
            __syncthreads();  // everyone is done with both LDS chunk buffers
            // restart the stream for the next row block now: the copy runs under this block's
            // epilogue and the next block's X loads
            if (rb + 1 < nrb) stage(0, 0);

            // ---- row block done: merge the two lane halves, decode, store -------------------
            if (active) {
#pragma unroll
                for (int xt = 0; xt < kXT; ++xt) {
                    const int ob = __shfl_xor(best[xt], 32);
                    const int os = __shfl_xor(sec[xt], 32);
                    const int ot = __shfl_xor(btile[xt], 32);
                    int b = best[xt], s = sec[xt], t = btile[xt];
                    const bool ow = (ob > b) || (ob == b && (unsigned)ot < (unsigned)t);
                    s = max(max(s, os), ow ? b : ob);
                    t = ow ? ot : t;
                    b = ow ? ob : b;
                    const int k = rowbase + xt * 32 + l31;
                    bool acc = false;
                    if (lh == 0 && k < nrows) {
                        const int row = MODE == 0 ? k : (int)list[k];
                        Top2 o;
#if defined(AMC_DIAG) && (AMC_DIAG & 1)
                        // timing diagnostic: every row "has no match" (the scan state only feeds the pad word)
                        o.best_v = 0; o.best_idx = 0xFFFFFFFFu; o.second_v = 0; o.pad = (uint32_t)(b ^ s ^ t) & 0u;
                        out[row] = o;
#else
                        o.best_v = (uint32_t)(b + xterm[xt]);
                        o.best_idx = o.best_v ? (uint32_t)t : 0xFFFFFFFFu;  // TILE of the best
                        o.second_v = (uint32_t)(s + xterm[xt]);
                        o.pad = 0;
                        out[row] = o;
                        // a larger second only ever rejects: rows failing now can be forgotten
                        if (MODE == 0) acc = one_way_accepts(o, lut, fp.max_ratio, fp.max_distance);
#endif
                    }
                    if (MODE == 0) {  // accept bits of rows rowbase + xt*32 .. +31 (lanes 0..31)
                        const uint32_t bits = (uint32_t)__ballot(acc);
                        if (lane == 0) accmask[(p.row_off + (uint64_t)(rowbase + xt * 32)) >> 5] = bits;
                    }
                }
            }
        }
    }
}

void launch_match_mfma(int mode, const ImageDev* imgs, const PairDev* pairs,
                       const uint32_t* order, uint32_t nitems, uint32_t* queue_head,
                       const uint32_t* cand_cnt, const uint32_t* candbuf, Top2* outbuf,
                       uint32_t* accmask, const float* acos_lut, FinalizeParams fp, hipStream_t s) {
    if (nitems == 0) return;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t grid = nitems < (uint32_t)cus ? nitems : (uint32_t)cus;  // 1 WG per CU
    (void)hipMemsetAsync(queue_head, 0, sizeof(uint32_t), s);
    if (mode == 0)
        hipLaunchKernelGGL((match_mfma_kernel<0>), dim3(grid), dim3(512), 0, s, imgs, pairs,
                           order, nitems, queue_head, cand_cnt, candbuf, outbuf, accmask, acos_lut, fp);
    else
        hipLaunchKernelGGL((match_mfma_kernel<1>), dim3(grid), dim3(512), 0, s, imgs, pairs,
                           order, nitems, queue_head, cand_cnt, candbuf, outbuf, accmask, acos_lut, fp);
}

}  // namespace amc
