// match_mfma_guided.hip — the one-way top-2 scan of match_mfma.hip on the FILTERED distance matrix of
// guided matching (COLMAP 3.9.1 SiftCPUFeatureMatcher::MatchGuided): an entry
// whose two keypoints violate the pair's geometry (float32 Sampson error under F, transfer error
// under H; guided_rejects in amc_internal.h) scores 0, everything else is the plain scan.
//
// Evaluating the filter for every entry costs ~20 VALU per distance and is what bounds the dot4
// kernel.  Here the int8 MFMA produces the unfiltered unit (32 Y rows x 32 X rows, 16 outputs per
// lane) as in the plain kernel, and the filter is only consulted where it can matter:
//   * m = maximum of the lane's 16 outputs (8 VALU).  If m does not exceed the lane's running second
//     of unit maxima, nothing in the unit - kept or zeroed - can change the lane's state: done.
//   * otherwise the lane walks its outputs from the largest down (keys = value << 4 | register,
//     so the maximum carries its position): the first one the filter keeps is the unit's filtered
//     maximum; one it rejects is struck out and the next largest is tried, until the candidates
//     fall to the running second.  A row sees ~2 ln(n) such events over a whole scan.
// The rest is the plain kernel's contract: per X row the best filtered value, the 32-row tile holding
// it and the largest filtered unit maximum outside that unit; resolve_index_kernel recomputes the
// winning tile (through the same filter) for the exact index and second.  MODE 0: X = image 1,
// Y = image 2.  MODE 1: X = the candidate rows of image 2, Y = image 1; the filter always sees
// (image-1 point, image-2 point).
//
// Shape: as match_mfma.hip (512 threads, Y streamed through LDS in 256-row chunks by DMA, double
// buffered; the chunk's keypoints ride along), but two resident X tiles per wave instead of four
// (the walk needs the registers) and no hand-interleaved phases: this kernel is bound by the walk,
// the two waves of a SIMD overlap one's MFMAs with the other's VALU on their own.
#include <climits>

#include "amc_internal.h"

namespace amc {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int gBN = 256;                // Y rows per LDS chunk (= kRowPad)
constexpr int gYT = gBN / 32;           // Y tiles per chunk
constexpr int gWaves = 8;
constexpr int gXT = 2;                  // resident X tiles per wave
constexpr int gWM = 32 * gXT;           // X rows per wave
constexpr int gBM = gWaves * gWM;       // 512 X rows per row block
constexpr int gChunkBytes = gBN * kDim; // 32 KiB

constexpr int gOffRs = 0;                          // 2 x rs128 chunks (gBN ints each)
constexpr int gOffQ = gOffRs + 2 * gBN * 4;        // queue slot
constexpr int gOffKp = gOffQ + 16;                 // 2 x keypoint chunks (gBN x (x, y) float32)
constexpr int gOffB = gOffKp + 2 * gBN * 8;        // 2 x descriptor chunks
constexpr int gLdsBytes = gOffB + 2 * gChunkBytes;
static_assert(gBN == kRowPad, "a chunk is the row padding unit");
static_assert((gChunkBytes / 1024) % gWaves == 0, "1 KiB DMA pieces per wave");

struct YFrag {
    i32x4 f[4];
    i32x16 ci;
};

template <int MODE>
__global__ __launch_bounds__(512) void match_mfma_guided_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const uint32_t* __restrict__ order, uint32_t nitems, uint32_t* __restrict__ queue_head,
    const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ candbuf,
    Top2* __restrict__ outbuf, uint32_t* __restrict__ accmask, const float* __restrict__ lut,
    FinalizeParams fp, const GuidedDev* __restrict__ guided) {
    __shared__ __attribute__((aligned(16))) char smem[gLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    volatile uint32_t* s_q = reinterpret_cast<volatile uint32_t*>(smem + gOffQ);

    for (;;) {
        if (tid == 0) *s_q = atomicAdd(queue_head, 1u);
        __syncthreads();
        const uint32_t q = *s_q;
        __syncthreads();
        if (q >= nitems) break;
        const uint32_t pi = order[q];
        const PairDev p = pairs[pi];
        const GuidedDev gd = guided[pi];
        const ImageDev X = imgs[MODE == 0 ? p.slot1 : p.slot2];
        const ImageDev Y = imgs[MODE == 0 ? p.slot2 : p.slot1];
        const int nrows = (int)(MODE == 0 ? X.rows : cand_cnt[pi]);
        const uint32_t* list = candbuf + p.col_off;
        Top2* out = outbuf + (MODE == 0 ? p.row_off : p.col_off);
        if (nrows == 0 || Y.rows == 0) continue;
        const int nchunks = (int)((Y.rows + gBN - 1) / gBN);
        const int nrb = (nrows + gBM - 1) / gBM;

        auto stage = [&](int c, int buf) {
            const char* src = reinterpret_cast<const char*>(Y.prep) + (size_t)c * gChunkBytes;
#pragma unroll
            for (int ps = 0; ps < gChunkBytes / 1024 / gWaves; ++ps) {
                const int piece = ps * gWaves + wid;
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + piece * 1024 + lane * 16),
                                                 (lvoid_t*)(smem + gOffB + buf * gChunkBytes + piece * 1024), 16, 0, 0);
            }
            if (wid == 0) {  // rs128 of the chunk's rows: 1 KiB
                const char* rsrc = reinterpret_cast<const char*>(Y.rs128 + (size_t)c * gBN);
                __builtin_amdgcn_global_load_lds((gvoid_t*)(rsrc + lane * 16),
                                                 (lvoid_t*)(smem + gOffRs + buf * gBN * 4), 16, 0, 0);
            } else if (wid <= 2) {  // the chunk's keypoints (x, y): 2 KiB, the array is zero-padded to whole chunks
                const char* ksrc = reinterpret_cast<const char*>(Y.kp) + (size_t)c * gBN * 8 + (wid - 1) * 1024;
                __builtin_amdgcn_global_load_lds((gvoid_t*)(ksrc + lane * 16),
                                                 (lvoid_t*)(smem + gOffKp + buf * gBN * 8 + (wid - 1) * 1024), 16, 0, 0);
            }
        };
        auto load_y = [&](YFrag& y, int buf, int yt) {
            const int row = yt * 32 + l31;
            const int sw = (row >> 1) & 7;
            const char* cp = smem + gOffB + buf * gChunkBytes + row * kDim;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                y.f[s] = *reinterpret_cast<const i32x4*>(cp + (((2 * s + lh) ^ sw) * 16));
            const int* rsb = reinterpret_cast<const int*>(smem + gOffRs + buf * gBN * 4) + yt * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const i32x4 v = *reinterpret_cast<const i32x4*>(rsb + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) y.ci[4 * g + e] = v[e];
            }
        };

        stage(0, 0);
        for (int rb = 0; rb < nrb; ++rb) {
            const int rowbase = rb * gBM + wid * gWM;
            const bool active = rowbase < nrows;  // wave-uniform

            i32x4 xf[gXT][4];
            int best[gXT], sec[gXT], btile[gXT], xterm[gXT];
            float xkx[gXT], xky[gXT];
            if (active) {
#pragma unroll
                for (int xt = 0; xt < gXT; ++xt) {
                    const int k = rowbase + xt * 32 + l31;
                    const int row = MODE == 0 ? min(k, (int)X.rows_pad - 1) : (int)list[min(k, nrows - 1)];
                    const char* rp = reinterpret_cast<const char*>(X.prep) + (size_t)row * kDim;
                    const int sw = (row >> 1) & 7;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        xf[xt][s] = *reinterpret_cast<const i32x4*>(rp + (((2 * s + lh) ^ sw) * 16));
                    xterm[xt] = X.rs128[row] - (1 << 21);
                    best[xt] = -xterm[xt];
                    sec[xt] = -xterm[xt];
                    btile[xt] = -1;
                    const int kr = min(row, (int)X.kp_rows - 1);
                    xkx[xt] = X.kp[2 * (size_t)kr];
                    xky[xt] = X.kp[2 * (size_t)kr + 1];
                }
            }

            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (nchunks > 1) stage(1, 1);
            for (int c = 0; c < nchunks; ++c) {
                const int buf = c & 1;
                if (active) {
                    const float* kpb = reinterpret_cast<const float*>(smem + gOffKp + buf * gBN * 8);
#pragma unroll 1
                    for (int yt = 0; yt < gYT; ++yt) {
                        YFrag y;
                        load_y(y, buf, yt);
                        const int tile = c * gYT + yt;
#pragma unroll
                        for (int xt = 0; xt < gXT; ++xt) {
                            i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[0], xf[xt][0], y.ci, 0, 0, 0);
#pragma unroll
                            for (int s = 1; s < 4; ++s)
                                a = __builtin_amdgcn_mfma_i32_32x32x32_i8(y.f[s], xf[xt][s], a, 0, 0, 0);
                            // unfiltered unit maximum
                            int m = a[0];
#pragma unroll
                            for (int i = 1; i < 16; ++i) m = max(m, a[i]);
                            if (m > sec[xt]) {
                                // walk the unit's outputs from the largest down; key = value << 4 | register
                                // (|value| < 2^27: a dot product of u8 vectors and its two zero-point terms)
                                int key[16];
#pragma unroll
                                for (int i = 0; i < 16; ++i) key[i] = (int)(((unsigned)a[i] << 4) | (unsigned)i);
                                const int floor_v = -xterm[xt];
                                m = floor_v;
                                for (;;) {
                                    int km = key[0];
#pragma unroll
                                    for (int i = 1; i < 16; ++i) km = max(km, key[i]);
                                    const int val = km >> 4;
                                    if (val <= sec[xt]) break;  // nothing left that could change the state
                                    const int idx = km & 15;
                                    // accumulator register r <-> Y row (r&3) + 8*(r>>2) + 4*lh of the tile
                                    const int yrow = yt * 32 + (idx & 3) + 8 * (idx >> 2) + 4 * lh;
                                    const float yx = kpb[2 * yrow], yy = kpb[2 * yrow + 1];
                                    const bool rej = MODE == 0 ? guided_rejects(gd, xkx[xt], xky[xt], yx, yy)
                                                               : guided_rejects(gd, yx, yy, xkx[xt], xky[xt]);
                                    if (!rej) {
                                        m = val;
                                        break;
                                    }
#pragma unroll
                                    for (int i = 0; i < 16; ++i) key[i] = (i == idx) ? INT_MIN : key[i];
                                }
                            }
                            // insertion of the unit's filtered maximum into (best, second, tile)
                            const int b0 = best[xt];
                            const int lo = min(b0, m), hi = max(b0, m);
                            sec[xt] = max(sec[xt], lo);   // sec <= best always
                            btile[xt] = (hi != b0) ? tile : btile[xt];  // strict: the first tile wins ties
                            best[xt] = hi;
                        }
                    }
                }
                if (c + 1 < nchunks) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();  // chunk c+1 landed, everyone is done with chunk c's buffer
                    if (c + 2 < nchunks) stage(c + 2, c & 1);
                }
            }
            __syncthreads();
            if (rb + 1 < nrb) stage(0, 0);

            // ---- row block done: merge the two lane halves, decode, store (as match_mfma.hip) ----
            if (active) {
#pragma unroll
                for (int xt = 0; xt < gXT; ++xt) {
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
                        o.best_v = (uint32_t)(b + xterm[xt]);
                        o.best_idx = o.best_v ? (uint32_t)t : 0xFFFFFFFFu;  // TILE of the best
                        o.second_v = (uint32_t)(s + xterm[xt]);
                        o.pad = 0;
                        out[row] = o;
                        if (MODE == 0) acc = one_way_accepts(o, lut, fp.max_ratio, fp.max_distance);
                    }
                    if (MODE == 0) {
                        const uint32_t bits = (uint32_t)__ballot(acc);
                        if (lane == 0) accmask[(p.row_off + (uint64_t)(rowbase + xt * 32)) >> 5] = bits;
                    }
                }
            }
        }
    }
}

}  // namespace

void launch_match_mfma_guided(int mode, const ImageDev* imgs, const PairDev* pairs,
                              const uint32_t* order, uint32_t nitems, uint32_t* queue_head,
                              const uint32_t* cand_cnt, const uint32_t* candbuf, Top2* outbuf,
                              uint32_t* accmask, const float* acos_lut, FinalizeParams fp,
                              const GuidedDev* guided, hipStream_t s) {
    if (nitems == 0) return;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t grid = nitems < (uint32_t)cus ? nitems : (uint32_t)cus;  // 1 WG per CU
    (void)hipMemsetAsync(queue_head, 0, sizeof(uint32_t), s);
    if (mode == 0)
        hipLaunchKernelGGL((match_mfma_guided_kernel<0>), dim3(grid), dim3(512), 0, s, imgs, pairs, order, nitems,
                           queue_head, cand_cnt, candbuf, outbuf, accmask, acos_lut, fp, guided);
    else
        hipLaunchKernelGGL((match_mfma_guided_kernel<1>), dim3(grid), dim3(512), 0, s, imgs, pairs, order, nitems,
                           queue_head, cand_cnt, candbuf, outbuf, accmask, acos_lut, fp, guided);
}

}  // namespace amc
