// match_mfma.hip — the hot kernel: all-pairs 128-D u8 descriptor dot products on the int8
// matrix cores (v_mfma_i32_32x32x32_i8) with the two one-way top-2 scans of COLMAP's
// FindBestMatchesBruteForce fused into the epilogue.  gfx950 only.
//
// Every dot product is computed ONCE and feeds both directions:
//   rows    (image 1 -> image 2): packed (value, index) keys, exact lowest-index tie break
//   columns (image 2 -> image 1): values only; finalize cross-checks by value equality,
//                                 which is exact whenever max_ratio <= 1.
//
// Exactness of the int8 path.  Descriptors are u8 in [0,255]; the matrix core is signed, so
// the prepared arena holds a' = a - 128 (bytes ^ 0x80) and
//     sum a*b = sum a'*b' + 128*SA_i + 128*SB_j - 2^21,   SA/SB = plain byte sums.
// All of it stays in int32 (|sum a'b'| <= 2^21).  The row term rides in for free as the MFMA's
// C operand (a persistent register block per 32-row tile, distinct from the D operand); the
// column term is folded into the per-column constant of the single v_lshl_add that packs the
// row key.  Epilogue cost: 2.5 VALU/output for rows + 1.5 for columns (two-at-a-time
// med3/max3 insertion), vs 0.25 MFMA issue/output.
//
// Shape.  One 512-thread workgroup per image pair (dynamic queue, pairs sorted by image 2 so
// co-resident workgroups stream the same B image out of L2).  8 waves x 64 rows = a 512-row
// block of image 1 held in registers (A fragments, C-init block, row top-2 state); image 2
// streams through LDS in 256-column chunks by direct-to-LDS DMA, double buffered, one barrier
// per chunk; the prepared arena is pre-swizzled so the linear DMA image is bank-conflict-free
// for ds_read_b128 fragment reads.  Column top-2 state for the whole of image 2 lives in LDS
// and is merged with two ds_max atomics per lane per 32-column tile.
#include "amc_internal.h"

namespace amc {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kBN = 256;                // columns per LDS chunk
constexpr int kWaves = 8;               // waves per workgroup
constexpr int kWM = 64;                 // rows per wave (two 32-row MFMA tiles)
constexpr int kBM = kWaves * kWM;       // 512 rows per row block
constexpr int kMaxCols = 8192;          // LDS column-state capacity
constexpr int kChunkBytes = kBN * kDim; // 32 KiB

// single LDS object (a second __shared__ object de-pipelines the DMA waits)
constexpr int kOffB = 0;                               // 2 x 32 KiB descriptor chunks
constexpr int kOffRs = 2 * kChunkBytes;                // 2 x 1 KiB rs128 chunks
constexpr int kOffColB = kOffRs + 2 * kBN * 4;         // int[kMaxCols] column best
constexpr int kOffColS = kOffColB + kMaxCols * 4;      // int[kMaxCols] column second
constexpr int kOffQ = kOffColS + kMaxCols * 4;         // queue slot
constexpr int kLdsBytes = kOffQ + 16;

size_t match_mfma_max_cols() { return kMaxCols; }

__device__ __forceinline__ int smed3(int a, int b, int c) {
    return max(min(a, b), min(max(a, b), c));
}
__device__ __forceinline__ int smax3(int a, int b, int c) { return max(max(a, b), c); }

// insert two candidates into a (best, second) pair: 3 VALU for 2 elements
__device__ __forceinline__ void top2_insert2(int& best, int& second, int x, int y) {
    const int t = smed3(best, x, y);
    best = smax3(best, x, y);
    second = max(second, t);
}

template <int SHIFT, bool CROSS>
__global__ __launch_bounds__(512) void match_mfma_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const uint32_t* __restrict__ order, uint32_t npairs, uint32_t* __restrict__ queue_head,
    Top2* __restrict__ rowbuf, Top2* __restrict__ colbuf) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
    constexpr int IDXMASK = (1 << SHIFT) - 1;
    constexpr int VB = 1 << (31 - SHIFT);  // value bias: key high part = v - VB, signed
    constexpr int INT_MIN_ = -2147483647 - 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int* colB = reinterpret_cast<int*>(smem + kOffColB);
    int* colS = reinterpret_cast<int*>(smem + kOffColS);
    volatile uint32_t* s_q = reinterpret_cast<volatile uint32_t*>(smem + kOffQ);

    for (;;) {
        if (tid == 0) *s_q = atomicAdd(queue_head, 1u);
        __syncthreads();
        const uint32_t q = *s_q;
        if (q >= npairs) break;
        const PairDev p = pairs[order[q]];
        const ImageDev A = imgs[p.slot1];
        const ImageDev B = imgs[p.slot2];
        const int nchunks = (int)((B.rows + kBN - 1) / kBN);
        const int ncols = nchunks * kBN;
        const int nrb = (int)((A.rows + kBM - 1) / kBM);

        if (CROSS) {
            // floor of the value-only column scan: v = 0  <=>  acc = 2^21 - rs128_B[j] - VB
            for (int j = tid; j < ncols; j += 512) {
                const int f = (1 << 21) - B.rs128[j] - VB;
                colB[j] = f;
                colS[j] = f;
            }
        }

        auto stage = [&](int c, int buf) {
            // 32 KiB descriptor chunk: 4 x (8 waves x 1 KiB); dest = wave-uniform base + lane*16
            const char* src = reinterpret_cast<const char*>(B.prep) + (size_t)c * kChunkBytes;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int piece = ps * kWaves + wid;
                __builtin_amdgcn_global_load_lds(
                    (gvoid_t*)(src + piece * 1024 + lane * 16),
                    (lvoid_t*)(smem + kOffB + buf * kChunkBytes + piece * 1024), 16, 0, 0);
            }
            if (wid == 0) {  // 1 KiB of rs128 for this chunk's 256 columns
                const char* rsrc = reinterpret_cast<const char*>(B.rs128 + (size_t)c * kBN);
                __builtin_amdgcn_global_load_lds((gvoid_t*)(rsrc + lane * 16),
                                                 (lvoid_t*)(smem + kOffRs + buf * kBN * 4), 16,
                                                 0, 0);
            }
        };

        for (int rb = 0; rb < nrb; ++rb) {
            const int rowbase = rb * kBM + wid * kWM;
            const bool active = rowbase < (int)A.rows;  // wave-uniform

            // ---- per-row-block register state -----------------------------------------
            i32x4 afrag[2][4];
            i32x16 cinit[2];
            int rbest[2][16], rsec[2][16];
            if (active) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int row = rowbase + mt * 32 + l31;
                    const char* rp = reinterpret_cast<const char*>(A.prep) + (size_t)row * kDim;
                    const int sw = (row >> 1) & 7;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int qs = (2 * s + lh) ^ sw;
                        afrag[mt][s] = *reinterpret_cast<const i32x4*>(rp + qs * 16);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row_r = rowbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        cinit[mt][r] = A.rs128[row_r] - VB;
                        rbest[mt][r] = INT_MIN_;
                        rsec[mt][r] = INT_MIN_;
                    }
                }
            }

            stage(0, 0);
            for (int c = 0; c < nchunks; ++c) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
                if (!active) continue;

                const char* bbase = smem + kOffB + (c & 1) * kChunkBytes;
                const int* rsb = reinterpret_cast<const int*>(smem + kOffRs + (c & 1) * kBN * 4);
#pragma unroll 1
                for (int np = 0; np < kBN / 64; ++np) {
                    // two 32-column tiles per step
                    i32x4 bfrag[2][4];
                    int cj[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = np * 64 + t * 32 + l31;  // within chunk
                        const int sw = (col >> 1) & 7;
                        const char* cp = bbase + col * kDim;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int qs = (2 * s + lh) ^ sw;
                            bfrag[t][s] = *reinterpret_cast<const i32x4*>(cp + qs * 16);
                        }
                        const int jg = c * kBN + col;
                        // ((128*SB_j - 2^21) << SHIFT) + (IDXMASK - j), wraps mod 2^32
                        cj[t] = (int)(((uint32_t)(rsb[col] - (1 << 21)) << SHIFT) +
                                      (uint32_t)(IDXMASK - jg));
                    }
                    int cb[2] = {INT_MIN_, INT_MIN_}, cs[2] = {INT_MIN_, INT_MIN_};
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        i32x16 acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(
                            afrag[mt][0], bfrag[0][0], cinit[mt], 0, 0, 0);
                        i32x16 acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(
                            afrag[mt][0], bfrag[1][0], cinit[mt], 0, 0, 0);
#pragma unroll
                        for (int s = 1; s < 4; ++s) {
                            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[mt][s], bfrag[0][s],
                                                                         acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[mt][s], bfrag[1][s],
                                                                         acc1, 0, 0, 0);
                        }
                        // rows: acc + colterm_j = v - VB; key = (that << SHIFT) + (IDXMASK - j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int x = (int)(((uint32_t)acc0[r] << SHIFT) + (uint32_t)cj[0]);
                            const int y = (int)(((uint32_t)acc1[r] << SHIFT) + (uint32_t)cj[1]);
                            top2_insert2(rbest[mt][r], rsec[mt][r], x, y);
                        }
                        if (CROSS) {
                            // columns: all 16 registers of a tile share this lane's column
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                top2_insert2(cb[0], cs[0], acc0[r], acc0[r + 1]);
                                top2_insert2(cb[1], cs[1], acc1[r], acc1[r + 1]);
                            }
                        }
                    }
                    if (CROSS) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int jg = c * kBN + np * 64 + t * 32 + l31;
                            const int old = atomicMax(&colB[jg], cb[t]);
                            const int loser = min(old, cb[t]);
                            atomicMax(&colS[jg], max(loser, cs[t]));
                        }
                    }
                }
            }

            // ---- row block done: merge the 32 lanes (l31) that share each row, store ----
            if (active) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int b = rbest[mt][r], s = rsec[mt][r];
#pragma unroll
                        for (int m = 1; m < 32; m <<= 1) {
                            const int ob = __shfl_xor(b, m);
                            const int os = __shfl_xor(s, m);
                            const int lo = min(b, ob);
                            b = max(b, ob);
                            s = smax3(s, os, lo);
                        }
                        if (l31 == 0) {
                            const int row_r = rowbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            Top2 t;
                            t.best_v = (uint32_t)((b >> SHIFT) + VB);
                            t.best_idx = (uint32_t)(IDXMASK - (b & IDXMASK));
                            t.second_v = (uint32_t)((s >> SHIFT) + VB);
                            t.pad = 0;
                            rowbuf[p.row_off + row_r] = t;
                        }
                    }
                }
            }
            __syncthreads();  // everyone is done with both LDS chunk buffers
        }

        if (CROSS) {
            for (int j = tid; j < ncols; j += 512) {
                const int colterm = B.rs128[j] - (1 << 21);
                Top2 t;
                t.best_v = (uint32_t)(colB[j] + colterm + VB);
                t.best_idx = 0xFFFFFFFFu;
                t.second_v = (uint32_t)(colS[j] + colterm + VB);
                t.pad = 0;
                colbuf[p.col_off + j] = t;
            }
        }
        __syncthreads();  // LDS column state and queue slot are reused by the next pair
    }
}

void launch_match_mfma(const ImageDev* imgs, const PairDev* pairs, const uint32_t* order,
                       uint32_t npairs, int shift, int cross_check, uint32_t* queue_head,
                       Top2* rowbuf, Top2* colbuf, hipStream_t s) {
    if (npairs == 0) return;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t grid = npairs < (uint32_t)cus ? npairs : (uint32_t)cus;  // 1 WG per CU (LDS)
    (void)hipMemsetAsync(queue_head, 0, sizeof(uint32_t), s);
#define AMC_LAUNCH(SH, CR)                                                                      \
    hipLaunchKernelGGL((match_mfma_kernel<SH, CR>), dim3(grid), dim3(512), 0, s, imgs, pairs,   \
                       order, npairs, queue_head, rowbuf, colbuf)
    if (cross_check) {
        switch (shift) {
            case 13: AMC_LAUNCH(13, true); break;
            case 12: AMC_LAUNCH(12, true); break;
            case 11: AMC_LAUNCH(11, true); break;
            default: AMC_LAUNCH(10, true); break;
        }
    } else {
        switch (shift) {
            case 13: AMC_LAUNCH(13, false); break;
            case 12: AMC_LAUNCH(12, false); break;
            case 11: AMC_LAUNCH(11, false); break;
            default: AMC_LAUNCH(10, false); break;
        }
    }
#undef AMC_LAUNCH
}

}  // namespace amc
