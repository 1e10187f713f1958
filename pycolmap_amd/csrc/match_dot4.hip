// match_dot4.hip — exact u8 x u8 one-way top-2 kernel on v_dot4_u32_u8 (no MFMA).
//
// This is (a) the general-purpose path: no precondition on descriptor norms, image sizes or
// thresholds — it tracks (value, index) explicitly with COLMAP's own scan semantics — and
// (b) the "LDS-tiled + dot4 + wavefront-shuffle" design BASELINE.json's north_star names, kept
// as the comparison point for the int8-MFMA kernel (match_mfma.hip).
//
// Work item = one 64-row block of image X against ALL rows of image Y (one direction of one
// pair).  256 threads; thread (ty, tx) owns a 4x4 micro-tile of each 64x64 tile; the X block
// stays in LDS for the whole scan, Y tiles stream through LDS.  Each thread keeps the running
// (best, best_idx, second) of its 4 rows in registers over all Y tiles; the 16 lanes that share
// a row merge with __shfl_xor at the end.  The cross-check direction is a second work item
// with X and Y swapped — the same "second pass over the transposed matrix" COLMAP does
// (FindBestMatchesBruteForce).
//
// LDS layout is k-major ([32 dwords][64 rows]) so a thread reads its 4 rows / 4 columns of one
// k-dword as one conflict-free ds_read_b128.
#include "amc_internal.h"

namespace amc {

struct RowState {
    uint32_t bv, bj, sv;
};

// COLMAP's scan step: strict '>' keeps the lowest index among equal bests;
// second = second-largest with multiplicity; both floored at 0.
__device__ __forceinline__ void scan_step(RowState& s, uint32_t d, uint32_t j) {
    if (d > s.bv) {
        s.sv = s.bv;
        s.bv = d;
        s.bj = j;
    } else if (d > s.sv) {
        s.sv = d;
    }
}

// Order-independent merge of two partial scans over disjoint column sets.
__device__ __forceinline__ void merge_state(RowState& a, const RowState b) {
    const bool b_wins = (b.bv > a.bv) || (b.bv == a.bv && b.bj < a.bj);
    const uint32_t loser_bv = b_wins ? a.bv : b.bv;
    const uint32_t win_sv = b_wins ? b.sv : a.sv;
    a.bj = b_wins ? b.bj : a.bj;
    a.bv = b_wins ? b.bv : a.bv;
    a.sv = max(win_sv, loser_bv);
}

template <bool GUIDED>
__global__ __launch_bounds__(256) void match_dot4_kernel(const ImageDev* __restrict__ imgs,
                                                         const PairDev* __restrict__ pairs,
                                                         const Dot4Work* __restrict__ work,
                                                         Top2* __restrict__ rowbuf,
                                                         Top2* __restrict__ colbuf,
                                                         const GuidedDev* __restrict__ guided) {
    __shared__ __attribute__((aligned(16))) uint32_t Ak[32][64];
    __shared__ __attribute__((aligned(16))) uint32_t Bk[32][64];

    const Dot4Work w = work[blockIdx.x];
    const PairDev p = pairs[w.pair];
    const ImageDev X = imgs[w.dir == 0 ? p.slot1 : p.slot2];
    const ImageDev Y = imgs[w.dir == 0 ? p.slot2 : p.slot1];
    Top2* out = (w.dir == 0 ? rowbuf + p.row_off : colbuf + p.col_off) + (size_t)w.rb * 64;

    const uint32_t tid = threadIdx.x;
    const uint32_t tx = tid & 15, ty = tid >> 4;
    const uint32_t lr = tid & 63, lk = tid >> 6;  // staging role: row lr, 16-B pieces lk, lk+4

    // stage the X block once (k-major transpose; consecutive lanes -> consecutive rows, so the
    // LDS stores are conflict-free)
    {
        const uint4* g = reinterpret_cast<const uint4*>(X.raw + (size_t)w.rb * 64 * kDim);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const uint32_t k4 = lk + 4 * it;
            const uint4 v = g[lr * 8 + k4];
            Ak[4 * k4 + 0][lr] = v.x;
            Ak[4 * k4 + 1][lr] = v.y;
            Ak[4 * k4 + 2][lr] = v.z;
            Ak[4 * k4 + 3][lr] = v.w;
        }
    }

    RowState st[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) st[r] = RowState{0u, 0xFFFFFFFFu, 0u};

    // guided matching: this thread's 4 X keypoints (rows past the end are never stored)
    GuidedDev gd;
    float xk[4][2];
    if (GUIDED) {
        gd = guided[w.pair];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = min(w.rb * 64 + ty * 4 + r, X.kp_rows - 1);
            xk[r][0] = X.kp[2 * (size_t)i];
            xk[r][1] = X.kp[2 * (size_t)i + 1];
        }
    }

    const uint32_t ntiles = (Y.rows + 63) / 64;
    for (uint32_t ct = 0; ct < ntiles; ++ct) {
        __syncthreads();  // previous tile fully consumed (first pass: orders the Ak stores)
        {
            const uint4* g = reinterpret_cast<const uint4*>(Y.raw + (size_t)ct * 64 * kDim);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const uint32_t k4 = lk + 4 * it;
                const uint4 v = g[lr * 8 + k4];
                Bk[4 * k4 + 0][lr] = v.x;
                Bk[4 * k4 + 1][lr] = v.y;
                Bk[4 * k4 + 2][lr] = v.z;
                Bk[4 * k4 + 3][lr] = v.w;
            }
        }
        __syncthreads();

        uint32_t acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0;

#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const uint4 a = *reinterpret_cast<const uint4*>(&Ak[k][ty * 4]);
            const uint4 b = *reinterpret_cast<const uint4*>(&Bk[k][tx * 4]);
            const uint32_t av[4] = {a.x, a.y, a.z, a.w};
            const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = __builtin_amdgcn_udot4(av[r], bv[c], acc[r][c], false);
        }

        const uint32_t j0 = ct * 64 + tx * 4;
        if (GUIDED) {
            // the filter always sees (image-1 point, image-2 point), whichever image is being scanned
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t j = min(j0 + c, Y.kp_rows - 1);
                const float yx = Y.kp[2 * (size_t)j], yy = Y.kp[2 * (size_t)j + 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool rej = w.dir == 0 ? guided_rejects(gd, xk[r][0], xk[r][1], yx, yy)
                                                : guided_rejects(gd, yx, yy, xk[r][0], xk[r][1]);
                    if (rej) acc[r][c] = 0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) scan_step(st[r], acc[r][c], j0 + c);
    }

    // merge the 16 lanes (tx = 0..15) that share each row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            RowState o;
            o.bv = __shfl_xor(st[r].bv, m);
            o.bj = __shfl_xor(st[r].bj, m);
            o.sv = __shfl_xor(st[r].sv, m);
            merge_state(st[r], o);
        }
        if (tx == 0) {
            Top2 t;
            t.best_v = st[r].bv;
            t.best_idx = st[r].bj;
            t.second_v = st[r].sv;
            t.pad = 0;
            out[ty * 4 + r] = t;
        }
    }
}

hipError_t launch_match_dot4(const ImageDev* imgs, const PairDev* pairs, const Dot4Work* work,
                             uint32_t nwork, Top2* rowbuf, Top2* colbuf, const GuidedDev* guided, hipStream_t s) {
    if (nwork == 0) return hipSuccess;
    if (guided)
        hipLaunchKernelGGL((match_dot4_kernel<true>), dim3(nwork), dim3(256), 0, s, imgs, pairs, work,
                           rowbuf, colbuf, guided);
    else
        hipLaunchKernelGGL((match_dot4_kernel<false>), dim3(nwork), dim3(256), 0, s, imgs, pairs, work,
                           rowbuf, colbuf, guided);
    return hipGetLastError();
}

}  // namespace amc
