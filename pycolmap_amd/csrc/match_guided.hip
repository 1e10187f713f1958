// match_guided.hip — guided matching (FeatureMatcher::MatchGuided) by candidate generation.
//
// MatchGuided scores the full n1 x n2 matrix and zeroes every entry the geometric filter rejects
// (oracle/match_oracle.c: oracle_match_guided; COLMAP 3.9.1 colmap/feature/sift.cc).  A zero entry can
// never become a best or a second (strict '>' against floors of 0), so the one-way top-2 of a row is the
// top-2 over the entries the filter ACCEPTS - about 1 % of them for an epipolar band a few pixels wide,
// far fewer for a homography.  The dense kernel (match_dot4.hip, GUIDED) computes all n1 x n2 dot
// products and filters afterwards; this one finds the accepted entries first:
//
//   * the image being searched has its keypoints bucketed on a 64 x 64 grid (GridDev, built at upload);
//   * one wave per row.  Lane gy looks at grid row gy and computes the interval of cells a conservative
//     superset of the filter's acceptance region covers there: the band |l . (x, y, 1)| <= W around the
//     epipolar line l for F (Sampson error <= T implies point-line distance^2 <= T (1 + |other|^2 / |l|^2),
//     `other` bounded over the image's keypoint box), the box around H p for H, the H^-1 image of the box
//     around p for the reverse direction.  Cells of one grid row are contiguous in the CSR, so each lane has
//     ONE range of sorted keypoints;
//   * the ranges are flattened into an LDS list, every listed keypoint goes through the exact float32
//     filter (the same guided_rejects the dense kernel uses), survivors are compacted, and only those get
//     their 128-byte dot product (v_dot4_u32_u8 against the row's descriptor held in scalar registers);
//   * per-lane (best, index, second) states merge order-independently, lowest index among equal bests.
//
// The result is the dense kernel's, bit for bit: the superset only has to contain every entry the float32
// filter accepts (the slack - 2 % and one pixel - is orders above float32 rounding of the filter for image
// coordinates; rows whose geometry degenerates scan the whole grid; pairs whose model could make the filter
// return NaN = "not rejected" never come here: GuidedDev::grid_ok, amc_api.hip).
#include "amc_internal.h"

namespace amc {

namespace {

constexpr int kCandCap = 512;  // keypoints staged per wave and round

struct GState {
    uint32_t bv, bj, sv;
};
// order-independent merge of two partial scans over disjoint column sets (as match_dot4.hip's)
__device__ __forceinline__ void gmerge(GState& a, const GState b) {
    const bool b_wins = (b.bv > a.bv) || (b.bv == a.bv && b.bj < a.bj);
    const uint32_t loser_bv = b_wins ? a.bv : b.bv;
    const uint32_t win_sv = b_wins ? b.sv : a.sv;
    a.bj = b_wins ? b.bj : a.bj;
    a.bv = b_wins ? b.bv : a.bv;
    a.sv = max(win_sv, loser_bv);
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

__global__ __launch_bounds__(256) void match_guided_grid_kernel(const ImageDev* __restrict__ imgs,
                                                                const GridDev* __restrict__ grids,
                                                                const PairDev* __restrict__ pairs,
                                                                const Dot4Work* __restrict__ work,
                                                                Top2* __restrict__ rowbuf, Top2* __restrict__ colbuf,
                                                                const GuidedDev* __restrict__ guided) {
    __shared__ uint32_t s_list[4][kCandCap];
    __shared__ uint32_t s_acc[4][kCandCap];

    const Dot4Work w = work[blockIdx.x];
    const PairDev p = pairs[w.pair];
    const uint32_t dir = w.dir;
    const ImageDev X = imgs[dir == 0 ? p.slot1 : p.slot2];
    const ImageDev Y = imgs[dir == 0 ? p.slot2 : p.slot1];
    const GridDev G = grids[dir == 0 ? p.slot2 : p.slot1];
    // the pair's model, field by field with constant indices (a run-time index into the struct would park all of
    // it - and the filter's nine coefficients with it - in scratch memory)
    GuidedDev gd;
    {
        const GuidedDev* __restrict__ gp = guided + w.pair;
        gd.kind = gp->kind;
        gd.max_residual = gp->max_residual;
#pragma unroll
        for (int k = 0; k < 9; ++k) gd.m[k] = gp->m[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) gd.minv[k] = gp->minv[k];
    }
    const double bound = w.dir == 0 ? guided[w.pair].bound[0] : guided[w.pair].bound[1];
    Top2* out = (dir == 0 ? rowbuf + p.row_off : colbuf + p.col_off) + (size_t)w.rb * 64;

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    uint32_t* list = s_list[wid];
    uint32_t* acc = s_acc[wid];

    double m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = (double)gd.m[k];
    const double T = (double)gd.max_residual;
    // lane = grid row: the y interval its keypoints lie in (1 % of a cell on either side covers the float
    // rounding of grid_cell)
    double ylo, yhi;
    guided::grid_row_interval(G.y0, G.ch, lane, ylo, yhi);

    for (int rr = 0; rr < 16; ++rr) {
        const uint32_t row = w.rb * 64 + (uint32_t)wid * 16 + (uint32_t)rr;  // wave-uniform
        if (row >= X.rows) {
            if (lane == 0) out[wid * 16 + rr] = Top2{0u, 0xFFFFFFFFu, 0u, 0u};
            continue;
        }
        const float pxf = X.kp[2 * (size_t)row], pyf = X.kp[2 * (size_t)row + 1];
        const double px = (double)pxf, py = (double)pyf;
        // the row's descriptor, one dword per scalar register
        uint32_t xd[32];
        {
            const uint32_t* xr = reinterpret_cast<const uint32_t*>(X.raw + (size_t)row * kDim);
            const uint32_t v = xr[lane & 31];
#pragma unroll
            for (int i = 0; i < 32; ++i) xd[i] = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
        }

        // ---- the acceptance region's cells in this lane's grid row: [xa, xb], or nothing (guided_region.h) ----
        double xa, xb;
        const bool none = !guided::guided_row_region(gd.kind, (int)dir, m, gd.minv, T, bound, px, py, ylo, yhi,
                                                     (double)G.cw + (double)G.ch, xa, xb);
        uint32_t s0 = 0, len = 0;
        int gx0, gx1;
        if (!none && guided::grid_cells_of(xa, xb, G.x0, G.bx1, G.inv_cw, gx0, gx1)) {
            s0 = G.cell_start[lane * kGridDim + gx0];
            len = G.cell_start[lane * kGridDim + gx1 + 1] - s0;
        }
        // exclusive prefix of the range lengths over the lanes
        uint32_t incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const uint32_t pre = incl - len;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);

        GState st{0u, 0xFFFFFFFFu, 0u};
        for (uint32_t base = 0; base < total; base += kCandCap) {
            const uint32_t k0 = max(pre, base), k1 = min(pre + len, base + (uint32_t)kCandCap);
            for (uint32_t k = k0; k < k1; ++k) list[k - base] = s0 + (k - pre);
            lds_sync();
            const uint32_t cnt = min(total - base, (uint32_t)kCandCap);
            // the exact float32 filter on every listed keypoint; survivors compacted into acc
            uint32_t nacc = 0;
            for (uint32_t k = lane; k < (cnt + 63u) / 64u * 64u; k += 64) {
                bool ok = false;
                uint32_t j = 0;
                if (k < cnt) {
                    const uint32_t pos = list[k];
                    const float qx = G.sxy[2 * (size_t)pos], qy = G.sxy[2 * (size_t)pos + 1];
                    j = G.sidx[pos];
                    const bool rej = dir == 0 ? guided_rejects(gd, pxf, pyf, qx, qy) : guided_rejects(gd, qx, qy, pxf, pyf);
                    ok = !rej && j < Y.rows;
                }
                const unsigned long long bal = __ballot(ok);
                if (ok) acc[nacc + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = j;
                nacc += (uint32_t)__popcll(bal);
            }
            lds_sync();
            for (uint32_t k = lane; k < nacc; k += 64) {
                const uint32_t j = acc[k];
                const uint4* yr = reinterpret_cast<const uint4*>(Y.raw + (size_t)j * kDim);
                uint32_t d = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint4 v = yr[q];
                    d = __builtin_amdgcn_udot4(xd[4 * q + 0], v.x, d, false);
                    d = __builtin_amdgcn_udot4(xd[4 * q + 1], v.y, d, false);
                    d = __builtin_amdgcn_udot4(xd[4 * q + 2], v.z, d, false);
                    d = __builtin_amdgcn_udot4(xd[4 * q + 3], v.w, d, false);
                }
                if (d > 0) gmerge(st, GState{d, j, 0u});
            }
            lds_sync();  // the next round rewrites both lists
        }
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            GState o;
            o.bv = __shfl_xor(st.bv, msk);
            o.bj = __shfl_xor(st.bj, msk);
            o.sv = __shfl_xor(st.sv, msk);
            gmerge(st, o);
        }
        if (lane == 0) out[wid * 16 + rr] = Top2{st.bv, st.bj, st.sv, 0u};
    }
}

hipError_t launch_match_guided_grid(const ImageDev* imgs, const GridDev* grids, const PairDev* pairs, const Dot4Work* work,
                                    uint32_t nwork, Top2* rowbuf, Top2* colbuf, const GuidedDev* guided, hipStream_t s) {
    if (nwork == 0) return hipSuccess;
    hipLaunchKernelGGL(match_guided_grid_kernel, dim3(nwork), dim3(256), 0, s, imgs, grids, pairs, work, rowbuf, colbuf,
                       guided);
    return hipGetLastError();
}

}  // namespace amc
