// match_common.hip — descriptor preparation and the finalize (thresholds + cross-check +
// ordered compaction) kernel shared by both match kernels.  gfx950 only.
#include "amc_internal.h"

namespace amc {

// ---------------------------------------------------------------------------------------
// prep: raw u8 rows -> (a) signed-offset copy (u8 ^ 0x80 == int8(u8 - 128)) in the LDS-bank
// swizzled slot order the MFMA kernel stages linearly, (b) rs128[r] = 128 * sum_k raw[r][k]
// (the MFMA kernel's exact zero-point correction, see match_mfma.hip), (c) max_r |raw[r]|^2
// (host-side exactness precondition of the packed-key fast path).
// One thread per 16-byte slot, 8 threads per row; HBM-bound, coalesced 16 B/lane both ways.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const uint8_t* __restrict__ raw,
                                                   uint8_t* __restrict__ prep,
                                                   int32_t* __restrict__ rs128,
                                                   uint32_t rows_pad,
                                                   uint32_t* __restrict__ maxsq_out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t r = gid >> 3, q = gid & 7;
    uint32_t sum = 0, sq = 0;
    if (r < rows_pad) {
        const uint4 v = *reinterpret_cast<const uint4*>(raw + (size_t)r * kDim + q * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sum = __builtin_amdgcn_udot4(w[i], 0x01010101u, sum, false);
            sq = __builtin_amdgcn_udot4(w[i], w[i], sq, false);
        }
        uint4 o;
        o.x = v.x ^ 0x80808080u; o.y = v.y ^ 0x80808080u;
        o.z = v.z ^ 0x80808080u; o.w = v.w ^ 0x80808080u;
        const uint32_t qs = q ^ ((r >> 1) & 7u);
        *reinterpret_cast<uint4*>(prep + (size_t)r * kDim + qs * 16) = o;
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        sum += __shfl_xor(sum, m);
        sq += __shfl_xor(sq, m);
    }
    if (r < rows_pad && q == 0) rs128[r] = (int32_t)(sum * 128u);
    // wave max of the per-row squared norms, one atomic per wave
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) sq = max(sq, (uint32_t)__shfl_xor(sq, m));
    if ((threadIdx.x & 63) == 0) atomicMax(maxsq_out, sq);
}

void launch_prep(const uint8_t* raw, uint8_t* prep, int32_t* rs128, uint32_t rows_pad,
                 uint32_t* maxsq_out, hipStream_t s) {
    if (rows_pad == 0) return;
    const uint32_t nthreads = rows_pad * 8u;
    hipLaunchKernelGGL(prep_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, s, raw, prep,
                       rs128, rows_pad, maxsq_out);
}

// ---------------------------------------------------------------------------------------
// finalize: one workgroup per pair.  Applies COLMAP's per-row acceptance tests
// (FindBestMatchesOneWayBruteForce) to both one-way tables, the cross check
// (FindBestMatchesBruteForce), and emits (idx1, idx2) ascending in idx1.
//
// acos thresholds: a LUT built on the HOST with the host libm, lut[d] = acosf(min(d/512^2,1));
// (float)d * 2^-18 is exact for d < 2^24, so indexing by min(d, 262144) reproduces COLMAP's
// float expression bit-for-bit without depending on the device's acosf.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool one_way_accepts(const Top2 t, const float* __restrict__ lut,
                                                float max_ratio, float max_distance) {
    if (t.best_v == 0u) return false;  // best_i2 == -1: nothing > 0
    const float a_best = lut[min(t.best_v, 262144u)];
    if (a_best > max_distance) return false;
    const float a_second = lut[min(t.second_v, 262144u)];
    // single IEEE multiply, nothing to contract with
    if (a_best >= max_ratio * a_second) return false;
    return true;
}

__global__ __launch_bounds__(256) void finalize_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const Top2* __restrict__ rowbuf, const Top2* __restrict__ colbuf,
    const float* __restrict__ lut, FinalizeParams fp, uint32_t* __restrict__ cursor,
    uint32_t capacity, uint32_t* __restrict__ pair_off, uint32_t* __restrict__ pair_cnt,
    uint32_t* __restrict__ matches) {
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t s_base;
    const PairDev p = pairs[blockIdx.x];
    const uint32_t n1 = imgs[p.slot1].rows, n2 = imgs[p.slot2].rows;
    const Top2* rows = rowbuf + p.row_off;
    const Top2* cols = colbuf + p.col_off;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool value_mode = (p.mode != 0);

    auto is_match = [&](uint32_t i, uint32_t& j_out) -> bool {
        if (i >= n1 || n2 == 0) return false;
        const Top2 t = rows[i];
        if (!one_way_accepts(t, lut, fp.max_ratio, fp.max_distance)) return false;
        const uint32_t j = t.best_idx;
        j_out = j;
        if (!fp.cross_check) return true;
        const Top2 c = cols[j];
        if (!one_way_accepts(c, lut, fp.max_ratio, fp.max_distance)) return false;
        // value mode: column j accepted => its maximum is unique (ties fail the ratio test
        // whenever max_ratio <= 1), and dist(i,j) == t.best_v lies in column j, so i is the
        // column's argmax iff the values agree.
        return value_mode ? (c.best_v == t.best_v) : (c.best_idx == i);
    };

    // pass 1: count
    uint32_t local = 0;
    for (uint32_t base = 0; base < n1; base += 256) {
        uint32_t j;
        local += is_match(base + tid, j) ? 1u : 0u;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) local += __shfl_xor(local, m);
    if (lane == 0) wave_cnt[wid] = local;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        const uint32_t b = total ? atomicAdd(cursor, total) : 0u;
        s_base = b;
        pair_off[blockIdx.x] = b;
        pair_cnt[blockIdx.x] = total;
    }
    __syncthreads();
    uint32_t running = s_base;
    // pass 2: ordered write
    for (uint32_t base = 0; base < n1; base += 256) {
        uint32_t j = 0;
        const bool m = is_match(base + tid, j);
        const unsigned long long bal = __ballot(m);
        const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wid; ++w) woff += wave_cnt[w];
        if (m) {
            const uint32_t dst = running + woff + before;
            if (dst < capacity) {
                matches[2ull * dst] = base + tid;
                matches[2ull * dst + 1] = j;
            }
        }
        running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

void launch_finalize(const ImageDev* imgs, const PairDev* pairs, uint32_t npairs,
                     const Top2* rowbuf, const Top2* colbuf, const float* acos_lut,
                     FinalizeParams fp, uint32_t* cursor, uint32_t capacity, uint32_t* pair_off,
                     uint32_t* pair_cnt, uint32_t* matches, hipStream_t s) {
    if (npairs == 0) return;
    hipLaunchKernelGGL(finalize_kernel, dim3(npairs), dim3(256), 0, s, imgs, pairs, rowbuf,
                       colbuf, acos_lut, fp, cursor, capacity, pair_off, pair_cnt, matches);
}

}  // namespace amc
