// match_common.hip — descriptor preparation and the finalize (thresholds + cross-check +
// ordered compaction) kernel shared by both match kernels.  gfx950 only.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>

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

// The test hook of memset_async / memcpy_async (amc_internal.h): true exactly once, on the k-th call made while the
// variable holds the positive integer k.
// The hooks exist only in a process that had AMC_TEST_HOOKS in its environment when the library made its first such
// call (read once): a production process pays one cached flag per memset / copy, not a getenv.
static bool fault_due(const char* name, std::atomic<long>& seen, std::atomic<long>& armed) {
    static const bool hooks = std::getenv("AMC_TEST_HOOKS") != nullptr;
    if (!hooks) return false;
    const char* e = std::getenv(name);
    const long k = e ? std::atol(e) : 0;
    if (k <= 0) {
        if (armed.load(std::memory_order_relaxed) != 0) {
            armed.store(0);
            seen.store(0);
        }
        return false;
    }
    if (armed.exchange(k) != k) seen.store(0);
    return seen.fetch_add(1) + 1 == k;
}
hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s) {
    static std::atomic<long> seen{0}, armed{0};
    if (fault_due("AMC_FAIL_NEXT_MEMSET", seen, armed)) return hipErrorInvalidValue;
    return hipMemsetAsync(p, value, bytes, s);
}
hipError_t memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    static std::atomic<long> seen{0}, armed{0};
    if (fault_due("AMC_FAIL_NEXT_MEMCPY", seen, armed)) return hipErrorInvalidValue;
    return hipMemcpyAsync(dst, src, bytes, kind, s);
}

hipError_t launch_prep(const uint8_t* raw, uint8_t* prep, int32_t* rs128, uint32_t rows_pad,
                       uint32_t* maxsq_out, hipStream_t s) {
    if (rows_pad == 0) return hipSuccess;
    const uint32_t nthreads = rows_pad * 8u;
    hipLaunchKernelGGL(prep_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, s, raw, prep,
                       rs128, rows_pad, maxsq_out);
    return hipGetLastError();
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
// Pass 1 writes accept words for rows up to round_up(n1, 256) (the block size): that stays inside the pair's own region of
// the accept mask because a pair's rows are padded to kRowPad - a multiple of it.  Change either and a workgroup would
// zero its neighbour pair's accept bits while that pair's workgroup reads them.
constexpr int kFinalizeThreads = 256;
static_assert(kRowPad % kFinalizeThreads == 0, "finalize_kernel's pass 1 rounds a pair's rows up to its block size");
__global__ __launch_bounds__(kFinalizeThreads) void finalize_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const Top2* __restrict__ rowbuf, const Top2* __restrict__ colbuf,
    uint32_t* accmask, const float* __restrict__ lut, FinalizeParams fp,
    uint32_t* __restrict__ cursor,
    uint32_t capacity, uint32_t* __restrict__ pair_off, uint32_t* __restrict__ pair_cnt,
    uint32_t* __restrict__ matches) {
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t s_base;
    const PairDev p = pairs[blockIdx.x];
    const uint32_t n1 = imgs[p.slot1].rows, n2 = imgs[p.slot2].rows;
    const Top2* rows = rowbuf + p.row_off;
    const Top2* cols = colbuf + p.col_off;
    uint32_t* mask = accmask + (p.row_off >> 5);  // read in pass 1, rewritten by it (the matches), read again in pass 2
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    auto is_match = [&](uint32_t i, uint32_t& j_out) -> bool {
        if (i >= n1 || n2 == 0) return false;
        Top2 t;
        if (p.mode) {  // mfma pair: the accept bits are final after resolve_index
            if (!((mask[i >> 5] >> (i & 31)) & 1u)) return false;
            t = rows[i];
        } else {
            t = rows[i];
            if (!one_way_accepts(t, lut, fp.max_ratio, fp.max_distance)) return false;
        }
        const uint32_t j = t.best_idx;
        j_out = j;
        if (j >= n2) return false;  // unresolved (counted as an internal error upstream)
        if (!fp.cross_check) return true;
        const Top2 c = cols[j];
        if (!one_way_accepts(c, lut, fp.max_ratio, fp.max_distance)) return false;
        // cols[j] exists for every j an accepted row points at: the dot4 path scans all columns,
        // the mfma path scans exactly those (select_candidates_kernel)
        return c.best_idx == i;
    };

    // pass 1: count - and leave the decisions in the pair's accept words (nobody reads them after this kernel; for
    // dot4 pairs the words exist and are unused): pass 2 then walks bits instead of repeating the gathers of the
    // column table (two 16-byte records in two 64-byte lines per accepted row, the bulk of this kernel's traffic)
    uint32_t local = 0;
    for (uint32_t base = 0; base < n1; base += 256) {
        uint32_t j;
        const bool mm = is_match(base + tid, j);
        local += mm ? 1u : 0u;
        const unsigned long long bal = __ballot(mm);
        if ((lane & 31u) == 0) mask[(base + tid) >> 5] = (uint32_t)(bal >> (lane & 32u));
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
    // pass 2: ordered write, by accept word: thread t owns words t, t + 256, ... of the pair (32 rows each, the bits
    // pass 1 left), a block-wide prefix of their popcounts places them, and a thread writes its word's matches in row
    // order.  A 4,096-row image is 128 words: one round and two barriers, where a round per 256 rows took thirty-two.
    const uint32_t nwords = (n1 + 31) / 32;
    for (uint32_t wbase = 0; wbase < nwords; wbase += 256) {
        const uint32_t w = wbase + tid;
        uint32_t bits = w < nwords ? mask[w] : 0u;
        const uint32_t c = __popc(bits);
        uint32_t inc = c;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t o = __shfl_up(inc, m);
            if (lane >= (uint32_t)m) inc += o;
        }
        __syncthreads();  // (wave_cnt is still being read from the round before)
        if (lane == 63) wave_cnt[wid] = inc;
        __syncthreads();
        uint32_t dst = running + inc - c;
        for (uint32_t k = 0; k < wid; ++k) dst += wave_cnt[k];
        while (bits) {
            const uint32_t i = w * 32 + (uint32_t)(__ffs(bits) - 1);
            bits &= bits - 1;
            if (dst < capacity) {
                matches[2ull * dst] = i;
                matches[2ull * dst + 1] = rows[i].best_idx;
            }
            ++dst;
        }
        running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// ---------------------------------------------------------------------------------------
// resolve_index: the mfma kernel reports, per X row, the best VALUE and the 32-row TILE of Y
// holding it.  For the rows that pass COLMAP's acceptance tests (the only ones whose index is
// ever used) recompute the 32 dot products of that tile on the raw u8 descriptors and take the
// lowest row whose value equals the best: COLMAP's strict-'>' scan keeps exactly that one.  The
// same 32 recomputed values complete the row's second-largest value (match_mfma.hip, valu16): the
// scan's second_v is exact except for values inside the winning tile, and it can only grow here,
// so a row rejected by the acceptance tests before this kernel stays rejected.
// side 0: X rows = all rows of image 1, table = rowbuf.  side 1: X rows = the candidate rows of
// image 2 (candbuf), table = colbuf.  One workgroup per pair, a wave per accepted row in turn:
// lane = (Y row of the tile, half of the 128 bytes).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolve_index_kernel(
    int side, const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    Top2* __restrict__ table, uint32_t* __restrict__ accmask, const float* __restrict__ lut,
    FinalizeParams fp, const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ candbuf,
    uint32_t* __restrict__ err_count,
    const uint32_t* __restrict__ order) {
    const uint32_t pi = order ? order[blockIdx.x] : blockIdx.x;  // (in the streamed image's order: neighbours share Y)
    const PairDev p = pairs[pi];
    if (p.mode == 0) return;  // dot4 pairs carry exact indices already
    const ImageDev X = imgs[side == 0 ? p.slot1 : p.slot2];
    const ImageDev Y = imgs[side == 0 ? p.slot2 : p.slot1];
    const uint32_t n = side == 0 ? X.rows : cand_cnt[pi];
    if (n == 0 || Y.rows == 0) return;
    Top2* tab = table + (side == 0 ? p.row_off : p.col_off);
    uint32_t* amask = accmask + (p.row_off >> 5);
    const uint32_t* list = candbuf + p.col_off;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t l31 = lane & 31, half = lane >> 5;
    for (uint32_t base = wid * 64; base < n; base += 256) {
        const uint32_t e = base + lane;
        uint32_t row = 0;
        Top2 t = Top2{0u, 0xFFFFFFFFu, 0u, 0u};
        bool acc = false;
        if (e < n) {
            row = side == 0 ? e : list[e];
            if (side == 0) {  // the scan left one accept bit per row: only those rows are read
                acc = (amask[e >> 5] >> (e & 31)) & 1u;
                if (acc) t = tab[row];
            } else {
                t = tab[row];
                acc = one_way_accepts(t, lut, fp.max_ratio, fp.max_distance);
            }
        }
        uint32_t resolved = 0xFFFFFFFFu;
        uint32_t second = t.second_v;  // so far: the largest value outside the best's scan unit
        unsigned long long mask = __ballot(acc);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const uint32_t xrow = __shfl(row, b);
            const uint32_t tile = __shfl(t.best_idx, b);
            const uint32_t val = __shfl(t.best_v, b);
            const uint32_t jj = tile * 32 + l31;
            uint32_t sum = 0;
            if (jj < Y.rows_pad) {
                const uint4* xp = reinterpret_cast<const uint4*>(X.raw + (size_t)xrow * kDim + half * 64);
                const uint4* yp = reinterpret_cast<const uint4*>(Y.raw + (size_t)jj * kDim + half * 64);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 a = xp[k], c = yp[k];
                    sum = __builtin_amdgcn_udot4(a.x, c.x, sum, false);
                    sum = __builtin_amdgcn_udot4(a.y, c.y, sum, false);
                    sum = __builtin_amdgcn_udot4(a.z, c.z, sum, false);
                    sum = __builtin_amdgcn_udot4(a.w, c.w, sum, false);
                }
            }
            sum += __shfl_xor(sum, 32);
            const bool eq = (sum == val) && (jj < Y.rows);
            const uint32_t m = (uint32_t)(__ballot(eq) & 0xFFFFFFFFull);
            const uint32_t first = m ? (uint32_t)(__ffs(m) - 1) : 0xFFFFFFFFu;
            const uint32_t j = m ? tile * 32 + first : 0xFFFFFFFFu;
            // second-largest of the tile's values, with multiplicity: everything but the one
            // element that is the best (0 = COLMAP's floor for padding rows and the best itself)
            uint32_t sw = (jj < Y.rows && l31 != first) ? sum : 0u;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) sw = max(sw, (uint32_t)__shfl_xor(sw, d));
            if ((int)lane == b) {
                resolved = j;
                second = max(second, sw);
                if (!m) atomicAdd(err_count, 1u);  // scan and recomputation disagree: a bug
            }
        }
        if (acc) {  // rows rejected above: index never used, second only ever grows
            tab[row].best_idx = resolved;
            tab[row].second_v = second;
        }
        if (side == 0) {  // narrow the accept bits to the rows that pass with the exact second
            t.second_v = second;
            const bool keep = acc && one_way_accepts(t, lut, fp.max_ratio, fp.max_distance);
            const unsigned long long kb = __ballot(keep);
            if ((lane & 31) == 0 && e < ((n + 31) & ~31u)) amask[e >> 5] = (uint32_t)(kb >> (lane & 32));
        }
    }
}

// ---------------------------------------------------------------------------------------
// resolve_index, grouped by tile.  The kernel above reads the 4 KB Y tile of every accepted row from L2 - fine
// when a pair has a dozen accepted rows, but on overlapping image pairs a third of the rows are accepted (1,300 of
// 4,096) and the launch becomes L2-bandwidth bound (340 GB per 62 k pairs, 9 TB/s, as long as the reverse scan
// itself).  Here the accepted rows of a pair are bucketed by their best tile first (counting sort in LDS, chunks
// of 4,096 rows), each wave then walks a contiguous range of the sorted list and keeps the current tile's 32 Y
// rows in registers (lane = (Y row, half), 64 bytes each) across all the X rows that point into it: per accepted
// row only its own 128 bytes are fetched.  Same arithmetic per row as above, so the same results; rows are
// independent, their order is free.  Used when the Y image has at most kResolveMaxTiles tiles.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kResolveChunk = 4096, kResolveMaxTiles = 2048;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void resolve_index_grouped_kernel(
    int side, const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    Top2* __restrict__ table, uint32_t* __restrict__ accmask, const float* __restrict__ lut,
    FinalizeParams fp, const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ candbuf,
    uint32_t* __restrict__ err_count,
    const uint32_t* __restrict__ order) {
    // 48 KB, three workgroups per CU.  While sorting: s_list | s_sorted | s_hist; while walking the rows the first
    // third holds the rows' dot products (`part`) and the last third the batch's X rows (over s_hist and beyond).
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[3 * kResolveChunk];
    uint32_t* const s_list = s_pool;                        // (tile << 12) | position in the chunk
    uint32_t* const s_sorted = s_pool + kResolveChunk;
    uint32_t* const s_hist = s_pool + 2 * kResolveChunk;    // kResolveMaxTiles entries
    uint4* const s_xall = reinterpret_cast<uint4*>(s_pool + 2 * kResolveChunk);  // 4 waves x 256 x 16 B
    static_assert(kResolveMaxTiles <= kResolveChunk && 4 * 256 * 4 <= kResolveChunk, "the last third holds either");
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_brow[4][32], s_btile[4][32];  // the batch's rows and tiles, by wave
    const uint32_t pi = order ? order[blockIdx.x] : blockIdx.x;  // (in the streamed image's order: neighbours share Y)
    const PairDev p = pairs[pi];
    if (p.mode == 0) return;  // dot4 pairs carry exact indices already
    const ImageDev X = imgs[side == 0 ? p.slot1 : p.slot2];
    const ImageDev Y = imgs[side == 0 ? p.slot2 : p.slot1];
    const uint32_t n = side == 0 ? X.rows : cand_cnt[pi];
    if (n == 0 || Y.rows == 0) return;
    Top2* tab = table + (side == 0 ? p.row_off : p.col_off);
    uint32_t* amask = accmask + (p.row_off >> 5);
    const uint32_t* list = candbuf + p.col_off;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t l31 = lane & 31, half = lane >> 5;
    const uint32_t ntiles = Y.rows_pad / 32;
    constexpr uint32_t kPer = kResolveMaxTiles / 256;  // histogram bins per thread in the scan
    for (uint32_t chunk0 = 0; chunk0 < n; chunk0 += kResolveChunk) {
        const uint32_t chunk_end = min(n, chunk0 + kResolveChunk);
        if (tid == 0) s_count = 0;
        for (uint32_t k = tid; k < kResolveMaxTiles; k += 256) s_hist[k] = 0;
        __syncthreads();
        // ---- accepted rows of the chunk and the histogram of their tiles (four rows per thread and round: their table
        // reads are independent and in flight together)
        for (uint32_t e0 = chunk0 + tid; e0 < chunk_end; e0 += 4 * 256) {
            bool acc[4];
            uint32_t tile[4];
            if (side == 0) {
                // the accept words first (coalesced, a few lines per pair), then the table only where a bit is set: on a
                // sparse set 1 row in 100 is accepted and the 64 KB of a pair's row table stay where they are
                uint32_t aw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) aw[u] = amask[min(e0 + (uint32_t)u * 256, chunk_end - 1) >> 5];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = e0 + (uint32_t)u * 256;
                    acc[u] = e < chunk_end && ((aw[u] >> (e & 31)) & 1u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) tile[u] = acc[u] ? tab[e0 + (uint32_t)u * 256].best_idx : 0u;
            } else {
                uint32_t le[4];
                Top2 tt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) le[u] = list[min(e0 + (uint32_t)u * 256, chunk_end - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) tt[u] = tab[le[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = e0 + (uint32_t)u * 256 < chunk_end && one_way_accepts(tt[u], lut, fp.max_ratio, fp.max_distance);
                    tile[u] = tt[u].best_idx;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!acc[u]) continue;
                const uint32_t e = e0 + (uint32_t)u * 256;
                if (tile[u] >= ntiles) {  // cannot happen (the scan only reports tiles it visited): counted, row left unresolved
                    atomicAdd(err_count, 1u);
                    continue;
                }
                const uint32_t pos = atomicAdd(&s_count, 1u);
                s_list[pos] = (tile[u] << 12) | (e - chunk0);
                atomicAdd(&s_hist[tile[u]], 1u);
            }
        }
        __syncthreads();
        const uint32_t cnt = s_count;
        if (cnt == 0) { __syncthreads(); continue; }
        // ---- exclusive scan of the histogram (thread t owns bins [t * kPer, (t + 1) * kPer))
        uint32_t h[kPer], c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { h[k] = s_hist[tid * kPer + k]; c += h[k]; }
        uint32_t inc = c;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t o = __shfl_up(inc, m);
            if (lane >= (uint32_t)m) inc += o;
        }
        if (lane == 63) s_wsum[wid] = inc;
        __syncthreads();
        uint32_t start = inc - c;
        for (uint32_t k = 0; k < wid; ++k) start += s_wsum[k];
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { s_hist[tid * kPer + k] = start; start += h[k]; }
        __syncthreads();
        for (uint32_t k = tid; k < cnt; k += 256) {
            const uint32_t v = s_list[k];
            s_sorted[atomicAdd(&s_hist[v >> 12], 1u)] = v;
        }
        __syncthreads();
        // ---- each wave walks a contiguous quarter of the sorted list in batches of 32 rows.  Lane (j, h) keeps Y row j
        // of ITS half's current tile in registers (all 128 bytes); half h handles rows h*16 .. h*16+15 of the batch, one
        // per step, every lane of the half loading the same X row (one request).  A step is 8 loads + 32 dot4 + one
        // LDS store of the lane's dot product; everything that needs the 32 values of a row together - first index
        // equal to the best, largest of the others, the acceptance test with its LUT read, the table update - is done
        // after the batch by the row's owner lane, 32 rows in parallel.  (The first grouped version did that per row:
        // a 32-lane reduction through seven LDS permutes and a lane-0 tail with three dependent memory round trips,
        // ~3,600 cycles per row.)
        uint32_t* part = s_list + wid * 1024;  // [32 rows][32], row r rotated by r: free of bank conflicts both ways
        uint4* xs = s_xall + wid * 256;
        const uint32_t kb = (uint32_t)(((uint64_t)cnt * wid) / 4), ke = (uint32_t)(((uint64_t)cnt * (wid + 1)) / 4);
        uint32_t cur_tile = 0xFFFFFFFFu;
        uint4 yv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) yv[q] = make_uint4(0, 0, 0, 0);
        for (uint32_t b0 = kb; b0 < ke; b0 += 32) {
            const uint32_t nb = min(32u, ke - b0);
            uint32_t my_row = 0, my_e = 0, my_tile = 0;
            Top2 my_t{0u, 0u, 0u, 0u};
            if (lane < nb) {
                const uint32_t v = s_sorted[b0 + lane];
                my_tile = v >> 12;
                my_e = chunk0 + (v & 4095u);
                my_row = side == 0 ? my_e : list[my_e];
                my_t = tab[my_row];
                s_brow[wid][lane] = my_row;
                s_btile[wid][lane] = my_tile;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the batch's X rows -> LDS with one round of independent loads (lane = 16-byte piece; 4 KB per wave)
            {
                uint4 piece[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t idx = (uint32_t)q * 64 + lane, r = idx >> 3;
                    const uint32_t rowq = s_brow[wid][r < nb ? r : 0];
                    piece[q] = *reinterpret_cast<const uint4*>(X.raw + (size_t)rowq * kDim + (idx & 7u) * 16);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[(uint32_t)q * 64 + lane] = piece[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {   // first step's Y tile
                const uint32_t r = half * 16;
                const uint32_t tile = s_btile[wid][r < nb ? r : 0];
                if (r < nb && tile != cur_tile) {
                    cur_tile = tile;
                    const uint4* yp = reinterpret_cast<const uint4*>(Y.raw + (size_t)(tile * 32 + l31) * kDim);
#pragma unroll
                    for (int q = 0; q < 8; ++q) yv[q] = yp[q];  // tile * 32 + l31 < rows_pad: padding rows are zero
                }
            }
            for (uint32_t st = 0; st < 16; ++st) {
                const uint32_t r = half * 16 + st;
                const bool live = r < nb;
                // the next step's Y tile, if it is another one, while this step is multiplied
                const uint32_t rn = r + 1;
                const bool live_n = st + 1 < 16 && rn < nb;
                const uint32_t tile_n = s_btile[wid][live_n ? rn : 0];
                const bool change = live_n && tile_n != cur_tile;
                uint4 yn[8];
                if (change) {
                    const uint4* yp = reinterpret_cast<const uint4*>(Y.raw + (size_t)(tile_n * 32 + l31) * kDim);
#pragma unroll
                    for (int q = 0; q < 8; ++q) yn[q] = yp[q];
                }
                uint32_t sum = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint4 xa = xs[(live ? r : 0u) * 8 + (uint32_t)q];
                    sum = __builtin_amdgcn_udot4(xa.x, yv[q].x, sum, false);
                    sum = __builtin_amdgcn_udot4(xa.y, yv[q].y, sum, false);
                    sum = __builtin_amdgcn_udot4(xa.z, yv[q].z, sum, false);
                    sum = __builtin_amdgcn_udot4(xa.w, yv[q].w, sum, false);
                }
                if (live) part[r * 32 + ((l31 + r) & 31u)] = sum;
                if (change) {
                    cur_tile = tile_n;
#pragma unroll
                    for (int q = 0; q < 8; ++q) yv[q] = yn[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < nb) {
                bool found = false;
                uint32_t first = 0, sw = 0;
#pragma unroll 8
                for (uint32_t jx = 0; jx < 32; ++jx) {
                    const uint32_t sv = part[lane * 32 + ((jx + lane) & 31u)];
                    const bool valid = my_tile * 32 + jx < Y.rows;
                    if (valid && !found && sv == my_t.best_v) {
                        found = true;
                        first = jx;
                    } else if (valid) {
                        sw = max(sw, sv);
                    }
                }
                Top2 w2 = my_t;
                w2.best_idx = found ? my_tile * 32 + first : 0xFFFFFFFFu;
                w2.second_v = max(my_t.second_v, sw);
                tab[my_row].best_idx = w2.best_idx;
                tab[my_row].second_v = w2.second_v;
                if (!found) atomicAdd(err_count, 1u);  // scan and recomputation disagree: a bug
                // side 0: narrow the accept bits to the rows that pass with the exact second
                if (side == 0 && !one_way_accepts(w2, lut, fp.max_ratio, fp.max_distance))
                    atomicAnd(&amask[my_e >> 5], ~(1u << (my_e & 31)));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// resolve_index on the matrix core (round 4).  Same sort by tile as the grouped kernel above; the dot products of a
// batch come from v_mfma_i32_32x32x32_i8 instead of 32 x 32 v_dot4 per accepted row.  Same values (the zero-point
// identity is exact), same decisions, same table updates: test_dense_overlap_and_both_resolve_kernels holds the three
// forms against each other.  Used whenever the grouped form would be (AMC_RESOLVE_DOT4=1 keeps that one, for A/B).
// ---------------------------------------------------------------------------------------
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kResolveFewRows = 96;   // up to so many accepted rows in a chunk are walked unsorted (below)
constexpr uint32_t kResolveChunkM = 3840;  // rows per sort chunk of the mfma form: two lists of it + the histogram stay under 40 KB
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void resolve_index_mfma_kernel(
    int side, const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    Top2* __restrict__ table, uint32_t* __restrict__ accmask, const float* __restrict__ lut,
    FinalizeParams fp, const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ candbuf,
    uint32_t* __restrict__ err_count,
    const uint32_t* __restrict__ order) {
    // 38 KB, four workgroups per CU: the chunk's accepted rows, the same sorted by tile, the tile histogram
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[2 * kResolveChunkM + kResolveMaxTiles];
    uint32_t* const s_list = s_pool;                        // (tile << 12) | position in the chunk
    uint32_t* const s_sorted = s_pool + kResolveChunkM;
    uint32_t* const s_hist = s_pool + 2 * kResolveChunkM;    // kResolveMaxTiles entries
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_brow[4][32];  // the batch's rows, by wave
    const uint32_t pi = order ? order[blockIdx.x] : blockIdx.x;  // (in the streamed image's order: neighbours share Y)
    const PairDev p = pairs[pi];
    if (p.mode == 0) return;  // dot4 pairs carry exact indices already
    const ImageDev X = imgs[side == 0 ? p.slot1 : p.slot2];
    const ImageDev Y = imgs[side == 0 ? p.slot2 : p.slot1];
    const uint32_t n = side == 0 ? X.rows : cand_cnt[pi];
    if (n == 0 || Y.rows == 0) return;
    Top2* tab = table + (side == 0 ? p.row_off : p.col_off);
    uint32_t* amask = accmask + (p.row_off >> 5);
    const uint32_t* list = candbuf + p.col_off;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t l31 = lane & 31, half = lane >> 5;
    const uint32_t ntiles = Y.rows_pad / 32;
    constexpr uint32_t kPer = kResolveMaxTiles / 256;  // histogram bins per thread in the scan
    for (uint32_t chunk0 = 0; chunk0 < n; chunk0 += kResolveChunkM) {
        const uint32_t chunk_end = min(n, chunk0 + kResolveChunkM);
        if (tid == 0) s_count = 0;
        __syncthreads();
        // ---- accepted rows of the chunk and the histogram of their tiles (four rows per thread and round: their table
        // reads are independent and in flight together)
        for (uint32_t e0 = chunk0 + tid; e0 < chunk_end; e0 += 4 * 256) {
            bool acc[4];
            uint32_t tile[4];
            if (side == 0) {
                // the accept words first (coalesced, a few lines per pair), then the table only where a bit is set: on a
                // sparse set 1 row in 100 is accepted and the 64 KB of a pair's row table stay where they are
                uint32_t aw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) aw[u] = amask[min(e0 + (uint32_t)u * 256, chunk_end - 1) >> 5];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = e0 + (uint32_t)u * 256;
                    acc[u] = e < chunk_end && ((aw[u] >> (e & 31)) & 1u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) tile[u] = acc[u] ? tab[e0 + (uint32_t)u * 256].best_idx : 0u;
            } else {
                uint32_t le[4];
                Top2 tt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) le[u] = list[min(e0 + (uint32_t)u * 256, chunk_end - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) tt[u] = tab[le[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = e0 + (uint32_t)u * 256 < chunk_end && one_way_accepts(tt[u], lut, fp.max_ratio, fp.max_distance);
                    tile[u] = tt[u].best_idx;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!acc[u]) continue;
                const uint32_t e = e0 + (uint32_t)u * 256;
                if (tile[u] >= ntiles) {  // cannot happen (the scan only reports tiles it visited): counted, row left unresolved
                    atomicAdd(err_count, 1u);
                    continue;
                }
                const uint32_t pos = atomicAdd(&s_count, 1u);
                s_list[pos] = (tile[u] << 12) | (e - chunk0);
            }
        }
        __syncthreads();
        const uint32_t cnt = s_count;
        if (cnt == 0) { __syncthreads(); continue; }
        // Few accepted rows (a pair without overlap: a handful of chance matches - 97 % of an exhaustive job's pairs):
        // they sit in as many tiles as there are rows, sorting could not merge a single visit, so the waves walk the
        // list as it was collected and the histogram is never touched (zeroing, counting, scanning and sorting it was
        // most of this kernel's time on such pairs).  The order of the rows is free: each is resolved on its own.
        const bool few = cnt <= kResolveFewRows;
        const uint32_t* const walk = few ? s_list : s_sorted;
        if (!few) {
            for (uint32_t k = tid; k < kResolveMaxTiles; k += 256) s_hist[k] = 0;
            __syncthreads();
            for (uint32_t k = tid; k < cnt; k += 256) atomicAdd(&s_hist[s_list[k] >> 12], 1u);
            __syncthreads();
            // ---- exclusive scan of the histogram (thread t owns bins [t * kPer, (t + 1) * kPer))
            uint32_t h[kPer], c = 0;
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) { h[k] = s_hist[tid * kPer + k]; c += h[k]; }
            uint32_t inc = c;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const uint32_t o = __shfl_up(inc, m);
                if (lane >= (uint32_t)m) inc += o;
            }
            if (lane == 63) s_wsum[wid] = inc;
            __syncthreads();
            uint32_t start = inc - c;
            for (uint32_t k = 0; k < wid; ++k) start += s_wsum[k];
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) { s_hist[tid * kPer + k] = start; start += h[k]; }
            __syncthreads();
            for (uint32_t k = tid; k < cnt; k += 256) {
                const uint32_t v = s_list[k];
                s_sorted[atomicAdd(&s_hist[v >> 12], 1u)] = v;
            }
            __syncthreads();
        }
        // ---- each wave walks a contiguous quarter of the sorted list in batches of 32 rows, and the batch's dot products
        // come from the int8 matrix core: B operand = the batch's 32 X rows (prepared arena, fetched once per batch),
        // A operand = one 32-row Y tile; four MFMAs give every (X row, Y row) product of the pair, the zero-point term
        // 128 SY_j rides in as the C operand and the lane's own term is added at decode - the scan's arithmetic
        // (match_mfma.hip), so the values are the exact u8 dot products.  The batch is sorted by tile: the rows of
        // one tile are a run, and each distinct tile of the batch costs one visit (four MFMAs + ~50 VALU for all 32
        // rows at once) instead of 32 x 32 v_dot4 per row.
        // A lane's 16 outputs become keys, value << 5 | (31 - j): their maximum is the best value at its LOWEST row
        // index (COLMAP's strict '>' keeps that one), their second-largest the second value with multiplicity.
        const uint32_t kb = (uint32_t)(((uint64_t)cnt * wid) / 4), ke = (uint32_t)(((uint64_t)cnt * (wid + 1)) / 4);
        int codes[16];  // 31 - (row of the tile that accumulator register r holds on this lane)
#pragma unroll
        for (int r = 0; r < 16; ++r) codes[r] = 31 - ((r & 3) + 8 * (r >> 2) + 4 * (int)half);
        for (uint32_t b0 = kb; b0 < ke; b0 += 32) {
            const uint32_t nb = min(32u, ke - b0);
            uint32_t my_row = 0, my_e = 0, my_tile = 0xFFFFFFFFu;
            Top2 my_t{0u, 0u, 0u, 0u};
            if (lane < nb) {
                const uint32_t v = walk[b0 + lane];
                my_tile = v >> 12;
                my_e = chunk0 + (v & 4095u);
                my_row = side == 0 ? my_e : list[my_e];
                my_t = tab[my_row];
                s_brow[wid][lane] = my_row;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // B operand: X row l31 of the batch (rows past nb repeat row 0; their results are never read)
            const uint32_t xrow = s_brow[wid][l31 < nb ? l31 : 0];
            i32x4 xf[4];
            {
                const char* rp = reinterpret_cast<const char*>(X.prep) + (size_t)xrow * kDim;
                const uint32_t sw = (xrow >> 1) & 7u;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) xf[sl] = *reinterpret_cast<const i32x4*>(rp + (((2 * sl + half) ^ sw) * 16));
            }
            // dot = accumulator + xterm, xterm = 128 * (sum of the row's bytes) - 2^21.  The sum comes from the fragments
            // just loaded (prep = raw ^ 0x80; this lane holds 64 of the row's 128 bytes) instead of a 4-byte gather from
            // rs128 - a 64-byte line per accepted row, a sixth of the kernel's traffic on overlapping pairs.
            uint32_t rsum = 0;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rsum = __builtin_amdgcn_udot4((uint32_t)xf[sl][e] ^ 0x80808080u, 0x01010101u, rsum, false);
            rsum += (uint32_t)__shfl_xor((int)rsum, 32);
            const int xterm = (int)(rsum * 128u) - (1 << 21);
            bool found = false;
            uint32_t first = 0, sw_val = 0;
            unsigned long long todo = __ballot(lane < nb);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)my_tile, src);  // the next tile of the batch
                todo &= ~__ballot(lane < nb && my_tile == T);
                // A operand + C block of tile T (rows T*32 .. +31 < rows_pad: padding rows are zero descriptors)
                const uint32_t yrow = T * 32 + l31;
                const char* yp = reinterpret_cast<const char*>(Y.prep) + (size_t)yrow * kDim;
                const uint32_t ysw = (yrow >> 1) & 7u;
                i32x4 yf[4];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) yf[sl] = *reinterpret_cast<const i32x4*>(yp + (((2 * sl + half) ^ ysw) * 16));
                i32x16 acc;
                const int* rsb = Y.rs128 + T * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const i32x4 v = *reinterpret_cast<const i32x4*>(rsb + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * g + e] = v[e];
                }
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(yf[sl], xf[sl], acc, 0, 0, 0);
                // top two keys of the lane's 16 outputs, then of the two lane halves (they hold different Y rows)
                int k1 = (acc[0] << 5) | codes[0], k2 = INT_MIN;
#pragma unroll
                for (int r = 1; r < 16; ++r) {
                    const int k = (acc[r] << 5) | codes[r];
                    k2 = max(k2, min(k1, k));
                    k1 = max(k1, k);
                }
                const int o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32);
                const int K1 = max(k1, o1);
                const int K2 = max(min(k1, o1), max(k2, o2));
                if (my_tile == T) {  // (lanes 0 .. nb-1 carry the batch's rows: l31 == lane there)
                    const uint32_t val = (uint32_t)((K1 >> 5) + xterm);
                    const uint32_t j = 31u - (uint32_t)(K1 & 31);
                    found = val == my_t.best_v && T * 32 + j < Y.rows;
                    first = j;
                    // the second of the tile: every other entry (a padding row counts 0, COLMAP's floor)
                    const int s2 = (K2 >> 5) + xterm;
                    sw_val = (uint32_t)max(s2, 0);
                }
            }
            if (lane < nb) {
                Top2 w2 = my_t;
                w2.best_idx = found ? my_tile * 32 + first : 0xFFFFFFFFu;
                w2.second_v = max(my_t.second_v, sw_val);
                tab[my_row].best_idx = w2.best_idx;
                tab[my_row].second_v = w2.second_v;
                if (!found) atomicAdd(err_count, 1u);  // scan and recomputation disagree: a bug
                // side 0: narrow the accept bits to the rows that pass with the exact second
                if (side == 0 && !one_way_accepts(w2, lut, fp.max_ratio, fp.max_distance))
                    atomicAnd(&amask[my_e >> 5], ~(1u << (my_e & 31)));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
    }
}

hipError_t launch_resolve_index(int side, const ImageDev* imgs, const PairDev* pairs, uint32_t npairs,
                          Top2* table, uint32_t* accmask, const float* acos_lut, FinalizeParams fp,
                          const uint32_t* cand_cnt, const uint32_t* candbuf, uint32_t* err_count,
                          bool grouped, const uint32_t* order, uint32_t norder, hipStream_t s) {
    // `order` (norder mfma pairs, sorted by the image this side STREAMS): a workgroup per listed pair, so that
    // workgroups running at the same time read the same Y tiles - on the dense set a pair's workgroup reads all of
    // image Y (512 KB of tiles) and a third of image X; in plain pair order (image 1 major) neighbours shared X only
    // and the launch moved ~0.7 MB per pair from HBM.  Without it: one workgroup per pair of the batch.
    const uint32_t grid = order ? norder : npairs;
    if (grid == 0) return hipSuccess;
    const bool dot4_form = std::getenv("AMC_RESOLVE_DOT4") != nullptr;  // (test hook / A/B: the v_dot4 form)
    if (grouped && !dot4_form)
        hipLaunchKernelGGL(resolve_index_mfma_kernel, dim3(grid), dim3(256), 0, s, side, imgs, pairs,
                           table, accmask, acos_lut, fp, cand_cnt, candbuf, err_count, order);
    else if (grouped)
        hipLaunchKernelGGL(resolve_index_grouped_kernel, dim3(grid), dim3(256), 0, s, side, imgs, pairs,
                           table, accmask, acos_lut, fp, cand_cnt, candbuf, err_count, order);
    else
        hipLaunchKernelGGL(resolve_index_kernel, dim3(grid), dim3(256), 0, s, side, imgs, pairs,
                           table, accmask, acos_lut, fp, cand_cnt, candbuf, err_count, order);
    return hipGetLastError();
}
uint32_t resolve_grouped_max_rows() { return kResolveMaxTiles * 32; }

// ---------------------------------------------------------------------------------------
// select_candidates: which rows of image 2 ("columns") does the cross check need?  Exactly
// those some accepted row of image 1 points at.  One workgroup per pair: rows that pass
// COLMAP's one-way tests set bit best_idx in an LDS bitmap; the bitmap is then compacted,
// ascending, into candbuf[col_off ...] and counted.  The bitmap is dynamic shared memory sized
// by the launch's largest image 2 (kSelectMaxCols = 1 Mi columns = 128 KiB bounds it): thread t
// owns `per` consecutive words, so ascending column order is thread order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_candidates_kernel(
    const ImageDev* __restrict__ imgs, const PairDev* __restrict__ pairs,
    const Top2* __restrict__ rowbuf, const uint32_t* __restrict__ accmask,
    const float* __restrict__ lut, FinalizeParams fp, uint32_t* __restrict__ cand_cnt,
    uint32_t* __restrict__ candbuf) {
    extern __shared__ uint32_t bits[];  // 256 * per words (the launch's bound; this pair uses what its n2 needs)
    __shared__ uint32_t wsum[4];
    const PairDev p = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (p.mode == 0) {  // dot4 pair: its column table is complete already
        if (tid == 0) cand_cnt[blockIdx.x] = 0;
        return;
    }
    const uint32_t n1 = imgs[p.slot1].rows, n2 = imgs[p.slot2].rows;
    const uint32_t per = ((n2 + 31) / 32 + 255) / 256;  // words per thread for THIS pair
    const Top2* rows = rowbuf + p.row_off;
    for (uint32_t k = tid; k < per * 256; k += 256) bits[k] = 0;
    __syncthreads();
    if (n2 != 0) {
        const uint32_t* mask = accmask + (p.row_off >> 5);  // final accept bits (resolve_index)
        for (uint32_t i = tid; i < n1; i += 256) {
            if (!((mask[i >> 5] >> (i & 31)) & 1u)) continue;
            const uint32_t j = rows[i].best_idx;
            if (j < n2) atomicOr(&bits[j >> 5], 1u << (j & 31));
        }
    }
    __syncthreads();
    uint32_t c = 0;
    for (uint32_t k = 0; k < per; ++k) c += __popc(bits[tid * per + k]);
    // inclusive scan over the 256 threads
    uint32_t inc = c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t o = __shfl_up(inc, m);
        if (lane >= (uint32_t)m) inc += o;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wid; ++k) base += wsum[k];
    uint32_t* dst = candbuf + p.col_off + base + inc - c;
    for (uint32_t k = 0; k < per; ++k) {
        uint32_t ww = bits[tid * per + k];
        while (ww) {
            const uint32_t b = __ffs(ww) - 1;
            *dst++ = (tid * per + k) * 32 + b;
            ww &= ww - 1;
        }
    }
    if (tid == 255) cand_cnt[blockIdx.x] = base + inc;
}

hipError_t launch_select_candidates(const ImageDev* imgs, const PairDev* pairs, uint32_t npairs, uint32_t max_cols,
                              const Top2* rowbuf, const uint32_t* accmask, const float* acos_lut,
                              FinalizeParams fp, uint32_t* cand_cnt, uint32_t* candbuf, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    const uint32_t per = ((max_cols + 31) / 32 + 255) / 256;
    const size_t shmem = (size_t)std::max(per, 1u) * 256 * sizeof(uint32_t);
    if (shmem > 48 * 1024) {  // above the default dynamic-LDS limit (images beyond 393,216 descriptors): opt in, per device
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_candidates_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kSelectMaxCols / 8));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(select_candidates_kernel, dim3(npairs), dim3(256), shmem, s, imgs, pairs,
                       rowbuf, accmask, acos_lut, fp, cand_cnt, candbuf);
    return hipGetLastError();
}

// Pair p's matches sit at src + 2 * src_off[p] (finalize_kernel's atomic-cursor order); move them to
// dst + 2 * dst_off[p]: the batch in pair order, i.e. the CSR layout of the result, so that one D2H copy lands them
// where the caller reads them (no per-pair scatter on the host) and the verification kernel can index them by pair.
// One wave per pair, 8-byte elements, consecutive lanes on consecutive matches.
__global__ __launch_bounds__(256) void reorder_matches_kernel(const uint32_t* __restrict__ src_off,
                                                              const uint32_t* __restrict__ cnt,
                                                              const uint64_t* __restrict__ dst_off, uint32_t npairs,
                                                              const uint2* __restrict__ src, uint2* __restrict__ dst) {
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;
    const uint32_t n = cnt[p];
    const uint2* s = src + src_off[p];
    uint2* d = dst + dst_off[p];
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) d[i] = s[i];
}
hipError_t launch_reorder_matches(const uint32_t* src_off, const uint32_t* cnt, const uint64_t* dst_off, uint32_t npairs,
                                  const uint32_t* src, uint32_t* dst, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    hipLaunchKernelGGL(reorder_matches_kernel, dim3((npairs + 3) / 4), dim3(256), 0, s, src_off, cnt, dst_off, npairs,
                       reinterpret_cast<const uint2*>(src), reinterpret_cast<uint2*>(dst));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// A batch's matches to the host, beside the NEXT batch's kernels.  hipMemcpyAsync(DeviceToHost) of a large buffer
// is executed by a runtime copy kernel whose grid covers the whole buffer: launched in the gap between two
// batches it takes every CU, moves 400 MB at PCIe speed for 10 ms, and the persistent scan behind it starts when
// it is done (kernel trace of the dense set: 19 ms of 292 exposed).  This kernel is the same copy with a grid of a
// few dozen workgroups and a dozen registers: it fits beside the scan's waves on CUs the scan fills (234 of 256
// registers x 2 waves per SIMD leave room for a small wave) and PCIe, not the CUs, bounds it.  dst: pinned host
// memory (hipHostMalloc: the same pointer on the device); bytes a multiple of 8.
// ---------------------------------------------------------------------------------------
constexpr int kHostCopyBlocks = 64;
__global__ __launch_bounds__(256) void host_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16,
                                                        uint2* __restrict__ dst_tail, const uint2* __restrict__ src_tail,
                                                        uint32_t tail8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane, then four posted writes
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail8) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
hipError_t launch_host_copy(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    // both pointers are 8-byte aligned (matches are uint32 pairs); the bulk runs from the first 16-byte boundary
    const uintptr_t a = reinterpret_cast<uintptr_t>(src_dev);
    const size_t head = (a & 15) && ((reinterpret_cast<uintptr_t>(dst_pinned) & 15) == (a & 15)) ? 16 - (a & 15) : 0;
    const bool same_phase = (reinterpret_cast<uintptr_t>(dst_pinned) & 15) == (a & 15);
    if (!same_phase || bytes < ((size_t)1 << 20)) {  // small, or the two sides are not 16-byte congruent: the runtime's copy
        return memcpy_async(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, s);
    }
    const char* sp = static_cast<const char*>(src_dev);
    char* dp = static_cast<char*>(dst_pinned);
    if (head) {
        const hipError_t e = memcpy_async(dp, sp, head, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
    }
    const size_t body = bytes - head, n16 = body / 16, tail = body - n16 * 16;
    hipLaunchKernelGGL(host_copy_kernel, dim3(kHostCopyBlocks), dim3(256), 0, s, reinterpret_cast<uint4*>(dp + head),
                       reinterpret_cast<const uint4*>(sp + head), n16, reinterpret_cast<uint2*>(dp + head + n16 * 16),
                       reinterpret_cast<const uint2*>(sp + head + n16 * 16), (uint32_t)(tail / 8));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The batch tables' small copies as a kernel (either side may be pinned host memory; both 4-byte aligned, bytes a
// multiple of 4).  The runtime's asynchronous copies of ALL streams go through one in-order DMA queue: a copy that waits
// for a kernel of its own stream holds up every other stream's copies behind it - with a batch's chain beside the
// next batch's scan (match_impl) that serialised the two (profiles/r06/overlap_timeline_v2.txt).  A kernel only
// waits for its own stream.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t nwords,
                                                         int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const size_t n16 = nwords / 4;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t j = i; j < n16; j += stride) d4[j] = s4[j];
        for (size_t j = n16 * 4 + i; j < nwords; j += stride) dst[j] = src[j];
    } else {
        for (; i < nwords; i += stride) dst[i] = src[i];
    }
}
hipError_t launch_copy_words(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(dst) & 3) || (reinterpret_cast<uintptr_t>(src) & 3)) return hipErrorInvalidValue;
    const size_t nwords = bytes / 4;
    const int vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    const size_t units = vec ? (nwords + 3) / 4 : nwords;
    const unsigned blocks = (unsigned)std::min<size_t>(32, (units + 255) / 256);
    hipLaunchKernelGGL(copy_words_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(dst),
                       static_cast<const uint32_t*>(src), nwords, vec);
    return hipGetLastError();
}

hipError_t launch_finalize(const ImageDev* imgs, const PairDev* pairs, uint32_t npairs,
                           const Top2* rowbuf, const Top2* colbuf, uint32_t* accmask,
                           const float* acos_lut, FinalizeParams fp, uint32_t* cursor, uint32_t capacity,
                           uint32_t* pair_off, uint32_t* pair_cnt, uint32_t* matches, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_kernel, dim3(npairs), dim3(kFinalizeThreads), 0, s, imgs, pairs, rowbuf,
                       colbuf, accmask, acos_lut, fp, cursor, capacity, pair_off, pair_cnt, matches);
    return hipGetLastError();
}

}  // namespace amc
