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
// Why this shape (measured on MI355X, tools/ubench_ops.hip, tools/ubench_mix.hip, tools/ubench_ladder.hip):
//   * every 32-bit min/max/med3/max3/shift/shift-add is HALF rate on gfx950 (4 clk / wave64);
//   * on one SIMD an int8 MFMA (32 clk) hides only ~6 VALU instructions; each further one costs
//     ~4.5 clk.  So the epilogue budget is 6 VALU per MFMA = 1.5 per output.
// A values-only top-2 insertion (v_med3_i32 + v_max3_i32 + v_max_i32 per TWO candidates) is
// exactly 1.5/output, i.e. VALU-bound.  The scan therefore does less: it reduces each lane's 16
// outputs of a unit to their maximum (8 x v_max3/v_max) and keeps the top two of those maxima
// plus the tile of the best, 12 VALU per 16 outputs, which leaves the matrix pipe as
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
// Work items (round 4).  The unit of work is no longer a pair.  Every pair's X side is cut into
// SEGMENTS of 128 rows; the segments of all pairs that stream the same Y image are packed, eight to
// an ITEM, by three small kernels (seg_count / seg_scan / seg_fill below) into 64-byte descriptors
// that carry every pointer a wave needs.  A workgroup pops an item, streams that Y image once
// through LDS, and its waves scan their own segments - which may belong to DIFFERENT pairs (they
// only share the Y stream).  So a 4,500-row image costs 36 segment-times, not 5 x 8; 512-row pairs
// fill a workgroup two at a time; and the reverse scan's candidate lists (known only on the
// device: the packing kernels read cand_cnt) are packed just as tightly.
//
// Shape.  One workgroup per item (dynamic queue; items in Y order so co-resident workgroups stream
// the same image out of L2; the next item's descriptor is fetched while the current one is
// scanned).  <W, XT> = waves per workgroup x resident 32-row X tiles per wave, W * XT = 32:
//   <8, 4>  512 threads, a segment per wave, 2 waves per SIMD (256 registers each)
//   <4, 8>  256 threads, two segments per wave, 1 wave per SIMD (512 registers): half the LDS
//           fragment traffic per MFMA, the wave's own VALU fills its own MFMA shadows
// Y streams through LDS in 256-row chunks by direct-to-LDS DMA, three buffers, the pieces of chunk c+2
// issued one per Y tile of chunk c, one barrier per chunk; the prepared arena is pre-swizzled so the linear DMA image is bank-conflict-free for
// ds_read_b128.  Each wave software-pipelines: the 4 MFMAs of unit u+1 are interleaved with the
// 12 VALU of unit u (unit = 32 Y rows x 32 X rows), two accumulator sets.
#include <climits>
#include <cstdlib>

#include "amc_internal.h"
#include "scan_accept.h"

namespace amc {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kBN = 256;                // Y rows per LDS chunk (= kRowPad); 512 measured 2 % slower
constexpr int kYT = kBN / 32;           // Y tiles per chunk
constexpr int kChunkBytes = kBN * kDim; // 32 KiB
constexpr int kSegTiles = kSegRows / 32;  // 32-row X tiles per segment
#ifndef AMC_KEY_INSERT
#define AMC_KEY_INSERT 1   // 0: the four-instruction insertion with an explicit tile register (A/B)
#endif
constexpr int kKeyShift = 7, kKeyCarried = 127;

// single LDS object (a second __shared__ object de-pipelines the DMA waits).  THREE chunk buffers: while chunk c is
// scanned, chunk c+1 has landed (or is landing) and chunk c+2 is being fetched into the buffer chunk c-1 left -
// so its DMA pieces need not wait for a barrier and are issued one per Y tile instead of all at the chunk
// boundary, where they stalled the wave's MFMA stream (2.7 % of the scan at two waves per SIMD, 5 % at one).
constexpr int kNB = 3;
constexpr int kOffRs = 0;                        // kNB x rs128 chunks (kBN ints each)
constexpr int kOffQ = kOffRs + kNB * kBN * 4;    // queue slot
constexpr int kOffB = kOffQ + 16;                // kNB x descriptor chunks
constexpr int kLdsBytes = kOffB + kNB * kChunkBytes;
static_assert(kBN == kRowPad, "a chunk is the row padding unit");
static_assert(kRowPad % kSegRows == 0 && kSegRows % 32 == 0, "segments tile the padded rows");
static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");
static_assert(sizeof(SegDesc) == 64, "a descriptor is 16 dwords: one per lane of a quarter wave");

// dword positions of the SegDesc fields (a wave holds its descriptors in ONE register: lane l has dword l)
enum : int { kDXprep = 0, kDXrs = 2, kDOut = 4, kDList = 6, kDYprep = 8, kDYrs = 10, kDCnt = 12, kDAccword = 13,
             kDYrows = 14 };

// v_max3_i32 pinned by hand (hipcc pattern-matches it only some of the time).
// These read MFMA results directly and hipcc does not pad hazards for inline asm, so the kernel
// is structured so that an accumulator is only ever read one full phase (>= 2 MFMA issues)
// after the MFMAs that produced it were issued; sched_barriers pin that order.
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

// First MFMA of a unit: C operand = the Y tile's 128*SY block, destination = a free accumulator.  Written as
// inline asm with an early-clobber output: hipcc selects the TIED form for this one (destination = C operand)
// and then copies the 16 C registers into the accumulator first - 8 v_mov_b64 and their wait states per Y tile,
// on the matrix pipe's critical path.  Operands come from LDS reads (the waitcnt pass tracks asm operands) and
// the result is next read by the dependent MFMA behind it (same destination: no wait states needed).
// BA: the B operand (the resident X fragment) lives in the accumulation half of the register file (AGPRs) - the
// one-wave-per-SIMD shape keeps its 128 fragment registers there, out of the way of everything the VALU touches
// (hipcc on its own shuttles accumulators through v_accvgpr_read/write, 32 extra VALU per unit).
template <bool BA>
__device__ __forceinline__ void mfma_first(i32x16& d, const i32x4& a, const i32x4& b, const i32x16& c) {
    if constexpr (BA)
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    else
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
// The other three MFMAs of a unit accumulate in place.
template <bool BA>
__device__ __forceinline__ void mfma_acc(i32x16& d, const i32x4& a, const i32x4& b) {
    if constexpr (BA)
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    else
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d, 0, 0, 0);
}

struct YFrag {
    i32x4 f[4];  // 32 Y rows x four 32-deep k-slices (MFMA A operand)
    i32x16 ci;   // 128*SY_j for this lane's 16 Y rows (MFMA C operand)
};

// The previous batch's matches on their way to the host (CopyJob, amc_internal.h).  The runtime's own copy kernel
// covers the buffer with its grid and takes every CU while PCIe moves 400 MB (10 ms, and the scan behind it waits:
// kernel trace of the dense set, 19 ms of 292 per call); a small copy kernel on a second stream shares a hardware queue
// with the scan's stream more often than not.  So the copy rides in the scan's own launch: the first `parts`
// workgroups to arrive take one part each - 512 lanes with four 16-byte loads in flight per lane saturate PCIe from
// a handful of workgroups - and then scan like the others; the scan loses parts x 10 ms of one workgroup's time.
__device__ __noinline__ void copy_part(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned long long b,
                                       unsigned long long e, int tid, int nthreads) {
    unsigned long long i = b + (unsigned long long)tid;
    const unsigned long long st = (unsigned long long)nthreads;
    for (; i + 3 * st < e; i += 4 * st) {
        const uint4 v0 = src[i], v1 = src[i + st], v2 = src[i + 2 * st], v3 = src[i + 3 * st];
        dst[i] = v0; dst[i + st] = v1; dst[i + 2 * st] = v2; dst[i + 3 * st] = v3;
    }
    for (; i < e; i += st) dst[i] = src[i];
}

template <int MODE, int W, int XT>
__global__ __launch_bounds__(64 * W) void match_mfma_kernel(const SegDesc* __restrict__ segs,
                                                            const uint32_t* __restrict__ nitems_p,
                                                            uint32_t* __restrict__ queue_head,
                                                            uint32_t* __restrict__ accmask,
                                                            const ScanAccept* __restrict__ accept,
                                                            CopyJob job, uint32_t* __restrict__ copy_head) {
    constexpr int SPW = XT / kSegTiles;  // segments per wave
    constexpr bool BA = (W == 4);        // X fragments in AGPRs (one wave per SIMD: 512 registers)
    static_assert(XT % kSegTiles == 0 && W * SPW == kSegsPerItem, "a workgroup takes one item");
    static_assert((kChunkBytes / 1024) % W == 0 && kYT % (kChunkBytes / 1024 / W) == 0, "a chunk splits into 1 KiB DMA pieces per wave, dealt evenly over its Y tiles");
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    volatile uint32_t* s_q = reinterpret_cast<volatile uint32_t*>(smem + kOffQ);
    if constexpr (MODE == 0) {
        if (job.parts) {  // (wave-uniform: a kernel argument)
            __shared__ uint32_t s_part;
            if (tid == 0) s_part = atomicAdd(copy_head, 1u);
            __syncthreads();
            const uint32_t part = s_part;
            if (part < job.parts)
                copy_part(static_cast<const uint4*>(job.src), static_cast<uint4*>(job.dst), job.n16 * part / job.parts,
                          job.n16 * (part + 1) / job.parts, tid, 64 * W);
        }
    }
    const uint32_t nitems = *nitems_p;
    const ScanAccept sa = *accept;  // twelve scalar registers for the whole kernel (re-read per row block it is four dependent scalar-cache round trips per X tile)

    // a wave's SPW descriptors of item q, one dword per lane (lanes 16*SPW.. hold 0)
    auto load_desc = [&](uint32_t q) -> int {
        int v = 0;
        if (q < nitems && lane < 16 * SPW)
            v = reinterpret_cast<const int*>(segs + (size_t)q * kSegsPerItem + wid * SPW)[lane];
        return v;
    };
    auto dword = [&](int dv, int h, int f) -> int { return __builtin_amdgcn_readlane(dv, 16 * h + f); };
    auto dptr = [&](int dv, int h, int f) -> const char* {
        const uint64_t lo = (uint32_t)__builtin_amdgcn_readlane(dv, 16 * h + f);
        const uint64_t hi = (uint32_t)__builtin_amdgcn_readlane(dv, 16 * h + f + 1);
        return reinterpret_cast<const char*>(lo | (hi << 32));
    };

    // chunk c of the Y image (prep / rs128 base pointers) -> LDS buffer `buf`, in 1 KiB pieces dealt to the waves
    // (dest = wave-uniform base + lane*16); a wave owns kPW pieces of a chunk, wave 0 also its rs128 block
    constexpr int kPW = kChunkBytes / 1024 / W;
    auto stage_piece = [&](const char* yprep, int c, int buf, int ps) __attribute__((always_inline)) {
        const int piece = ps * W + wid;
        __builtin_amdgcn_global_load_lds((gvoid_t*)(yprep + (size_t)c * kChunkBytes + piece * 1024 + lane * 16),
                                         (lvoid_t*)(smem + kOffB + buf * kChunkBytes + piece * 1024), 16, 0, 0);
    };
    auto stage_rs = [&](const char* yrs, int c, int buf) __attribute__((always_inline)) {
        if (wid == 0)  // rs128 of this chunk's rows: 1 KiB
            __builtin_amdgcn_global_load_lds((gvoid_t*)(yrs + (size_t)c * kBN * 4 + lane * 16),
                                             (lvoid_t*)(smem + kOffRs + buf * kBN * 4), 16, 0, 0);
    };
    auto stage = [&](const char* yprep, const char* yrs, int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < kPW; ++ps) stage_piece(yprep, c, buf, ps);
        stage_rs(yrs, c, buf);
    };

    // A-operand fragments + C-init block of Y tile `yt` of the chunk in LDS buffer `buf`, as eight 16-byte reads:
    // parts 0..3 = the C block (needed first), 4..7 = the four k-slices.  The scan spreads them over the phases
    // of the previous tile, one or two behind each phase's last MFMA, so they never pile up in one MFMA shadow.
    auto load_y_part = [&](YFrag& y, int buf, int yt, int part) __attribute__((always_inline)) {
        if (part < 4) {
            // accumulator register r <-> Y row (r&3) + 8*(r>>2) + 4*lh of the tile
            const int* rsb = reinterpret_cast<const int*>(smem + kOffRs + buf * kBN * 4) + yt * 32 + 4 * lh;
            const i32x4 v = *reinterpret_cast<const i32x4*>(rsb + 8 * part);
#pragma unroll
            for (int e = 0; e < 4; ++e) y.ci[4 * part + e] = v[e];  // the -2^21 lives in xterm
        } else {
            const int s = part - 4;
            const int row = yt * 32 + l31;  // within chunk
            const int sw = (row >> 1) & 7;
            const char* cp = smem + kOffB + buf * kChunkBytes + row * kDim;
            y.f[s] = *reinterpret_cast<const i32x4*>(cp + (((2 * s + lh) ^ sw) * 16));
        }
    };
    auto load_y = [&](YFrag& y, int buf, int yt) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 8; ++part) load_y_part(y, buf, yt, part);
    };

    // resident X state: B-operand fragments, one (best, second, tile) per lane and X tile
    i32x4 xf[XT][4];
    int best[XT], sec[XT], btile[XT], xterm[XT];
    auto load_x = [&](int dv) __attribute__((always_inline)) {
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
            const int h = xt / kSegTiles;
            const int kl = (xt % kSegTiles) * 32 + l31;  // row within the segment
            const int cnt = dword(dv, h, kDCnt);
            const char* xprep = dptr(dv, h, kDXprep);
            const int* xrs = reinterpret_cast<const int*>(dptr(dv, h, kDXrs));
            int row;
            if (MODE == 0) {
                row = kl;  // xprep / xrs point at the segment's first row; rows past the image's end are zero padding
            } else {
                // candidate rows; a segment's tail repeats its last row (results never stored), an empty one reads row 0
                const uint32_t* list = reinterpret_cast<const uint32_t*>(dptr(dv, h, kDList));
                row = cnt > 0 ? (int)list[min(kl, cnt - 1)] : 0;
            }
            const char* rp = xprep + (size_t)row * kDim;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const char* fp = rp + (((2 * s + lh) ^ sw) * 16);
                if constexpr (BA)
                    // straight into the accumulation registers; hipcc does not count this load, the
                    // s_waitcnt vmcnt(0) in front of the scan's first barrier covers it
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(xf[xt][s]) : "v"(fp) : "memory");
                else
                    xf[xt][s] = *reinterpret_cast<const i32x4*>(fp);
            }
            // acc = sum a'b' + 128*SY_j = v - (128*SX_i - 2^21)
            xterm[xt] = xrs[row] - (1 << 21);
            // COLMAP's floor best = second = 0  <=>  acc = -xterm
#if AMC_KEY_INSERT
            best[xt] = ((-xterm[xt]) << kKeyShift) | kKeyCarried;  // a key: value << 7 | tile code (see `insert`)
            sec[xt] = best[xt];
#else
            best[xt] = -xterm[xt];
            sec[xt] = -xterm[xt];
#endif
            btile[xt] = -1;
        }
    };
    auto any_rows = [&](int dv) -> bool {
        bool a = false;
#pragma unroll
        for (int h = 0; h < SPW; ++h) a = a || dword(dv, h, kDCnt) > 0;
        return a;
    };

    if (tid == 0) *s_q = atomicAdd(queue_head, 1u);
    __syncthreads();
    uint32_t q = *s_q;
    int dv = load_desc(q);
    if (q < nitems) {
        stage(dptr(dv, 0, kDYprep), dptr(dv, 0, kDYrs), 0, 0);
        if (dword(dv, 0, kDYrows) > kBN) stage(dptr(dv, 0, kDYprep), dptr(dv, 0, kDYrs), 1, 1);
        if (any_rows(dv)) load_x(dv);
    }
    __syncthreads();  // everyone has read the slot before it is rewritten

    while (q < nitems) {
        // chunks 0 and 1 of this item's Y image are on their way and the X fragments are being loaded (issued by
        // the previous iteration or the prologue)
        const char* yprep = dptr(dv, 0, kDYprep);  // the same in every descriptor of the item
        const char* yrs = dptr(dv, 0, kDYrs);
        const int yrows_item = dword(dv, 0, kDYrows);
        const int nchunks = (yrows_item + kBN - 1) / kBN;
        // Tiles of the LAST chunk that hold rows, rounded up to a pair of steps: what follows them in the chunk is the
        // image's zero padding (rows_pad is a multiple of 256) - a zero row can never become a best nor raise a second,
        // so its tiles need not be scanned (round 6: an image of 4,000 rows paid for 4,096; n ~ U[2000, 6000]: 2.4 % of
        // the scan).  An image whose rows fill its last chunk scans all kYT tiles as before.
#ifndef AMC_SKIP_PAD_TILES
#define AMC_SKIP_PAD_TILES 1
#endif
        const int last_tiles = AMC_SKIP_PAD_TILES ? ((((yrows_item - (nchunks - 1) * kBN) + 31) / 32 + 1) & ~1) : kYT;
        const bool active = any_rows(dv);  // wave-uniform
        if (tid == 0) *s_q = atomicAdd(queue_head, 1u);  // the next item, behind the loads already in flight

        i32x16 acc[2];
        auto mfma4 = [&](i32x16& a, const YFrag& y, int xt) __attribute__((always_inline)) {
            mfma_first<BA>(a, y.f[0], xf[xt][0], y.ci);
#pragma unroll
            for (int s = 1; s < 4; ++s) mfma_acc<BA>(a, y.f[s], xf[xt][s]);
        };
        // One phase = the 4 MFMAs of the NEXT unit (into `an`) with the 12 VALU of the CURRENT
        // unit (reading `ac`, completed by the previous phase) spread between them, 4/4/4/-.
        // Interleaved, the matrix pipe stays fed (15.6 ns per MFMA in tools/ubench_mix.hip);
        // "4 MFMAs, then the VALU" leaves it idle whenever every wave of the SIMD is in its VALU stretch.
        //
        // 12 VALU for 16 outputs: the scan only keeps, per lane, the top two of the per-unit
        // MAXIMA (8 v_max3/v_max for the maximum of the unit's 16 outputs, then one insertion)
        // and the tile of the best.  The second-largest VALUE of the whole row is either the
        // maximum of another unit - which the running `sec` then holds, ties included - or
        // sits inside the winning tile, where resolve_index_kernel recomputes the 32 dot
        // products anyway to find the index: it takes the second of those 32 as well and the
        // row's second is the larger of the two.
        // Insertion of a unit maximum into a lane's (best, second, tile) state, in place (no copies for the
        // register allocator to make).
#if AMC_KEY_INSERT
        // Three instructions: the state holds KEYS, value << 7 | code, code = 126 - (tile mod 64).  The accumulators stay
        // below 2^24 in magnitude (|sum a'b'| <= 2^21, 128 SY < 2^22), so a key fits 32 bits; a larger value is a
        // larger key, equal values are ordered first tile first (strict '>' of the reference scan), and the second
        // largest key carries the second largest value, equal ones included.  Every 64 tiles (and at the end of the
        // item) `flush` moves the tile of a best found since the last flush to btile and marks the key "carried"
        // (code 127: it beats equal values of later tiles).  The tile code is wave-uniform: a scalar operand.
        auto insert = [&](int xt, int m, int code) __attribute__((always_inline)) {
            asm volatile(
                "v_lshl_or_b32 %2, %2, 7, %3\n\t"
                "v_med3_i32 %1, %0, %1, %2\n\t"  // sec <= best always: the new second of the maxima
                "v_max_i32 %0, %0, %2"
                : "+v"(best[xt]), "+v"(sec[xt]), "+v"(m)
                : "s"(code));
        };
        auto flush = [&](int sb) __attribute__((always_inline)) {  // sb: the 64-tile block that ends here
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
                const int c = best[xt] & 127;
                btile[xt] = c != kKeyCarried ? sb * 64 + (126 - c) : btile[xt];
                best[xt] |= kKeyCarried;
            }
        };
#else
        // The compare comes first: on gfx950 a VALU read of VCC needs two
        // instructions between it and the VALU write.  Strict '>': the first tile wins ties.
        auto insert = [&](int xt, int m, int tile) __attribute__((always_inline)) {
            asm volatile(
                "v_cmp_gt_i32 vcc, %3, %0\n\t"
                "v_med3_i32 %1, %0, %1, %3\n\t"  // sec <= best always: the new second of the maxima
                "v_max_i32 %0, %0, %3\n\t"
                "v_cndmask_b32 %2, %2, %4, vcc"
                : "+v"(best[xt]), "+v"(sec[xt]), "+v"(btile[xt])
                : "v"(m), "v"(tile)
                : "vcc");
        };
#endif
        // The insertion of a unit's maximum does not touch accumulators, so it is deferred into
        // the hazard slot of the NEXT phase (between its first two MFMAs): pm / ptile carry the
        // pending maximum (of X tile xtc - 1) from one phase to the next.
#if AMC_KEY_INSERT
        int pm = -(1 << 24), ptile = 0;  // a pending "maximum" below every accumulator: its key is below every floor key
#else
        int pm = INT_MIN, ptile = 0;
#endif
        auto phase = [&](i32x16& an, const YFrag& y, int xtn, const i32x16& ac, int xtc, int tile)
                         __attribute__((always_inline)) {
            mfma_first<BA>(an, y.f[0], xf[xtn][0], y.ci);
            __builtin_amdgcn_sched_barrier(0);
#if defined(AMC_DIAG) && (AMC_DIAG & 2)
            mfma_acc<BA>(an, y.f[1], xf[xtn][1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_acc<BA>(an, y.f[2], xf[xtn][2]);
            __builtin_amdgcn_sched_barrier(0);
            pm = smax2(pm, ac[0]);  // timing diagnostic: one VALU per unit keeps the accumulators live
            ptile = tile;
            (void)insert;
#else
#if AMC_KEY_INSERT
            insert((xtc + XT - 1) % XT, pm, 126 - (ptile & 63));
            // the last unit of a 64-tile block has just gone in: settle the tiles before the next block reuses the codes
            if (xtc == 0 && tile != 0 && (tile & 63) == 0) flush((tile >> 6) - 1);
#else
            insert((xtc + XT - 1) % XT, pm, ptile);
#endif
            __builtin_amdgcn_sched_barrier(0);
            mfma_acc<BA>(an, y.f[1], xf[xtn][1]);
            __builtin_amdgcn_sched_barrier(0);
            // first read of `ac`: two MFMAs have issued since the one that completed it.  One asm block per
            // MFMA shadow (hipcc pads dependent asm statements with s_nops it cannot know to be needless)
            int t0, t1, t2, t3;
            asm volatile(
                "v_max3_i32 %0, %4, %5, %6\n\t"
                "v_max3_i32 %1, %7, %8, %9\n\t"
                "v_max3_i32 %2, %10, %11, %12\n\t"
                "v_max3_i32 %3, %13, %14, %15"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "v"(ac[0]), "v"(ac[1]), "v"(ac[2]), "v"(ac[3]), "v"(ac[4]), "v"(ac[5]), "v"(ac[6]), "v"(ac[7]),
                  "v"(ac[8]), "v"(ac[9]), "v"(ac[10]), "v"(ac[11]));
            __builtin_amdgcn_sched_barrier(0);
            mfma_acc<BA>(an, y.f[2], xf[xtn][2]);
            __builtin_amdgcn_sched_barrier(0);
            int m4;
            asm volatile(
                "v_max3_i32 %1, %4, %5, %6\n\t"  // m4 = max of outputs 12..14
                "v_max3_i32 %0, %0, %2, %3\n\t"  // t0 = max(t0, t1, t2)
                "v_max3_i32 %1, %8, %1, %7\n\t"  // m4 = max(t3, m4, output 15)
                "v_max_i32 %0, %0, %1"
                : "+v"(t0), "=&v"(m4)
                : "v"(t1), "v"(t2), "v"(ac[12]), "v"(ac[13]), "v"(ac[14]), "v"(ac[15]), "v"(t3));
            pm = t0;
            ptile = tile;
#endif
            __builtin_amdgcn_sched_barrier(0);
            mfma_acc<BA>(an, y.f[3], xf[xtn][3]);
            __builtin_amdgcn_sched_barrier(0);
        };

        // ---- software-pipelined scan over all of Y -----------------------------------
        // units in order (ytile, xt = 0..XT-1); acc[0] holds even xt, acc[1] odd xt.  Each phase
        // issues the MFMAs of the NEXT unit, then runs the VALU of the current one.
        YFrag y0, y1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk 0 and this item's X rows landed
        __syncthreads();
        const uint32_t qn = *s_q;
        const int dvn = load_desc(qn);  // in flight during the scan
        if (active) {
            load_y(y0, 0, 0);
            mfma4(acc[0], y0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The sched_barriers pin the interleaving.  An accumulator is read (by inline asm, which
        // hipcc does not hazard-check) only after TWO further MFMAs have issued behind the one
        // that completed it - the matrix pipe runs MFMAs in order, 32 clk each, so the value
        // has been written back for well over the 11 wait states an 8-pass MFMA needs.
        // one step = one Y tile held in `yc`; prefetches the next tile into `yn`
        // cb / nb: LDS buffers of chunk c and c + 1 (c % 3, (c + 1) % 3)
        auto step = [&](YFrag& yc, YFrag& yn, int c, int cb, int nb, int yt, bool even) __attribute__((always_inline)) {
            const bool lastt = (yt == kYT - 1);
            const bool cross = lastt && (c + 1 < nchunks);
            const bool fetch = c + 2 < nchunks;  // chunk c + 2 goes to the buffer chunk c - 1 left: (c + 2) % 3
#if !(defined(AMC_DIAG) && (AMC_DIAG & 8))
            if (fetch) {
                const int fb = cb == 0 ? 2 : cb - 1;
                constexpr int kEvery = kYT / kPW;  // a piece every tile (one wave per SIMD) or every other tile
                if (kEvery == 1) stage_piece(yprep, c + 2, fb, yt);
                else if (even) stage_piece(yprep, c + 2, fb, yt / 2);
                if (yt == 0) stage_rs(yrs, c + 2, fb);
            }
#endif
            if (cross) {
#if !(defined(AMC_DIAG) && (AMC_DIAG & 4))
                // chunk c + 1 must have landed: everything but this chunk's own kPW (+ 1) pieces of chunk c + 2
                if (!fetch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (wid == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPW + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPW) : "memory");
                // a bare barrier: __syncthreads() is a fence and would drain the pieces just issued (vmcnt(0)).
                // Nothing is stored to LDS here by a wave itself; the DMA'd bytes are ordered by the waits above.
                asm volatile("s_barrier" ::: "memory");
#endif
            }
            if (active) {
                // very last tile: re-read itself (result unused) to stay branch-free
                const int nbuf = cross ? nb : cb;
                const int nyt = lastt ? (cross ? 0 : yt) : yt + 1;
                const int tile = c * kYT + yt;
                constexpr int RPP = 16 / XT;  // reads behind each phase of the step's first half
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    if (xt + 1 < XT)
                        phase(acc[(xt + 1) & 1], yc, xt + 1, acc[xt & 1], xt, tile);
                    else
                        phase(acc[0], yn, 0, acc[xt & 1], xt, tile);
                    if (xt < XT / 2) {
#pragma unroll
                        for (int r = 0; r < RPP; ++r) load_y_part(yn, nbuf, nyt, xt * RPP + r);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        for (int c = 0, cb = 0; c < nchunks; ++c) {
            const int nb = cb == 2 ? 0 : cb + 1;
            const int yend = (c == nchunks - 1) ? last_tiles : kYT;
#pragma unroll 1
            for (int yt = 0; yt < yend; yt += 2) {
                step(y0, y1, c, cb, nb, yt, true);
                step(y1, y0, c, cb, nb, yt + 1, false);
            }
            cb = nb;
        }
#if AMC_KEY_INSERT
        if (active) {
            insert(XT - 1, pm, 126 - (ptile & 63));  // the last unit's maximum is still pending
            flush(ptile >> 6);
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {  // keys -> values
                best[xt] >>= kKeyShift;
                sec[xt] >>= kKeyShift;
            }
        }
#else
        if (active) insert(XT - 1, pm, ptile);  // the last unit's maximum is still pending
#endif
        __syncthreads();  // everyone is done with both LDS chunk buffers (and has read the queue slot)

        // ---- item done.  Start the next one's loads, then decode and store this one under them ----
        int eb[XT], es[XT], et[XT], ex[XT];
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
            // (asm: hipcc otherwise sinks these copies into the scan loop, 3 x XT moves per Y tile)
            asm volatile("v_mov_b32 %0, %1" : "=v"(eb[xt]) : "v"(best[xt]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(es[xt]) : "v"(sec[xt]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(et[xt]) : "v"(btile[xt]));
            ex[xt] = xterm[xt];
        }
        if (qn < nitems) {
            stage(dptr(dvn, 0, kDYprep), dptr(dvn, 0, kDYrs), 0, 0);
            if (dword(dvn, 0, kDYrows) > kBN) stage(dptr(dvn, 0, kDYprep), dptr(dvn, 0, kDYrs), 1, 1);
            if (any_rows(dvn)) load_x(dvn);
        }
        if (active) {
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
                const int h = xt / kSegTiles;
                const int kl = (xt % kSegTiles) * 32 + l31;
                const int cnt = dword(dv, h, kDCnt);
                // merge the two lane halves (they saw different Y rows of every tile)
                const int ob = __shfl_xor(eb[xt], 32);
                const int os = __shfl_xor(es[xt], 32);
                const int ot = __shfl_xor(et[xt], 32);
                int b = eb[xt], s = es[xt], t = et[xt];
                const bool ow = (ob > b) || (ob == b && (unsigned)ot < (unsigned)t);
                s = max(max(s, os), ow ? b : ob);
                t = ow ? ot : t;
                b = ow ? ob : b;
                bool acc_bit = false;
                if (lh == 0 && kl < cnt) {
                    Top2* out = reinterpret_cast<Top2*>(const_cast<char*>(dptr(dv, h, kDOut)));
                    int row = kl;  // MODE 0: `out` points at the segment's first row
                    if (MODE == 1) row = (int)reinterpret_cast<const uint32_t*>(dptr(dv, h, kDList))[kl];
                    Top2 o;
#if defined(AMC_DIAG) && (AMC_DIAG & 1)
                    // timing diagnostic: every row "has no match" (the scan state only feeds the pad word)
                    o.best_v = 0; o.best_idx = 0xFFFFFFFFu; o.second_v = 0; o.pad = (uint32_t)(b ^ s ^ t) & 0u;
                    out[row] = o;
#else
                    o.best_v = (uint32_t)(b + ex[xt]);
                    o.best_idx = o.best_v ? (uint32_t)t : 0xFFFFFFFFu;  // TILE of the best
                    o.second_v = (uint32_t)(s + ex[xt]);
                    o.pad = 0;
                    out[row] = o;
                    // a larger second only ever rejects: rows failing now can be forgotten
                    // (scan_accept.h: thresholds instead of acos; a superset of what the exact tests keep)
                    if (MODE == 0) acc_bit = scan_may_accept(sa, o.best_v, o.second_v);
#endif
                }
                if (MODE == 0 && cnt > 0) {  // accept bits of the 32 rows of this X tile (lanes 0..31)
                    const uint32_t bits = (uint32_t)__ballot(acc_bit);
                    if (lane == 0) accmask[(uint32_t)dword(dv, h, kDAccword) + (xt % kSegTiles)] = bits;
                }
            }
        }
        q = qn;
        dv = dvn;
    }
}

// ---------------------------------------------------------------------------------------------------
// Packing the segments of a launch into items (three tiny kernels, all on the match stream).
// `order` lists the launch's pairs sorted by the streamed image; `grp_start` (host-built, ngroups + 1
// entries) cuts it where that image changes.  A pair contributes ceil(rows / 128) segments - rows = image 1's
// rows (mode 0) or the pair's candidate count (mode 1, known only here) - none if the streamed image is empty.
// Items never mix streamed images: every group is padded to a multiple of 8 segments with null descriptors
// (cnt = 0, but a valid Y stream, so any wave can read it from its own descriptor).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pair_segments(int mode, const ImageDev* imgs, const PairDev& p, uint32_t pi,
                                                  const uint32_t* cand_cnt, uint32_t* nrows_out) {
    const uint32_t nx = mode == 0 ? imgs[p.slot1].rows : cand_cnt[pi];
    const uint32_t ny = imgs[mode == 0 ? p.slot2 : p.slot1].rows;
    *nrows_out = nx;
    return ny ? (nx + kSegRows - 1) / kSegRows : 0u;
}

// one 256-thread block per group: seg_base[i] = segments of the group before pair order[i]; group totals
__global__ __launch_bounds__(256) void seg_count_kernel(int mode, const ImageDev* __restrict__ imgs,
                                                        const PairDev* __restrict__ pairs,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ grp_start,
                                                        const uint32_t* __restrict__ cand_cnt,
                                                        uint32_t* __restrict__ seg_base, uint32_t* __restrict__ grp_segs) {
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t i0 = grp_start[g], i1 = grp_start[g + 1];
    uint32_t running = 0;
    for (uint32_t base = i0; base < i1; base += 256) {
        const uint32_t i = base + tid;
        uint32_t s = 0, nr;
        if (i < i1) {
            const uint32_t pi = order[i];
            s = pair_segments(mode, imgs, pairs[pi], pi, cand_cnt, &nr);
        }
        uint32_t inc = s;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t o = __shfl_up(inc, m);
            if (lane >= (uint32_t)m) inc += o;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < wid) wbase += wsum[k];
            total += wsum[k];
        }
        if (i < i1) seg_base[i] = running + wbase + inc - s;
        running += total;
        __syncthreads();
    }
    if (tid == 0) grp_segs[g] = running;
}

// one block: grp_item_base = exclusive scan of ceil(grp_segs / 8); the launch's item count
__global__ __launch_bounds__(1024) void seg_scan_kernel(const uint32_t* __restrict__ grp_segs, uint32_t ngroups,
                                                        uint32_t* __restrict__ grp_item_base,
                                                        uint32_t* __restrict__ nitems_out) {
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t running = 0;
    for (uint32_t base = 0; base < ngroups; base += 1024) {
        const uint32_t g = base + tid;
        const uint32_t s = g < ngroups ? (grp_segs[g] + kSegsPerItem - 1) / kSegsPerItem : 0u;
        uint32_t inc = s;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t o = __shfl_up(inc, m);
            if (lane >= (uint32_t)m) inc += o;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (uint32_t k = 0; k < 16; ++k) {
            if (k < wid) wbase += wsum[k];
            total += wsum[k];
        }
        if (g < ngroups) grp_item_base[g] = running + wbase + inc - s;
        running += total;
        __syncthreads();
    }
    if (tid == 0) *nitems_out = running;
}

// kFillY 256-thread blocks per group: a wave per pair in turn writes the pair's descriptors (lane = segment);
// the group's first block pads its last item.  (One block per group left the 500 groups of an exhaustive launch
// to 500 blocks: 2 ms per launch on the dense set.)
constexpr uint32_t kFillY = 8;
__global__ __launch_bounds__(256) void seg_fill_kernel(int mode, const ImageDev* __restrict__ imgs,
                                                       const PairDev* __restrict__ pairs,
                                                       const uint32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ grp_start,
                                                       const uint32_t* __restrict__ cand_cnt,
                                                       const uint32_t* __restrict__ candbuf,
                                                       const uint32_t* __restrict__ seg_base,
                                                       const uint32_t* __restrict__ grp_segs,
                                                       const uint32_t* __restrict__ grp_item_base,
                                                       Top2* __restrict__ outbuf, SegDesc* __restrict__ segs) {
    const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = blockIdx.y * 4 + (tid >> 6);
    const uint32_t i0 = grp_start[g], i1 = grp_start[g + 1];
    if (i0 >= i1) return;
    const uint32_t total = grp_segs[g];
    if (total == 0) return;
    SegDesc* gs = segs + (size_t)grp_item_base[g] * kSegsPerItem;
    // the group's streamed image (the same for all of its pairs)
    const PairDev p0 = pairs[order[i0]];
    const ImageDev Y = imgs[mode == 0 ? p0.slot2 : p0.slot1];
    for (uint32_t i = i0 + wid; i < i1; i += 4 * kFillY) {
        const uint32_t pi = order[i];
        const PairDev p = pairs[pi];
        uint32_t nx;
        const uint32_t ns = pair_segments(mode, imgs, p, pi, cand_cnt, &nx);
        const ImageDev X = imgs[mode == 0 ? p.slot1 : p.slot2];
        const uint32_t sb = seg_base[i];
        for (uint32_t k = lane; k < ns; k += 64) {
            const uint32_t row0 = k * kSegRows;
            SegDesc d;
            d.yprep = Y.prep;
            d.yrs = Y.rs128;
            d.yrows = Y.rows;
            d.cnt = min(nx - row0, (uint32_t)kSegRows);
            d.pad_ = 0;
            if (mode == 0) {
                d.xprep = X.prep + (size_t)row0 * kDim;
                d.xrs = X.rs128 + row0;
                d.out = outbuf + p.row_off + row0;
                d.list = nullptr;
                d.accword = (uint32_t)((p.row_off + row0) >> 5);
            } else {
                d.xprep = X.prep;
                d.xrs = X.rs128;
                d.out = outbuf + p.col_off;  // rows scattered by their index
                d.list = candbuf + p.col_off + row0;
                d.accword = 0;
            }
            gs[sb + k] = d;
        }
    }
    const uint32_t padded = (total + kSegsPerItem - 1) / kSegsPerItem * kSegsPerItem;
    if (blockIdx.y == 0 && tid < padded - total) {
        SegDesc d;
        d.xprep = Y.prep;  // any mapped rows: a null segment's results are never stored
        d.xrs = Y.rs128;
        d.out = nullptr;
        d.list = nullptr;
        d.yprep = Y.prep;
        d.yrs = Y.rs128;
        d.cnt = 0;
        d.accword = 0;
        d.yrows = Y.rows;
        d.pad_ = 0;
        gs[total + tid] = d;
    }
}

hipError_t launch_build_segments(int mode, const ImageDev* imgs, const PairDev* pairs, const uint32_t* order,
                                 const uint32_t* grp_start, uint32_t ngroups, const uint32_t* cand_cnt,
                                 const uint32_t* candbuf, Top2* outbuf, uint32_t* seg_base, uint32_t* grp_segs,
                                 uint32_t* grp_item_base, SegDesc* segs, uint32_t* nitems_dev, hipStream_t s) {
    if (ngroups == 0) return memset_async(nitems_dev, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(seg_count_kernel, dim3(ngroups), dim3(256), 0, s, mode, imgs, pairs, order, grp_start,
                       cand_cnt, seg_base, grp_segs);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, s, grp_segs, ngroups, grp_item_base, nitems_dev);
    hipLaunchKernelGGL(seg_fill_kernel, dim3(ngroups, kFillY), dim3(256), 0, s, mode, imgs, pairs, order, grp_start,
                       cand_cnt, candbuf, seg_base, grp_segs, grp_item_base, outbuf, segs);
    return hipGetLastError();
}

int match_mfma_shape() {
    static const int shape = [] {
        const char* e = std::getenv("AMC_MFMA_SHAPE");  // A/B switch: "8x4" or "4x8"
        if (e && e[0] == '4') return 4;
        if (e && e[0] == '8') return 8;
        return AMC_MFMA_DEFAULT_WAVES;
    }();
    return shape;
}

hipError_t launch_match_mfma(int mode, const SegDesc* segs, const uint32_t* nitems_dev, uint32_t max_items,
                             uint32_t* queue_head, uint32_t* accmask, const ScanAccept* accept_dev, hipStream_t s,
                             const CopyJob& job_in, uint32_t* copy_head, int leave_cus) {
    if (max_items == 0) return hipSuccess;  // (the caller checks: a job is only handed to a launch that happens)
    CopyJob job = (mode == 0 && copy_head) ? job_in : CopyJob();
    // The persistent workgroups pop from these two counters: nothing is launched unless both were reset.
    hipError_t e = job.parts ? memset_async(copy_head, 0, sizeof(uint32_t), s) : hipSuccess;
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    // 1 WG per CU.  AMC_SCAN_GRID (A/B hook, read once): fewer workgroups - the CUs left over stay free for whatever
    // else is in flight (round 6's question: does a power-bound scan lose less than its share of CUs?  DESIGN.md section 6)
    static const int grid_cap = [] {
        const char* e = std::getenv("AMC_SCAN_GRID");
        return e ? std::atoi(e) : 0;
    }();
    if (grid_cap > 0 && grid_cap < cus) cus = grid_cap;
    if (leave_cus > 0) cus = std::max(cus - leave_cus, cus / 2);
    const uint32_t grid = max_items < (uint32_t)cus ? max_items : (uint32_t)cus;
    if (job.parts > grid) job.parts = grid;  // every part needs a workgroup
    if ((e = memset_async(queue_head, 0, sizeof(uint32_t), s)) != hipSuccess) return e;
    const bool w4 = match_mfma_shape() == 4;
#define AMC_LAUNCH(M, W, XT)                                                                             \
    hipLaunchKernelGGL((match_mfma_kernel<M, W, XT>), dim3(grid), dim3(64 * W), 0, s, segs, nitems_dev, \
                       queue_head, accmask, accept_dev, job, copy_head)
    if (mode == 0) {
        if (w4) AMC_LAUNCH(0, 4, 8); else AMC_LAUNCH(0, 8, 4);
    } else {
        if (w4) AMC_LAUNCH(1, 4, 8); else AMC_LAUNCH(1, 8, 4);
    }
#undef AMC_LAUNCH
    return hipGetLastError();
}

}  // namespace amc
