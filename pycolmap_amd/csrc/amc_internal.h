// amc_internal.h — shared between the host-side C-ABI implementation and the HIP kernels.
// gfx950-only; no CPU fallback anywhere in this directory.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amc.h"
#include "guided_region.h"

namespace amc {

// ----- arena geometry -------------------------------------------------------------------
// Every image's descriptor block is zero-padded to a multiple of kRowPad rows.  A zero
// descriptor has dot product 0 with everything, and COLMAP's scan only admits values > 0
// (best=0, second=0 floors, strict '>'), so padded rows/columns can never
// become a best or raise a second: no bounds checks inside the kernels.
constexpr int kDim = AMC_DESC_DIM;   // 128 bytes per descriptor
constexpr int kRowPad = 256;         // = MFMA kernel's column chunk; multiple of every tile
constexpr int kAcosLutSize = 262145; // d in [0, 512*512]

// Device-side view of one image slot.
struct ImageDev {
    const uint8_t* raw;    // rows_pad x 128 u8, row-major, zero padded (dot4 kernel)
    const uint8_t* prep;   // rows_pad x 128, bytes ^0x80 (= u8-128 as i8), 16-B slots of row r
                           // stored at slot (q ^ ((r>>1)&7)) (LDS-bank swizzle), zero rows = 0x80
    const int32_t* rs128;  // rows_pad: 128 * sum_k raw[r][k]
    const float* kp;       // kp_rows x 2 float32 keypoints (x, y), or nullptr (guided matching only)
    uint32_t rows;
    uint32_t rows_pad;
    uint32_t kp_rows;
    uint32_t pad_;
};

// Guided matching (SiftCPUFeatureMatcher::MatchGuided): the pair's float32 filter model.
enum : int { kGuidedNone = 0, kGuidedF = 1, kGuidedH = 2 };
struct GuidedDev {
    int32_t kind;         // kGuidedF: Sampson error under F; kGuidedH: forward transfer error under H
    float max_residual;   // (float)(max_error * max_error)
    float m[9];           // F or H cast to float, row-major
    int32_t grid_ok;      // the candidate-generation kernel may run this pair (match_guided.hip), else the dense one
    // candidate generation only (doubles of the float model above):
    double bound[2];      // F: [0] max |F^T x2|_12^2 over image 2's keypoint box, [1] max |F x1|_12^2 over image 1's
    double minv[9];       // H: inverse of the float model, row-major
};

// Guided matching by candidate generation: an image's float32 keypoints bucketed on a kGridDim x kGridDim grid
// over their bounding box (built once per upload, amc_upload_keypoints).  Cell (gx, gy) has id gy * kGridDim + gx;
// cell_start is the CSR of the keypoints sorted by cell id, so the keypoints of cells gx0..gx1 of one grid row
// are one contiguous range of sxy / sidx.
using guided::kGridDim;
struct GridDev {
    const float* sxy;            // n x 2: keypoints in cell order
    const uint32_t* sidx;        // n: their original indices
    const uint32_t* cell_start;  // kGridDim^2 + 1
    float x0, y0, inv_cw, inv_ch;    // cell of (x, y) = clamp(floor((x - x0) * inv_cw)), clamp(floor((y - y0) * inv_ch))
    float cw, ch;
    float bx1, by1;              // bounding box is [x0, bx1] x [y0, by1]
    uint32_t n;                  // keypoints on the grid (0: no grid - non-finite coordinates or no keypoints)
    uint32_t pad_;
};
using guided::grid_cell;
#if defined(__HIPCC__)
// Guided matching's float32 filter (SiftCPUFeatureMatcher::MatchGuided; oracle_guided_filter in
// oracle/match_oracle.c spells out the operation order): true = this (image-1 point, image-2 point)
// pairing is rejected and its distance is forced to 0.
__device__ __forceinline__ bool guided_rejects(const GuidedDev& g, float x1, float y1, float x2, float y2) {
    const float* m = g.m;
    if (g.kind == kGuidedF) {
        const float Fx1_0 = m[0] * x1 + m[1] * y1 + m[2] * 1.0f;
        const float Fx1_1 = m[3] * x1 + m[4] * y1 + m[5] * 1.0f;
        const float Fx1_2 = m[6] * x1 + m[7] * y1 + m[8] * 1.0f;
        const float Ftx2_0 = m[0] * x2 + m[3] * y2 + m[6] * 1.0f;
        const float Ftx2_1 = m[1] * x2 + m[4] * y2 + m[7] * 1.0f;
        const float x2tFx1 = x2 * Fx1_0 + y2 * Fx1_1 + 1.0f * Fx1_2;
        return x2tFx1 * x2tFx1 / (Fx1_0 * Fx1_0 + Fx1_1 * Fx1_1 + Ftx2_0 * Ftx2_0 + Ftx2_1 * Ftx2_1) > g.max_residual;
    }
    const float Hp_0 = m[0] * x1 + m[1] * y1 + m[2] * 1.0f;
    const float Hp_1 = m[3] * x1 + m[4] * y1 + m[5] * 1.0f;
    const float Hp_2 = m[6] * x1 + m[7] * y1 + m[8] * 1.0f;
    const float e0 = Hp_0 / Hp_2 - x2;
    const float e1 = Hp_1 / Hp_2 - y2;
    return e0 * e0 + e1 * e1 > g.max_residual;
}
#endif

// One-way top-2 record, the common intermediate of both match kernels (16 B).
//   best_v   : best dot product (true value), 0 if none > 0
//   best_idx : its index (lowest index among ties); 0xFFFFFFFF if none.  Straight out of the
//              mfma kernel it is the 32-row TILE holding the best; resolve_index_kernel
//              replaces it by the exact index for the rows that pass the acceptance tests.
//   second_v : second-largest value with multiplicity, floor 0
struct Top2 {
    uint32_t best_v;
    uint32_t best_idx;
    uint32_t second_v;
    uint32_t pad;
};

// A pair inside one batch.
struct PairDev {
    uint32_t slot1, slot2;
    uint32_t mode;     // 1: mfma kernel (column table computed lazily for candidates), 0: dot4
    uint32_t pad;
    uint64_t row_off;  // into the batch's Top2 row buffer (rows_pad(slot1) entries)
    uint64_t col_off;  // into the batch's Top2 col buffer (rows_pad(slot2) entries)
};

// dot4 work item: one 64-row block of one direction of one pair.
struct Dot4Work {
    uint32_t pair;   // index into PairDev[]
    uint32_t dir;    // 0: rows = image1 vs image2 ; 1: rows = image2 vs image1
    uint32_t rb;     // 64-row block index
};

struct FinalizeParams {
    float max_ratio;
    float max_distance;
    int cross_check;
    int reserved;
};

// ----- for the translation units beside amc_api.hip that implement C-ABI entry points (amc_comm.hip) --------------
// api_fail: sets the calling thread's amc_last_error() message and returns `code`.
int api_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
struct CtxView {
    int device;
    hipStream_t stream;        // the stream the ctx launches on (its own, or the host's: amc_ctx_set_stream)
    const uint32_t* resident;  // the last match call's table in device memory (amc_ctx_resident_matches), 2 uint32 per row
    uint64_t resident_rows;
};
CtxView ctx_view(amc_ctx* c);
// What the last amc_verify_pairs / amc_match_verify_pairs call left in device memory, in the caller's pair order (the
// exchange step's verification half reads it there): the packed amc_tvg records, the masks at the input's CSR offsets
// `moff`, the pairs' TvgPair records (match_off / M index `matches`).  npairs == 0: nothing resident (no such call yet,
// or a later match / verification call / trim has reused the buffers).
struct TvgPair;
struct VerifyResident {
    size_t npairs = 0;
    uint64_t total = 0;
    const amc_tvg* tvg = nullptr;
    const uint8_t* mask = nullptr;
    const uint64_t* moff = nullptr;
    const TvgPair* tp = nullptr;
    const uint32_t* matches = nullptr;
};
VerifyResident verify_resident(amc_ctx* c);

// ----- launchers (defined in the .hip files) ---------------------------------------------
// Every launcher returns the status of what it enqueued: a failed memset of a queue head or counter in front of a
// persistent kernel must fail the call (AMC_E_HIP), not let the kernel pop from a stale counter.
//
// memset_async / memcpy_async: hipMemsetAsync / hipMemcpyAsync with a fault-injection hook in front (tests only):
// with AMC_FAIL_NEXT_MEMSET=k (or AMC_FAIL_NEXT_MEMCPY=k) in the environment the k-th such call made while the
// variable holds that value returns hipErrorInvalidValue without enqueuing anything - once; the count restarts
// when the variable is unset or changed.  Every memset / small copy whose target a kernel of the match or
// verification path reads goes through them.
hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s);
hipError_t memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);

hipError_t launch_prep(const uint8_t* raw, uint8_t* prep, int32_t* rs128, uint32_t rows_pad,
                 uint32_t* maxsq_out, hipStream_t s);

// guided: nullptr, or one GuidedDev per PairDev of the batch (entries the filter rejects score 0)
hipError_t launch_match_dot4(const ImageDev* imgs, const PairDev* pairs, const Dot4Work* work,
                       uint32_t nwork, Top2* rowbuf, Top2* colbuf, const GuidedDev* guided, hipStream_t s);
// guided matching by candidate generation (match_guided.hip): same work items and Top2 output as the dot4 kernel
hipError_t launch_match_guided_grid(const ImageDev* imgs, const GridDev* grids, const PairDev* pairs, const Dot4Work* work,
                              uint32_t nwork, Top2* rowbuf, Top2* colbuf, const GuidedDev* guided, hipStream_t s);

// mfma work items (match_mfma.hip).  A pair's X side - image 1's rows (mode 0: -> rowbuf + row_off) or the
// candidate rows of image 2 (mode 1: candbuf + col_off, cand_cnt[pair] of them -> colbuf + col_off, scattered by
// row index) - is cut into segments of kSegRows rows; the segments of all pairs that stream the same image are
// packed kSegsPerItem to an item.  One 64-byte descriptor per segment holds every pointer a wave needs, so a
// workgroup that pops an item is one round trip from its data whatever pairs its segments belong to.
constexpr int kSegRows = 128;
constexpr int kSegsPerItem = 8;
struct alignas(64) SegDesc {
    const uint8_t* xprep;   // mode 0: the segment's first prepared row; mode 1: the image's row 0
    const int32_t* xrs;     // rs128, likewise
    Top2* out;              // mode 0: the segment's first row of the row table; mode 1: the pair's column table
    const uint32_t* list;   // mode 1: the segment's first entry of the pair's candidate list
    const uint8_t* yprep;   // the streamed image (the same in all descriptors of an item)
    const int32_t* yrs;
    uint32_t cnt;           // rows of this segment, 1..kSegRows; 0: padding descriptor, nothing is stored
    uint32_t accword;       // mode 0: accmask word of the segment's first row
    uint32_t yrows;
    uint32_t pad_;
};
#ifndef AMC_MFMA_DEFAULT_WAVES
#define AMC_MFMA_DEFAULT_WAVES 8   // waves per workgroup of the mfma scan: 8 (x 4 X tiles) or 4 (x 8); AMC_MFMA_SHAPE overrides
#endif
// COLMAP's per-row acceptance tests (FindBestMatchesOneWayBruteForce) on a (best,
// second) pair.  acos thresholds: lut[d] = acosf(min(d/512^2, 1)) built on the HOST with the host
// libm; (float)d * 2^-18 is exact for d < 2^24, so indexing by min(d, 262144) reproduces COLMAP's
// float expression bit-for-bit without depending on the device's acosf.
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

// Packs the segments of the pairs listed in `order` (sorted by the streamed image; grp_start cuts it where that
// image changes) into items: segs[0 .. 8 * *nitems_dev).  seg_base (one per order entry), grp_segs and
// grp_item_base (one per group) are scratch.  All on stream s; mode 1 reads cand_cnt on the device.
hipError_t launch_build_segments(int mode, const ImageDev* imgs, const PairDev* pairs, const uint32_t* order,
                           const uint32_t* grp_start, uint32_t ngroups, const uint32_t* cand_cnt,
                           const uint32_t* candbuf, Top2* outbuf, uint32_t* seg_base, uint32_t* grp_segs,
                           uint32_t* grp_item_base, SegDesc* segs, uint32_t* nitems_dev, hipStream_t s);
// The scan over the packed items (persistent workgroups, dynamic queue).  max_items bounds the grid only.
// accmask: one bit per row of the mfma pairs' row tables (bit r of word (row_off + r) / 32): set by the scan (mode 0)
// for the rows that may pass the acceptance tests (scan_accept.h: a superset, with the preliminary second value),
// narrowed by resolve_index (side 0) to the rows that pass with the exact one.  The later kernels walk the bits
// instead of re-reading and re-testing every row.
// accept_dev: the scan's accept-bit thresholds for the call's options (scan_accept.h), in device memory.
struct ScanAccept;
// A device -> pinned-host copy that rides in a forward scan launch: the first `parts` workgroups to start copy one part
// each (16-byte units, PCIe-bound) and then join the scan's queue like the others (match_mfma.hip, "the previous
// batch's matches").  parts = 0: nothing to copy.
struct CopyJob {
    const void* src = nullptr;
    void* dst = nullptr;
    unsigned long long n16 = 0;
    uint32_t parts = 0, pad_ = 0;
};
hipError_t launch_match_mfma(int mode, const SegDesc* segs, const uint32_t* nitems_dev, uint32_t max_items,
                       uint32_t* queue_head, uint32_t* accmask, const ScanAccept* accept_dev, hipStream_t s,
                       const CopyJob& job = CopyJob(), uint32_t* copy_head = nullptr, int leave_cus = 0);
// leave_cus: launch that many workgroups fewer than the device has CUs (a workgroup owns its CU): the CUs stay free for
// what runs beside the scan - amc_match_verify_pairs' verification of the batch before
int match_mfma_shape();  // waves per workgroup in use (8 or 4)

hipError_t launch_resolve_index(int side, const ImageDev* imgs, const PairDev* pairs, uint32_t npairs,
                          Top2* table, uint32_t* accmask, const float* acos_lut, FinalizeParams fp,
                          const uint32_t* cand_cnt, const uint32_t* candbuf, uint32_t* err_count,
                          bool grouped, const uint32_t* order, uint32_t norder, hipStream_t s);
// largest image (padded rows) the tile-grouped variant of resolve_index handles (its LDS histogram)
uint32_t resolve_grouped_max_rows();
constexpr uint32_t kSelectMaxCols = 1u << 20;  // select_candidates' LDS bitmap: at most 128 KiB of dynamic shared memory

// max_cols: the largest image 2 (rows) among the launch's pairs - sizes the kernel's LDS bitmap
hipError_t launch_select_candidates(const ImageDev* imgs, const PairDev* pairs, uint32_t npairs, uint32_t max_cols,
                              const Top2* rowbuf, const uint32_t* accmask, const float* acos_lut,
                              FinalizeParams fp, uint32_t* cand_cnt, uint32_t* candbuf, hipStream_t s);

hipError_t launch_finalize(const ImageDev* imgs, const PairDev* pairs, uint32_t npairs,
                     const Top2* rowbuf, const Top2* colbuf, uint32_t* accmask,
                     const float* acos_lut, FinalizeParams fp, uint32_t* cursor, uint32_t capacity,
                     uint32_t* pair_off, uint32_t* pair_cnt, uint32_t* matches, hipStream_t s);

// match_common.hip: device -> pinned host copy by a small-grid kernel that co-resides with the scan (bytes % 8 == 0)
hipError_t launch_host_copy(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s);
// a small copy as a kernel on the stream (pinned host or device memory on either side, 4-byte units): match_common.hip
hipError_t launch_copy_words(void* dst, const void* src, size_t bytes, hipStream_t s);
hipError_t launch_reorder_matches(const uint32_t* src_off, const uint32_t* cnt, const uint64_t* dst_off, uint32_t npairs,
                            const uint32_t* src, uint32_t* dst, hipStream_t s);

// ----- two-view verification (tvg_core.h; kernels in tvg_e.hip, tvg_fh.hip and their _big builds) ------------------------------------------------------
struct CameraDev {
    int32_t model_id;    // COLMAP camera model id 0..10 (camera_math.h)
    int32_t has_prior;
    uint64_t width, height;
    double params[12];
};
struct TvgImage {
    const float* kp;     // rows x 2 float32 (x, y), or
    const double* kp64;  // rows x 2 float64 when the points were uploaded in double precision
    const double* kpn;   // rows x 2 float64: Camera::CamFromImg of every keypoint, for cameras with distortion
                         // parameters (lifted once per image, camera.hip / amc_api.hip ensure_normalized);
                         // nullptr for SIMPLE_PINHOLE / PINHOLE, whose lift is two divisions done in place
    uint32_t rows;
    uint32_t pad;
    CameraDev cam;
};
// Camera::CamFromImg of every keypoint of one image (polynomial distortion models only; camera.hip)
hipError_t launch_undistort(const float* kp, const double* kp64, uint32_t rows, const CameraDev& cam, double* kpn,
                            hipStream_t s);
hipError_t launch_project(const double* uv, uint32_t n, const CameraDev& cam, double* xy, hipStream_t s);
struct TvgPair {
    uint32_t slot1, slot2;
    uint64_t match_off;   // into the batch's match array (in matches, not uint32s)
    uint64_t mask_off;    // into the device mask buffer; 128-B aligned so that no two pairs (=
                          // two waves, possibly on different XCDs with non-coherent L2s) ever
                          // write bytes of the same cache line
    uint32_t M;
    uint32_t tab_off[3];  // dyn_max_num_trials tables for the E (k=5), F (k=7), H (k=4) RANSACs
    uint32_t orig;        // index of the pair's TvgOut record (the caller's pair index)
};
// ----- relative pose of verified pairs (pose.hip) ----------------------------------------------
// One pair of EstimateTwoViewGeometryPose: the geometry's config / E / H and its inlier matches
// (rows match_off .. match_off + M of the batch's inlier-match array).
struct PosePair {
    uint32_t slot1, slot2;
    uint64_t match_off;
    uint64_t mask_off;   // with a mask: the pair's inlier bytes (amc_verify_pairs' device mask), row k counts if non-zero
    uint64_t ws_off;     // the pair's M doubles of the cosine workspace
    uint32_t M;
    int32_t config;
    double E[9], H[9];
};
// R, t of the winning candidate, its quaternion, the number of points in front of both cameras and
// the one or two cosines the median triangulation angle is the acos of (host libm; pose_math.h)
struct alignas(128) PoseOut {
    int32_t ok;
    int32_t t_is_zero;   // ||t|| == 0 (PLANAR_OR_PANORAMIC -> PANORAMIC)
    uint32_t num_points3D;
    uint32_t pad;
    double R[9], t[3], q[4], cmed[2];
};
// mask == nullptr: every listed match is an inlier match
hipError_t launch_pose(const TvgImage* imgs, const PosePair* pairs, uint32_t npairs, const uint32_t* matches,
                       const uint8_t* mask, double* cosine_ws, PoseOut* out, hipStream_t s);

// device-side result record: amc_tvg padded to its own cache lines (same reason)
struct alignas(128) TvgOut {
    amc_tvg g;
    // shader-clock cycles spent per phase (diagnostics; printed with AMC_TVG_PROFILE=1):
    // 0 sampling, 1 minimal solvers, 2 scoring of sample models, 3 local optimisation, 4 total
    unsigned long long prof[8];
    // algorithmic work of the pair (tvg_core.h WK_*): residual evaluations by kind (Sampson, homography transfer,
    // translation), minimal solves (5-point, 7-point, 4-point), local solves (5-point, 8-point, DLT), the inlier
    // points those local solves summed over, 1-point trials - the inputs of bench.py's FP64 roofline
    unsigned long long work[12];
};
struct TvgParams {
    int32_t min_num_inliers, detect_watermark, force_H_use, min_num_trials;
    int32_t max_trials[4];  // E, F, H, watermark translation — clamped as the RANSAC ctor does
    double min_E_F_inlier_ratio, max_H_inlier_ratio, watermark_min_inlier_ratio,
        watermark_border_size, max_error;
    int32_t force_slow_sampler;  // test hook (AMC_TVG_SLOW_SAMPLER=1): draw-by-draw sampler path only
    int32_t no_fast_count;       // test hook (AMC_TVG_EXACT_COUNT=1): no division-free test in the counting loop
    int32_t no_fast32;           // test hook (AMC_TVG_NO_S32=1): no FP32 Sampson pre-filter in the F / E counting loops
    int32_t mode;                // 0: EstimateTwoViewGeometry; 1 / 2 / 3: a single F / H / E LO-RANSAC
    uint32_t* bad_index_count;   // += 1 per pair whose matches index past an image's keypoints (the pair is skipped)
    // dyn_max_num_trials of the watermark (translation, 1-point) RANSAC.  Its sample count is the pair's inlier
    // count, known only on the device, so a table by match count cannot be laid out ahead; but
    // ComputeNumTrials depends on (num_inliers, num_samples) only through r = num_inliers / num_samples and is
    // non-increasing in r, so the host (host libm, like every other trial table) bisects, for every trial
    // count T in [0, max_trials[3]], the smallest double r with ComputeNumTrials(r) <= T: wm_cut[T] (2.0 when
    // there is none).  The kernel then has dyn_max = the first T with r >= wm_cut[T].
    const double* wm_cut;
    // The sample stream: std::mt19937(seed) is re-seeded for every pair (D4), so every pair consumes the SAME
    // sequence - laid out once by the host as tempered 32-bit words.  A RANSAC's generator state is then a position
    // in this table (no twist, no snapshot, no roll-back of a 624-word state; the position travels from the E kernel
    // to the F/H kernel in TvgEState).  stream_len covers every RANSAC running to its trial cap; a wave that would
    // read past it (only through Lemire's rejection loop) counts in *stream_err and the host retries with more.
    const uint32_t* stream;
    uint32_t stream_len;
    uint32_t* stream_err;
};
// What the essential-matrix kernel hands to the F/H kernel for one pair: the RANSAC report (the mask goes to the
// pair's region of a second mask buffer) and the stream position it stopped at.
struct alignas(64) TvgEState {
    double model[9];
    double sum;
    int32_t cnt, success, num_trials;
    uint32_t soff;
};
void tvg_diag_report();    // diagnostic builds (-DAMC_TVG_LODIAG): stage cycles of the local estimators, on stderr
void tvg_diag_report_e();  // ... of the essential-matrix kernel
size_t tvg_ws_doubles_host(uint32_t mcap);
size_t tvg_ws_doubles_e_host(uint32_t mcap);   // the essential-matrix kernel's waves (larger model / staging region)
size_t tvg_ws_mask_bytes_host(uint32_t mcap);
// Target occupancy (waves per SIMD) of the two verification kernels: sets their VGPR budgets and LDS shares.
// tvg_e_kernel holds the 5-point solver and its root finder (256 VGPRs): 2 (round 6, final code: 3 waves -1.7 %).
// tvg_fh_kernel: 4 since round 6 (128 VGPRs; 3 waves -5.5 %, same box).  Rounds 3 to 5 ran it at 3: the 7-point solver's
// 7 x 9 matrix (126 VGPRs) spills at 4 waves, and with the scratch traffic those builds had (wave-uniform state and
// models spilled around every chunk's calls) more waves only added to it (round 3: 4 -> 3 waves +2.4 %).  With that
// traffic gone (tvg_core.h lo_ransac) the fourth wave hides what latency is left.
#ifndef AMC_E_WAVES
#define AMC_E_WAVES 2
#endif
#ifndef AMC_FH_WAVES
#define AMC_FH_WAVES 4
#endif
constexpr int kTvgEWavesPerSimd = AMC_E_WAVES;
constexpr int kTvgFhWavesPerSimd = AMC_FH_WAVES;
size_t tvg_lds_bytes(uint32_t mcap, int waves);    // F/H kernel
size_t tvg_lds_bytes_e(uint32_t mcap, int waves);  // essential-matrix kernel (+ its root finder's scratch)
// pose.hip: a verification call's records and masks in the caller's layout, work counters summed
hipError_t launch_pack_verify(const TvgOut* out, const TvgPair* tp, uint32_t npairs, const uint8_t* mask_src,
                              const uint64_t* moff, amc_tvg* tvg_dst, uint8_t* mask_dst, unsigned long long* work,
                              int32_t trivial_below, hipStream_t s);
// pose.hip: PoseFromHomographyMatrix on given points; in27 = H, K1, K2; out16 = R, t, n, count
hipError_t launch_homography_decomposition(const double* in27, const double* p1, const double* p2, uint32_t n, double* out16,
                                           double* points3D, hipStream_t s);
hipError_t launch_sampson(const double* p1, const double* p2, size_t n, const double* E9, double* out,
                          hipStream_t s);
// the essential-matrix RANSAC of the listed (calibrated) pairs -> estate[pair.orig], emask + pair.mask_off
hipError_t launch_tvg_e(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                        const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint32_t mcap, uint32_t num_waves,
                        int waves_per_block, uint32_t* queue_head, TvgEState* estate, uint8_t* emask, TvgOut* out,
                        uint8_t* out_mask, hipStream_t s);
// F and H RANSACs, model selection, watermark test of the listed pairs (after launch_tvg_e on the same stream)
hipError_t launch_tvg_fh(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                         const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint8_t* mask_ws, uint32_t mcap,
                         uint32_t num_waves, int waves_per_block, uint32_t* queue_head, const TvgEState* estate,
                         const uint8_t* emask, TvgOut* out, uint8_t* out_mask, hipStream_t s);

// the "big" builds of the two kernels (tvg_e_big.hip / tvg_fh_big.hip): index arrays in global memory, for pairs of
// ~38,000 .. 65,535 matches.  ws holds, behind the num_waves point workspaces, tvg_big_idx_doubles_host(mcap) doubles per wave.
size_t tvg_big_lds_bytes(int waves);
size_t tvg_big_lds_bytes_e(int waves);
size_t tvg_big_idx_doubles_host(uint32_t mcap);
hipError_t launch_tvg_e_big(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                            const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint32_t mcap, uint32_t num_waves,
                            int waves_per_block, uint32_t* queue_head, TvgEState* estate, uint8_t* emask, TvgOut* out,
                            uint8_t* out_mask, hipStream_t s);
hipError_t launch_tvg_fh_big(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                             const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint8_t* mask_ws, uint32_t mcap,
                             uint32_t num_waves, int waves_per_block, uint32_t* queue_head, const TvgEState* estate,
                             const uint8_t* emask, TvgOut* out, uint8_t* out_mask, hipStream_t s);

}  // namespace amc
