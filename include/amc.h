/*
 * amc.h — C ABI of libamc.so: the MI355X (gfx950) match + verify core.
 *
 * This is the drop-in boundary for pycolmap's exhaustive SIFT matching / two-view
 * verification hot path.  Plain C: `int` status codes, plain pointers and sizes, no torch or
 * C++ types.  The entry points stand in for exactly what the reference's binding layer
 * reaches through COLMAP's C++ API for this path:
 *
 *   reference call site (file:line)                              replaced by
 *   -----------------------------------------------------------  -------------------------------
 *   CreateExhaustiveFeatureMatcher / CreateSequentialFeature-    amc_ctx_create + amc_upload_*
 *     Matcher(opts, sift_opts, tvg_opts, db_path)->Start()        + amc_match_pairs (+ amc_verify_pairs)
 *     /root/reference/pycolmap/pipeline/match_features.h:45-47,    driven by the host scheduler
 *     :220, :229                                                   (pycolmap_amd/csrc/host)
 *   CreateImagePairsFeatureMatcher(...) in verify_matches        amc_verify_pairs
 *     /root/reference/pycolmap/pipeline/match_features.h:64-66
 *   SiftMatchingOptions{max_ratio,max_distance,cross_check}      amc_match_opts
 *     /root/reference/pycolmap/pipeline/match_features.h:73-98
 *   FeatureMatches <-> N x 2 uint32 row-major                    amc_match_result.matches
 *     /root/reference/pycolmap/estimators/two_view_geometry.h:19-38
 *
 * The operator-level seam inside COLMAP 3.9.1 that amc_match_pairs covers is
 * FeatureMatcher::Match(descriptors1, descriptors2, &matches) as called by
 * FeatureMatcherWorker::Run; semantics are those of
 * FindBestMatchesBruteForce and results are bit-identical to
 * oracle/match_oracle.c.
 *
 * Conventions
 *   - every function returns AMC_OK (0) or a negative AMC_E_* code; amc_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - the library owns all device memory.  Host buffers passed in are caller-owned and may be
 *     freed as soon as the call returns.  Result buffers are library-allocated (pinned host
 *     memory) and released with the matching *_free.
 *   - calls are blocking unless stated otherwise; a ctx is thread-compatible (one thread at a
 *     time), different ctxs are independent.
 *   - there is NO CPU fallback: if no gfx950 device/kernel image is available the calls fail
 *     with AMC_E_HIP.
 */
#ifndef AMC_H_
#define AMC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMC_ABI_VERSION 1
#define AMC_DESC_DIM 128 /* SIFT descriptor bytes; /root/reference/pycolmap/feature/sift.h:76-77 */

enum {
    AMC_OK = 0,
    AMC_E_INVALID = -1, /* bad argument (maps to ValueError, as THROW_CHECK does:
                           /root/reference/pycolmap/log_exceptions.h:54-76) */
    AMC_E_HIP = -2,     /* HIP runtime / kernel failure, or no device */
    AMC_E_NOMEM = -3,
    AMC_E_STATE = -4    /* e.g. slot not uploaded */
};

/* Which match kernel to run.  AUTO picks the int8-MFMA kernel (exact for any u8 values) unless
 * image 2 has more than 8192 descriptors and cross_check is on (candidate bitmap limit), where
 * it uses the u8 dot4 kernel (see DESIGN.md "match kernels"). */
enum { AMC_KERNEL_AUTO = 0, AMC_KERNEL_MFMA = 1, AMC_KERNEL_DOT4 = 2 };

typedef struct amc_ctx amc_ctx;

/* Mirrors SiftMatchingOptions' matching fields
 * (/root/reference/pycolmap/pipeline/match_features.h:82-91).  Doubles, cast to float at the
 * comparison exactly as COLMAP does. */
typedef struct amc_match_opts {
    double max_ratio;    /* default 0.8 */
    double max_distance; /* default 0.7 */
    int32_t cross_check; /* default 1 */
    int32_t kernel;      /* AMC_KERNEL_* (default AUTO) */
} amc_match_opts;

/* CSR match table for a list of image pairs: pair p owns matches[2*offsets[p] ..
 * 2*offsets[p+1]) as (idx1, idx2) uint32 rows, ascending in idx1 — the layout of COLMAP's
 * `matches` blob and of pycolmap's N x 2 uint32 view. */
typedef struct amc_match_result {
    size_t npairs;
    uint64_t* offsets;   /* npairs + 1 */
    uint32_t* matches;   /* 2 * offsets[npairs] */
    uint64_t num_distances;   /* sum over pairs of n1*n2 (the BASELINE.json metric's unit) */
    uint64_t pairs_mfma;      /* pairs routed to the int8-MFMA kernel */
    uint64_t pairs_dot4;      /* pairs routed to the u8 dot4 kernel */
    double device_ms;         /* first kernel launch -> last result byte on host, HIP events */
    double match_kernel_ms;   /* sum of one-way match-kernel launch durations (image 1 -> image 2
                                 scan), HIP events on the stream */
    double cross_kernel_ms;   /* candidate selection + reverse scan of candidate columns */
    uint32_t match_kernel_launches;
    void* _priv;
} amc_match_result;

const char* amc_last_error(void);
int amc_abi_version(void);

/* Number of visible HIP devices (>= 0), or AMC_E_HIP. */
int amc_device_count(void);

int amc_ctx_create(int device_id, amc_ctx** out);
void amc_ctx_destroy(amc_ctx* ctx);

/* Launch all of this ctx's work on `hip_stream` (a hipStream_t) instead of the ctx's own
 * stream; pass NULL to restore.  Lets a host that owns streams (e.g. torch) order/time us. */
int amc_ctx_set_stream(amc_ctx* ctx, void* hip_stream);

/* Size the image-slot table. Slots are dense ids 0..num_slots-1 chosen by the caller (the host
 * layer maps COLMAP image_ids to slots). Discards previously uploaded data. */
int amc_ctx_reserve_slots(amc_ctx* ctx, uint32_t num_slots);

/* Upload an image's descriptors: rows x 128 uint8 row-major (COLMAP FeatureDescriptors /
 * the `descriptors` blob).  rows may be 0.  The library copies. */
int amc_upload_descriptors(amc_ctx* ctx, uint32_t slot, const uint8_t* host_desc, uint32_t rows);

/* Same, but the source already lives in this device's memory (e.g. a torch uint8 tensor).
 * The library still makes its own prepared copy; the source may be freed after return. */
int amc_upload_descriptors_device(amc_ctx* ctx, uint32_t slot, const void* dev_desc,
                                  uint32_t rows);

/* Brute-force match every listed pair (slot1[p] = image 1 = match rows, slot2[p] = image 2).
 * Semantics: FindBestMatchesBruteForce.  Pairs with an empty image yield 0
 * matches.  Blocking.  `out` is filled on success and must be released with
 * amc_match_result_free. */
int amc_match_pairs(amc_ctx* ctx, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                    const amc_match_opts* opts, amc_match_result* out);
void amc_match_result_free(amc_match_result* r);

/* Defaults identical to COLMAP's SiftMatchingOptions. */
void amc_match_opts_default(amc_match_opts* opts);

/* The host-built acosf table the thresholds use: lut[d] = acosf(min(d / 512^2, 1)),
 * d in [0, 262144].  Copies 262145 floats. (Test hook: must equal the oracle's table.) */
int amc_get_acos_lut(amc_ctx* ctx, float* out);

#ifdef __cplusplus
}
#endif
#endif /* AMC_H_ */
