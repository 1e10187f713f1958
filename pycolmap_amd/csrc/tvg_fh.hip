// tvg_fh.hip — second kernel of the two-view verification (tvg_core.h): for every pair the fundamental-matrix and
// homography LO-RANSACs, COLMAP's model selection (EstimateCalibrated/UncalibratedTwoViewGeometry) and the watermark
// test.  Calibrated pairs arrive with their essential-matrix RANSAC already run by tvg_e_kernel (tvg_e.hip): its report,
// inlier mask and the position it left the sample stream at are read from the pair's TvgEState.  Nothing here needs the
// 5-point solver's registers, so the kernel is built for AMC_FH_WAVES waves per SIMD.
#include "tvg_core.h"

namespace amc {

__device__ __forceinline__ bool in_bbox(double x, double y, double minx, double maxx, double miny,
                                        double maxy) {
    return x >= minx && x <= maxx && y >= miny && y <= maxy;
}

// EstimateTwoViewGeometry for pair q, by one wave
__device__ __noinline__ void process_pair_fh(Wave& w, uint32_t q, const TvgImage* __restrict__ imgs,
                                             const TvgPair* __restrict__ pairs,
                                             const uint32_t* __restrict__ matches,
                                             const uint32_t* __restrict__ trial_tabs, const TvgParams& P,
                                             const TvgEState* __restrict__ estate, const uint8_t* __restrict__ emask,
                                             TvgOut* __restrict__ out, uint8_t* __restrict__ out_mask) {
    const int lane = w.lane;
    const uint32_t mcap = w.mcap;
    for (int i = 0; i < 8; ++i) w.prof[i] = 0;
    const TvgPair pr = pairs[q];
    const uint32_t oq = pr.orig;  // results are stored by the caller's pair index, whatever the queue order
    w.work = out[oq].work;        // zeroed by the host before the launches; both kernels add to it
    const unsigned long long tstart = prof_clock();
    // the image records are read field by field where they are needed (wave-uniform scalar loads): a by-value
    // copy of both would hold 2 x 38 dwords of camera parameters in scalar registers for the whole pair
    const TvgImage* __restrict__ pim1 = imgs + pr.slot1;
    const TvgImage* __restrict__ pim2 = imgs + pr.slot2;
    const int M = (int)pr.M;
    amc_tvg g;
    g.config = AMC_TVG_UNDEFINED;
    g.num_inliers = 0;
    for (int i = 0; i < 9; ++i) { g.E[i] = 0; g.F[i] = 0; g.H[i] = 0; }
    for (int i = 0; i < 4; ++i) g.num_trials[i] = 0;
    for (int i = 0; i < 3; ++i) g.model_inliers[i] = 0;
    uint8_t* omask = out_mask + pr.mask_off;
    for (int k = lane; k < M; k += 64) omask[k] = 0;

    if (P.mode == 0 && M < P.min_num_inliers) {
        g.config = AMC_TVG_DEGENERATE;
        if (lane == 0) out[oq].g = g;
        return;
    }
    // ---- matched points (FeatureKeypointsToPointsVector: float -> double) ------------------
    PtRec* XP = ws_pts(w, 0);  // (x1, y1, x2, y2) records, pixels
    const double* X1 = &XP[0].x1;
    const uint32_t* mm = matches + 2 * pr.match_off;
    // The indices are checked here, where they are read anyway (the host only checks the pairs that
    // return above): a pair with a match past an image's keypoints is counted and not estimated - the
    // host then fails the whole call with AMC_E_INVALID, naming the match.
    bool bad = false;
    {
        const float* __restrict__ kp1 = pim1->kp;
        const float* __restrict__ kp2 = pim2->kp;
        const double* __restrict__ kd1 = pim1->kp64;
        const double* __restrict__ kd2 = pim2->kp64;
        const uint32_t rows1 = pim1->rows, rows2 = pim2->rows;
        for (int k = lane; k < M; k += 64) {
            const uint32_t i1 = mm[2 * k], i2 = mm[2 * k + 1];
            if (i1 >= rows1 || i2 >= rows2) {
                bad = true;
                continue;
            }
            PtRec r;
            r.x1 = kd1 ? kd1[2 * (size_t)i1] : (double)kp1[2 * (size_t)i1];
            r.y1 = kd1 ? kd1[2 * (size_t)i1 + 1] : (double)kp1[2 * (size_t)i1 + 1];
            r.x2 = kd2 ? kd2[2 * (size_t)i2] : (double)kp2[2 * (size_t)i2];
            r.y2 = kd2 ? kd2[2 * (size_t)i2 + 1] : (double)kp2[2 * (size_t)i2 + 1];
            XP[k] = r;
        }
    }
    if (__any(bad)) {
        g.config = AMC_TVG_UNDEFINED;
        if (lane == 0) {
            atomicAdd(P.bad_index_count, 1u);
            out[oq].g = g;
        }
        return;
    }
    wave_mem_sync();

    // mode 0: the EstimateTwoViewGeometry dispatch; modes 1 / 2: exactly one of F / H (mode 3, E alone, is the other
    // kernel's)
    const bool calibrated = P.mode == 0 && !P.force_H_use && pim1->cam.has_prior && pim2->cam.has_prior;
    const bool run_F = P.mode == 0 ? !P.force_H_use : P.mode == 1;
    const bool run_H = P.mode == 0 || P.mode == 2;
    const uint8_t* maskE = emask + pr.mask_off;
    uint8_t *maskF = w.masks + mcap, *maskH = w.masks + 2 * (size_t)mcap;
    Report E_rep, F_rep, H_rep;
    E_rep.success = F_rep.success = H_rep.success = false;
    E_rep.support.cnt = F_rep.support.cnt = H_rep.support.cnt = 0;
    E_rep.num_trials = F_rep.num_trials = 0;
    for (int i = 0; i < 9; ++i) { E_rep.model[i] = 0; F_rep.model[i] = 0; }

    // SetPRNGSeed(seed): the stream starts over for every pair; a calibrated pair's E RANSAC has consumed its share
    w.soff = 0;
    if (calibrated) {
        const TvgEState* es = estate + oq;
        E_rep.success = es->success != 0;
        E_rep.support.cnt = es->cnt;
        E_rep.num_trials = es->num_trials;
        for (int i = 0; i < 9; ++i) E_rep.model[i] = es->model[i];
        w.soff = es->soff;
        for (int i = 0; i < 9; ++i) g.E[i] = E_rep.model[i];
        g.num_trials[0] = E_rep.num_trials;
        g.model_inliers[0] = E_rep.support.cnt;
    }
    RansacCfg cfg;
    cfg.wm_cut = nullptr;
    cfg.min_trials = P.min_num_trials;
    cfg.force_slow_sampler = P.force_slow_sampler;
    cfg.no_fast_count = P.no_fast_count;
    cfg.no_fast32 = P.no_fast32;
    if (run_F) {
        cfg.max_res = P.max_error * P.max_error;
        cfg.max_trials = P.max_trials[1];
        cfg.dyn_tab = trial_tabs + pr.tab_off[1];
        F_rep = lo_ransac<K_F7, K_F8>(w, cfg, X1, mcap, M, maskF);
        for (int i = 0; i < 9; ++i) g.F[i] = F_rep.model[i];
        g.num_trials[1] = F_rep.num_trials;
        g.model_inliers[1] = F_rep.support.cnt;
    }
    H_rep.num_trials = 0;
    for (int i = 0; i < 9; ++i) H_rep.model[i] = 0;
    if (run_H) {
        cfg.max_res = P.max_error * P.max_error;
        cfg.max_trials = P.max_trials[2];
        cfg.dyn_tab = trial_tabs + pr.tab_off[2];
        H_rep = lo_ransac<K_H, K_H>(w, cfg, X1, mcap, M, maskH);
        for (int i = 0; i < 9; ++i) g.H[i] = H_rep.model[i];
        g.num_trials[2] = H_rep.num_trials;
        g.model_inliers[2] = H_rep.support.cnt;
    }
    if (P.mode != 0) {
        // single-RANSAC report: config carries report.success, the mask is report.inlier_mask
        const Report& r = P.mode == 1 ? F_rep : H_rep;
        const uint8_t* rm = P.mode == 1 ? maskF : maskH;
        g.config = r.success ? 1 : 0;
        g.num_inliers = r.support.cnt;
        if (r.success)
            for (int k = lane; k < M; k += 64) omask[k] = rm[k];
        if (lane == 0) {
            out[oq].g = g;
            w.prof[4] = prof_clock() - tstart;
            if (AMC_TVG_PROF_ON) for (int i = 0; i < 8; ++i) out[oq].prof[i] += w.prof[i];
        }
        return;
    }

    // ---- model selection (two_view_geometry.cc), wave-uniform --------------------------------
    const int minI = P.min_num_inliers;
    const int Ei = E_rep.support.cnt, Fi = F_rep.support.cnt, Hi = H_rep.support.cnt;
    const uint8_t* best_mask = nullptr;
    bool best_ok = false;  // best_mask non-null and non-empty (its RANSAC succeeded)
    int num_inliers = 0;
    bool done = false;
    if (P.force_H_use) {
        if (!H_rep.success || Hi < minI) { g.config = AMC_TVG_DEGENERATE; done = true; }
        else { g.config = AMC_TVG_PLANAR_OR_PANORAMIC; best_mask = maskH; best_ok = true; num_inliers = Hi; }
    } else if (calibrated) {
        if ((!E_rep.success && !F_rep.success && !H_rep.success) || (Ei < minI && Fi < minI && Hi < minI)) {
            g.config = AMC_TVG_DEGENERATE; done = true;
        } else {
            const double E_F = (double)Ei / (double)Fi, H_F = (double)Hi / (double)Fi, H_E = (double)Hi / (double)Ei;
            if (E_rep.success && E_F > P.min_E_F_inlier_ratio && Ei >= minI) {
                if (Ei >= Fi) { num_inliers = Ei; best_mask = maskE; best_ok = E_rep.success; }
                else { num_inliers = Fi; best_mask = maskF; best_ok = F_rep.success; }
                if (H_E > P.max_H_inlier_ratio) {
                    g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                    if (Hi > num_inliers) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
                } else g.config = AMC_TVG_CALIBRATED;
            } else if (F_rep.success && Fi >= minI) {
                num_inliers = Fi; best_mask = maskF; best_ok = true;
                if (H_F > P.max_H_inlier_ratio) {
                    g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                    if (Hi > num_inliers) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
                } else g.config = AMC_TVG_UNCALIBRATED;
            } else if (H_rep.success && Hi >= minI) {
                num_inliers = Hi; best_mask = maskH; best_ok = true; g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
            } else { g.config = AMC_TVG_DEGENERATE; done = true; }
        }
    } else {
        if ((!F_rep.success && !H_rep.success) || (Fi < minI && Hi < minI)) {
            g.config = AMC_TVG_DEGENERATE; done = true;
        } else {
            const double H_F = (double)Hi / (double)Fi;
            best_mask = maskF; best_ok = F_rep.success; num_inliers = Fi;
            if (H_F > P.max_H_inlier_ratio) {
                g.config = AMC_TVG_PLANAR_OR_PANORAMIC;
                if (Hi >= Fi) { num_inliers = Hi; best_mask = maskH; best_ok = H_rep.success; }
            } else g.config = AMC_TVG_UNCALIBRATED;
        }
    }
    if (!done) {
        if (best_ok) {
            g.num_inliers = num_inliers;
            for (int k = lane; k < M; k += 64) omask[k] = best_mask[k];
        } else {
            g.num_inliers = 0;
        }
        // ---- DetectWatermark -----------------------------------------------------------------
        if (P.detect_watermark && best_ok) {
            const uint64_t w1 = pim1->cam.width, h1 = pim1->cam.height, w2 = pim2->cam.width, h2 = pim2->cam.height;
            const double diagonal1 = dsqrt((double)(w1 * w1 + h1 * h1));
            const double diagonal2 = dsqrt((double)(w2 * w2 + h2 * h2));
            const double minx1 = P.watermark_border_size * diagonal1, miny1 = minx1;
            const double maxx1 = (double)w1 - minx1, maxy1 = (double)h1 - miny1;
            const double minx2 = P.watermark_border_size * diagonal2, miny2 = minx2;
            const double maxx2 = (double)w2 - minx2, maxy2 = (double)h2 - miny2;
            PtRec* IP = ws_pts(w, 1);  // the inlier subset, same record layout
            const double* ix1 = &IP[0].x1;
            int basep = 0, border = 0;
            for (int k0 = 0; k0 < M; k0 += 64) {
                const int k = k0 + lane;
                const bool in = k < M && best_mask[k];
                const unsigned long long bal = __ballot(in);
                if (in) {
                    const int pos = basep + __popcll(bal & ((1ull << lane) - 1ull));
                    const PtRec r = XP[k];
                    IP[pos] = r;
                    if (!in_bbox(r.x1, r.y1, minx1, maxx1, miny1, maxy1) &&
                        !in_bbox(r.x2, r.y2, minx2, maxx2, miny2, maxy2))
                        ++border;
                }
                basep += __popcll(bal);
            }
            wave_mem_sync();
            border = wave_sum_int(border);
            const double ratio = (double)border / (double)num_inliers;
            if (!(ratio < P.watermark_min_inlier_ratio)) {
                cfg.max_res = P.max_error * P.max_error;
                cfg.max_trials = P.max_trials[3];
                cfg.dyn_tab = nullptr;
                cfg.wm_cut = P.wm_cut;
                const Report T_rep = lo_ransac<K_T, K_T>(w, cfg, ix1, mcap, num_inliers, w.masks + 3 * (size_t)mcap);
                g.num_trials[3] = T_rep.num_trials;
                const double inlier_ratio = (double)T_rep.support.cnt / (double)num_inliers;
                if (inlier_ratio >= P.watermark_min_inlier_ratio) g.config = AMC_TVG_WATERMARK;
            }
        }
    }
    if (lane == 0) {
        out[oq].g = g;
        w.prof[4] = prof_clock() - tstart;
        if (AMC_TVG_PROF_ON) for (int i = 0; i < 8; ++i) out[oq].prof[i] += w.prof[i];
    }
}

#if defined(AMC_TVG_BIG)   // second build of this file (tvg_fh_big.hip): index arrays in global memory, own symbol names
#define tvg_fh_kernel tvg_fh_big_kernel
#define launch_tvg_fh launch_tvg_fh_big
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kTvgFhWavesPerSimd, kTvgFhWavesPerSimd))) void tvg_fh_kernel(
    const TvgImage* __restrict__ imgs, const TvgPair* __restrict__ pairs, uint32_t npairs,
    const uint32_t* __restrict__ matches, const uint32_t* __restrict__ trial_tabs, TvgParams P,
    double* __restrict__ ws_all, uint8_t* __restrict__ mask_ws_all, uint32_t mcap, uint32_t* __restrict__ queue_head,
    const TvgEState* __restrict__ estate, const uint8_t* __restrict__ emask, TvgOut* __restrict__ out,
    uint8_t* __restrict__ out_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    Wave w;
    w.lane = lane;
    wave_carve(w, (AMC_LDS char*)smem + (size_t)wid * tvg_lds_per_wave(mcap), mcap);
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wid;
    w.ws = ws_all + gw * tvg_ws_doubles(mcap);
#if defined(AMC_TVG_BIG)
    wave_carve_idx(w, ws_all + (size_t)gridDim.x * (blockDim.x >> 6) * tvg_ws_doubles(mcap) + gw * tvg_idx_doubles(mcap), mcap);
#endif
    w.masks = mask_ws_all + gw * tvg_ws_bytes_extra(mcap);
    w.stream = P.stream;
    w.stream_len = P.stream_len;
    w.err = P.stream_err;
    w.soff = 0;
    w.rootscr = RootScratch{};  // (the essential-matrix kernel's)

    LODIAG_WAVE_START(gw);
    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(queue_head, 1u);
        q = __shfl(q, 0);
        if (q >= npairs) break;
        process_pair_fh(w, q, imgs, pairs, matches, trial_tabs, P, estate, emask, out, out_mask);
    }
    LODIAG_WAVE_END(gw);
}

#if !defined(AMC_TVG_BIG)
size_t tvg_ws_doubles_host(uint32_t mcap) { return tvg_ws_doubles(mcap); }
size_t tvg_ws_mask_bytes_host(uint32_t mcap) { return tvg_ws_bytes_extra(mcap); }
size_t tvg_lds_bytes(uint32_t mcap, int waves) { return (size_t)waves * tvg_lds_per_wave(mcap); }
#else
size_t tvg_big_lds_bytes(int waves) { return (size_t)waves * tvg_lds_per_wave(0); }
#endif

#if defined(AMC_TVG_BIG)
// (the diagnostics report and the Sampson kernel belong to the regular build)
#elif defined(AMC_TVG_LODIAG)
void tvg_diag_report() {
    unsigned long long h[48];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lo_diag), sizeof h) != hipSuccess) return;
    std::fprintf(stderr, "[amc tvg lodiag tvg_diag_report] local 5-point solves %llu: cycles per solve ata %.0f jacobi %.0f build %.0f roots %.0f models %.0f | "
                 "8-point solves %llu: ata %.0f jacobi %.0f finish %.0f | DLT solves %llu: ata %.0f jacobi %.0f finish %.0f\n",
                 h[0], (double)h[1] / (h[0] ? h[0] : 1), (double)h[2] / (h[0] ? h[0] : 1), (double)h[3] / (h[0] ? h[0] : 1),
                 (double)h[4] / (h[0] ? h[0] : 1), (double)h[5] / (h[0] ? h[0] : 1), h[8], (double)h[9] / (h[8] ? h[8] : 1),
                 (double)h[10] / (h[8] ? h[8] : 1), (double)h[11] / (h[8] ? h[8] : 1), h[12], (double)h[13] / (h[12] ? h[12] : 1),
                 (double)h[14] / (h[12] ? h[12] : 1), (double)h[15] / (h[12] ? h[12] : 1));
    const char* nm[4] = {"F", "H", "E", "T"};
    for (int k = 0; k < 4; ++k) {
        const unsigned long long* q = h + 16 + 8 * k;
        if (!q[0]) continue;
        std::fprintf(stderr, "[amc tvg lodiag] %s RANSACs %llu: cycles per RANSAC tables %.0f, scalar sync %.0f, exact re-scores %.1f x %.0f, final mask %.0f, minimal solves %.0f, counting %.0f\n",
                     nm[k], q[0], (double)q[1] / q[0], (double)q[2] / q[0], (double)q[3] / q[0], q[3] ? (double)q[4] / q[3] : 0.0, (double)q[5] / q[0],
                     (double)q[6] / q[0], (double)q[7] / q[0]);
    }
    unsigned long long z[48] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lo_diag), z, sizeof z);
    lodiag_report_spans("tvg_fh_kernel");
}
#else
void tvg_diag_report() {}
#endif

#if !defined(AMC_TVG_BIG)
// ComputeSquaredSampsonError over n correspondences (points n x 2, E row-major)
__global__ __launch_bounds__(256) void sampson_kernel(const double* __restrict__ p1, const double* __restrict__ p2,
                                                      size_t n, const double* __restrict__ E9,
                                                      double* __restrict__ out) {
    double e[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = E9[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = sampson(e, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
}
hipError_t launch_sampson(const double* p1, const double* p2, size_t n, const double* E9, double* out,
                          hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(sampson_kernel, dim3(blocks), dim3(256), 0, s, p1, p2, n, E9, out);
    return hipGetLastError();
}
#endif

hipError_t launch_tvg_fh(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                         const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint8_t* mask_ws, uint32_t mcap,
                         uint32_t num_waves, int waves_per_block, uint32_t* queue_head, const TvgEState* estate,
                         const uint8_t* emask, TvgOut* out, uint8_t* out_mask, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    const uint32_t blocks = (num_waves + waves_per_block - 1) / waves_per_block;
    const size_t lds = (size_t)waves_per_block * tvg_lds_per_wave(mcap);  // (this build's own layout: the big one has no index arrays in LDS)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tvg_fh_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = memset_async(queue_head, 0, sizeof(uint32_t), s);  // (the persistent waves pop pairs from it)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tvg_fh_kernel, dim3(blocks), dim3(64 * waves_per_block), lds, s, imgs, pairs, npairs, matches,
                       trial_tabs, P, ws, mask_ws, mcap, queue_head, estate, emask, out, out_mask);
    return hipGetLastError();
}

}  // namespace amc
