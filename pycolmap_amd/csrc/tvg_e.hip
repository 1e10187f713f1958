// tvg_e.hip — first kernel of the two-view verification (tvg_core.h): the essential-matrix LO-RANSAC
// (LORANSAC<EssentialMatrixFivePointEstimator, EssentialMatrixFivePointEstimator>) of every calibrated pair, on the
// CamFromImg-lifted correspondences.  The report, the inlier mask and the position the RANSAC left the sample stream
// at go to the pair's TvgEState / mask region, where tvg_fh_kernel (tvg_fh.hip) picks them up: COLMAP runs E, then F,
// then H on one generator.  The 5-point solver keeps ~200 doubles live per lane, so this kernel - and only this one -
// is built for 2 waves per SIMD (256 VGPRs).
#include "tvg_core.h"

namespace amc {

__device__ __noinline__ void process_pair_e(Wave& w, uint32_t q, const TvgImage* __restrict__ imgs,
                                            const TvgPair* __restrict__ pairs, const uint32_t* __restrict__ matches,
                                            const uint32_t* __restrict__ trial_tabs, const TvgParams& P,
                                            TvgEState* __restrict__ estate, uint8_t* __restrict__ emask,
                                            TvgOut* __restrict__ out, uint8_t* __restrict__ out_mask) {
    const int lane = w.lane;
    const uint32_t mcap = w.mcap;
    for (int i = 0; i < 8; ++i) w.prof[i] = 0;
    const TvgPair pr = pairs[q];
    const uint32_t oq = pr.orig;
    w.work = out[oq].work;
    const unsigned long long tstart = prof_clock();
    const TvgImage* __restrict__ pim1 = imgs + pr.slot1;
    const TvgImage* __restrict__ pim2 = imgs + pr.slot2;
    const int M = (int)pr.M;
    // the pairs tvg_fh_kernel leaves without an estimation are left alone here too (it reports them)
    if (P.mode == 0 && M < P.min_num_inliers) return;
    uint8_t* maskE = emask + pr.mask_off;
    // ---- Camera::CamFromImg of the matched points.  SIMPLE_PINHOLE / PINHOLE: (x - c) / f of the keypoint
    //      (FeatureKeypointsToPointsVector: float -> double).  Cameras with distortion parameters: the keypoints
    //      were lifted once per image (kpn), gather from there.
    PtRec* NP = ws_pts(w, 0);  // (x1, y1, x2, y2) records, normalised
    const uint32_t* mm = matches + 2 * pr.match_off;
    bool bad = false;
    {
        const float* __restrict__ kp1 = pim1->kp;
        const float* __restrict__ kp2 = pim2->kp;
        const double* __restrict__ kd1 = pim1->kp64;
        const double* __restrict__ kd2 = pim2->kp64;
        const double* __restrict__ kn1 = pim1->kpn;
        const double* __restrict__ kn2 = pim2->kpn;
        const uint32_t rows1 = pim1->rows, rows2 = pim2->rows;
        const int model1 = pim1->cam.model_id, model2 = pim2->cam.model_id;
        const int nf1 = cam::num_focal(model1), nf2 = cam::num_focal(model2);
        const double f1x = pim1->cam.params[0], f1y = pim1->cam.params[nf1 - 1];
        const double c1x = pim1->cam.params[nf1], c1y = pim1->cam.params[nf1 + 1];
        const double f2x = pim2->cam.params[0], f2y = pim2->cam.params[nf2 - 1];
        const double c2x = pim2->cam.params[nf2], c2y = pim2->cam.params[nf2 + 1];
        for (int k = lane; k < M; k += 64) {
            const uint32_t i1 = mm[2 * k], i2 = mm[2 * k + 1];
            if (i1 >= rows1 || i2 >= rows2) {
                bad = true;
                continue;
            }
            if (kn1) {
                NP[k].x1 = kn1[2 * (size_t)i1];
                NP[k].y1 = kn1[2 * (size_t)i1 + 1];
            } else {
                const double x = kd1 ? kd1[2 * (size_t)i1] : (double)kp1[2 * (size_t)i1];
                const double y = kd1 ? kd1[2 * (size_t)i1 + 1] : (double)kp1[2 * (size_t)i1 + 1];
                NP[k].x1 = (x - c1x) / f1x;
                NP[k].y1 = (y - c1y) / f1y;
            }
            if (kn2) {
                NP[k].x2 = kn2[2 * (size_t)i2];
                NP[k].y2 = kn2[2 * (size_t)i2 + 1];
            } else {
                const double x = kd2 ? kd2[2 * (size_t)i2] : (double)kp2[2 * (size_t)i2];
                const double y = kd2 ? kd2[2 * (size_t)i2 + 1] : (double)kp2[2 * (size_t)i2 + 1];
                NP[k].x2 = (x - c2x) / f2x;
                NP[k].y2 = (y - c2y) / f2y;
            }
        }
    }
    if (__any(bad)) {
        if (P.mode != 0 && lane == 0) {   // E alone (amc_ransac_pairs): this kernel reports the pair
            amc_tvg g;
            g.config = AMC_TVG_UNDEFINED;
            g.num_inliers = 0;
            for (int i = 0; i < 9; ++i) { g.E[i] = 0; g.F[i] = 0; g.H[i] = 0; }
            for (int i = 0; i < 4; ++i) g.num_trials[i] = 0;
            for (int i = 0; i < 3; ++i) g.model_inliers[i] = 0;
            atomicAdd(P.bad_index_count, 1u);
            out[oq].g = g;
        }
        return;
    }
    wave_mem_sync();
    // SetPRNGSeed(seed): the E RANSAC is the first consumer of the pair's stream
    w.soff = 0;
    RansacCfg cfg;
    cfg.wm_cut = nullptr;
    cfg.min_trials = P.min_num_trials;
    cfg.force_slow_sampler = P.force_slow_sampler;
    cfg.no_fast_count = P.no_fast_count;
    cfg.no_fast32 = P.no_fast32;
    // E threshold: (cam1.CamFromImgThreshold(e) + cam2.CamFromImgThreshold(e)) / 2
    const double e_err = (cam::cam_from_img_threshold(pim1->cam.model_id, pim1->cam.params, P.max_error) +
                          cam::cam_from_img_threshold(pim2->cam.model_id, pim2->cam.params, P.max_error)) / 2;
    cfg.max_res = e_err * e_err;
    cfg.max_trials = P.max_trials[0];
    cfg.dyn_tab = trial_tabs + pr.tab_off[0];
    const Report E_rep = lo_ransac<K_E5, K_E5>(w, cfg, &NP[0].x1, mcap, M, maskE);
    if (lane == 0) {
        TvgEState* es = estate + oq;
        for (int i = 0; i < 9; ++i) es->model[i] = E_rep.model[i];
        es->sum = E_rep.support.sum;
        es->cnt = E_rep.support.cnt;
        es->success = E_rep.success ? 1 : 0;
        es->num_trials = E_rep.num_trials;
        es->soff = w.soff;
    }
    if (P.mode != 0) {
        // single-RANSAC report (amc_ransac_pairs, E alone): config carries report.success, the mask is
        // report.inlier_mask
        amc_tvg g;
        for (int i = 0; i < 9; ++i) { g.E[i] = E_rep.model[i]; g.F[i] = 0; g.H[i] = 0; }
        for (int i = 0; i < 4; ++i) g.num_trials[i] = 0;
        for (int i = 0; i < 3; ++i) g.model_inliers[i] = 0;
        g.num_trials[0] = E_rep.num_trials;
        g.model_inliers[0] = E_rep.support.cnt;
        g.config = E_rep.success ? 1 : 0;
        g.num_inliers = E_rep.support.cnt;
        uint8_t* omask = out_mask + pr.mask_off;
        for (int k = lane; k < M; k += 64) omask[k] = E_rep.success ? maskE[k] : 0;
        if (lane == 0) out[oq].g = g;
    }
    if (lane == 0) {
        w.prof[4] = prof_clock() - tstart;
        if (AMC_TVG_PROF_ON) for (int i = 0; i < 8; ++i) out[oq].prof[i] += w.prof[i];
    }
}

#if defined(AMC_TVG_BIG)   // second build of this file (tvg_e_big.hip): index arrays in global memory, own symbol names
#define tvg_e_kernel tvg_e_big_kernel
#define launch_tvg_e launch_tvg_e_big
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kTvgEWavesPerSimd, kTvgEWavesPerSimd))) void tvg_e_kernel(
    const TvgImage* __restrict__ imgs, const TvgPair* __restrict__ pairs, uint32_t npairs,
    const uint32_t* __restrict__ matches, const uint32_t* __restrict__ trial_tabs, TvgParams P,
    double* __restrict__ ws_all, uint32_t mcap, uint32_t* __restrict__ queue_head, TvgEState* __restrict__ estate,
    uint8_t* __restrict__ emask, TvgOut* __restrict__ out, uint8_t* __restrict__ out_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    Wave w;
    w.lane = lane;
    AMC_LDS char* lds = (AMC_LDS char*)smem + (size_t)wid * tvg_lds_per_wave_e(mcap);
    wave_carve(w, lds, mcap);
    w.rootscr = root_scratch_carve(lds, lds + tvg_lds_per_wave(mcap));
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wid;
    w.ws = ws_all + gw * tvg_ws_doubles_e(mcap);
#if defined(AMC_TVG_BIG)
    wave_carve_idx(w, ws_all + (size_t)gridDim.x * (blockDim.x >> 6) * tvg_ws_doubles_e(mcap) + gw * tvg_idx_doubles(mcap), mcap);
#endif
    w.masks = nullptr;
    w.stream = P.stream;
    w.stream_len = P.stream_len;
    w.err = P.stream_err;
    w.soff = 0;

    LODIAG_WAVE_START(gw);
    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(queue_head, 1u);
        q = __shfl(q, 0);
        if (q >= npairs) break;
        process_pair_e(w, q, imgs, pairs, matches, trial_tabs, P, estate, emask, out, out_mask);
    }
    LODIAG_WAVE_END(gw);
}


#if defined(AMC_TVG_BIG)
// (the diagnostics report belongs to the regular build)
#elif defined(AMC_TVG_LODIAG)
void tvg_diag_report_e() {
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lo_diag), sizeof h) != hipSuccess) return;
    if (h[48]) std::fprintf(stderr, "[amc tvg lodiag] minimal 5-point chunks %llu: cycles per chunk rows %.0f, elimination %.0f, finish %.0f, roots %.0f, models %.0f\n", h[48],
                            (double)h[49] / h[48], (double)h[50] / h[48], (double)h[51] / h[48], (double)h[52] / h[48], (double)h[53] / h[48]);
    std::fprintf(stderr, "[amc tvg lodiag tvg_diag_report_e] local 5-point solves %llu: cycles per solve ata %.0f jacobi %.0f build %.0f roots %.0f models %.0f | "
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
    unsigned long long z[64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lo_diag), z, sizeof z);
    lodiag_report_spans("tvg_e_kernel");
}
#else
void tvg_diag_report_e() {}
#endif

#if !defined(AMC_TVG_BIG)
size_t tvg_ws_doubles_e_host(uint32_t mcap) { return tvg_ws_doubles_e(mcap); }
size_t tvg_lds_bytes_e(uint32_t mcap, int waves) { return (size_t)waves * tvg_lds_per_wave_e(mcap); }
#else
size_t tvg_big_lds_bytes_e(int waves) { return (size_t)waves * tvg_lds_per_wave_e(0); }
size_t tvg_big_idx_doubles_host(uint32_t mcap) { return tvg_idx_doubles(mcap); }
#endif

hipError_t launch_tvg_e(const TvgImage* imgs, const TvgPair* pairs, uint32_t npairs, const uint32_t* matches,
                        const uint32_t* trial_tabs, const TvgParams& P, double* ws, uint32_t mcap, uint32_t num_waves,
                        int waves_per_block, uint32_t* queue_head, TvgEState* estate, uint8_t* emask, TvgOut* out,
                        uint8_t* out_mask, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    const uint32_t blocks = (num_waves + waves_per_block - 1) / waves_per_block;
    const size_t lds = (size_t)waves_per_block * tvg_lds_per_wave_e(mcap);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tvg_e_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = memset_async(queue_head, 0, sizeof(uint32_t), s);  // (the persistent waves pop pairs from it)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tvg_e_kernel, dim3(blocks), dim3(64 * waves_per_block), lds, s, imgs, pairs, npairs, matches,
                       trial_tabs, P, ws, mcap, queue_head, estate, emask, out, out_mask);
    return hipGetLastError();
}

}  // namespace amc
