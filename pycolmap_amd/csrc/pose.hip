// pose.hip — relative pose of verified image pairs: COLMAP 3.9.1 EstimateTwoViewGeometryPose
// (TwoViewGeometryOptions.compute_relative_pose, /root/reference/pycolmap/estimators/
// two_view_geometry.h:59-62; estimate_two_view_geometry_pose, :153-159; the cam2_from_cam1 of
// essential_matrix_estimation, /root/reference/pycolmap/estimators/essential_matrix.h:63-83).
//
// One wave per pair.
//   1. lane 0 decomposes the model into its candidate poses (E: 2 rotations x 2 translation signs;
//      H: Malis-Vargas, 1 or 4 candidates) and parks them in LDS;
//   2. per candidate the wave triangulates the inlier correspondences, one per lane (4 x 4 DLT,
//      Jacobi eigen-decomposition in registers), and counts the points in front of both cameras
//      with a ballot; the last candidate with the largest count wins (COLMAP's >=);
//   3. the winner's points are triangulated once more and the cosine of each triangulation angle is
//      written to the pair's slice of a global workspace;
//   4. the median angle belongs to the middle element(s) of the cosines ordered by |c|: a 63-step
//      bitwise search on the keys, counting with ballots.  The host takes acos of the one or two
//      selected cosines (libm, as the acos table of the matcher: pose_math.h).
// FP64 throughout, no contraction; bit-exact with oracle/tvg_oracle.cc (tests/test_pose_gpu.py).
// Bytes per pair are tiny (16 B per inlier match + 2 x 16 B keypoints, read five times from L2):
// the kernel is bound by the FP64 Jacobi sweeps, ~5 triangulations per inlier.
#include <hip/hip_runtime.h>

#include "amc_internal.h"
#include "camera_math.h"
#include "pose_math.h"

namespace amc {
namespace {

using namespace tvg;

__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct Corr { double x1, y1, x2, y2; };

// What load_corr needs of one image, read once per pair (wave-uniform)
struct LiftSrc {
    const float* kp;
    const double* kp64;
    const double* kpn;   // lifted keypoints of a camera with distortion parameters, else nullptr
    double fx, fy, cx, cy;
};
__device__ __forceinline__ LiftSrc lift_src(const TvgImage* im) {
    LiftSrc s;
    s.kp = im->kp; s.kp64 = im->kp64; s.kpn = im->kpn;
    const int nf = cam::num_focal(im->cam.model_id);
    s.fx = im->cam.params[0]; s.fy = im->cam.params[nf - 1];
    s.cx = im->cam.params[nf]; s.cy = im->cam.params[nf + 1];
    return s;
}
__device__ __forceinline__ void lift_point(const LiftSrc& s, uint32_t i, double& u, double& v) {
    if (s.kpn) {
        u = s.kpn[2 * (size_t)i];
        v = s.kpn[2 * (size_t)i + 1];
        return;
    }
    const double X = s.kp64 ? s.kp64[2 * (size_t)i] : (double)s.kp[2 * (size_t)i];
    const double Y = s.kp64 ? s.kp64[2 * (size_t)i + 1] : (double)s.kp[2 * (size_t)i + 1];
    u = (X - s.cx) / s.fx;
    v = (Y - s.cy) / s.fy;
}
// k-th correspondence of the pair in camera coordinates (Camera::CamFromImg of both cameras)
__device__ __forceinline__ Corr load_corr(const LiftSrc& im1, const LiftSrc& im2, const uint32_t* mm, int k) {
    Corr c;
    lift_point(im1, mm[2 * (size_t)k], c.x1, c.y1);
    lift_point(im2, mm[2 * (size_t)k + 1], c.x2, c.y2);
    return c;
}

// the cosine with rank `rank` (0-based) among n cosines ordered by |c|, largest first
__device__ double select_cosine(const double* cs, uint32_t n, uint32_t rank, int lane) {
    uint64_t K = 0;
    for (int bit = 62; bit >= 0; --bit) {
        const uint64_t T = K | (1ull << bit);
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const bool ge = i < n && cosine_key(cs[i]) >= T;
            cnt += (uint32_t)__popcll(__ballot(ge));
        }
        if (cnt >= rank + 1) K = T;
    }
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const double c = i < n ? cs[i] : 0.0;
        const unsigned long long hit = __ballot(i < n && cosine_key(c) == K);
        if (hit) {
            const int src = __builtin_ctzll(hit);
            union { double d; int w[2]; } b;
            b.d = c;
            b.w[0] = __builtin_amdgcn_readlane(b.w[0], src);
            b.w[1] = __builtin_amdgcn_readlane(b.w[1], src);
            return b.d;
        }
    }
    return 0.0;
}

__global__ __launch_bounds__(64) void pose_kernel(const TvgImage* __restrict__ imgs, const PosePair* __restrict__ pairs,
                                                  uint32_t npairs, const uint32_t* __restrict__ matches,
                                                  const uint8_t* __restrict__ mask_all,
                                                  double* __restrict__ cosine_ws, PoseOut* __restrict__ out) {
    __shared__ PoseCands cands;
    const uint32_t p = blockIdx.x;
    if (p >= npairs) return;
    const int lane = threadIdx.x;
    const PosePair& pr = pairs[p];
    const int config = pr.config;
    PoseOut* o = out + p;
    const bool has_geometry = config == AMC_TVG_CALIBRATED || config == AMC_TVG_UNCALIBRATED ||
                              config == AMC_TVG_PLANAR || config == AMC_TVG_PANORAMIC ||
                              config == AMC_TVG_PLANAR_OR_PANORAMIC;
    if (!has_geometry) {  // EstimateTwoViewGeometryPose returns false: the geometry keeps its defaults
        if (lane == 0) {
            o->ok = 0; o->t_is_zero = 1; o->num_points3D = 0; o->pad = 0;
            for (int i = 0; i < 9; ++i) o->R[i] = (i % 4 == 0) ? 1.0 : 0.0;
            for (int i = 0; i < 3; ++i) o->t[i] = 0.0;
            o->q[0] = 1.0; o->q[1] = 0.0; o->q[2] = 0.0; o->q[3] = 0.0;
            o->cmed[0] = 1.0; o->cmed[1] = 1.0;
        }
        return;
    }
    const LiftSrc im1 = lift_src(imgs + pr.slot1), im2 = lift_src(imgs + pr.slot2);
    const uint32_t* mm = matches + 2 * pr.match_off;
    // inlier matches: all M rows, or (behind amc_verify_pairs) the rows of the pair's mask that are set
    const uint8_t* mask = mask_all ? mask_all + pr.mask_off : nullptr;
    const int M = (int)pr.M;
    if (lane == 0) {
        PoseCands c;
        double m9[9];
        if (config == AMC_TVG_CALIBRATED || config == AMC_TVG_UNCALIBRATED) {
            for (int i = 0; i < 9; ++i) m9[i] = pr.E[i];
            pose_candidates_E(m9, c);
        } else {
            double K1[9], K2[9];
            for (int i = 0; i < 9; ++i) m9[i] = pr.H[i];
            cam::calibration_matrix(imgs[pr.slot1].cam.model_id, imgs[pr.slot1].cam.params, K1);
            cam::calibration_matrix(imgs[pr.slot2].cam.model_id, imgs[pr.slot2].cam.params, K2);
            pose_candidates_H(m9, K1, K2, c);
        }
        cands = c;
    }
    __syncthreads();
    const int ncand = cands.n;
    int best = 0;
    uint32_t best_count = 0;
    for (int k = 0; k < ncand; ++k) {
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = cands.R[k][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = cands.t[k][i];
        const CheiralityBounds b = cheirality_bounds(R, t);
        uint32_t cnt = 0;
        for (int base = 0; base < M; base += 64) {
            const int i = base + lane;
            bool ok = false;
            if (i < M && (!mask || mask[i])) {
                const Corr c = load_corr(im1, im2, mm, i);
                double X[3];
                ok = cheirality_point(R, t, b, c.x1, c.y1, c.x2, c.y2, X);
            }
            cnt += (uint32_t)__popcll(__ballot(ok));
        }
        if (cnt >= best_count) { best = k; best_count = cnt; }
    }
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cands.R[best][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cands.t[best][i];
    const CheiralityBounds b = cheirality_bounds(R, t);
    double c2[3], baseline2;
    second_centre(R, t, c2, &baseline2);
    double* cs = cosine_ws + pr.ws_off;
    uint32_t n = 0;
    for (int base = 0; base < M; base += 64) {
        const int i = base + lane;
        bool ok = false;
        double cosine = 0.0;
        if (i < M && (!mask || mask[i])) {
            const Corr c = load_corr(im1, im2, mm, i);
            double X[3];
            ok = cheirality_point(R, t, b, c.x1, c.y1, c.x2, c.y2, X);
            if (ok) cosine = triangulation_cosine(c2, baseline2, X);
        }
        const unsigned long long bal = __ballot(ok);
        if (ok) cs[n + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = cosine;
        n += (uint32_t)__popcll(bal);
    }
    wave_mem_sync();
    double cmed0 = 1.0, cmed1 = 1.0;
    if (n > 0) {
        cmed0 = select_cosine(cs, n, n / 2, lane);
        if (n % 2 == 0) cmed1 = select_cosine(cs, n, n / 2 - 1, lane);
    }
    if (lane == 0) {
        o->ok = 1;
        o->t_is_zero = vec3_norm(t) == 0.0 ? 1 : 0;
        o->num_points3D = n;
        o->pad = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o->R[i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) o->t[i] = t[i];
        double q[4];
        rotation_to_quaternion(R, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) o->q[i] = q[i];
        o->cmed[0] = cmed0;
        o->cmed[1] = cmed1;
    }
}

// PoseFromHomographyMatrix on given points (pycolmap.homography_decomposition): one wave.  in: H, K1, K2 (27 doubles);
// out: R (9), t (3), n (3), then the number of points3D as a double.  points3D: the winner's points, in input order.
__global__ __launch_bounds__(64) void homography_decomposition_kernel(const double* __restrict__ in, const double* __restrict__ p1,
                                                                      const double* __restrict__ p2, uint32_t n,
                                                                      double* __restrict__ out, double* __restrict__ points3D) {
    __shared__ PoseCands cands;
    const int lane = threadIdx.x;
    if (lane == 0) {
        PoseCands c;
        double H[9], K1[9], K2[9];
        for (int i = 0; i < 9; ++i) { H[i] = in[i]; K1[i] = in[9 + i]; K2[i] = in[18 + i]; }
        pose_candidates_H<true>(H, K1, K2, c);
        cands = c;
    }
    __syncthreads();
    const int ncand = cands.n;
    int best = 0;
    uint32_t best_count = 0;
    for (int k = 0; k < ncand; ++k) {
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = cands.R[k][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = cands.t[k][i];
        const CheiralityBounds b = cheirality_bounds(R, t);
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            bool ok = false;
            if (i < n) {
                double X[3];
                ok = cheirality_point(R, t, b, p1[2 * (size_t)i], p1[2 * (size_t)i + 1], p2[2 * (size_t)i], p2[2 * (size_t)i + 1], X);
            }
            cnt += (uint32_t)__popcll(__ballot(ok));
        }
        if (cnt >= best_count) { best = k; best_count = cnt; }  // CheckCheirality's >=: later candidates win ties
    }
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cands.R[best][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cands.t[best][i];
    const CheiralityBounds b = cheirality_bounds(R, t);
    uint32_t m = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        bool ok = false;
        double X[3] = {0.0, 0.0, 0.0};
        if (i < n) ok = cheirality_point(R, t, b, p1[2 * (size_t)i], p1[2 * (size_t)i + 1], p2[2 * (size_t)i], p2[2 * (size_t)i + 1], X);
        const unsigned long long bal = __ballot(ok);
        if (ok) {
            double* dst = points3D + 3 * (size_t)(m + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)));
            dst[0] = X[0]; dst[1] = X[1]; dst[2] = X[2];
        }
        m += (uint32_t)__popcll(bal);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { out[9 + i] = t[i]; out[12 + i] = cands.nrm[best][i]; }
        out[15] = (double)m;
    }
}

// amc_verify_pairs' results in the caller's layout: one wave per pair copies the record without its counters
// (amc_tvg: 70 dwords of the 512-byte TvgOut) and the pair's mask bytes from their 128-byte aligned slot to the
// input's CSR offset, and adds the pair's work counters to the call's sums (one atomic per counter and workgroup).
// trivial_below: a pair with fewer matches was never handed to a kernel (mode 0: EstimateTwoViewGeometry returns DEGENERATE
// for it before it looks at a point) - its record is the zeroed one with that config, its mask bytes are zero.
__global__ __launch_bounds__(256) void pack_verify_kernel(const TvgOut* __restrict__ out, const TvgPair* __restrict__ tp,
                                                          uint32_t npairs, const uint8_t* __restrict__ mask_src,
                                                          const uint64_t* __restrict__ moff, amc_tvg* __restrict__ tvg_dst,
                                                          uint8_t* __restrict__ mask_dst, unsigned long long* __restrict__ work,
                                                          uint32_t trivial_below) {
    __shared__ unsigned long long s_work[12];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (tid < 12) s_work[tid] = 0ull;
    __syncthreads();
    const uint32_t p = blockIdx.x * 4 + (tid >> 6);
    if (p < npairs) {
        static_assert(sizeof(amc_tvg) % 4 == 0, "amc_tvg is copied in dwords");
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&out[p].g);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tvg_dst + p);
        const uint32_t M = tp[p].M;
        const bool trivial = M < trivial_below;
        static_assert(offsetof(amc_tvg, config) == 0, "the record's first dword is its config");
        for (uint32_t i = lane; i < sizeof(amc_tvg) / 4; i += 64) dst[i] = (trivial && i == 0) ? (uint32_t)AMC_TVG_DEGENERATE : src[i];
        const uint8_t* ms = mask_src + tp[p].mask_off;
        uint8_t* md = mask_dst + moff[p];
        for (uint32_t i = lane; i < M; i += 64) md[i] = trivial ? (uint8_t)0 : ms[i];
        if (lane < 12) atomicAdd(&s_work[lane], out[p].work[lane]);
    }
    __syncthreads();
    if (tid < 12 && s_work[tid]) atomicAdd(&work[tid], s_work[tid]);
}

}  // namespace

hipError_t launch_pack_verify(const TvgOut* out, const TvgPair* tp, uint32_t npairs, const uint8_t* mask_src,
                              const uint64_t* moff, amc_tvg* tvg_dst, uint8_t* mask_dst, unsigned long long* work,
                              int32_t trivial_below, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_verify_kernel, dim3((npairs + 3) / 4), dim3(256), 0, s, out, tp, npairs, mask_src, moff, tvg_dst,
                       mask_dst, work, (uint32_t)(trivial_below > 0 ? trivial_below : 0));
    return hipGetLastError();
}

hipError_t launch_homography_decomposition(const double* in27, const double* p1, const double* p2, uint32_t n, double* out16,
                                           double* points3D, hipStream_t s) {
    hipLaunchKernelGGL(homography_decomposition_kernel, dim3(1), dim3(64), 0, s, in27, p1, p2, n, out16, points3D);
    return hipGetLastError();
}

hipError_t launch_pose(const TvgImage* imgs, const PosePair* pairs, uint32_t npairs, const uint32_t* matches,
                       const uint8_t* mask, double* cosine_ws, PoseOut* out, hipStream_t s) {
    if (npairs == 0) return hipSuccess;
    hipLaunchKernelGGL(pose_kernel, dim3(npairs), dim3(64), 0, s, imgs, pairs, npairs, matches, mask, cosine_ws, out);
    return hipGetLastError();
}

}  // namespace amc
