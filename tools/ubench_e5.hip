// ubench_e5.hip — where the minimal 5-point solve's time goes (the verification kernel's E RANSAC: 2 x 64 solves per
// pair, ~2 M cycles per 64).  Times the stages of estimate_e5_minimal (pycolmap_amd/csrc/tvg_math.h) cumulatively,
// one problem per lane, at the verification kernel's occupancy (2 waves per SIMD, 256 VGPRs).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ipycolmap_amd/csrc -Iinclude tools/ubench_e5.hip -o /tmp/ubench_e5
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

#include "tvg_math.h"

using namespace amc::tvg;

// STAGE 0: null space only; 1: + constraint matrix, elimination, determinant; 2: + roots; 3: everything
template <int STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_e5(const double* pts, double* out,
                                                                                       unsigned long long* cycles, int reps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double x1[5], y1[5], x2[5], y2[5];
    for (int i = 0; i < 5; ++i) {
        x1[i] = pts[(size_t)t * 20 + i];
        y1[i] = pts[(size_t)t * 20 + 5 + i];
        x2[i] = pts[(size_t)t * 20 + 10 + i];
        y2[i] = pts[(size_t)t * 20 + 15 + i];
    }
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        double A[5][9];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            A[i][0] = x2[i] * x1[i]; A[i][1] = x2[i] * y1[i]; A[i][2] = x2[i];
            A[i][3] = y2[i] * x1[i]; A[i][4] = y2[i] * y1[i]; A[i][5] = y2[i];
            A[i][6] = x1[i]; A[i][7] = y1[i]; A[i][8] = 1;
        }
        double ns[4][9];
        nullspace_reg<5>(A, ns);
        double nsp[36];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 9; ++j) nsp[k * 9 + j] = ns[k][j];
        if (STAGE == 0) {
            for (int k = 0; k < 36; ++k) acc += nsp[k];
        } else {
            E5Polys P;
            e5_build(nsp, P);
            if (STAGE == 1) {
                for (int k = 0; k < 11; ++k) acc += P.det[k];
            } else {
                double roots[10];
                const int nr = real_roots_t<10>(P.det, roots);
                if (STAGE == 2) {
                    for (int k = 0; k < nr; ++k) acc += roots[k];
                } else {
                    double models[90];
                    const int nm = e5_models(nsp, P, roots, nr, models);
                    for (int k = 0; k < nm * 9; ++k) acc += models[k];
                }
            }
        }
        x1[0] += acc * 1e-300;  // keep the iterations dependent
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[t] = acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(cycles, t1 - t0);
}

int main() {
    const int blocks = 512, threads = 256, n = blocks * threads, reps = 4;
    std::mt19937 rng(1);
    std::uniform_real_distribution<double> u(-0.6, 0.6);
    std::normal_distribution<double> g(0.0, 1e-3);
    std::vector<double> pts((size_t)n * 20);
    for (int t = 0; t < n; ++t) {
        // a real relative pose: points in front of both cameras, small rotation, unit baseline
        const double rx = u(rng) * 0.2, ry = u(rng) * 0.2, tx = 1.0, ty = u(rng), tz = u(rng);
        for (int i = 0; i < 5; ++i) {
            const double X = u(rng) * 4, Y = u(rng) * 3, Z = 5 + u(rng) * 3;
            const double X2 = X + ry * Z + tx, Y2 = Y - rx * Z + ty, Z2 = Z - ry * X + rx * Y + tz;
            pts[(size_t)t * 20 + i] = X / Z + g(rng);
            pts[(size_t)t * 20 + 5 + i] = Y / Z + g(rng);
            pts[(size_t)t * 20 + 10 + i] = X2 / Z2 + g(rng);
            pts[(size_t)t * 20 + 15 + i] = Y2 / Z2 + g(rng);
        }
    }
    double *d_pts, *d_out;
    unsigned long long* d_cyc;
    hipMalloc(&d_pts, pts.size() * 8);
    hipMalloc(&d_out, (size_t)n * 8);
    hipMalloc(&d_cyc, 8);
    hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice);
    const char* names[4] = {"null space (5 x 9 Gauss-Jordan)", "+ constraints, 10 x 20 elimination, det", "+ real roots (degree 10)",
                            "+ models"};
    double prev = 0.0;
    for (int s = 0; s < 4; ++s) {
        for (int pass = 0; pass < 2; ++pass) {
            hipMemset(d_cyc, 0, 8);
            if (s == 0) hipLaunchKernelGGL(k_e5<0>, dim3(blocks), dim3(threads), 0, 0, d_pts, d_out, d_cyc, reps);
            if (s == 1) hipLaunchKernelGGL(k_e5<1>, dim3(blocks), dim3(threads), 0, 0, d_pts, d_out, d_cyc, reps);
            if (s == 2) hipLaunchKernelGGL(k_e5<2>, dim3(blocks), dim3(threads), 0, 0, d_pts, d_out, d_cyc, reps);
            if (s == 3) hipLaunchKernelGGL(k_e5<3>, dim3(blocks), dim3(threads), 0, 0, d_pts, d_out, d_cyc, reps);
            hipDeviceSynchronize();
        }
        unsigned long long cyc = 0;
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)cyc / ((double)n / 64 * reps);
        std::printf("%-48s %9.0f cycles per 64 solves (wave, 2 waves/SIMD)   stage alone %9.0f\n", names[s], per, per - prev);
        prev = per;
    }
    return 0;
}
