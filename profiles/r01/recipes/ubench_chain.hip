// ubench_chain.hip — cost of the kernel's real issue pattern: groups of 4 MFMAs (dependent K-chain
// on one accumulator, or NCH independent chains interleaved) followed by a block of NV half-rate
// VALU ops that form ONE dependent chain (like the top-2 state) or 2 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NCH, int NV, int VCH>
__global__ __launch_bounds__(1024) void k(int* out, int iters) {
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, 7};
    i32x16 acc[NCH];
    for (int i = 0; i < NCH; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = threadIdx.x + i;
    const int w = threadIdx.x * 7;
    for (int it = 0; it < iters; ++it) {
        // NCH units' worth of MFMAs: 4 dependent K-slices each, chains interleaved
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NV * NCH; ++u)
            asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u % VCH]) : "v"(w), "v"(w));
        __builtin_amdgcn_sched_barrier(0);
    }
    int s = 0;
    for (int i = 0; i < NCH; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, int NV, int VCH>
void run(int* out, int wps) {
    const int CUS = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCH, NV, VCH>), dim3(CUS), dim3(256 * wps), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, NV, VCH>), dim3(CUS), dim3(256 * wps), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 4 * NCH * wps;
    printf("chains=%d valu/unit=%2d valu-chains=%d waves/SIMD=%d : %7.3f ms  %6.1f clk per MFMA per SIMD @2.4GHz\n",
           NCH, NV, VCH, wps, ms, ms * 1e-3 * 2.4e9 / mf);
}
int main() {
    int* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(int));
    for (int w = 1; w <= 3; ++w) {
        run<1, 0, 1>(out, w); run<2, 0, 1>(out, w);
        run<1, 24, 1>(out, w); run<1, 24, 2>(out, w); run<1, 24, 4>(out, w);
        run<2, 24, 1>(out, w); run<2, 24, 2>(out, w); run<2, 24, 4>(out, w);
    }
    return 0;
}
