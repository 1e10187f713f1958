// ubench_mix.hip — do int8 MFMA and half-rate VALU overlap on one SIMD?  Per iteration: 4 MFMAs
// (32x32x32 i8, 4 independent accumulators) each followed by N independent v_max3_i32.
// If the pipes overlap: cycles/MFMA = max(~35, 4N [or 2N]); if they serialise: ~35 + 4N.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N, int OPK>
__global__ __launch_bounds__(1024) void k(int* out, int iters) {
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, 7};
    i32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (OPK == 0) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 3) & 7]), "v"(v[(u + 5) & 7]));
                if (OPK == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(v[(u + 3) & 7]));
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the match kernel's issue pattern: per unit 4 MFMAs into ONE accumulator (dependent chain, DEP=1) or
// into 4 different ones (DEP=0), then NV half-rate VALU; two accumulator sets alternate
template <int DEP, int NV>
__global__ __launch_bounds__(1024) void k2(int* out, int iters) {
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, 7};
    i32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int t = DEP ? u * 4 : u * 4 + m;
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NV; ++q)
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(v[(q + 3) & 7]), "v"(v[(q + 5) & 7]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int DEP, int NV>
void run2(int* out, int waves_per_simd) {
    const int CUS = 256, iters = 3000;
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<DEP, NV>), dim3(CUS), dim3(threads), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k2<DEP, NV>), dim3(CUS), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf_per_simd = (double)iters * 8 * waves_per_simd;
    printf("unit pattern dep=%d NV=%2d waves/SIMD=%d: %7.3f ms  %5.2f ns per MFMA per SIMD\n", DEP, NV, waves_per_simd, ms,
           ms * 1e6 / mf_per_simd);
}
template <int N, int OPK>
void run(int* out, int waves_per_simd) {
    const int CUS = 256, iters = 3000;
    const int threads = 256 * waves_per_simd;  // one block per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, OPK>), dim3(CUS), dim3(threads), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, OPK>), dim3(CUS), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf_per_simd = (double)iters * 4 * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / mf_per_simd;
    printf("%s N=%2d waves/SIMD=%d: %7.3f ms  (%5.2f ns per MFMA per SIMD)  cycles per (MFMA + N valu) per SIMD @2.4GHz = %6.1f   (serial model %3d, overlap model %3d)\n",
           OPK == 0 ? "v_max3_i32" : "v_add_u32 ", N, waves_per_simd, ms, ms * 1e6 / mf_per_simd, cyc, 35 + (OPK == 0 ? 4 : 2) * N, (OPK == 0 ? 4 : 2) * N > 35 ? (OPK == 0 ? 4 : 2) * N : 35);
}
int main() {
    int* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(int));
    for (int w = 1; w <= 3; ++w) {
        run2<1, 0>(out, w); run2<0, 0>(out, w); run2<1, 13>(out, w); run2<0, 13>(out, w); run2<1, 26>(out, w); run2<0, 26>(out, w);
    }
    for (int w = 1; w <= 4; ++w) {
        run<0, 0>(out, w); run<3, 0>(out, w); run<4, 0>(out, w); run<6, 0>(out, w); run<8, 0>(out, w); run<10, 0>(out, w); run<12, 0>(out, w); run<16, 0>(out, w);
        run<10, 1>(out, w); run<20, 1>(out, w);
    }
    return 0;
}
