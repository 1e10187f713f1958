// ubench_valu.hip — issue-rate microbenchmark for the integer VALU ops the match epilogue uses,
// and the int8 MFMA they run beside.  Build: hipcc --offload-arch=gfx950 -O3 profiles/r01/recipes/ubench_valu.hip
// Prints lane-ops per clock per CU at the measured wall time (assumes 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k_valu(int* out, int iters, int seed) {
    int a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (i + 1) + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 12, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(long long*)&a[i & 6]) : "v"(*(long long*)&a[(i & 6)]) );
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma(int* out, int iters, int valu_per_mfma) {
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, 7};
    i32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    int v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        for (int u = 0; u < valu_per_mfma; ++u) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
        for (int u = 0; u < valu_per_mfma; ++u) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc2, 0, 0, 0);
        for (int u = 0; u < valu_per_mfma; ++u) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
        acc3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc3, 0, 0, 0);
        for (int u = 0; u < valu_per_mfma; ++u) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int* out; hipMalloc(&out, 256 * 8 * 1024 * sizeof(int));
    const int CUS = 256, iters = 4000;
    const char* names[] = {"v_max_i32", "v_med3_i32", "v_max3_i32", "v_lshl_add_u32", "v_pk_max_i16", "v_add_u32", "v_fma_f32", "v_pk_fma_f32"};
    for (int wpc = 4; wpc <= 32; wpc *= 2) {  // waves per CU: blocks of 256 threads (4 waves)
        const int blocks = CUS * wpc / 4;
        printf("waves/CU=%d\n", wpc);
#define RUN(OP) { float ms = time_ms([&] { hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1); }); \
        double ops = (double)blocks * 256 * iters * 64; printf("  %-16s %8.3f ms  %6.1f lane-ops/clk/CU @2.4GHz  (%.2f Tlaneop/s)\n", names[OP], ms, ops / (ms * 1e-3) / 2.4e9 / CUS, ops / (ms * 1e-3) / 1e12); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    }
    for (int vpm : {0, 4, 8, 12, 16, 24, 32}) {
        for (int wpc : {4, 8}) {
            const int blocks = CUS * wpc / 4, it2 = 2000;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, it2, vpm); });
            double mf = (double)blocks * 4 * it2 * 4;  // wave-level MFMAs
            double tops = mf * 65536 / (ms * 1e-3) / 1e12;
            double valu = mf * vpm * 64 / (ms * 1e-3) / 2.4e9 / CUS;
            printf("mfma_i32_32x32x32_i8 + %2d v_max3/mfma, waves/CU=%d: %7.3f ms  %7.1f TOP/s  valu %5.1f lane-ops/clk/CU  cyc/mfma/SIMD@2.4=%.1f\n",
                   vpm, wpc, ms, tops, valu, (ms * 1e-3) * 2.4e9 / (it2 * 4.0 * wpc / 4));
        }
    }
    return 0;
}
