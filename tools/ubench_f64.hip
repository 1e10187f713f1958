// ubench_f64.hip — issue rate and latency of the FP64 VALU ops the verification kernels are made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
// For each op: independent chains (throughput) at 8 / 16 waves per CU, and one dependent chain (latency), and the
// IEEE division sequence the compiler emits for 1.0 / x.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double seed) {
    double a[8], b = seed + threadIdx.x * 1e-3, c = seed * 0.5 + 1.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double& x = a[CHAINS == 1 ? 0 : i % CHAINS];
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(b));
                if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
                if (OP == 4) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                if (OP == 5) { unsigned long long m; asm volatile("v_cmp_ge_f64 %0, %1, %2" : "=s"(m) : "v"(x), "v"(b)); asm volatile("" :: "s"(m)); }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the compiler's IEEE 1.0 / x (div_scale, rcp, 4 fma, mul, fma, div_fmas, div_fixup), CHAINS independent values
template <int CHAINS>
__global__ __launch_bounds__(256) void k_div(double* out, int iters, double seed) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.5 + threadIdx.x * 1e-4 * (i + 1) + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = 1.0 / a[i] + 0.75;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
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
    double* out; hipMalloc(&out, 256 * 8 * 1024 * sizeof(double));
    const int CUS = 256, iters = 2000;
    const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_max_f64", "v_cmp_ge_f64"};
    for (int wpc : {4, 8, 16}) {
        const int blocks = CUS * wpc / 4;
        printf("waves/CU=%d (%d per SIMD)\n", wpc, wpc / 4);
#define RUN(OP, CH) { float ms = time_ms([&] { hipLaunchKernelGGL((k_f64<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }); \
        double instr_per_simd = (double)iters * 64 * (wpc / 4.0); \
        printf("  %-14s chains=%d %8.3f ms  cycles/instr/SIMD@2.4GHz = %.2f\n", names[OP], CH, ms, ms * 1e-3 * 2.4e9 / instr_per_simd); }
        RUN(0, 8) RUN(1, 8) RUN(2, 8) RUN(3, 8) RUN(4, 8) RUN(5, 8)
        RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(3, 1)
        { float ms = time_ms([&] { hipLaunchKernelGGL(k_div<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
          printf("  1.0/x (IEEE) + add, 4 independent: %8.3f ms  cycles per division per SIMD = %.1f\n", ms, ms * 1e-3 * 2.4e9 / ((double)iters * 4 * (wpc / 4.0))); }
        { float ms = time_ms([&] { hipLaunchKernelGGL(k_div<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
          printf("  1.0/x (IEEE) + add, dependent:     %8.3f ms  cycles per division per SIMD = %.1f\n", ms, ms * 1e-3 * 2.4e9 / ((double)iters * 1 * (wpc / 4.0))); }
    }
    return 0;
}
