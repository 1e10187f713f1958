// ubench_ops.hip — which gfx950 VALU ops issue at full rate (2 clk / wave64) vs half rate?
// Each kernel: 8 independent dependency chains per lane, 16 waves/CU, asm volatile so nothing folds.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CUS 256
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    int a[8]; int b = seed + threadIdx.x, c = seed * 3 + 1;
    long long a64[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (i + 1) + seed;
#pragma unroll
    for (int i = 0; i < 4; ++i) a64[i] = threadIdx.x * (i + 1) + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define A(n, s) if (OP == n) asm volatile(s : "+v"(a[i]) : "v"(b), "v"(c));
                A(0, "v_max_f32 %0, %0, %1")
                A(1, "v_min_f32 %0, %0, %1")
                A(2, "v_med3_f32 %0, %0, %1, %2")
                A(3, "v_max3_f32 %0, %0, %1, %2")
                A(4, "v_max_u32 %0, %0, %1")
                A(5, "v_and_b32 %0, %0, %1")
                A(6, "v_or_b32 %0, %0, %1")
                A(7, "v_lshlrev_b32 %0, 8, %0")
                A(8, "v_lshl_or_b32 %0, %0, 8, %1")
                A(9, "v_and_or_b32 %0, %0, %1, %2")
                A(10, "v_add3_u32 %0, %0, %1, %2")
                A(11, "v_mad_i32_i24 %0, %0, %1, %2")
                A(12, "v_mad_u32_u24 %0, %0, %1, %2")
                A(13, "v_sub_u32 %0, %0, %1")
                A(14, "v_cndmask_b32 %0, %0, %1, vcc")
                A(15, "v_bfe_u32 %0, %0, %1, %2")
                A(16, "v_perm_b32 %0, %0, %1, %2")
                A(17, "v_alignbit_b32 %0, %0, %1, %2")
                A(18, "v_pk_max_f16 %0, %0, %1")
                A(19, "v_max_f16 %0, %0, %1")
                A(20, "v_pk_add_u16 %0, %0, %1")
                A(21, "v_mul_lo_u32 %0, %0, %1")
                A(22, "v_mul_u32_u24 %0, %0, %1")
                A(23, "v_add_lshl_u32 %0, %0, %1, 8")
                A(24, "v_xad_u32 %0, %0, %1, %2")
                A(25, "v_mul_f32 %0, %0, %1")
                A(26, "v_add_f32 %0, %0, %1")
                A(27, "v_min3_f32 %0, %0, %1, %2")
                A(28, "v_mov_b32 %0, %1")
                A(29, "v_mad_u32_u16 %0, %0, %1, %2")
                A(30, "v_sad_u32 %0, %0, %1, %2")
                A(31, "v_pk_max_i16 %0, %0, %1")
                A(32, "v_cvt_f32_i32 %0, %0")
                A(33, "v_fma_f32 %0, %0, %1, %2")
                A(34, "v_max_i32 %0, %0, %1")
            }
            if (OP == 40) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a64[i]) : "v"(a64[(i + 1) & 3]));
            }
            if (OP == 41) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a64[i]) : "v"(a64[(i + 1) & 3]));
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += (int)a64[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int* out, int per_iter) {
    const int blocks = CUS * 4, iters = 2000;  // 16 waves/CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("%-18s FAILED\n", name); return; }
    double ops = (double)blocks * 256 * iters * per_iter;
    printf("%-18s %7.3f ms  %6.1f lane-instr/clk/CU @2.4GHz\n", name, ms, ops / (ms * 1e-3) / 2.4e9 / CUS);
}
int main() {
    int* out; (void)hipMalloc(&out, CUS * 4 * 256 * sizeof(int));
#define R(n, s) run<n>(s, out, 64);
    R(0,"v_max_f32") R(1,"v_min_f32") R(2,"v_med3_f32") R(3,"v_max3_f32") R(4,"v_max_u32") R(5,"v_and_b32") R(6,"v_or_b32")
    R(7,"v_lshlrev_b32") R(8,"v_lshl_or_b32") R(9,"v_and_or_b32") R(10,"v_add3_u32") R(11,"v_mad_i32_i24") R(12,"v_mad_u32_u24")
    R(13,"v_sub_u32") R(14,"v_cndmask_b32") R(15,"v_bfe_u32") R(16,"v_perm_b32") R(17,"v_alignbit_b32") R(18,"v_pk_max_f16")
    R(19,"v_max_f16") R(20,"v_pk_add_u16") R(21,"v_mul_lo_u32") R(22,"v_mul_u32_u24") R(23,"v_add_lshl_u32") R(24,"v_xad_u32")
    R(25,"v_mul_f32") R(26,"v_add_f32") R(27,"v_min3_f32") R(28,"v_mov_b32") R(29,"v_mad_u32_u16") R(30,"v_sad_u32") R(31,"v_pk_max_i16")
    R(32,"v_cvt_f32_i32") R(33,"v_fma_f32") R(34,"v_max_i32")
    run<40>("v_max_f64", out, 32); run<41>("v_pk_fma_f32", out, 32);
    return 0;
}
