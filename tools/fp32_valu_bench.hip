// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (independent accumulators pinned by inline asm) at 1, 2, 4
// waves per SIMD — the peak the fp32 per-pair kernels (k_psi32.hip) are priced against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp32_valu_bench.hip -o tools/fp32_valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
// mode 0: v_fma_f32   1: v_pk_fma_f32   2: v_pk_fma_f32 with an op_sel broadcast   3: v_pk_mul_f32   4: alternating fma / pk_fma
__global__ __launch_bounds__(1024) void k(float *out, int iters, int mode) {
    float a = 1.0000001f, b = 1e-9f;
    f2 a2 = {1.0000001f, 0.9999999f}, b2 = {1e-9f, 2e-9f};
    float acc[16];
    f2 acc2[16];
    for (int i = 0; i < 16; ++i) { acc[i] = threadIdx.x + i; acc2[i].x = threadIdx.x + i; acc2[i].y = i; }
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    } else if (mode == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        }
    } else if (mode == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        }
    } else if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(a2));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    float *out; (void)hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel", "v_pk_mul_f32", "fma/pk_fma alternating"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode)
        for (int wps : {1, 2, 4}) {
            const int threads = 64 * 4 * wps;              // one workgroup per CU
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, mode); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr = 16.0 * iters;             // per wave
            const double flop_per_instr = (mode == 0) ? 128 : (mode == 3) ? 128 : (mode == 4) ? 192 : 256;
            const double waves = 256.0 * 4 * wps;
            printf("%-26s %d waves/SIMD: %.3f ms  %.2f cycles/instr/SIMD-slot @2.4GHz  %.1f TFLOP/s\n", names[mode], wps, ms,
                   ms * 1e-3 * 2.4e9 / (instr * wps), instr * waves * flop_per_instr / (ms * 1e-3) / 1e12);
        }
    return 0;
}
