// Microbenchmark: does the v_mfma_f64_16x16x4_f64 issue rate depend on the operand pattern of a GEMM K loop?
//   mode 0: 16 accumulators, one A and one B register pair for all (tools/mfma_f64_bench.hip's pattern)
//   mode 1: 8 accumulators (64 VGPRs), 4 A x 2 B register pairs, the k_tgemm / k_syrk burst
//   mode 2: as 1, accumulators in AGPRs
//   mode 3: as 1, A/B pairs rewritten (v_mov) between bursts like freshly loaded fragments
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_operands.hip -o tools/mfma_f64_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define MFV(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(double *out, int iters) {
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.0 + threadIdx.x * 1e-4, b1 = b0 + 1;
    d4_t z = {0, 0, 0, 0};
    d4_t c0 = z, c1 = z, c2 = z, c3 = z, c4 = z, c5 = z, c6 = z, c7 = z, c8 = z, c9 = z, c10 = z, c11 = z, c12 = z, c13 = z, c14 = z, c15 = z;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            MFV(c0, a0, b0); MFV(c1, a0, b0); MFV(c2, a0, b0); MFV(c3, a0, b0); MFV(c4, a0, b0); MFV(c5, a0, b0); MFV(c6, a0, b0); MFV(c7, a0, b0);
            MFV(c8, a0, b0); MFV(c9, a0, b0); MFV(c10, a0, b0); MFV(c11, a0, b0); MFV(c12, a0, b0); MFV(c13, a0, b0); MFV(c14, a0, b0); MFV(c15, a0, b0);
        } else if (MODE == 1 || MODE == 3) {
            MFV(c0, a0, b0); MFV(c1, a1, b0); MFV(c2, a2, b0); MFV(c3, a3, b0); MFV(c4, a0, b1); MFV(c5, a1, b1); MFV(c6, a2, b1); MFV(c7, a3, b1);
            if (MODE == 3) asm volatile("v_mov_b64 %0, %0\n v_mov_b64 %1, %1\n v_mov_b64 %2, %2\n v_mov_b64 %3, %3\n v_mov_b64 %4, %4\n v_mov_b64 %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1));
            MFV(c0, a0, b0); MFV(c1, a1, b0); MFV(c2, a2, b0); MFV(c3, a3, b0); MFV(c4, a0, b1); MFV(c5, a1, b1); MFV(c6, a2, b1); MFV(c7, a3, b1);
            if (MODE == 3) asm volatile("v_mov_b64 %0, %0\n v_mov_b64 %1, %1\n v_mov_b64 %2, %2\n v_mov_b64 %3, %3\n v_mov_b64 %4, %4\n v_mov_b64 %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1));
        } else {
            MFA(c0, a0, b0); MFA(c1, a1, b0); MFA(c2, a2, b0); MFA(c3, a3, b0); MFA(c4, a0, b1); MFA(c5, a1, b1); MFA(c6, a2, b1); MFA(c7, a3, b1);
            MFA(c0, a0, b0); MFA(c1, a1, b0); MFA(c2, a2, b0); MFA(c3, a3, b0); MFA(c4, a0, b1); MFA(c5, a1, b1); MFA(c6, a2, b1); MFA(c7, a3, b1);
        }
    }
    d4_t s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE>
void run(double *out, int iters, const char *what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256, 512}) {   // 2 and 4 waves per SIMD
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", what, blocks / 128, blocks * 8.0 * iters * 16 * 2048.0 / ms * 1e-9, ms);
    }
}
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    double *out; (void)hipMalloc(&out, 512 * 512 * sizeof(double));
    run<0>(out, iters, "16 acc, one A/B pair");
    run<1>(out, iters, "8 acc, 4 A x 2 B pairs (GEMM burst)");
    run<2>(out, iters, "8 acc in AGPRs, 4 A x 2 B pairs");
    run<3>(out, iters, "8 acc, 4 A x 2 B pairs rewritten between bursts");
    return 0;
}
