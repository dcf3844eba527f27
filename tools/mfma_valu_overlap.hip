// Developer tool: do v_mfma_f32_4x4x1_16b_f32 and vector instructions of DIFFERENT waves of one SIMD overlap, and what does a wave pay
// for switching between runs of the two?  (k_psi32m.hip's instruction stream is ~545 MFMA + ~480 VALU per sample.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o build/mfma_overlap && build/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0)
#define FMA(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d))
// role of a wave: 0 = 16 MFMA per iteration, 1 = 16 v_fma per iteration, 2 = runs: RUN mfma then RUN v_fma (16 + 16 per iteration in all)
template <int RUN>
__global__ __launch_bounds__(512) void k(float *out, int iters, int roleA, int roleB) {
    f4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = (f4){0.f, 1.f, 2.f, 3.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f, c = 1.0f + threadIdx.x * 2e-6f, d = threadIdx.x * 1e-9f;
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = threadIdx.x + q;
    const int role = (threadIdx.x >> 8) ? roleB : roleA;     // waves 0-3 / 4-7: one of each on every SIMD
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) MF(q);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) FMA(q);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16 / RUN; ++g) {
#pragma unroll
                for (int q = 0; q < RUN; ++q) MF(g * RUN + q);
#pragma unroll
                for (int q = 0; q < RUN; ++q) FMA(g * RUN + q);
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 16; ++q) s += acc[q][0] + acc[q][3] + v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int RUN>
float run(float *out, int threads, int roleA, int roleB) {
    const int iters = 10000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<RUN>, dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<RUN>, dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (16.0f * iters);   // cycles @2.4 GHz per "slot" (one of the 16 per iteration) of one wave
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    printf("cycles @2.4GHz per instruction slot of a wave (16 per iteration)\n");
    printf("1 wave/SIMD  mfma only            : %.2f\n", run<1>(out, 256, 0, 0));
    printf("1 wave/SIMD  v_fma only           : %.2f\n", run<1>(out, 256, 1, 1));
    printf("2 waves/SIMD mfma | mfma          : %.2f\n", run<1>(out, 512, 0, 0));
    printf("2 waves/SIMD v_fma | v_fma        : %.2f\n", run<1>(out, 512, 1, 1));
    printf("2 waves/SIMD mfma | v_fma         : %.2f   (max of the two alone = full overlap, sum = none)\n", run<1>(out, 512, 0, 1));
    printf("per (mfma + v_fma) PAIR of a wave, runs of R mfma then R v_fma:\n");
    printf("1 wave/SIMD  R=1 : %.2f   R=2 : %.2f   R=4 : %.2f   R=8 : %.2f   R=16 : %.2f\n", run<1>(out, 256, 2, 2), run<2>(out, 256, 2, 2),
           run<4>(out, 256, 2, 2), run<8>(out, 256, 2, 2), run<16>(out, 256, 2, 2));
    printf("2 waves/SIMD R=1 : %.2f   R=2 : %.2f   R=4 : %.2f   R=8 : %.2f   R=16 : %.2f\n", run<1>(out, 512, 2, 2), run<2>(out, 512, 2, 2),
           run<4>(out, 512, 2, 2), run<8>(out, 512, 2, 2), run<16>(out, 512, 2, 2));
    return 0;
}
