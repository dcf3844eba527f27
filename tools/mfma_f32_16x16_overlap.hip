// Developer tool: do the LARGE fp32 matrix instructions (v_mfma_f32_16x16x4_f32: 8 passes; v_mfma_f32_32x32x2_f32: 16 passes) overlap with
// fp32 vector work (v_pk_fma_f32 / v_fma_f32) on one SIMD - across waves and inside one wave's stream?  (The 2-pass 4x4x1 shape does
// not, tools/mfma_valu_overlap.hip.)  Decides whether config 5's pair kernels can form A = I + sum_k psi_k r_k r_k' as a K = 20 product
// on the matrix pipe while the vector ALU factorises the previous pairs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_16x16_overlap.hip -o build/mfma_f32_16x16_overlap && build/mfma_f32_16x16_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define PK(q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d))
// SHAPE 0: 16x16x4 (1024 multiply-adds), 1: 32x32x2 (2048).  role 0 = 8 MFMA per iteration, 1 = 8*NF v_pk_fma_f32, 2 = 1 MFMA then NF pk_fma, 8 times
template <int SHAPE, int NF>
__global__ __launch_bounds__(512) void k(float *out, int iters, int roleA, int roleB) {
    f4 acc4[8];
    f16v acc16[4];
    for (int q = 0; q < 8; ++q) acc4[q] = (f4){0.f, 1.f, 2.f, 3.f};
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) acc16[q][e] = (float)e;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    f2 c = {1.0f + threadIdx.x * 2e-6f, 1.0f}, d = {threadIdx.x * 1e-9f, 0.f};
    f2 v[16];
    for (int q = 0; q < 16; ++q) v[q] = (f2){(float)threadIdx.x + q, (float)q};
    const int role = (threadIdx.x >> 8) ? roleB : roleA;
    auto mf = [&](int g) {
        if (SHAPE == 0) acc4[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[g], 0, 0, 0);
        else acc16[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[g & 3], 0, 0, 0);
    };
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) mf(g);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int q = 0; q < NF; ++q) PK((g * NF + q) & 15);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                mf(g);
#pragma unroll
                for (int q = 0; q < NF; ++q) PK((g * NF + q) & 15);
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 8; ++q) s += acc4[q][0] + acc4[q][3];
    for (int q = 0; q < 4; ++q) s += acc16[q][0] + acc16[q][15];
    for (int q = 0; q < 16; ++q) s += v[q][0] + v[q][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int NF>
float run(float *out, int threads, int roleA, int roleB) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NF>), dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NF>), dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (8.0f * iters);   // cycles @2.4 GHz per group (1 MFMA and/or NF v_pk_fma) of one wave
}
template <int SHAPE, int NF>
void row(float *out) {
    printf("%s NF=%2d | 1w mfma %.1f | 1w pk %.1f | 2w mfma,mfma %.1f | 2w pk,pk %.1f | 2w mfma,pk %.1f | 1w interleaved %.1f | 2w interleaved %.1f\n",
           SHAPE ? "32x32x2" : "16x16x4", NF, run<SHAPE, NF>(out, 256, 0, 0), run<SHAPE, NF>(out, 256, 1, 1), run<SHAPE, NF>(out, 512, 0, 0),
           run<SHAPE, NF>(out, 512, 1, 1), run<SHAPE, NF>(out, 512, 0, 1), run<SHAPE, NF>(out, 256, 2, 2), run<SHAPE, NF>(out, 512, 2, 2));
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    printf("cycles @2.4GHz per group of one wave; a group = 1 MFMA (role mfma), NF v_pk_fma_f32 (role pk), or both (interleaved)\n");
    printf("(2w columns: two waves on every SIMD; 'mfma,pk' = one wave of each kind: max(alone) = full overlap, sum = none)\n");
    row<0, 4>(out); row<0, 8>(out); row<0, 16>(out);
    row<1, 8>(out); row<1, 16>(out); row<1, 32>(out);
    return 0;
}
