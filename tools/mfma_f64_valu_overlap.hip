// Developer tool: do v_mfma_f64_16x16x4_f64 and fp64 vector instructions overlap on one SIMD - between different waves, and inside one
// wave's stream?  (Decides whether the covariance kinds' PHI build pays as a GEMM over row monomials with exp() in the epilogue:
// the quadratic form costs the same multiply-adds on either pipe, so the gain is exactly the overlap.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_valu_overlap.hip -o build/mfma_f64_overlap && build/mfma_f64_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0)
#define FMA(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d))
// role: 0 = 8 MFMA per iteration, 1 = 64 v_fma_f64 per iteration (the same 16 x 8 = 128 issue cycles if an MFMA is 16 and an fma 2... measured),
// 2 = interleaved in one wave: 1 MFMA then NF v_fma, 8 times
template <int NF>
__global__ __launch_bounds__(512) void k(double *out, int iters, int roleA, int roleB) {
    d4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = (d4){0., 1., 2., 3.};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, c = 1.0 + threadIdx.x * 2e-9, d = threadIdx.x * 1e-12;
    double v[16];
    for (int q = 0; q < 16; ++q) v[q] = threadIdx.x + q;
    const int role = (threadIdx.x >> 8) ? roleB : roleA;
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) MF(q);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int q = 0; q < NF; ++q) FMA((g * NF + q) & 15);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                MF(g);
#pragma unroll
                for (int q = 0; q < NF; ++q) FMA((g * NF + q) & 15);
            }
        }
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][3];
    for (int q = 0; q < 16; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NF>
float run(double *out, int threads, int roleA, int roleB) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NF>, dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NF>, dim3(256), dim3(threads), 0, 0, out, iters, roleA, roleB);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (8.0f * iters);   // cycles @2.4 GHz per group (1 MFMA and/or NF v_fma) of one wave
}
template <int NF>
void row() {
    double *out; (void)hipMalloc(&out, 256 * 512 * sizeof(double));
    printf("NF=%2d | 1w mfma %.1f | 1w fma %.1f | 2w mfma,mfma %.1f | 2w fma,fma %.1f | 2w mfma,fma %.1f | 1w interleaved %.1f | 2w interleaved %.1f\n", NF,
           run<NF>(out, 256, 0, 0), run<NF>(out, 256, 1, 1), run<NF>(out, 512, 0, 0), run<NF>(out, 512, 1, 1), run<NF>(out, 512, 0, 1),
           run<NF>(out, 256, 2, 2), run<NF>(out, 512, 2, 2));
    (void)hipFree(out);
}
int main() {
    printf("cycles @2.4GHz per group of one wave; a group = 1 v_mfma_f64_16x16x4 (role mfma), NF v_fma_f64 (role fma), or both (interleaved)\n");
    printf("(2w columns: per group with two waves on every SIMD, i.e. divide by 2 for the per-SIMD cost of one group)\n");
    row<2>(); row<4>(); row<8>(); row<16>();
    return 0;
}
