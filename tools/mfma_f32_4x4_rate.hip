// Developer tool: issue rate of v_mfma_f32_4x4x1_16b_f32 per SIMD at 1 / 2 / 4 waves per SIMD, alone and with VALU work
// interleaved from the same wave - the budget k_psi32m.hip is planned against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_4x4_rate.hip -o build/mfma_rate32 && build/mfma_rate32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0)
#define FMA(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d))
#define MUL(q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[q]) : "v"(c))
#define DPPF(q) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(v[q]) : "v"(c), "v"(d))
// MODE 0: 8 MFMA   1: 8 x (MFMA, v_fma)   2: 8 x (MFMA, 2 v_fma)   3: 16 v_fma   4: 8 x (MFMA, dpp v_fmac)   5: 8 x (MFMA, v_mul)
//      6: 8 x (2 MFMA ... 1 v_fma) = 16 MFMA + 8 v_fma   7: 16 dpp v_fmac   8: dependent chain MFMA -> rcp_dpp -> mul -> MFMA (8 links)
//      9: the chain of 8 with 3 independent MFMAs and 2 v_fma between the links
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    f4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = (f4){0.f, 1.f, 2.f, 3.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f, c = 1.0f + threadIdx.x * 2e-6f, d = threadIdx.x * 1e-9f;
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = threadIdx.x + q;
    f4 t = acc[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) { MF(q); }
            if (MODE == 1) { MF(q); FMA(q); }
            if (MODE == 2) { MF(q); FMA(q); FMA(q + 8); }
            if (MODE == 3) { FMA(q); FMA(q + 8); }
            if (MODE == 4) { MF(q); DPPF(q); }
            if (MODE == 5) { MF(q); MUL(q); }
            if (MODE == 6) { MF(q); MF(q + 8); FMA(q); }
            if (MODE == 7) { DPPF(q); DPPF(q + 8); }
            if (MODE == 8 || MODE == 9) {
                const float p = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t[1]), 0x55, 0xf, 0xf, true));
                const float r = __builtin_amdgcn_rcpf(p);
                const float na = -t[1] * r;
                t = __builtin_amdgcn_mfma_f32_4x4x1f32(na, t[1], t, 0, 0, 0);
                t[1] += 2.0f;
                if (MODE == 9) { MF(q); MF(q + 8); MF((q + 4) & 7); FMA(q); FMA(q + 8); }
            }
        }
    }
    float s = t[0] + t[2];
    for (int q = 0; q < 16; ++q) s += acc[q][0] + acc[q][3];
    for (int q = 0; q < 16; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *out) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 64 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %d waves/SIMD: %.3f ms  %.2f cycles per group per SIMD @2.4GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / (8.0 * iters * wps));
    }
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("group = 1 mfma", out);
    run<1>("group = 1 mfma + 1 v_fma", out);
    run<2>("group = 1 mfma + 2 v_fma", out);
    run<3>("group = 2 v_fma", out);
    run<4>("group = 1 mfma + 1 dpp v_fmac", out);
    run<5>("group = 1 mfma + 1 v_mul", out);
    run<6>("group = 2 mfma + 1 v_fma", out);
    run<7>("group = 2 dpp v_fmac", out);
    run<8>("group = chain link mfma>rcp_dpp>mul>mfma", out);
    run<9>("group = chain link + 3 mfma + 2 v_fma", out);
    return 0;
}
