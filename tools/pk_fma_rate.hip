// Developer tool: v_pk_fma_f32 (with and without op_sel broadcast) and a 2:1 mix with v_fma_f32 per SIMD at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_fma_rate.hip -o build/pk_fma_rate && build/pk_fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    f2 acc[16];
    float v[16];
    for (int q = 0; q < 16; ++q) { acc[q].x = threadIdx.x + q; acc[q].y = q; v[q] = q + threadIdx.x; }
    f2 a2 = {1.0000001f, 0.9999999f}, b2 = {1e-9f, 2e-9f};
    float a = 1.0000001f, b = 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[q]) : "v"(a2), "v"(b2));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[q]) : "v"(a2), "v"(b2));
            if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[q]) : "v"(a2), "v"(b2)); if (q & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b)); }
            if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[q]) : "v"(a2));
        }
    }
    float s = 0;
    for (int q = 0; q < 16; ++q) s += acc[q].x + acc[q].y + v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *out, double instr_per_iter, double fma_per_iter_lane) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 64 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-34s %d waves/SIMD: %.2f cycles@2.4GHz per instruction per SIMD, %.1f FMA/cycle/SIMD, %.1f TFLOP/s\n", name, wps,
               cyc / (instr_per_iter * iters * wps), fma_per_iter_lane * 64 * iters * wps / cyc,
               2.0 * fma_per_iter_lane * 64 * iters * wps * 1024 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("v_pk_fma_f32", out, 16, 32);
    run<1>("v_pk_fma_f32 op_sel broadcast", out, 16, 32);
    run<2>("2 v_pk_fma_f32 : 1 v_fma_f32", out, 24, 40);
    run<3>("v_fma_f32", out, 16, 16);
    run<4>("v_pk_mul_f32", out, 16, 32);
    return 0;
}
