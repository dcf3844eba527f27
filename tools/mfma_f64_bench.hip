// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (the peak the roofline fractions are quoted against),
// the f64 VALU FMA rate, and both together (MFMA waves and VALU waves co-resident on every SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4_t __attribute__((ext_vector_type(4)));

// 16 independent accumulators pinned in registers by inline asm (no accumulator shuffling by the compiler)
#define MF(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__device__ __forceinline__ double mfma_loop(int iters, double a, double b) {
    d4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    d4_t c8 = c0, c9 = c0, c10 = c0, c11 = c0, c12 = c0, c13 = c0, c14 = c0, c15 = c0;
    for (int it = 0; it < iters; ++it) {
        MF(c0); MF(c1); MF(c2); MF(c3); MF(c4); MF(c5); MF(c6); MF(c7);
        MF(c8); MF(c9); MF(c10); MF(c11); MF(c12); MF(c13); MF(c14); MF(c15);
    }
    d4_t s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
    return s[0] + s[1] + s[2] + s[3];
}
__device__ __forceinline__ double fma_loop(int iters, double seed) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = seed + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    return s;
}
// mode 0: all waves MFMA; 1: all waves VALU FMA; 2: waves 0-3 MFMA, waves 4-7 FMA
__global__ __launch_bounds__(512) void k_mix(double *out, int iters_mfma, int iters_fma, int mode) {
    const int wave = threadIdx.x >> 6;
    double r;
    const bool do_mfma = (mode == 0) || (mode == 2 && wave < 4);   // waves 0-3 and 4-7 each cover all four SIMDs
    if (do_mfma) r = mfma_loop(iters_mfma, threadIdx.x * 1e-3, 1.0 + threadIdx.x * 1e-4);
    else r = fma_loop(iters_fma, threadIdx.x * 1e-3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(64) void k_dep(double *out, int iters) {
    d4_t c = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0;
    for (int it = 0; it < iters; ++it) { MF(c); MF(c); MF(c); MF(c); }
    out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}
template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main(int argc, char **argv) {
    double *out; (void)hipMalloc(&out, 256 * 8 * 512 * sizeof(double));
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;   // 4000: 2-7 ms launches; 40000 shows the sustained (power-limited) rate
    const double mf = 16 * 2.0 * 16 * 16 * 4;      // flops per wave per mfma_loop iteration
    const double ff = 16 * 64 * 2.0;               // flops per wave per fma_loop iteration
    for (int wps : {1, 2, 3, 4}) {                 // waves per SIMD
        int threads = 256, blocks = 256 * wps;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(threads), 0, 0, out, iters, 0, 0); });
        printf("mfma_f64 16x16x4  %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", wps, blocks * 4.0 * iters * mf / ms * 1e-9, ms);
    }
    {
        double ms = timeit([&] { hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, out, iters); });
        printf("dependent mfma_f64 chain: %.1f ns per instruction\n", ms * 1e6 / (4.0 * iters));
    }
    const int itf = iters * 8;
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, 0, itf, 1); });
        printf("v_fma_f64         %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", wps, blocks * 4.0 * itf * ff / ms * 1e-9, ms);
    }
    // co-residency: 512-thread blocks, waves 0-3 MFMA, waves 4-7 FMA -> each SIMD hosts one of each per block
    for (int bpc : {1, 2}) {
        int blocks = 256 * bpc;
        // balance iteration counts so both halves run about equally long: mfma iter ~ 16*64 cycles, fma iter ~ 16*4 cycles
        int im = iters, ifm = iters * 16;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, im, ifm, 2); });
        double fl_m = blocks * 4.0 * im * mf, fl_f = blocks * 4.0 * ifm * ff;
        printf("mixed (%d x 512-thread blocks/CU): mfma %.1f + fma %.1f = %.1f TFLOP/s (%.2f ms)\n", bpc, fl_m / ms * 1e-9,
               fl_f / ms * 1e-9, (fl_m + fl_f) / ms * 1e-9, ms);
        double ms_m = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, im, 0, 2); });
        double ms_f = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, 0, ifm, 2); });
        printf("   same launch with only the mfma half: %.2f ms; only the fma half: %.2f ms\n", ms_m, ms_f);
    }
    return 0;
}
