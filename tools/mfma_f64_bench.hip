// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (the peak the roofline fractions are quoted against)
// and the f64 VALU FMA rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o mfma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters) {
    d4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double *out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {   // waves per CU
        int blocks = 256 * wpc / 4;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double flops = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64 16x16x4, %2d waves/CU, 8 acc: %.1f TFLOP/s  (%.2f ms)\n", wpc, flops / ms * 1e-9, ms);
    }
    {
        int blocks = 256 * 1;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double per = ms * 1e-3 / iters;   // seconds per dependent mfma
        printf("dependent-chain mfma_f64 latency: %.1f ns (~%.0f cycles at 2.4 GHz)\n", per * 1e9, per * 2.4e9);
    }
    for (int wpc : {4, 8, 16}) {
        int blocks = 256 * wpc / 4;
        double ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double flops = (double)blocks * 256 * iters * 16 * 2.0;
        printf("v_fma_f64, %2d waves/CU: %.1f TFLOP/s  (%.2f ms)\n", wpc, flops / ms * 1e-9, ms);
    }
    return 0;
}
