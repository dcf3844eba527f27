// Developer tool: operand / result lane maps and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by probing.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_4x4_probe.hip -o build/mfma_probe32 && build/mfma_probe32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *a, const float *b, float *o) {
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) o[threadIdx.x * 4 + r] = c[r];
}
template <int NACC>
__global__ void rate(float *o, int iters, long long *cyc) {
    f4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (f4){0.f, 1.f, 2.f, 3.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][3];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ void rate64(double *o, int iters, long long *cyc) {   // the f64 4x4x4 form for comparison (16 cycles by the PMC counters)
    double acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = q;
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float *a, *b, *o; long long *cyc; double *od;
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&o, 4096); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&od, 4096);
    std::vector<float> ha(64), hb(64), ho(256);
    for (int la : {0, 1, 2, 3, 4, 5, 17, 63}) {
        for (int l = 0; l < 64; ++l) { ha[l] = (l == la) ? 1.f : 0.f; hb[l] = 1.f; }
        (void)hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(a, b, o); (void)hipMemcpy(ho.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("A lane %2d -> D (lane.reg):", la); for (int e = 0; e < 256; ++e) if (ho[e] != 0) printf(" %d.%d", e / 4, e % 4); printf("\n");
    }
    for (int lb : {0, 1, 2, 3, 4, 5, 17, 63}) {
        for (int l = 0; l < 64; ++l) { hb[l] = (l == lb) ? 1.f : 0.f; ha[l] = 1.f; }
        (void)hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(a, b, o); (void)hipMemcpy(ho.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("B lane %2d -> D (lane.reg):", lb); for (int e = 0; e < 256; ++e) if (ho[e] != 0) printf(" %d.%d", e / 4, e % 4); printf("\n");
    }
    for (int nacc : {1, 4, 8}) {
        int iters = 20000; long long hc;
        if (nacc == 1) rate<1><<<1, 64>>>(o, iters, cyc);
        if (nacc == 4) rate<4><<<1, 64>>>(o, iters, cyc);
        if (nacc == 8) rate<8><<<1, 64>>>(o, iters, cyc);
        (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("f32 4x4x1_16b, %d independent accumulators: %.1f clock64 ticks per MFMA\n", nacc, (double)hc / iters / nacc);
        if (nacc == 1) rate64<1><<<1, 64>>>(od, iters, cyc);
        if (nacc == 4) rate64<4><<<1, 64>>>(od, iters, cyc);
        if (nacc == 8) rate64<8><<<1, 64>>>(od, iters, cyc);
        (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("f64 4x4x4_4b,  %d independent accumulators: %.1f clock64 ticks per MFMA\n", nacc, (double)hc / iters / nacc);
    }
    return 0;
}
