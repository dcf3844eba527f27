// Developer tool: operand / result lane maps and issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by probing.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o build/mfma_probe && build/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const double *a, const double *b, double *o) {
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    o[threadIdx.x] = c;
}
template <int NACC>
__global__ void rate(double *o, int iters, long long *cyc) {
    double acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = threadIdx.x * 1e-3 + q;
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q];
    o[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *a, *b, *o; long long *cyc;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&o, 1 << 20); hipMalloc(&cyc, 8);
    std::vector<double> ha(64), hb(64), ho(64);
    // A map: put 1 at lane la of A, all-ones B -> which outputs light up tells (i) of lane la; then B likewise
    int amap_i[16], bmap_j[16];
    int dmap_i[16], dmap_j[16];
    for (int la = 0; la < 16; ++la) {
        for (int l = 0; l < 64; ++l) { ha[l] = (l == la) ? 1.0 : 0.0; hb[l] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(a, b, o); hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d -> D lanes:", la); for (int l = 0; l < 64; ++l) if (ho[l] != 0) printf(" %d", l); printf("\n");
    }
    for (int lb = 0; lb < 16; ++lb) {
        for (int l = 0; l < 64; ++l) { hb[l] = (l == lb) ? 1.0 : 0.0; ha[l] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(a, b, o); hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
        printf("B lane %2d -> D lanes:", lb); for (int l = 0; l < 64; ++l) if (ho[l] != 0) printf(" %d", l); printf("\n");
    }
    // which (A lane, B lane) pairs share k: product nonzero
    printf("k pairing (A lane la, B lane lb in block 0 giving a nonzero result):\n");
    for (int la = 0; la < 16; ++la) {
        printf("  la=%2d:", la);
        for (int lb = 0; lb < 16; ++lb) {
            for (int l = 0; l < 64; ++l) { ha[l] = (l == la) ? 1.0 : 0.0; hb[l] = (l == lb) ? 1.0 : 0.0; }
            hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
            probe<<<1, 64>>>(a, b, o); hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) if (ho[l] != 0) printf(" (lb=%d->D%d)", lb, l);
        }
        printf("\n");
    }
    for (int nacc : {1, 2, 4, 8}) {
        int iters = 20000;
        if (nacc == 1) rate<1><<<1, 64>>>(o, iters, cyc);
        if (nacc == 2) rate<2><<<1, 64>>>(o, iters, cyc);
        if (nacc == 4) rate<4><<<1, 64>>>(o, iters, cyc);
        if (nacc == 8) rate<8><<<1, 64>>>(o, iters, cyc);
        long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("independent accumulators %d: %.1f clock64 ticks per MFMA (1 wave)\n", nacc, (double)hc / iters / nacc);
    }
    return 0;
}
