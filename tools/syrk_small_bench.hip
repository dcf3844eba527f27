// Developer tool: k_syrk_small (+ its reduce) against k_syrk + k_syrk_reduce at a given shape, from the product sources: largest
// difference of the two results and the time of each pair.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/syrk_small_bench.hip -o build/syrk_small_bench
// Run:   build/syrk_small_bench [rows=100000] [mp=208] [old row splits=85]
#define GPZ_SYRK_SMALL_TRACE 1
#include "../gpz_amd/csrc/gpz_options.hip"
#include "../gpz_amd/csrc/k_gemm.hip"
#include "../gpz_amd/csrc/k_syrk_small.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

__global__ void k_fill(double *p, size_t n, unsigned seed, double scale, double off) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffffff) * (1.0 / 16777216.0) + off) * scale;
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, mp = argc > 2 ? atoi(argv[2]) : 208, ns = argc > 3 ? atoi(argv[3]) : 85;
    const int n_pad = (n + 127) / 128 * 128;
    double *Phi, *w, *slab, *S0, *S1;
    const int nt = (mp + 127) / 128;
    const int rps = ((n_pad + ns - 1) / ns + 15) / 16 * 16, nsp = (n_pad + rps - 1) / rps;
    size_t slab_n = (size_t)nsp * mp * mp, s2 = syrk_small_slab_count(n_pad, mp);
    if (s2 > slab_n) slab_n = s2;
    (void)hipMalloc(&Phi, (size_t)n_pad * mp * 8); (void)hipMalloc(&w, (size_t)n_pad * 8); (void)hipMalloc(&slab, slab_n * 8);
    (void)hipMalloc(&S0, (size_t)mp * mp * 8); (void)hipMalloc(&S1, (size_t)mp * mp * 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Phi, (size_t)n_pad * mp, 1u, 1.0, -0.4);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, w, (size_t)n, 5u, 1.0, 0.1);
    (void)hipMemset(w + n, 0, (size_t)(n_pad - n) * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    float ms_old = 0, ms_new = 0;
    for (int pass = 0; pass < 2; ++pass) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) {
            launch_syrk(0, Phi, mp, w, n_pad, mp, nsp, rps, nsp, rps, slab, false, false);
            launch_syrk_reduce(0, slab, nsp, nsp, mp, S0, mp, 0);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_old, e0, e1);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch_syrk_small(0, Phi, mp, w, n_pad, mp, slab, S1, mp, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_new, e0, e1);
    }
    printf("status: %s\n", hipGetErrorString(hipGetLastError()));
    std::vector<double> h0((size_t)mp * mp), h1((size_t)mp * mp);
    (void)hipMemcpy(h0.data(), S0, h0.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), S1, h1.size() * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0, asym = 0;
    for (int i = 0; i < mp; ++i)
        for (int j = 0; j < mp; ++j) {
            md = fmax(md, fabs(h0[(size_t)i * mp + j] - h1[(size_t)i * mp + j]));
            mx = fmax(mx, fabs(h0[(size_t)i * mp + j]));
            asym = fmax(asym, fabs(h1[(size_t)i * mp + j] - h1[(size_t)j * mp + i]));
        }
    if (md > 1e-11 * mx) {
        printf("blocks with differences (row block, column block: max difference):\n");
        for (int bi = 0; bi < mp / 16; ++bi)
            for (int bj = bi; bj < mp / 16; ++bj) {
                double d = 0;
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) d = fmax(d, fabs(h0[(size_t)(16 * bi + i) * mp + 16 * bj + j] - h1[(size_t)(16 * bi + i) * mp + 16 * bj + j]));
                if (d > 1e-11 * mx) printf(" (%d,%d: %.2g)", bi, bj, d);
            }
        printf("\n");
    }
    int kpw; const int nwg = syrk_small_plan(n_pad, &kpw);
    const double fl = (double)n * mp * (mp + 1);
    printf("n=%d mp=%d: k_syrk + reduce (%d tiles x %d splits) %.1f us (%.1f TFLOP/s); k_syrk_small + reduce (%d workgroups x %d K steps) %.1f us (%.1f TFLOP/s)\n",
           n, mp, nt * (nt + 1) / 2, nsp, ms_old / reps * 1e3, fl / (ms_old / reps) * 1e-9, nwg, kpw, ms_new / reps * 1e3, fl / (ms_new / reps) * 1e-9);
    printf("max |S_small - S_syrk| = %.3g of max |S| = %.3g (%.2g relative); asymmetry of S_small %.3g\n", md, mx, md / mx, asym);
    // per-kernel time of the new pair
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
        switch (mp / 16) {
        case 13: launch_syrk_small_nb<13>(0, Phi, mp, w, n_pad, nwg, kpw, slab); break;
        case 16: launch_syrk_small_nb<16>(0, Phi, mp, w, n_pad, nwg, kpw, slab); break;
        case 8: launch_syrk_small_nb<8>(0, Phi, mp, w, n_pad, nwg, kpw, slab); break;
        case 4: launch_syrk_small_nb<4>(0, Phi, mp, w, n_pad, nwg, kpw, slab); break;
        default: break;
        }
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms_k; (void)hipEventElapsedTime(&ms_k, e0, e1);
    printf("k_syrk_small alone (mp = 208 / 256 / 128 / 64 only): %.1f us\n", ms_k / reps * 1e3);
    return md <= 1e-11 * mx ? 0 : 1;
}
