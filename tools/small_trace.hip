// Developer tool: where does k_small_tail's time go?  Runs the kernel at a given shape from the product source (k_small.hip compiled
// with GPZ_SMALL_TRACE: s_memtime stamps of every wave at the phase boundaries of its first 8 row blocks) and prints the mean phase times.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/small_trace.hip -o build/small_trace
// Run:   build/small_trace [rows=100000] [m=200] [d=10]
#define GPZ_SMALL_TRACE 1
int gpz_cu_count() { return 256; }
#include "../gpz_amd/csrc/k_small.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_fill(double *p, size_t n, unsigned seed, double scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffffff) * (1.0 / 16777216.0) + 1e-3) * scale;
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, m = argc > 2 ? atoi(argv[2]) : 200, d = argc > 3 ? atoi(argv[3]) : 10, stg = argc > 4 ? atoi(argv[4]) : 0, wpc = argc > 5 ? atoi(argv[5]) : 2;
    const int mp = (m + 1 + 15) / 16 * 16, n_pad = (n + 1023) / 1024 * 1024, nwg = wpc * gpz_cu_count();
    SmallTailArgs a{};
    double *Phi, *B, *Xr, *xmu, *y, *lnb, *wb, *w, *v, *phiw, *slab, *partial;
    a.nf = small_tail_features(GPZ_KIND_DIAG, d, false); a.xs_ld = d + 2; a.missing = 0;
    (void)hipMalloc(&Phi, (size_t)n_pad * mp * 8); (void)hipMalloc(&B, (size_t)mp * mp * 8); (void)hipMalloc(&Xr, (size_t)n_pad * (d + 2) * 8);
    (void)hipMalloc(&xmu, d * 8); (void)hipMalloc(&y, n_pad * 8); (void)hipMalloc(&lnb, n_pad * 8); (void)hipMalloc(&wb, n_pad * 8);
    (void)hipMalloc(&w, m * 8); (void)hipMalloc(&v, m * 8); (void)hipMalloc(&phiw, n_pad * 8);
    (void)hipMalloc(&slab, (size_t)nwg * m * (a.nf + 2) * 8); (void)hipMalloc(&partial, (size_t)nwg * GPZ_NS * 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Phi, (size_t)n_pad * mp, 1u, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, B, (size_t)mp * mp, 2u, 1e-2);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, Xr, (size_t)n_pad * (d + 2), 3u, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, 0, xmu, (size_t)d, 4u, 0.5);
    for (double *q : {y, lnb, wb}) hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, q, (size_t)n_pad, 5u, 1.0);
    for (double *q : {w, v}) hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, 0, q, (size_t)m, 6u, 1.0);
    a.Phi = Phi; a.ld = mp; a.B = B; a.ldb = mp; a.n = n; a.n_pad = n_pad; a.m = m; a.mp = mp; a.d = d; a.kind = GPZ_KIND_DIAG; a.mcol = m;
    a.Xs = Xr; a.y = y; a.omega = nullptr; a.lnbeta = lnb; a.wbeta = wb; a.w = w; a.v = v; a.vscale = 1.0; a.phiw = phiw;
    a.slab = slab; a.partial = partial; a.stagger = stg;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        launch_small_tail(0, a, nwg);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_small_tail n=%d m=%d (mp %d) d=%d: %.1f us  %.1f TFLOP/s algorithmic (2 n m^2)  [%s]\n", n, m, mp, d, ms * 1e3, 2.0 * n * (double)m * m / ms * 1e-9, hipGetErrorString(hipGetLastError()));
    }
    const size_t nrec = (size_t)nwg * 4 * 8 * 8;
    unsigned long long *tr; (void)hipMalloc(&tr, nrec * 8); (void)hipMemset(tr, 0, nrec * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_small_trace), &tr, sizeof(tr));
    launch_small_tail(0, a, nwg);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nrec);
    (void)hipMemcpy(h.data(), tr, nrec * 8, hipMemcpyDeviceToHost);
    const char *names[7] = {"stage PHI + features + first B loads", "wait at barrier 1", "K loop", "nu partials", "barrier 2 + row scalars + barrier 3",
                            "dPHI + moment MFMAs", "barrier 4"};
    for (int wv = 0; wv < 4; ++wv) {
        double sum[7] = {0}, tot = 0; long cnt = 0;
        for (int wg = 0; wg < nwg; ++wg)
            for (int it = 0; it < 8; ++it) {
                const unsigned long long *r = &h[(((size_t)wg * 4 + wv) * 8 + it) * 8];
                if (!r[0] || !r[7]) continue;
                for (int p = 0; p < 7; ++p) sum[p] += (double)(r[p + 1] - r[p]);
                tot += (double)(r[7] - r[0]);
                ++cnt;
            }
        printf("wave %d: %ld blocks, %.0f memtime ticks per block (100 MHz ticks: x 24 = core cycles at 2.4 GHz)\n", wv, cnt, tot / cnt);
        for (int p = 0; p < 7; ++p) printf("    %-40s %8.1f ticks  %5.1f %%\n", names[p], sum[p] / cnt, 100.0 * sum[p] / tot);
    }
    // block-to-block: start of block it+1 minus end of block it, and the kernel's span
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t i = 0; i < nrec; ++i) if (h[i]) { tmin = h[i] < tmin ? h[i] : tmin; tmax = h[i] > tmax ? h[i] : tmax; }
    printf("span of all stamps: %llu ticks\n", tmax - tmin);
    return 0;
}
