// Developer tool: where does k_tgemm's wall time go?  Runs T = PHI * B at c4's shape from the product source (k_gemm.hip compiled
// with GPZ_GEMM_TRACE: s_memtime stamps per wave at kernel entry / K-loop start / K-loop end / exit, plus HW_ID) and prints the
// timeline statistics per compute unit: resident-workgroup coverage, prologue / loop / epilogue shares, gaps between workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/gemm_trace.hip gpz_amd/csrc/gpz_options.hip -o build/gemm_trace
// Run:   build/gemm_trace [rows=1000000] [m=1000]
#define GPZ_GEMM_TRACE 1
#include "../gpz_amd/csrc/k_gemm.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>

__global__ void k_fill(double *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0xffffff) * (1.0 / 16777216.0) + 1e-3;
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, m = argc > 2 ? atoi(argv[2]) : 1000;
    const int mp = (m + 1 + 15) / 16 * 16, n_pad = (n + 127) / 128 * 128;
    double *Phi, *B, *T, *nupart, *phiw;
    (void)hipMalloc(&Phi, (size_t)n_pad * mp * 8); (void)hipMalloc(&T, (size_t)n_pad * mp * 8); (void)hipMalloc(&B, (size_t)mp * mp * 8);
    const int nct = (mp + 127) / 128;
    (void)hipMalloc(&nupart, (size_t)gpz_gemm_wave_cols() * nct * n_pad * 8); (void)hipMalloc(&phiw, (size_t)n_pad * 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Phi, (size_t)n_pad * mp, 1u);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, B, (size_t)mp * mp, 2u);
    const int W = (n_pad / 128) * nct;
    const size_t nrec = (size_t)(W + 4 * 512) * 8 * 6;
    unsigned long long *tr; (void)hipMalloc(&tr, nrec * 8); (void)hipMemset(tr, 0, nrec * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {   // untraced timing first
        (void)hipEventRecord(e0);
        launch_tgemm(0, Phi, mp, B, mp, T, n_pad, mp, nupart, phiw, m, m);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_tgemm n=%d m=%d (mp %d): %.3f ms  %.1f TFLOP/s algorithmic (2 n m^2)  [%s]\n", n, m, mp, ms, 2.0 * n * (double)m * m / ms * 1e-9, hipGetErrorString(hipGetLastError()));
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &tr, sizeof(tr));
    (void)hipEventRecord(e0);
    launch_tgemm(0, Phi, mp, B, mp, T, n_pad, mp, nupart, phiw, m, m);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("traced launch: %.3f ms\n", ms);
    std::vector<unsigned long long> h(nrec);
    (void)hipMemcpy(h.data(), tr, nrec * 8, hipMemcpyDeviceToHost);

    // group the workgroups by compute unit: key = (xcc, se, sh, cu) from wave 0's HW_ID
    struct WG { unsigned long long t0, t1, t2, t3, t2max, t3min; };
    std::map<unsigned, std::vector<WG>> cus;
    size_t nwg = 0;
    for (size_t b = 0; b * 48 < nrec; ++b) {
        const unsigned long long *r = &h[b * 48];
        if (!r[0]) continue;
        ++nwg;
        const unsigned hw = (unsigned)r[4], xcc = (unsigned)(r[4] >> 32) & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        WG w{r[0], r[1], r[2], r[3], 0, ~0ull};
        for (int v = 0; v < 8; ++v) {
            w.t0 = std::min(w.t0, r[v * 6 + 0]); w.t3 = std::max(w.t3, r[v * 6 + 3]);
            w.t1 = std::max(w.t1, r[v * 6 + 1]); w.t2 = std::min(w.t2, r[v * 6 + 2]);
            w.t2max = std::max(w.t2max, r[v * 6 + 2]); w.t3min = std::min(w.t3min, r[v * 6 + 3]);
        }
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(w);
        if (b == 0 || b == 256) printf("item %zu runs on xcc %u se %u sh %u cu %u\n", b, xcc, se, sh, cu);
    }
    printf("%zu workgroups on %zu compute units\n", nwg, cus.size());
    double pro = 0, loop = 0, epi = 0, life = 0, skew2 = 0, skew3 = 0;
    double cover2 = 0, cover1 = 0, cover0 = 0, span_sum = 0;
    std::vector<double> gaps;
    for (auto &kv : cus) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [](const WG &a, const WG &b) { return a.t0 < b.t0; });
        unsigned long long lo = ~0ull, hi = 0;
        std::vector<std::pair<unsigned long long, int>> ev;
        for (auto &w : v) {
            pro += w.t1 - w.t0; loop += w.t2max - w.t1; epi += w.t3 - w.t2max; life += w.t3 - w.t0;
            skew2 += w.t2max - w.t2; skew3 += w.t3 - w.t3min;
            lo = std::min(lo, w.t0); hi = std::max(hi, w.t3);
            ev.push_back({w.t0, +1}); ev.push_back({w.t3, -1});
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0; unsigned long long prev = lo;
        for (auto &e : ev) {
            const double dt = (double)(e.first - prev);
            if (cur >= 2) cover2 += dt; else if (cur == 1) cover1 += dt; else cover0 += dt;
            cur += e.second; prev = e.first;
        }
        span_sum += (double)(hi - lo);
        // slot hand-over: time from a workgroup's exit to the next workgroup start after it on this CU
        std::vector<unsigned long long> starts, ends;
        for (auto &w : v) { starts.push_back(w.t0); ends.push_back(w.t3); }
        std::sort(ends.begin(), ends.end());
        for (size_t i = 0; i + 2 < v.size(); ++i) gaps.push_back((double)starts[i + 2] - (double)ends[i]);   // i-th exit frees the slot the (i+2)-th start takes
    }
    {   // the first compute unit's timeline: (start, loop start, loop end, exit) of its first and last workgroups, relative to its first start
        auto &v = cus.begin()->second;
        for (size_t i = 0; i < v.size(); ++i)
            if (i < 8 || i + 6 >= v.size())
                printf("  cu0 wg %3zu: start %9lld  loop %9lld .. %9lld  exit %9lld\n", i, (long long)(v[i].t0 - v[0].t0), (long long)(v[i].t1 - v[0].t0),
                       (long long)(v[i].t2max - v[0].t0), (long long)(v[i].t3 - v[0].t0));
    }
    std::sort(gaps.begin(), gaps.end());
    printf("per workgroup (cycles of s_memtime): prologue %.0f  K loop %.0f  epilogue %.0f  lifetime %.0f   wave skew at loop end %.0f, at exit %.0f\n", pro / nwg, loop / nwg,
           epi / nwg, life / nwg, skew2 / nwg, skew3 / nwg);
    printf("CU coverage: two workgroups resident %.2f %%, one %.2f %%, none %.2f %% of the per-CU span\n", 100 * cover2 / span_sum, 100 * cover1 / span_sum, 100 * cover0 / span_sum);
    if (!gaps.empty())
        printf("slot hand-over (exit -> next start on the CU): median %.0f, p10 %.0f, p90 %.0f, mean %.0f cycles\n", gaps[gaps.size() / 2], gaps[gaps.size() / 10],
               gaps[gaps.size() * 9 / 10], [&] { double s = 0; for (double g : gaps) s += g; return s / gaps.size(); }());
    // ideal K-loop time: 63 slices x 32 MFMA x 64 cycles x 4 waves per SIMD
    const int nst = mp / 16;
    printf("ideal MFMA time of one K loop with the CU shared by two workgroups: %d cycles (x 100/2400 in s_memtime ticks if the counter runs at 100 MHz)\n", nst * 32 * 64 * 4);
    return 0;
}
