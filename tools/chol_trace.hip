// Developer tool: where does one step of the blocked Cholesky (k_chol_step, 32-wide panels) spend its time?  Compiles the product
// source with GPZ_CHOL_TRACE (s_memtime stamps per wave: entry / rows loaded / diagonal block factored / barrier / rows solved /
// barrier / rank-32 update / stores issued) and prints the phase durations of tile (0, 0)'s workgroup per step, plus the
// launch-to-launch time of the chain eager and as one hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/chol_trace.hip gpz_amd/csrc/gpz_options.hip -o build/chol_trace
// Run:   build/chol_trace [m=1000]
#define GPZ_CHOL_TRACE 1
#include "../gpz_amd/csrc/k_chol.hip"
#include "../gpz_amd/csrc/k_gemm.hip"
#include <stdio.h>
#include <vector>
#include <math.h>

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 1000;
    const int mq = (m + CH_NB - 1) / CH_NB * CH_NB, nsteps = mq / CH_NB;
    std::vector<double> S((size_t)m * m), al(m, 1.0);
    // SPD: S = G G' / m + I with a fixed pseudo-random G (m x 64)
    std::vector<double> G((size_t)m * 64);
    unsigned h = 12345u;
    for (auto &g : G) { h = h * 1664525u + 1013904223u; g = ((h >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0; for (int q = 0; q < 64; ++q) s += G[(size_t)i * 64 + q] * G[(size_t)j * 64 + q];
            S[(size_t)i * m + j] = S[(size_t)j * m + i] = s;
        }
    double *dS, *dal, *A, *Lm, *Wz, *logdet; int *info;
    (void)hipMalloc(&dS, S.size() * 8); (void)hipMalloc(&dal, m * 8); (void)hipMalloc(&A, (size_t)mq * mq * 8); (void)hipMalloc(&Lm, (size_t)mq * mq * 8);
    (void)hipMalloc(&Wz, (size_t)mq * mq * 8); (void)hipMalloc(&logdet, 8); (void)hipMalloc(&info, 4);
    (void)hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dal, al.data(), m * 8, hipMemcpyHostToDevice);
    (void)hipMemset(info, 0, 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    auto chain = [&]() {
        launch_build_sigma(st, dS, m, dal, m, mq, A, mq, Wz, logdet);
        for (int k0 = 0; k0 < mq; k0 += CH_NB) launch_chol_step(st, A, Lm, Wz, mq, mq, k0, logdet, info, false);
    };
    double *Tmp; (void)hipMalloc(&Tmp, (size_t)mq * mq * 8);
    auto inverse = [&]() { for (int gs = CH_NB; gs < mq; gs *= 2) launch_trtri_level(st, Lm, Wz, Tmp, mq, mq, gs); };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, st); chain(); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("eager chain m=%d (%d steps): %.1f us = %.2f us per step  [%s]\n", m, nsteps, ms * 1e3, ms * 1e3 / nsteps, hipGetErrorString(hipGetLastError()));
    }
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); chain(); (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("graph chain: %.1f us = %.2f us per step\n", ms * 1e3, ms * 1e3 / nsteps);
    }
    {   // the levels of the triangular inverse behind the factorisation, as a graph of their own
        hipGraph_t g2; hipGraphExec_t ge2;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); inverse(); (void)hipStreamEndCapture(st, &g2);
        (void)hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipGraphLaunch(ge, st);
            (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge2, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("graph of the inverse's levels (gs = 32 .. %d): %.1f us\n", mq / 2, ms * 1e3);
        }
    }
    double ld; int inf; (void)hipMemcpy(&ld, logdet, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
    printf("logdet %.12g info %d\n", ld, inf);
    {   // the factor and the diagonal blocks of its inverse against the input
        std::vector<double> hL((size_t)mq * mq), hW((size_t)mq * mq);
        (void)hipMemcpy(hL.data(), Lm, hL.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hW.data(), Wz, hW.size() * 8, hipMemcpyDeviceToHost);
        double eA = 0, eW = 0;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0; for (int q = 0; q <= j; ++q) s += hL[(size_t)i * mq + q] * hL[(size_t)j * mq + q];
                eA = fmax(eA, fabs(s - (S[(size_t)i * m + j] + (i == j ? al[i] : 0.0))));
            }
        for (int b = 0; b < nsteps; ++b)
            for (int i = 0; i < CH_NB; ++i)
                for (int j = 0; j < CH_NB; ++j) {
                    double s = 0; for (int q = 0; q < CH_NB; ++q) s += hL[(size_t)(b * CH_NB + i) * mq + b * CH_NB + q] * hW[(size_t)(b * CH_NB + q) * mq + b * CH_NB + j];
                    eW = fmax(eW, fabs(s - (i == j ? 1.0 : 0.0)));
                }
        double eF = 0;
        for (int i = 0; i < mq; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0; for (int q = j; q <= i; ++q) s += hL[(size_t)i * mq + q] * hW[(size_t)q * mq + j];
                eF = fmax(eF, fabs(s - (i == j ? 1.0 : 0.0)));
            }
        printf("max |L L' - A| = %.3g   max |L_kk W_kk - I| over the diagonal blocks = %.3g   max |L W - I| = %.3g\n", eA, eW, eF);
    }
    // traced pass
    const size_t nrec = (size_t)nsteps * 4 * 4 * 8;
    unsigned long long *tr; (void)hipMalloc(&tr, nrec * 8); (void)hipMemset(tr, 0, nrec * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_trace), &tr, sizeof(tr));
    chain(); (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> t(nrec);
    (void)hipMemcpy(t.data(), tr, nrec * 8, hipMemcpyDeviceToHost);
    auto at = [&](int step, int wg, int wave, int slot) { return t[(((size_t)step * 4 + wg) * 4 + wave) * 8 + slot]; };
    printf("step  | wg0: load   chol  solve-tail  bar2   mfma  store | total wg0 (cycles of s_memtime) | wg1 total\n");
    double sum[8] = {0}; int cnt = 0;
    for (int s = 0; s < nsteps - 1; ++s) {
        const unsigned long long b = at(s, 0, 0, 0);
        const long load = at(s, 0, 0, 1) - b, chol = at(s, 0, 0, 2) - at(s, 0, 0, 1);
        const long tail = (long)(at(s, 0, 1, 4) - at(s, 0, 0, 2));          // wave 1's solve ends this long after the factorisation
        const long bar2 = at(s, 0, 1, 5) - at(s, 0, 1, 4), mf = at(s, 0, 0, 6) - at(s, 0, 0, 5), stv = at(s, 0, 0, 7) - at(s, 0, 0, 6);
        const long tot = at(s, 0, 0, 7) - b;
        const long tot1 = at(s, 1, 0, 0) ? (long)(at(s, 1, 0, 7) - at(s, 1, 0, 0)) : 0;
        if (s < 6 || s % 8 == 0) printf("%4d  | %6ld %6ld %6ld %6ld %6ld %6ld | %7ld | %7ld\n", s, load, chol, tail, bar2, mf, stv, tot, tot1);
        const long v[7] = {load, chol, tail, bar2, mf, stv, tot};
        for (int q = 0; q < 7; ++q) sum[q] += v[q];
        ++cnt;
    }
    printf("mean  | %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f | %7.0f\n", sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt);
    return 0;
}
