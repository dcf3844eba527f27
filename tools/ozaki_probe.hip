// Probe (developer tool, not product): what rate does an int8-sliced (Ozaki scheme I) form of T = PHI * B reach on gfx950?
//
// fp64 operands are cut into S = 7 balanced base-256 digit planes (int8); the product is the sum over the 28 plane pairs (p, q) with
// p + q <= 6 of exact int32 GEMMs on v_mfma_i32_32x32x32_i8, one int32 accumulator per LEVEL p + q (7 accumulators), recombined in fp64.
// Workgroup: 8 waves (2 per SIMD), tile 128 rows x 64 columns, each wave a 32 x 32 block with all 7 levels (112 accumulator registers);
// per K step of 32 the workgroup stages 7 x 4 KB of A planes + 7 x 2 KB of B planes by LDS-DMA (global_load_lds_dwordx4) from plane
// tiles that are stored in the order the LDS wants them (swizzled for conflict-free ds_read_b128), three stages deep.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/ozaki_probe.hip -o build/ozaki_probe ;  run: build/ozaki_probe [rows] [cols] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
#define S_PL 7
#define A_BLK 4096                     // one A plane of one (row panel, K step): 128 rows x 32 k bytes
#define B_BLK 2048                     // one B plane of one (column panel, K step): 64 columns x 32 k bytes
#define STAGE (S_PL * (A_BLK + B_BLK)) // 43008 bytes
#define NSTAGE 3
#define GLDS(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g), (__attribute__((address_space(3))) void *)(l), 16, 0, (VAR & 64) ? 2 : 0)

// chunk (16 bytes) of element (r, half h) inside a plane block: r*2 + (h ^ bit3(r))
__host__ __device__ inline int chunk_of(int r, int h) { return r * 2 + (h ^ ((r >> 3) & 1)); }

// VAR bits: 16 = every workgroup reads the SAME plane tiles (L2-resident), 32 = staging through registers (global_load_dwordx4 + ds_write_b128) instead of LDS-DMA,
// 64 = DMA with the nt policy; 1 = DMA inside the loop, 2 = barrier, 4 = LDS fragment reads inside the loop, 8 = fragments of step ks + 1 read under the MFMAs of step ks
template <int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void k_oz(const char *__restrict__ Apl, const char *__restrict__ Bpl, double *__restrict__ T, int ldt,
                                               int ksteps, int ncp, const double *__restrict__ rs, const double *__restrict__ cs) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware map: the ncp column panels of one row panel run on one XCD at the same time
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int rp = lb / ncp, cp = lb % ncp;
    const char *ga = Apl + (size_t)((VAR & 16) ? 0 : rp) * ksteps * (S_PL * A_BLK);
    const char *gb = Bpl + (size_t)((VAR & 16) ? 0 : cp) * ksteps * (S_PL * B_BLK);

    i16_t acc[S_PL];
#pragma unroll
    for (int l = 0; l < S_PL; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[l][r] = 0;

    // DMA pieces of 1 KiB: 28 of A, 14 of B per stage; wave w issues pieces w, w + 8, ...
    auto dma = [&](int ks, int st) {
        const char *sa = ga + (size_t)ks * (S_PL * A_BLK), *sb = gb + (size_t)ks * (S_PL * B_BLK);
        char *l0 = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < 5; ++i) {   // pieces 0 .. 39: every wave five
            const int pc = wave + 8 * i;
            const char *src = pc < 28 ? sa + pc * 1024 : sb + (pc - 28) * 1024;
            GLDS(src + lane * 16, l0 + pc * 1024);
        }
        if (wave < 2) GLDS(sb + (wave + 12) * 1024 + lane * 16, l0 + (wave + 40) * 1024);
    };
    auto wait_next = [&](bool more) {   // all but the newest stage's pieces of this wave have landed
        if (more) {
            if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int ar = 32 * wr + (lane & 31), bj = 32 * wc + (lane & 31), h = lane >> 5;
    const int aoff = chunk_of(ar, h) * 16, boff = S_PL * A_BLK + chunk_of(bj, h) * 16;
    auto rdfrag = [&](int st, i4_t (&a)[S_PL], i4_t (&b)[S_PL]) {
        const char *l0 = lds + st * STAGE;
#pragma unroll
        for (int p = 0; p < S_PL; ++p) a[p] = *reinterpret_cast<const i4_t *>(l0 + aoff + p * A_BLK);
#pragma unroll
        for (int q = 0; q < S_PL; ++q) b[q] = *reinterpret_cast<const i4_t *>(l0 + boff + q * B_BLK);
    };
    auto mfmas = [&](const i4_t (&a)[S_PL], const i4_t (&b)[S_PL]) {
#pragma unroll
        for (int p = 0; p < S_PL; ++p)
#pragma unroll
            for (int q = 0; q < S_PL - p; ++q) acc[p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[q], acc[p + q], 0, 0, 0);
    };

    if (VAR & 32) {
        i4_t g[6], a[S_PL], b[S_PL];
        auto gld = [&](int ks) {
            const char *sa = ga + (size_t)ks * (S_PL * A_BLK), *sb = gb + (size_t)ks * (S_PL * B_BLK) - 1792 * 16;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int idx = tid + 512 * i;
                g[i] = *reinterpret_cast<const i4_t *>((idx < 1792 ? sa : sb) + idx * 16);
            }
            if (tid < 128) g[5] = *reinterpret_cast<const i4_t *>(sb + (tid + 2560) * 16);
        };
        auto lst = [&](int st) {
            char *l0 = lds + st * STAGE;
#pragma unroll
            for (int i = 0; i < 5; ++i) *reinterpret_cast<i4_t *>(l0 + (tid + 512 * i) * 16) = g[i];
            if (tid < 128) *reinterpret_cast<i4_t *>(l0 + (tid + 2560) * 16) = g[5];
        };
        gld(0); lst(0);
        if (ksteps > 1) { gld(1); lst(1); }
        __syncthreads();
        for (int ks = 0; ks < ksteps; ++ks) {
            if (ks + 2 < ksteps) gld(ks + 2);
            rdfrag(ks % NSTAGE, a, b);
            mfmas(a, b);
            if (ks + 2 < ksteps) lst((ks + 2) % NSTAGE);
            __syncthreads();
        }
    } else if (!(VAR & 8)) {
        dma(0, 0);
        if (ksteps > 1) dma(1, 1);
        i4_t a[S_PL], b[S_PL];
        if (!(VAR & 4)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); rdfrag(0, a, b); }
        for (int ks = 0; ks < ksteps; ++ks) {
            const int st = ks % NSTAGE;
            // this wave's pieces of stage ks have landed when at most the pieces of stage ks + 1 are outstanding
            if (VAR & 1) wait_next(ks + 1 < ksteps);
            if (VAR & 2) __builtin_amdgcn_s_barrier();   // (not __syncthreads(): its fence would wait for the DMA of the NEXT stage as well)
            if ((VAR & 1) && ks + 2 < ksteps) dma(ks + 2, (ks + 2) % NSTAGE);
            if (VAR & 4) rdfrag(st, a, b);
            else asm volatile("" : "+v"(a[0]), "+v"(b[0]));
            mfmas(a, b);
        }
    } else {
        i4_t a0[S_PL], b0[S_PL], a1[S_PL], b1[S_PL];
        dma(0, 0);
        if (ksteps > 1) dma(1, 1);
        wait_next(ksteps > 1);
        __builtin_amdgcn_s_barrier();
        if (ksteps > 2) dma(2, 2);
        rdfrag(0, a0, b0);
        auto step = [&](int ks, i4_t (&ca)[S_PL], i4_t (&cb)[S_PL], i4_t (&na)[S_PL], i4_t (&nb)[S_PL]) {
            // in registers: fragments of step ks; outstanding DMA: stages ks + 1 (perhaps) and ks + 2
            if (VAR & 1) wait_next(ks + 2 < ksteps);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stage ks are done before anyone refills its buffer
            if (VAR & 2) __builtin_amdgcn_s_barrier();
            if ((VAR & 1) && ks + 3 < ksteps) dma(ks + 3, ks % NSTAGE);
            if (ks + 1 < ksteps) rdfrag((ks + 1) % NSTAGE, na, nb);
            mfmas(ca, cb);
        };
        int ks = 0;
        for (; ks + 1 < ksteps; ks += 2) {
            step(ks, a0, b0, a1, b1);
            step(ks + 1, a1, b1, a0, b0);
        }
        if (ks < ksteps) step(ks, a0, b0, a1, b1);
    }
    if (EPI == 0) {   // keep the accumulators alive, store almost nothing
        int s = 0;
#pragma unroll
        for (int l = 0; l < S_PL; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[l][r];
        if (s == 0x7fffffff) T[0] = s;
        return;
    }
    // T[i][j] = rs[i] * cs[j] * sum_l acc_l * 2^(-8 (l + 2)),  smallest level first
    const int col = cp * 64 + 32 * wc + (lane & 31);
    const double csj = cs[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rp * 128 + 32 * wr + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        double v = 0.0;
#pragma unroll
        for (int l = S_PL - 1; l >= 0; --l) v = (v + (double)acc[l][r]) * 0x1p-8;   // Horner in 2^-8, smallest level first: sum_l acc_l 2^(-8 (l + 1))
        T[(size_t)row * ldt + col] = v * 0x1p-8 * rs[row] * csj;
    }
}

template <int EPI, int VAR>
static float run(dim3 grid, size_t shm, const char *dA, const char *dB, double *dT, int m, int ksteps, int ncp, const double *drs, const double *dcs) {
    (void)hipFuncSetAttribute((const void *)k_oz<EPI, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_oz<EPI, VAR>), grid, dim3(512), shm, 0, dA, dB, dT, m, ksteps, ncp, drs, dcs);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

// ---- host side: digit planes of random fp64 matrices, packed tiles, reference on sampled entries ----
static void digits7(double x, double scale_inv, int8_t d[S_PL]) {
    // x * scale_inv in (-1/2, 1/2) -> N = rint(x * scale_inv * 2^56); balanced base-256 digits, most significant first
    long double v = (long double)x * scale_inv;
    __int128 N = (__int128)llrintl(v * 0x1p56L);
    for (int s = S_PL - 1; s >= 0; --s) {
        int dd = (int)(((N % 256) + 256 + 128) % 256) - 128;
        d[s] = (int8_t)dd;
        N = (N - dd) / 256;
    }
    if (N != 0) { fprintf(stderr, "digit overflow\n"); exit(1); }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 131072, m = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int nrp = n / 128, ncp = m / 64, ksteps = K / 32;
    printf("T (%d x %d) = A (%d x %d) * B (%d x %d), 7 x 7 int8 digit planes, 28 products\n", n, m, n, K, K, m);
    // small dense fp64 operands for the check (first 256 rows only), random planes elsewhere
    const int ncheck = 256;
    std::vector<double> A((size_t)ncheck * K), B((size_t)K * m), rs(n), cs(m);
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX) - 0.5; };
    for (auto &v : A) v = rnd() * exp(8 * rnd());
    for (auto &v : B) v = rnd() * exp(8 * rnd());
    size_t abytes = (size_t)nrp * ksteps * S_PL * A_BLK, bbytes = (size_t)ncp * ksteps * S_PL * B_BLK;
    std::vector<int8_t> Apl(abytes), Bpl(bbytes);
    for (auto &v : Apl) v = (int8_t)(rand() & 255);
    // row scales: power of two above 2 max|row|
    for (int i = 0; i < n; ++i) rs[i] = 1.0;
    for (int i = 0; i < ncheck; ++i) {
        double mx = 0; for (int k = 0; k < K; ++k) mx = fmax(mx, fabs(A[(size_t)i * K + k]));
        int e; frexp(mx, &e); rs[i] = ldexp(1.0, e + 2);
        for (int k = 0; k < K; ++k) {
            int8_t d[S_PL]; digits7(A[(size_t)i * K + k], 1.0 / rs[i], d);
            const int rp = i / 128, r = i % 128, ks = k / 32, kk = k % 32;
            for (int p = 0; p < S_PL; ++p)
                Apl[((size_t)(rp * ksteps + ks) * S_PL + p) * A_BLK + chunk_of(r, kk / 16) * 16 + kk % 16] = d[p];
        }
    }
    for (int j = 0; j < m; ++j) {
        double mx = 0; for (int k = 0; k < K; ++k) mx = fmax(mx, fabs(B[(size_t)k * m + j]));
        int e; frexp(mx, &e); cs[j] = ldexp(1.0, e + 2);
        for (int k = 0; k < K; ++k) {
            int8_t d[S_PL]; digits7(B[(size_t)k * m + j], 1.0 / cs[j], d);
            const int cp = j / 64, c = j % 64, ks = k / 32, kk = k % 32;
            for (int q = 0; q < S_PL; ++q)
                Bpl[((size_t)(cp * ksteps + ks) * S_PL + q) * B_BLK + chunk_of(c, kk / 16) * 16 + kk % 16] = d[q];
        }
    }
    char *dA, *dB; double *dT, *drs, *dcs;
    (void)hipMalloc(&dA, abytes); (void)hipMalloc(&dB, bbytes); (void)hipMalloc(&dT, (size_t)n * m * 8);
    (void)hipMalloc(&drs, n * 8); (void)hipMalloc(&dcs, m * 8);
    (void)hipMemcpy(dA, Apl.data(), abytes, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bpl.data(), bbytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(drs, rs.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dcs, cs.data(), m * 8, hipMemcpyHostToDevice);
    const size_t shm = NSTAGE * STAGE;
    dim3 grid(nrp * ncp);
    const double ops = 28.0 * 2.0 * n * (double)m * K;
    auto report = [&](const char *name, float ms) {
        printf("%-58s %.3f ms  %.0f TOP/s int8 (%.3f of 5033)  fp64-equivalent %.1f TFLOP/s  [%s]\n", name, ms, ops / ms * 1e-9, ops / ms * 1e-9 / 5033.0,
               2.0 * n * (double)m * K / ms * 1e-9, hipGetErrorString(hipGetLastError()));
    };
#define RUN(E, V, NAME) report(NAME, run<E, V>(grid, shm, dA, dB, dT, m, ksteps, ncp, drs, dcs))
    RUN(1, 7, "dma + barrier + lds reads, epilogue");
    {
        std::vector<double> T((size_t)ncheck * m);
        (void)hipMemcpy(T.data(), dT, (size_t)ncheck * m * 8, hipMemcpyDeviceToHost);
        double emax = 0, tmax = 0, e64 = 0;
        for (int i = 0; i < ncheck; i += 7)
            for (int j = 0; j < m; j += 5) {
                long double s = 0; double s64 = 0;
                for (int k = 0; k < K; ++k) { s += (long double)A[(size_t)i * K + k] * B[(size_t)k * m + j]; s64 = fma(A[(size_t)i * K + k], B[(size_t)k * m + j], s64); }
                emax = fmax(emax, fabs((double)(T[(size_t)i * m + j] - s))); tmax = fmax(tmax, fabs((double)s)); e64 = fmax(e64, fabs((double)(s64 - s)));
            }
        printf("check on %d rows: max|T - ref| / max|T| = %.3e   (a plain fp64 fma chain: %.3e)\n", ncheck, emax / tmax, e64 / tmax);
    }
    RUN(0, 7, "dma + barrier + lds reads");
    RUN(0, 6, "      barrier + lds reads (no dma)");
    RUN(0, 4, "                lds reads (no dma, no barrier)");
    RUN(0, 0, "mfma only");
    RUN(0, 5, "dma           + lds reads (racy, timing only)");
    RUN(0, 3, "dma + barrier, fragments constant");
    RUN(0, 7 + 16, "dma + barrier + lds reads, all workgroups the same tiles");
    RUN(0, 7 + 64, "dma (nt) + barrier + lds reads");
    RUN(1, 32, "register staging (global_load_dwordx4 + ds_write_b128), epilogue");
    RUN(0, 32, "register staging (global_load_dwordx4 + ds_write_b128)");
    RUN(0, 32 + 16, "register staging, all workgroups the same tiles");
    RUN(1, 15, "prefetched fragments: dma + barrier + lds, epilogue");
    {
        std::vector<double> T((size_t)ncheck * m);
        (void)hipMemcpy(T.data(), dT, (size_t)ncheck * m * 8, hipMemcpyDeviceToHost);
        double emax = 0, tmax = 0;
        for (int i = 0; i < ncheck; i += 7)
            for (int j = 0; j < m; j += 5) {
                long double s = 0;
                for (int k = 0; k < K; ++k) s += (long double)A[(size_t)i * K + k] * B[(size_t)k * m + j];
                emax = fmax(emax, fabs((double)(T[(size_t)i * m + j] - s))); tmax = fmax(tmax, fabs((double)s));
            }
        printf("check (prefetched form): max|T - ref| / max|T| = %.3e\n", emax / tmax);
    }
    RUN(0, 15, "prefetched fragments: dma + barrier + lds");
    RUN(0, 14, "prefetched fragments:       barrier + lds (no dma)");
    RUN(0, 12, "prefetched fragments:                 lds (no dma, no barrier)");
    return 0;
}
