// Developer probe: can config 5's PHI pair kernel hold TWO waves per SIMD (<= 256 registers per lane) if the
// d = 20 pair matrix A = I + R Psi R' is generated and factorised in two 10-column blocks, so that at most L11 (55) + L21 (100)
// or L21 (100) + S22 (55) entries are live instead of the whole 210-entry triangle?  The probe is the blocked algorithm in scalar
// fp32 (the live-entry count is the same as in the packed row-pair form of k_psi32.hip), compiled for two workgroups per CU; its
// register / scratch figures come from the compiler, its time from a run against the one-block form of the same code.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/psi32_blocked_probe.hip -o build/psi32_blocked_probe   (-save-temps for the .s)
// Run:   build/psi32_blocked_probe [rows=250000] [m=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

#define TR(r, c) ((r) * ((r) + 1) / 2 + (c))   // packed lower triangle

// A_ab = delta_ab + sum_{k >= max(a,b)} R_ak psi_k R_bk   (R upper triangular, row-major D x D in LDS)
template <int D>
__device__ __forceinline__ float a_entry(const float *__restrict__ R, const float (&pd)[D], int a, int b) {
    float s = (a == b) ? 1.f : 0.f;
    const int k0 = a > b ? a : b;
#pragma unroll
    for (int k = k0; k < D; ++k) s = fmaf(R[a * D + k] * pd[k], R[b * D + k], s);
    return s;
}

// in-place Cholesky of an N x N packed lower triangle, forward substitution of z alongside; returns sum ln l_cc, adds |y|^2 to quad
template <int N>
__device__ __forceinline__ float chol_solve(float (&L)[N * (N + 1) / 2], float (&z)[N], float &quad) {
    float hl = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const float piv = L[TR(c, c)];
        const float rd = rsqrtf(piv);
        hl += 0.5f * __logf(piv);
        L[TR(c, c)] = rd;                                  // the reciprocal diagonal is what the later solves want
        const float yc = z[c] * rd;
        quad = fmaf(yc, yc, quad);
        z[c] = yc;
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const float l = L[TR(r, c)] * rd;
            L[TR(r, c)] = l;
            z[r] = fmaf(-l, yc, z[r]);
        }
#pragma unroll
        for (int r = c + 1; r < N; ++r)
#pragma unroll
            for (int q = c + 1; q <= r; ++q) L[TR(r, q)] = fmaf(-L[TR(r, c)], L[TR(q, c)], L[TR(r, q)]);
    }
    return hl;
}

template <int D, bool BLOCKED, int WGS, bool XL>
__global__ __launch_bounds__(256, WGS) void k_probe(const double *__restrict__ Xr, const float *__restrict__ PsiT, long ldp, int n, int m,
                                                     const float *__restrict__ Rg, const double *__restrict__ P, double *__restrict__ out) {
    constexpr int H = D / 2, JB = 8;
    __shared__ float sR[JB][D * D];
    __shared__ double sP[JB][D];
    __shared__ double sX[XL ? D : 1][256];                 // XL: the row's inputs live in LDS (dimension-major: conflict-free), not in 40 registers
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned ic = (unsigned)(i < n ? i : 0);
    double x[XL ? 1 : D];
    float pd[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (XL) sX[c][threadIdx.x] = Xr[(size_t)ic * D + c]; else x[c] = Xr[(size_t)ic * D + c];
        pd[c] = PsiT[(size_t)c * ldp + ic];
    }
#define XV(k) (XL ? sX[k][threadIdx.x] : x[XL ? 0 : (k)])
    double acc = 0.0;
    for (int j0 = 0; j0 < m; j0 += JB) {
        __syncthreads();
        for (int e = threadIdx.x; e < JB * D * D; e += 256) sR[e / (D * D)][e % (D * D)] = Rg[(size_t)(j0 + e / (D * D)) * D * D + e % (D * D)];
        for (int e = threadIdx.x; e < JB * D; e += 256) sP[e / D][e % D] = P[(size_t)(j0 + e / D) * D + e % D];
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const float *R = sR[jj];
            float quad = 0.f, hl = 0.f;
            if (!BLOCKED) {
                float L[D * (D + 1) / 2], z[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int k = a; k < D; ++k) s = fmaf(R[a * D + k], (float)(XV(k) - sP[jj][k]), s);
                    z[a] = s;
#pragma unroll
                    for (int b = 0; b <= a; ++b) L[TR(a, b)] = a_entry<D>(R, pd, a, b);
                }
                hl = chol_solve<D>(L, z, quad);
            } else {
                float z1[H], z2[H];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int k = a; k < D; ++k) s = fmaf(R[a * D + k], (float)(XV(k) - sP[jj][k]), s);
                    if (a < H) z1[a] = s; else z2[a - H] = s;
                }
                __builtin_amdgcn_sched_barrier(0);           // phase fences: the scheduler otherwise hoists the generation of later blocks
                float L21[H][H];
                {
                    float L11[H * (H + 1) / 2];
#pragma unroll
                    for (int a = 0; a < H; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) L11[TR(a, b)] = a_entry<D>(R, pd, a, b);
                    hl = chol_solve<H>(L11, z1, quad);      // z1 becomes y1, the diagonal holds reciprocals
                    __builtin_amdgcn_sched_barrier(0);
                    // L21 = A21 inv(L11)', row by row (forward substitution), and z2 -= L21 y1
#pragma unroll
                    for (int a = 0; a < H; ++a) {
#pragma unroll
                        for (int b = 0; b < H; ++b) {
                            float s = a_entry<D>(R, pd, H + a, b);
#pragma unroll
                            for (int q = 0; q < b; ++q) s = fmaf(-L21[a][q], L11[TR(b, q)], s);
                            L21[a][b] = s * L11[TR(b, b)];
                        }
#pragma unroll
                        for (int b = 0; b < H; ++b) z2[a] = fmaf(-L21[a][b], z1[b], z2[a]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }                                           // L11 is dead here
                float S[H * (H + 1) / 2];
#pragma unroll
                for (int a = 0; a < H; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        float s = a_entry<D>(R, pd, H + a, H + b);
#pragma unroll
                        for (int q = 0; q < H; ++q) s = fmaf(-L21[a][q], L21[b][q], s);
                        S[TR(a, b)] = s;
                        if (b == a) __builtin_amdgcn_sched_barrier(0);
                    }
                hl += chol_solve<H>(S, z2, quad);
            }
            acc += exp(-0.5 * (double)quad - (double)hl);
        }
    }
    if (i < n) out[i] = acc;
}

template <bool BLOCKED, int WGS, bool XL>
static float run(const double *Xr, const float *Psi, int n, int m, const float *R, const double *P, double *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<20, BLOCKED, WGS, XL>), dim3((n + 255) / 256), dim3(256), 0, 0, Xr, Psi, (long)n, n, m, R, P, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 250000, m = argc > 2 ? atoi(argv[2]) : 256, D = 20;
    std::vector<double> X((size_t)n * D), P((size_t)m * D);
    std::vector<float> Psi((size_t)D * n), R((size_t)m * D * D, 0.f);
    unsigned h = 99u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 8) & 0xffff) / 65536.0; };
    for (auto &v : X) v = rnd() * 2 - 1;
    for (auto &v : P) v = rnd() * 2 - 1;
    for (auto &v : Psi) v = (float)(0.01 + 0.05 * rnd());
    for (int j = 0; j < m; ++j)
        for (int a = 0; a < D; ++a)
            for (int k = a; k < D; ++k) R[((size_t)j * D + a) * D + k] = (float)(a == k ? 1.0 + rnd() : 0.3 * (rnd() - 0.5));
    double *dX, *dP, *o1, *o2; float *dPsi, *dR;
    (void)hipMalloc(&dX, X.size() * 8); (void)hipMalloc(&dP, P.size() * 8); (void)hipMalloc(&dPsi, Psi.size() * 4); (void)hipMalloc(&dR, R.size() * 4);
    (void)hipMalloc(&o1, (size_t)n * 8); (void)hipMalloc(&o2, (size_t)n * 8);
    (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dPsi, Psi.data(), Psi.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice);
    const float t1 = run<false, 1, false>(dX, dPsi, n, m, dR, dP, o1);
    const float t2 = run<true, 2, false>(dX, dPsi, n, m, dR, dP, o2);
    const float t3 = run<true, 1, false>(dX, dPsi, n, m, dR, dP, o2);
    const float t4 = run<true, 2, true>(dX, dPsi, n, m, dR, dP, o2);
    std::vector<double> a(n), b(n);
    (void)hipMemcpy(a.data(), o1, (size_t)n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), o2, (size_t)n * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (int i = 0; i < n; ++i) { md = fmax(md, fabs(a[i] - b[i])); mx = fmax(mx, fabs(a[i])); }
    printf("n=%d m=%d d=20 (scalar fp32 pair factorisation, one pair per lane and basis function)  [%s]\n", n, m, hipGetErrorString(hipGetLastError()));
    printf("one block (210 live entries), one workgroup per CU:          %.2f ms\n", t1);
    printf("two 10-column blocks (<= 155 live), two workgroups per CU:   %.2f ms\n", t2);
    printf("two 10-column blocks, one workgroup per CU:                  %.2f ms\n", t3);
    printf("two 10-column blocks, x in LDS, two workgroups per CU:       %.2f ms\n", t4);
    printf("max |difference| of the row sums %.3g of %.3g\n", md, mx);
    return 0;
}
