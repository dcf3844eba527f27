// Templates of the four-pairs-per-wave 4 x 4-tile pair kernels: see k_cpsi4.hip for the description.  Included by k_cpsi4.hip (ND = 3 .. 8,
// d <= 32) and k_cpsi4w.hip (ND = 9 .. 12, d <= 48), which only differ in what they instantiate.
#pragma once
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double c4_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

__device__ __forceinline__ void c4_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// W = inv(L) for A = L L' (4 x 4, lower triangle read), entries above the diagonal not written; returns det A.
__device__ __forceinline__ double c4_inv4(const double (&a)[4][4], double (&w)[4][4]) {
    const double p0 = a[0][0];
    const double r0 = c4_rsqrt(p0);
    const double l10 = a[1][0] * r0, l20 = a[2][0] * r0, l30 = a[3][0] * r0;
    const double p1 = fma(-l10, l10, a[1][1]);
    const double r1 = c4_rsqrt(p1);
    const double l21 = fma(-l20, l10, a[2][1]) * r1, l31 = fma(-l30, l10, a[3][1]) * r1;
    const double p2 = fma(-l21, l21, fma(-l20, l20, a[2][2]));
    const double r2 = c4_rsqrt(p2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, a[3][2])) * r2;
    const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a[3][3])));
    const double r3 = c4_rsqrt(p3);
    w[0][0] = r0; w[1][1] = r1; w[2][2] = r2; w[3][3] = r3;
    w[1][0] = -l10 * r0 * r1;
    w[2][1] = -l21 * r1 * r2;
    w[3][2] = -l32 * r2 * r3;
    w[2][0] = -fma(l21, w[1][0], l20 * r0) * r2;
    w[3][1] = -fma(l32, w[2][1], l31 * r1) * r3;
    w[3][0] = -fma(l32, w[2][0], fma(l31, w[1][0], l30 * r0)) * r3;
    return (p0 * p1) * (p2 * p3);
}

constexpr int c4_lt(int I, int J) { return I * (I + 1) / 2 + J; }   // tile (I, J), J <= I, of the lower triangle (tile row ND = Delta)
#define C4_NT(ND) (((ND) + 1) * ((ND) + 2) / 2)
// Register budget.  Tile loads are issued without a branch (clamped index, then a select) so that a pair's loads go out back to
// back: n = 1e5, m = 256, d = 20: 56 -> 39 ms per evaluation, d = 32: 188 -> 134 ms; that keeps up to two registers per tile in
// flight.  Two waves per SIMD (256 registers) hide the latency of the pivot-block chain where tiles, sums and loads fit without
// spilling (ND <= 5, d <= 20); beyond that one wave per SIMD with everything in flight wins (d = 28: 106 ms against 166 ms with
// two waves and 484 bytes of scratch).  -D overrides for experiments.
#ifndef C4_MINB_PHI
#define C4_MINB_PHI(ND) ((ND) <= 5 ? 2 : 1)
#endif
#ifndef C4_MINB_MOM
#define C4_MINB_MOM(ND) ((ND) <= 5 ? 2 : 1)
#endif
#ifndef C4_UNCOND
#define C4_UNCOND(ND) 1
#endif

struct C4Lane {
    int lane, hi, b, lo, tl;          // tl: the lane that holds the transposed element of a tile
    double mh[4], ml[4];              // 1.0 where hi == x / lo == x, else 0.0: W[lo][hi] is picked out of the lane-uniform W by multiply-adds
};
__device__ __forceinline__ C4Lane c4_lane() {
    C4Lane L;
    L.lane = threadIdx.x & 63;
    L.hi = L.lane >> 4; L.b = (L.lane >> 2) & 3; L.lo = L.lane & 3;
    L.tl = 16 * L.lo + 4 * L.b + L.hi;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        L.mh[x] = L.hi == x ? 1.0 : 0.0;
        L.ml[x] = L.lo == x ? 1.0 : 0.0;
    }
    return L;
}

// pivot tile -> every lane of the pair (through LDS) -> W = inv(chol), and the determinant into a running mantissa / exponent pair
__device__ __forceinline__ void c4_factor(double tile, double *__restrict__ ex, const C4Lane &L, double (&W)[4][4], double *mant,
                                          int *expo) {
    c4_sync();
    ex[L.lane] = tile;                               // slot 16 hi + 4 b + lo
    c4_sync();
    double A[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y <= x; ++y) A[x][y] = ex[16 * x + 4 * L.b + y];
    const double det = c4_inv4(A, W);
    *mant *= __builtin_amdgcn_frexp_mant(det);       // ln|M| = ln(prod mant) + (sum exp) ln 2: one logarithm per pair instead of one per panel
    *expo += __builtin_amdgcn_frexp_exp(det);
}

// Sweep of the ND pivot tiles of T (lower tiles, result layout).  INV: the whole matrix (T(0..ND-1, .) becomes -M^-1, row 0 of tile
// row ND (M^-1 Delta)', the corner -Delta' M^-1 Delta); else only the tiles below / right of the pivot are updated (the corner still
// ends as -Delta' M^-1 Delta).  *logdet = ln|M|.  ex: 64 doubles of LDS private to the wave.
template <int ND, bool INV>
__device__ __forceinline__ void c4_sweep(double (&T)[C4_NT(ND)], double *__restrict__ ex, const C4Lane &L, double *logdet) {
    double mant = 1.0;
    int expo = 0;
    double W[4][4];
    c4_factor(T[c4_lt(0, 0)], ex, L, W, &mant, &expo);
#pragma unroll
    for (int p = 0; p < ND; ++p) {
        // result-layout register of W' (lane (hi, lo) holds W[lo][hi], zero below the diagonal): 14 multiply-adds with the lane masks
        // (a select chain costs 60 v_cndmask); W itself is the lane transpose of that
        const double rl0 = fma(L.ml[3], W[3][0], fma(L.ml[2], W[2][0], fma(L.ml[1], W[1][0], L.ml[0] * W[0][0])));
        const double rl1 = fma(L.ml[3], W[3][1], fma(L.ml[2], W[2][1], L.ml[1] * W[1][1]));
        const double rl2 = fma(L.ml[3], W[3][2], L.ml[2] * W[2][2]);
        const double rl3 = L.ml[3] * W[3][3];
        const double Wt = fma(L.mh[3], rl3, fma(L.mh[2], rl2, fma(L.mh[1], rl1, L.mh[0] * rl0)));
        const double Wd = INV ? __shfl(Wt, L.tl, 64) : 0.0;                           // W[hi][lo]
        double Yt[ND + 1];
        // the next pivot tile first: its factorisation (a chain of dependent scalar work) then runs beside the other updates
        if (p + 1 < ND) {
            Yt[p + 1] = MFMA4(Wt, __shfl(T[c4_lt(p + 1, p)], L.tl, 64), 0.0);
            T[c4_lt(p + 1, p + 1)] = MFMA4(-Yt[p + 1], Yt[p + 1], T[c4_lt(p + 1, p + 1)]);
            c4_factor(T[c4_lt(p + 1, p + 1)], ex, L, W, &mant, &expo);
        }
#pragma unroll
        for (int J = 0; J <= ND; ++J) {
            if (J == p || (J == p + 1 && p + 1 < ND) || (!INV && J < p)) continue;
            const double a1 = J < p ? T[c4_lt(p, J)] : __shfl(T[c4_lt(J, p)], L.tl, 64);
            Yt[J] = MFMA4(Wt, a1, 0.0);                                               // Y_J' = W A_1J
        }
#pragma unroll
        for (int I = 0; I <= ND; ++I) {
            if (I == p || (!INV && I < p)) continue;
            const double ny = -Yt[I];
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                if (J == p || (!INV && J < p) || (I == p + 1 && J == p + 1 && p + 1 < ND)) continue;
                T[c4_lt(I, J)] = MFMA4(ny, Yt[J], T[c4_lt(I, J)]);                    // A_IJ - Y_I Y_J'
            }
        }
        if (INV) {
#pragma unroll
            for (int J = 0; J <= ND; ++J) {
                if (J < p) T[c4_lt(p, J)] = MFMA4(Wd, Yt[J], 0.0);                    // W' Y_J'
                else if (J > p) T[c4_lt(J, p)] = MFMA4(Yt[J], Wd, 0.0);               // Y_J W
            }
            T[c4_lt(p, p)] = MFMA4(-Wd, Wd, 0.0);                                     // -W'W
        }
    }
    *logdet = log(mant) + GPZ_LOG2 * (double)expo;
}

// PHI: block b of a wave = one sample, loop over the basis functions.  Arguments as k_psi_phi (k_psi.hip).
//   SHARED   GC: every basis function has the same covariance, so M = Sigma + Psi_i is swept ONCE per sample WITH the inverse and a basis
//            function costs Delta' M^-1 Delta as ND(ND+1)/2 tile products (as the GC branch of k_cpsi4_predict_noisy)
template <int ND, bool MISS, bool SHARED = false>
__global__ __launch_bounds__(256, C4_MINB_PHI(ND)) void k_cpsi4_phi(const double *__restrict__ Xr, int de, const double *__restrict__ Psi3, int n,
                                                    int m, int d, const double *__restrict__ P, const double *__restrict__ Sig,
                                                    const double *__restrict__ lnS, double *__restrict__ Phi, int ld,
                                                    const int *__restrict__ gid, const unsigned char *__restrict__ pat) {
    constexpr int NTD = ND * (ND + 1) / 2;
    static_assert(!MISS || NTD <= 64, "the pattern mask holds one bit per tile");
    __shared__ double ex_all[4][64];
    const C4Lane L = c4_lane();
    const int wave = threadIdx.x >> 6;
    double *ex = ex_all[wave];
    const int i = (blockIdx.x * 4 + wave) * 4 + L.b;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const int g = MISS ? gid[ic] : 0;
    const unsigned char *ob = MISS ? pat + (size_t)g * d : nullptr;
    const double *ps = Psi3 + (size_t)ic * d * d;
    // which elements of the lower tiles are Psi + Sigma (the rest is the identity padding / a marginalised dimension): one bit per tile
    unsigned long long kpm = 0ull;
#pragma unroll
    for (int I = 0; I < ND; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int row = 4 * I + L.hi, col = 4 * J + L.lo;
            bool k = row < d && col < d;
            if (MISS && k) k = ob[row] && ob[col];
            if (MISS && k) kpm |= 1ull << c4_lt(I, J);
        }
    const int eoff = L.hi * d + L.lo;                 // element (4I + hi, 4J + lo) of a row-major d x d matrix: eoff + 4 (I d + J)
    double xv[ND];
    bool obx[ND];
#pragma unroll
    for (int J = 0; J < ND; ++J) {
        const int col = 4 * J + L.lo;
        bool k = L.hi == 0 && col < d;
        if (MISS && k) k = ob[col];
        obx[J] = k;
        xv[J] = k ? Xr[(size_t)ic * de + col] : 0.0;
    }
    double cmiss = 0.0;
    if (MISS) {
        int nmiss = 0;
        for (int c = 0; c < d; ++c) nmiss += ob[c] ? 0 : 1;
        cmiss = -0.5 * GPZ_LOG2 * nmiss;                                               // -1/2 |u| ln 2   (getPHI.m:87)
    }
    const int slot = 4 * L.hi + L.lo;
    double held = 0.0;
    auto build = [&](double (&T)[C4_NT(ND)], const double *sg) {
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                // MISS: the pattern's bit (ND <= 10: at most 55 tiles); else just the padding test
                const bool k = MISS ? (bool)((kpm >> c4_lt(I, J)) & 1ull) : (4 * I + L.hi < d && 4 * J + L.lo < d);
                const double idn = (I == J && L.hi == L.lo) ? 1.0 : 0.0;
                if (C4_UNCOND(ND)) {
                    const int e = min(eoff + 4 * (I * d + J), d * d - 1);
                    const double sv = ps[e] + sg[e];
                    T[c4_lt(I, J)] = k ? sv : idn;                                     // Psi(o,o,i) + Sigma(o,o)   getPHI.m:84 (both symmetric)
                } else {
                    const int e = eoff + 4 * (I * d + J);
                    T[c4_lt(I, J)] = k ? ps[e] + sg[e] : idn;
                }
            }
    };
    double S[SHARED ? C4_NT(ND) : 1], xc[SHARED ? ND : 1];
    bool obc[SHARED ? ND : 1];
    double logdet = 0.0;
    if (SHARED) {
        double (&Sf)[C4_NT(ND)] = reinterpret_cast<double (&)[C4_NT(ND)]>(S);
        build(Sf, Sig);
#pragma unroll
        for (int J = 0; J <= ND; ++J) Sf[c4_lt(ND, J)] = 0.0;
        c4_sweep<ND, true>(Sf, ex, L, &logdet);                      // -M^-1, M = Sigma + Psi_i (missing dimensions: identity block)
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) Sf[c4_lt(I, J)] *= (I == J) ? -1.0 : -2.0;
#pragma unroll
        for (int I = 0; I < ND; ++I) {
            const int row = 4 * I + L.hi;
            bool k = L.lo == 0 && row < d;
            if (MISS && k) k = ob[row];
            obc[I] = k;
            xc[I] = k ? Xr[(size_t)ic * de + row] : 0.0;             // x as columns: lane (hi, lo = 0) holds x[4I + hi]
        }
    }
    for (int j = 0; j < m; ++j) {
        double quad;
        if (SHARED) {
            double qq = 0.0, dc[ND];
#pragma unroll
            for (int I = 0; I < ND; ++I) dc[I] = obc[SHARED ? I : 0] ? xc[SHARED ? I : 0] - P[(size_t)j * de + min(4 * I + L.hi, de - 1)] : 0.0;
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                const double dr = obx[J] ? xv[J] - P[(size_t)j * de + min(4 * J + L.lo, de - 1)] : 0.0;
                double h = 0.0;
#pragma unroll
                for (int I = J; I < ND; ++I) h = MFMA4(dc[I], S[SHARED ? c4_lt(I, J) : 0], h);   // row 0: sum_I Delta_I' S_IJ
                qq = fma(h, dr, qq);
            }
            qq += __shfl_xor(qq, 1, 64);
            qq += __shfl_xor(qq, 2, 64);
            quad = __shfl(qq, 4 * L.b, 64);
        } else {
            double T[C4_NT(ND)];
            build(T, Sig + (size_t)j * d * d);
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                const double pj = P[(size_t)j * de + min(4 * J + L.lo, de - 1)];
                T[c4_lt(ND, J)] = obx[J] ? xv[J] - pj : 0.0;
            }
            T[c4_lt(ND, ND)] = 0.0;
            c4_sweep<ND, false>(T, ex, L, &logdet);
            quad = -__shfl(T[c4_lt(ND, ND)], 4 * L.b, 64);
        }
        const double lns = MISS ? lnS[(size_t)g * m + j] : lnS[j];
        const double lp = -0.5 * quad + 0.5 * lns - 0.5 * logdet + cmiss;               // getPHI.m:86
        if ((j & 15) == slot) held = lp;
        if ((j & 15) == 15 || j == m - 1) {
            const int jb = j & ~15;
            if (valid && jb + slot <= j) Phi[(size_t)i * ld + jb + slot] = exp(held);
        }
    }
}

// GC (one covariance for every basis function): M_i = Sigma + Psi_i depends on the sample only, so its inverse is swept ONCE per sample
// here and the moment kernel below (SHARED) reads it back instead of sweeping once per (sample, basis function):
//     Minv[i][I][J][4 hi + lo] = -(M_i^-1)[4I + hi][4J + lo]      all ND x ND tiles (the upper ones are the lane transposes of the lower)
// Block b of a wave = one sample.  Missing dimensions: identity block, as everywhere in this file.
//   QROW   (no missing dimensions) also writes the row of the dense form of the PHI build,
//              ln PHI_ij = -1/2 ( sum_{a>=b} c_ab M^-1_ab p_a p_b  - 2 (M^-1 x)' p  +  x' M^-1 x + ln|M| - ln|Sigma| ),   c_aa = 1, c_ab = 2,
//          A[i] = [ c_ab M_i^-1_ab (d(d+1)/2) | M_i^-1 x_i (d) | x_i' M_i^-1 x_i + ln|M_i| - ln|Sigma| ]  against the m-sized table
//          [ p_a p_b ; -2 p_a ; 1 ] of k_gcq_tab: one product on k_tgemm and an exp replace ND(ND+1)/2 tile products per pair.
//          x rides along as the extra tile row of the sweep, which leaves (M^-1 x)' and -x' M^-1 x in it.
template <int ND, bool MISS, bool QROW = false>
__global__ __launch_bounds__(256, C4_MINB_PHI(ND)) void k_cpsi4_minv(const double *__restrict__ Psi3, int n, int d,
                                                                      const double *__restrict__ Sig, const int *__restrict__ gid,
                                                                      const unsigned char *__restrict__ pat,
                                                                      double *__restrict__ Minv, const double *__restrict__ Xr, int de,
                                                                      const double *__restrict__ lnS, double *__restrict__ A, int lda, const double *__restrict__ ctr) {
    static_assert(!(MISS && QROW), "the dense form of the PHI build is for rows without missing dimensions");
    __shared__ double ex_all[4][64];
    const C4Lane L = c4_lane();
    const int wave = threadIdx.x >> 6;
    double *ex = ex_all[wave];
    const int i = (blockIdx.x * 4 + wave) * 4 + L.b;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    const unsigned char *ob = MISS ? pat + (size_t)gid[ic] * d : nullptr;
    const double *ps = Psi3 + (size_t)ic * d * d;
    const int eoff = L.hi * d + L.lo;
    double T[C4_NT(ND)];
#pragma unroll
    for (int I = 0; I < ND; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int row = 4 * I + L.hi, col = 4 * J + L.lo;
            bool k = row < d && col < d;
            if (MISS && k) k = ob[row] && ob[col];
            const int e = min(eoff + 4 * (I * d + J), d * d - 1);
            const double sv = ps[e] + Sig[e];
            T[c4_lt(I, J)] = k ? sv : ((row == col) ? 1.0 : 0.0);                       // Sigma + Psi_i   GPz.m:170
        }
#pragma unroll
    for (int J = 0; J <= ND; ++J) T[c4_lt(ND, J)] = 0.0;
    if (QROW) {
#pragma unroll
        for (int J = 0; J < ND; ++J) T[c4_lt(ND, J)] = (L.hi == 0 && 4 * J + L.lo < d) ? Xr[(size_t)ic * de + 4 * J + L.lo] - ctr[4 * J + L.lo] : 0.0;   // x - c (see k_gcq_tab)
    }
    double logdet;
    c4_sweep<ND, true>(T, ex, L, &logdet);                                              // lower tiles: -M^-1
    if (QROW && valid) {
        double *ar = A + (size_t)ic * lda;
        const int K1 = d * (d + 1) / 2;
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                const int row = 4 * I + L.hi, col = 4 * J + L.lo;
                if (row < d && col <= row) ar[row * (row + 1) / 2 + col] = (row == col ? -1.0 : -2.0) * T[c4_lt(I, J)];
            }
#pragma unroll
        for (int J = 0; J < ND; ++J)
            if (L.hi == 0 && 4 * J + L.lo < d) ar[K1 + 4 * J + L.lo] = T[c4_lt(ND, J)];  // (M^-1 x)[4J + lo]
        if (L.hi == 0 && L.lo == 0) ar[K1 + d] = -T[c4_lt(ND, ND)] + logdet - lnS[0];
    }
    double *out = Minv + (size_t)ic * ND * ND * 16 + 4 * L.hi + L.lo;
#pragma unroll
    for (int I = 0; I < ND; ++I)
#pragma unroll
        for (int J = 0; J < ND; ++J) {
            const double v = (I >= J) ? T[c4_lt(I >= J ? I : J, I >= J ? J : I)] : __shfl(T[c4_lt(I >= J ? I : J, I >= J ? J : I)], L.tl, 64);
            if (valid) out[(I * ND + J) * 16] = v;
        }
}

// Moment records for k_gen_finish: block b of a wave = one basis function, rows of a chunk.  rec = [A0 | Acc1 (d) | Cacc (d*d) | r1 | r2].
//   SHARED   GC: -M_i^-1 comes from k_cpsi4_minv's table (Minv); a (sample, basis) pair then costs u' = Delta' M^-1 as ND^2 tile products
//            and the ND(ND+1)/2 products of the sums, no sweep: O(d^2) per pair instead of O(d^3)
template <int ND, bool MISS, bool SHARED = false>
__global__ __launch_bounds__(256, C4_MINB_MOM(ND)) void k_cpsi4_moments(const double *__restrict__ Phi, const double *__restrict__ Tm, int ld,
                                                        const double *__restrict__ rowscal, const double *__restrict__ w,
                                                        const double *__restrict__ v, const double *__restrict__ Xr, int de,
                                                        const double *__restrict__ Psi3, int n, int m, int d,
                                                        const double *__restrict__ P, const double *__restrict__ Sig,
                                                        int rows_per_chunk, double *__restrict__ slab, int nrec,
                                                        const int *__restrict__ gid, const unsigned char *__restrict__ pat,
                                                        const int *__restrict__ chunktab, const double *__restrict__ Minv) {
    constexpr int NTD = ND * (ND + 1) / 2;
    __shared__ double ex_all[4][64];
    const C4Lane L = c4_lane();
    const int wave = threadIdx.x >> 6;
    double *ex = ex_all[wave];
    const int j = (blockIdx.y * 4 + wave) * 4 + L.b;
    const bool valid = j < m;
    const int jc = valid ? j : m - 1;
    const int chunk = blockIdx.x;
    double cacc[NTD], acc1[ND], pv[ND];
#pragma unroll
    for (int e = 0; e < NTD; ++e) cacc[e] = 0.0;
    const int eoff = L.hi * d + L.lo;                 // element (4I + hi, 4J + lo) of a row-major d x d matrix: eoff + 4 (I d + J)
    const double *sg = Sig + (size_t)jc * d * d;
#pragma unroll
    for (int J = 0; J < ND; ++J) {
        const int col = 4 * J + L.lo;
        pv[J] = (L.hi == 0 && col < d) ? P[(size_t)jc * de + col] : 0.0;
        acc1[J] = 0.0;
    }
    double pvc[SHARED ? ND : 1];                     // SHARED: p_j as columns, lane (hi, lo = 0) holds p[4I + hi]
    if (SHARED) {
#pragma unroll
        for (int I = 0; I < ND; ++I) pvc[SHARED ? I : 0] = (L.lo == 0 && 4 * I + L.hi < d) ? P[(size_t)jc * de + 4 * I + L.hi] : 0.0;
    }
    const int slot16 = 4 * L.hi + L.lo;
    const double wj = w ? w[jc] : 0.0, vj = v ? v[jc] : 0.0;
    double a0 = 0.0, r1 = 0.0, r2 = 0.0;
    int r0 = chunk * rows_per_chunk, rend = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; rend = chunktab[2 * chunk + 1]; }   // chunks that end at pattern boundaries
    for (int i = r0; i < rend; ++i) {
        const double ph = Phi[(size_t)i * ld + jc];
        double dp;
        if (rowscal) {
            const double *rs = rowscal + (size_t)i * 4;
            dp = (-rs[0] * Tm[(size_t)i * ld + jc] - rs[1] * wj + rs[2] * vj) * ph;    // GPz.m:72,90,106,113
            r1 = fma(ph, rs[1], r1);
            r2 = fma(ph, rs[2], r2);
        } else {
            dp = Tm[(size_t)i * ld + jc];
        }
        const unsigned char *ob = MISS ? pat + (size_t)gid[i] * d : nullptr;
        double T[C4_NT(ND)];
        if (SHARED) {
            // -M_i^-1 from the table: the lower tiles stay in T (the sums below read them), every tile feeds u' = Delta' M^-1
            const double *mi = Minv + (size_t)i * ND * ND * 16 + slot16;
            double dc[ND];
#pragma unroll
            for (int I = 0; I < ND; ++I) {
                const int row = 4 * I + L.hi;
                bool k = L.lo == 0 && row < d;
                if (MISS && k) k = ob[row];
                const double xi = Xr[(size_t)i * de + min(row, de - 1)];
                dc[I] = k ? xi - pvc[SHARED ? I : 0] : 0.0;
            }
#pragma unroll
            for (int I = 0; I < ND; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J) T[c4_lt(I, J)] = mi[(I * ND + J) * 16];
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                double h = 0.0;
#pragma unroll
                for (int I = 0; I < ND; ++I) {
                    const double z = (I >= J) ? T[c4_lt(I >= J ? I : J, I >= J ? J : I)] : mi[(I * ND + J) * 16];
                    h = MFMA4(dc[I], z, h);                                            // row 0: sum_I Delta_I' (-M^-1)_IJ
                }
                T[c4_lt(ND, J)] = -h;                                                  // (M^-1 Delta)' in row 0, rows 1..3 zero
            }
            T[c4_lt(ND, ND)] = 0.0;
        } else {
        const double *ps = Psi3 + (size_t)i * d * d;
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                const int row = 4 * I + L.hi, col = 4 * J + L.lo;
                bool k = row < d && col < d;
                if (MISS && k) k = ob[row] && ob[col];
                const double idn = (row == col) ? 1.0 : 0.0;
                if (C4_UNCOND(ND)) {
                    const int e = min(eoff + 4 * (I * d + J), d * d - 1);
                    const double sv = sg[e] + ps[e];
                    T[c4_lt(I, J)] = k ? sv : idn;                                     // Sigma + Psi_i   GPz.m:170 (both symmetric)
                } else {
                    const int e = eoff + 4 * (I * d + J);
                    T[c4_lt(I, J)] = k ? sg[e] + ps[e] : idn;
                }
            }
#pragma unroll
        for (int J = 0; J < ND; ++J) {
            const int col = 4 * J + L.lo;
            bool k = L.hi == 0 && col < d;
            if (MISS && k) k = ob[col];
            const double xi = Xr[(size_t)i * de + min(col, de - 1)];
            T[c4_lt(ND, J)] = k ? xi - pv[J] : 0.0;
        }
        T[c4_lt(ND, ND)] = 0.0;
        double logdet;
        c4_sweep<ND, true>(T, ex, L, &logdet);                                         // tiles: -M^-1; tile row ND, row 0: (M^-1 Delta)'
        }
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                const double t = MFMA4(T[c4_lt(ND, I)], T[c4_lt(ND, J)], T[c4_lt(I, J)]);   // u_I u_J' - (M^-1)_IJ
                cacc[c4_lt(I, J)] = fma(dp, t, cacc[c4_lt(I, J)]);                     // GPz.m:174
            }
#pragma unroll
        for (int J = 0; J < ND; ++J) acc1[J] = fma(dp, T[c4_lt(ND, J)], acc1[J]);      // GPz.m:172
        a0 += dp;
    }
    if (!valid) return;
    double *rec = slab + ((size_t)chunk * m + j) * nrec;
    if (L.hi == 0 && L.lo == 0) {
        rec[0] = a0;
        rec[1 + d + d * d] = r1;
        rec[2 + d + d * d] = r2;
    }
#pragma unroll
    for (int J = 0; J < ND; ++J)
        if (L.hi == 0 && 4 * J + L.lo < d) rec[1 + 4 * J + L.lo] = acc1[J];
#pragma unroll
    for (int I = 0; I < ND; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int row = 4 * I + L.hi, col = 4 * J + L.lo;
            if (row < d && col < d) {
                rec[1 + d + row * d + col] = cacc[c4_lt(I, J)];
                if (I > J) rec[1 + d + col * d + row] = cacc[c4_lt(I, J)];
            }
        }
}

// predictNoisy, covariance kinds (predictCov.m:70-132) for 10 < d <= 32: block b of a wave = one sample, the pairs [p0, p1) of this
// chunk; partial sums part[chunk][3][k][ldx] as k_predict_noisy_cov (k_psi.hip).  tab record: [lnz | cij (d) | Cij (d x d)].
// N(x; cij, Cij + Psi_i) = exp(-1/2 Delta' M^-1 Delta - 1/2 ln|M|): one sweep without the inverse per (sample, pair).
//   SHARED   GC: every basis function has the same covariance, so Cij = Sigma/2 for every pair: M = Sigma/2 + Psi_i is swept ONCE per
//            sample WITH the inverse, and a pair costs Delta' M^-1 Delta = sum_J (sum_{I>J} 2 Delta_I' S_IJ + Delta_J' S_JJ) Delta_J on the
//            tiles S of M^-1 (ND(ND+1)/2 instructions instead of a sweep)
template <int ND, int KM, bool SHARED>
__global__ __launch_bounds__(256, C4_MINB_PHI(ND)) void k_cpsi4_predict_noisy(int n, long ldx, int m, int d, int de, int k,
                                                                               const double *__restrict__ Xr,
                                                                               const double *__restrict__ Psi3,
                                                                               const double *__restrict__ tab, int rec,
                                                                               const double *__restrict__ w,
                                                                               const double *__restrict__ v,
                                                                               const double *__restrict__ iS, long pairs_per_chunk,
                                                                               double *__restrict__ part) {
    __shared__ double ex_all[4][64];
    const C4Lane L = c4_lane();
    const int wave = threadIdx.x >> 6;
    double *ex = ex_all[wave];
    const int i = (blockIdx.x * 4 + wave) * 4 + L.b;
    const bool act = i < n;
    const int ic = act ? i : n - 1;
    const long npair = (long)m * (m + 1) / 2;
    const long p0 = (long)blockIdx.y * pairs_per_chunk, p1 = min(npair, p0 + pairs_per_chunk);
    const double *ps = Psi3 + (size_t)ic * d * d;
    const int eoff = L.hi * d + L.lo;
    double xv[ND];
#pragma unroll
    for (int J = 0; J < ND; ++J) xv[J] = (L.hi == 0 && 4 * J + L.lo < d) ? Xr[(size_t)ic * de + 4 * J + L.lo] : 0.0;
    double ga[KM], vl[KM], nu[KM];
#pragma unroll
    for (int o = 0; o < KM; ++o) { ga[o] = 0.0; vl[o] = 0.0; nu[o] = 0.0; }
    long a = (long)((sqrt(8.0 * (double)p0 + 1.0) - 1.0) * 0.5);   // (a, b) of the first pair, then walk
    while (a * (a + 1) / 2 > p0) --a;
    while ((a + 1) * (a + 2) / 2 <= p0) ++a;
    long bb = p0 - a * (a + 1) / 2;
    double S[C4_NT(ND)];
    double logdet = 0.0;
    auto build = [&](double (&T)[C4_NT(ND)], const double *t) {
        const double *cc = t + 1 + d;
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                const int row = 4 * I + L.hi, col = 4 * J + L.lo;
                const int q = min(eoff + 4 * (I * d + J), d * d - 1);   // unconditional loads: clamp, then select
                const double sv = cc[q] + ps[q];
                T[c4_lt(I, J)] = (row < d && col < d) ? sv : ((row == col) ? 1.0 : 0.0);   // Cij + Psi   predictCov.m:109 (both symmetric)
            }
    };
    if (SHARED && p0 < p1) {
        build(S, tab + (size_t)p0 * rec);
#pragma unroll
        for (int J = 0; J <= ND; ++J) S[c4_lt(ND, J)] = 0.0;
        c4_sweep<ND, true>(S, ex, L, &logdet);                       // S = -M^-1
#pragma unroll
        for (int I = 0; I < ND; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) S[c4_lt(I, J)] *= (I == J) ? -1.0 : -2.0;
    }
    double xc[SHARED ? ND : 1];                                      // x as columns: lane (hi, lo = 0) holds x[4I + hi]
    if (SHARED) {
#pragma unroll
        for (int I = 0; I < ND; ++I) xc[I] = (L.lo == 0 && 4 * I + L.hi < d) ? Xr[(size_t)ic * de + 4 * I + L.hi] : 0.0;
    }
#pragma unroll 1
    for (long e = p0; e < p1; ++e) {
        const double *t = tab + (size_t)e * rec;
        double quad;
        if (SHARED) {
            double dr[ND], dc[ND];                                   // Delta_J' as row 0 of a tile, Delta_I as column 0
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                dr[J] = (L.hi == 0 && 4 * J + L.lo < d) ? xv[J] - t[1 + 4 * J + L.lo] : 0.0;
                dc[J] = (L.lo == 0 && 4 * J + L.hi < d) ? xc[J] - t[1 + 4 * J + L.hi] : 0.0;
            }
            double qq = 0.0;
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                double h = 0.0;
#pragma unroll
                for (int I = J; I < ND; ++I) h = MFMA4(dc[I], S[c4_lt(I, J)], h);      // row 0: sum_I Delta_I' S_IJ
                qq = fma(h, dr[J], qq);                                                // lanes (0, lo): h_J[lo] Delta_J[lo]
            }
            qq += __shfl_xor(qq, 1, 64);
            qq += __shfl_xor(qq, 2, 64);
            quad = __shfl(qq, 4 * L.b, 64);
        } else {
            double T[C4_NT(ND)];
            build(T, t);
#pragma unroll
            for (int J = 0; J < ND; ++J) T[c4_lt(ND, J)] = (L.hi == 0 && 4 * J + L.lo < d) ? xv[J] - t[1 + 4 * J + L.lo] : 0.0;
            T[c4_lt(ND, ND)] = 0.0;
            c4_sweep<ND, false>(T, ex, L, &logdet);
            quad = -__shfl(T[c4_lt(ND, ND)], 4 * L.b, 64);
        }
        const double z = ((a == bb) ? 1.0 : 2.0) * exp(t[0] - 0.5 * quad - 0.5 * logdet);      // :111, 2x in the loop (:113-119)
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                ga[o] = fma(z, w[a + (size_t)m * o] * w[bb + (size_t)m * o], ga[o]);
                vl[o] = fma(z, v ? v[a + (size_t)m * o] * v[bb + (size_t)m * o] : 0.0, vl[o]);
                nu[o] = fma(z, iS[a + (size_t)m * bb + (size_t)m * m * o], nu[o]);
            }
        if (++bb > a) { ++a; bb = 0; }
    }
    if (act && L.hi == 0 && L.lo == 0) {
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                part[(((size_t)blockIdx.y * 3 + 0) * k + o) * ldx + i] = ga[o];
                part[(((size_t)blockIdx.y * 3 + 1) * k + o) * ldx + i] = vl[o];
                part[(((size_t)blockIdx.y * 3 + 2) * k + o) * ldx + i] = nu[o];
            }
    }
}

