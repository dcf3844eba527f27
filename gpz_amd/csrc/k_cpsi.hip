// GC/VC with input noise in fp64 for 10 < d <= 64: one WAVE per (sample, basis) pair, the pair matrix in MFMA accumulators.
//
//   getPHI.m:78-89   ln PHI_ij = -1/2 Delta' M^-1 Delta + 1/2 ln|Sigma_j| - 1/2 ln|M|,   M = Psi_i + Sigma_j
//   GPz.m:164-185    sum_i dPHI_ij * [1, M^-1 Delta, (M^-1 Delta)(M^-1 Delta)' - M^-1]
//
// The reference factorises a d x d matrix for each of the n*m pairs.  Up to d = 10 a lane holds the packed triangle in
// registers (k_psi.hip); a d = 20 triangle is 420 VGPRs and does not fit, and the general kernels (k_gen.hip) keep their
// matrices in a global-memory workspace - two orders of magnitude below what the arithmetic needs.  Here the matrix, padded
// with identity to D~ = 16 NT (NT = 1..4), lives in the accumulator registers of v_mfma_f64_16x16x4_f64 as NT x NT tiles
// (lane l, register r of tile (t, ct) = element (16t + (l>>4) + 4r, 16ct + (l&15))) and is eliminated in PANELS OF FOUR pivots:
//
//   panel s (pivots 4s .. 4s+3), symmetric sweep:   A11 = L L', W = inv(L) (4 x 4, every lane computes it from 10 broadcast LDS reads)
//       B_ij = A_ij - Y_i Y_j',  Y = A_i1 W'     one MFMA per tile: both operands are rows of Y (K = 4)
//       B_i1 = Y W (= A_i1 inv(A11)),   B_11 = -W'W      (only when the inverse is wanted)
//   after the last panel the tiles hold -M^-1; an extra column carried one element per lane turns into M^-1 Delta, and
//   Delta' M^-1 Delta, ln|M| fall out of the pivot blocks on the way (sum of |W Delta_1|^2, sum of ln det A11).
//
// The pivot columns reach the operand layout through a 4-column LDS panel private to the wave (no workgroup barrier anywhere);
// the four pivot ROWS of a panel are exactly accumulator register s mod 4 of tile row s div 4 in the f64 MFMA layout, so
// their replacement is a register write for all lanes.  PHI needs no inverse: it updates the trailing tiles only and skips
// the replacements.  Missing dimensions are marginalised as in k_psi.hip (identity block in M, zero Delta).
// PHI: wave = one sample, loop over the basis functions (Psi_i tiles stay in registers, PHI is written 64 columns at a time).
// Moments: wave = one basis function and a chunk of rows (Sigma_j tiles and the 3 + d + d*d sums stay in registers).
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

__device__ __forceinline__ double cpsi_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

// LDS accesses of a wave are executed in program order; this only keeps the compiler from moving them across a phase boundary.
__device__ __forceinline__ void cpsi_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// W = inv(L), A = L L' the Cholesky factorisation of a symmetric positive definite 4 x 4 block (lower triangle of A read; W lower
// triangular, its upper part set to zero); returns the product of the four pivots (= det A).  A non-positive pivot gives NaN
// as sqrt() would.  inv(A) = W' W is never formed for the rank-4 update: with Y = A_i1 W' the update is Y Y' (as accurate as
// a Cholesky factorisation of the whole matrix; forming inv(A11) first loses cond(A11) digits in the Schur complement).
__device__ __forceinline__ double cpsi_inv4(const double (&a)[4][4], double (&w)[4][4]) {
    const double p0 = a[0][0];
    const double r0 = cpsi_rsqrt(p0);
    const double l10 = a[1][0] * r0, l20 = a[2][0] * r0, l30 = a[3][0] * r0;
    const double p1 = fma(-l10, l10, a[1][1]);
    const double r1 = cpsi_rsqrt(p1);
    const double l21 = fma(-l20, l10, a[2][1]) * r1, l31 = fma(-l30, l10, a[3][1]) * r1;
    const double p2 = fma(-l21, l21, fma(-l20, l20, a[2][2]));
    const double r2 = cpsi_rsqrt(p2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, a[3][2])) * r2;
    const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a[3][3])));
    const double r3 = cpsi_rsqrt(p3);
    w[0][0] = r0; w[1][1] = r1; w[2][2] = r2; w[3][3] = r3;
    w[0][1] = w[0][2] = w[0][3] = w[1][2] = w[1][3] = w[2][3] = 0.0;
    w[1][0] = -l10 * r0 * r1;
    w[2][1] = -l21 * r1 * r2;
    w[3][2] = -l32 * r2 * r3;
    w[2][0] = -fma(l21, w[1][0], l20 * r0) * r2;
    w[3][1] = -fma(l32, w[2][1], l31 * r1) * r3;
    w[3][0] = -fma(l32, w[2][0], fma(l31, w[1][0], l30 * r0)) * r3;
    return (p0 * p1) * (p2 * p3);
}

// per-wave LDS: Pb | Yop | PErep (each ROWS x 4) | dbuf (64)
template <int NT>
struct CpsiLds {
    static constexpr int ROWS = 16 * NT;
    static constexpr int SIZE = 3 * ROWS * 4 + 64;
};

// Block elimination of the padded pair matrix `a` (see the header).  dl: element `lane` of Delta on entry; on exit (INV) element
// `lane` of M^-1 Delta.  *quad = Delta' M^-1 Delta, *logdet = ln|M| (over the panels that cover dimensions < d; the padding is identity).
template <int NT, bool INV>
__device__ __forceinline__ void cpsi_eliminate(d4_t (&a)[NT][NT], double &dl, int d, double *__restrict__ lds, int lane,
                                               double *quad, double *logdet) {
    constexpr int ROWS = 16 * NT;
    double *Pb = lds, *Yop = lds + ROWS * 4, *PErep = lds + 2 * ROWS * 4, *dbuf = lds + 3 * ROWS * 4;
    const int lr = lane >> 4, lc = lane & 15;
    double q = 0.0, ld_ = 0.0;
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s) {
        if (4 * s < d) {
            const int pt = s / 4, pr = s % 4, p0 = 4 * s;
            // pivot columns -> LDS panel
            if ((lc >> 2) == pr) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Pb[(16 * t + lr + 4 * r) * 4 + (lc & 3)] = a[t][pt][r];
            }
            dbuf[lane] = dl;
            cpsi_sync();
            double A[4][4], W[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y <= x; ++y) A[x][y] = Pb[(p0 + x) * 4 + y];
            const double det = cpsi_inv4(A, W);
            ld_ += log(det);
            double dpn[4], z[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) dpn[x] = dbuf[p0 + x];
#pragma unroll
            for (int x = 0; x < 4; ++x) {                               // z = W Delta_1:  Delta_1' inv(A11) Delta_1 = |z|^2
                double zz = 0.0;
#pragma unroll
                for (int y = 0; y <= x; ++y) zz = fma(W[x][y], dpn[y], zz);
                z[x] = zz;
                q = fma(zz, zz, q);
            }
            if (lane < ROWS) {
                double prow[4], yv[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) prow[x] = Pb[lane * 4 + x];
#pragma unroll
                for (int x = 0; x < 4; ++x) {                           // Y = A_i1 W'
                    double yy = 0.0;
#pragma unroll
                    for (int y = 0; y <= x; ++y) yy = fma(prow[y], W[x][y], yy);
                    yv[x] = yy;
                }
                const int k = lane - p0;
                const bool inpanel = k >= 0 && k < 4;
                const bool zero = INV ? inpanel : (lane < p0 + 4);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    Yop[lane * 4 + x] = zero ? 0.0 : yv[x];
                    if (INV) {
                        double pe = 0.0, ek = 0.0;                      // (A_i1 inv(A11))[x] = (Y W)[x];  inv(A11)[k][x] = (W' W)[k][x]
#pragma unroll
                        for (int y = x; y < 4; ++y) {
                            pe = fma(yv[y], W[y][x], pe);
                            const double wyk = k == 0 ? W[y][0] : (k == 1 ? W[y][1] : (k == 2 ? W[y][2] : W[y][3]));
                            ek = fma(wyk, W[y][x], ek);
                        }
                        PErep[lane * 4 + x] = inpanel ? -ek : pe;
                    }
                }
                double upd = dl, xk = 0.0;                              // other rows: Delta_i -= Y_i z;  pivot rows: (W' z)[k]
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    upd = fma(-yv[x], z[x], upd);
                    const double wxk = k == 0 ? W[x][0] : (k == 1 ? W[x][1] : (k == 2 ? W[x][2] : W[x][3]));
                    xk = fma(wxk, z[x], xk);
                }
                dl = inpanel ? xk : upd;
            }
            cpsi_sync();
            double at[NT], bt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bt[t] = Yop[(16 * t + lc) * 4 + lr];
                at[t] = -bt[t];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    if (INV || (t >= pt && ct >= pt)) a[t][ct] = MFMA_F64(at[t], bt[ct], a[t][ct]);
            if (INV) {
                if ((lc >> 2) == pr) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[t][pt][r] = PErep[(16 * t + lr + 4 * r) * 4 + (lc & 3)];
                }
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) a[pt][ct][pr] = PErep[(16 * ct + lc) * 4 + lr];
            }
            cpsi_sync();
        }
    }
    *quad = q;
    *logdet = ld_;
}

// PHI: one wave per sample.  Xr: n_pad x de; Psi3: n_pad x d*d; Sig: m x d*d; lnS: [m] (or [G][m] with MISS); P: m x de.
template <int NT, bool MISS>
__global__ __launch_bounds__(256) void k_cpsi_phi(const double *__restrict__ Xr, int de, const double *__restrict__ Psi3, int n,
                                                   int m, int d, const double *__restrict__ P, const double *__restrict__ Sig,
                                                   const double *__restrict__ lnS, double *__restrict__ Phi, int ld,
                                                   const int *__restrict__ gid, const unsigned char *__restrict__ pat) {
    __shared__ double lds_all[4][CpsiLds<NT>::SIZE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    double *lds = lds_all[wave];
    const int g = MISS ? gid[i] : 0;
    const unsigned char *ob = MISS ? pat + (size_t)g * d : nullptr;
    d4_t psi[NT][NT];
    bool keep[NT][NT][4];
    const double *ps = Psi3 + (size_t)i * d * d;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + lr + 4 * r, col = 16 * ct + lc;
                bool kp = row < d && col < d;
                if (MISS && kp) kp = ob[row] && ob[col];
                keep[t][ct][r] = kp;
                psi[t][ct][r] = kp ? ps[col + d * row] : ((row == col) ? 1.0 : 0.0);   // Psi_i is symmetric: consecutive lanes read consecutive doubles
            }
    bool obl = lane < d;
    double cmiss = 0.0;
    if (MISS) {
        obl = obl && ob[lane];
        int nmiss = 0;
        for (int c = 0; c < d; ++c) nmiss += ob[c] ? 0 : 1;
        cmiss = -0.5 * GPZ_LOG2 * nmiss;                                               // -1/2 |u| ln 2   (getPHI.m:87)
    }
    const double xl = obl ? Xr[(size_t)i * de + lane] : 0.0;
    double held = 0.0;
    for (int j = 0; j < m; ++j) {
        d4_t a[NT][NT];
        const double *sg = Sig + (size_t)j * d * d;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + lr + 4 * r, col = 16 * ct + lc;
                    a[t][ct][r] = keep[t][ct][r] ? psi[t][ct][r] + sg[row * d + col] : psi[t][ct][r];   // Psi(o,o,i) + Sigma(o,o)   getPHI.m:84
                }
        double dl = obl ? xl - P[(size_t)j * de + lane] : 0.0;
        double quad, logdet;
        cpsi_eliminate<NT, false>(a, dl, d, lds, lane, &quad, &logdet);
        const double lns = MISS ? lnS[(size_t)g * m + j] : lnS[j];
        const double lp = -0.5 * quad + 0.5 * lns - 0.5 * logdet + cmiss;               // getPHI.m:86
        if ((j & 63) == lane) held = lp;
        if ((j & 63) == 63 || j == m - 1) {
            const int jb = j & ~63;
            if (jb + lane <= j) Phi[(size_t)i * ld + jb + lane] = exp(held);
        }
    }
}

// Moment records for k_gen_finish: one wave per (chunk of rows, basis function).  rec = [A0 | Acc1 (d) | Cacc (d*d) | r1 | r2].
template <int NT, bool MISS>
__global__ __launch_bounds__(256) void k_cpsi_moments(const double *__restrict__ Phi, const double *__restrict__ T, int ld,
                                                       const double *__restrict__ rowscal, const double *__restrict__ w,
                                                       const double *__restrict__ v, const double *__restrict__ Xr, int de,
                                                       const double *__restrict__ Psi3, int n, int m, int d,
                                                       const double *__restrict__ P, const double *__restrict__ Sig,
                                                       int rows_per_chunk, double *__restrict__ slab, int nrec,
                                                       const int *__restrict__ gid, const unsigned char *__restrict__ pat,
                                                       const int *__restrict__ chunktab) {
    __shared__ double lds_all[4][CpsiLds<NT>::SIZE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
    const int j = blockIdx.y * 4 + wave;
    if (j >= m) return;
    double *lds = lds_all[wave];
    double *ubuf = lds + 3 * CpsiLds<NT>::ROWS * 4;
    const int chunk = blockIdx.x;
    d4_t sg[NT][NT], cacc[NT][NT];
    bool inb[NT][NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + lr + 4 * r, col = 16 * ct + lc;
                inb[t][ct][r] = row < d && col < d;
                sg[t][ct][r] = inb[t][ct][r] ? Sig[(size_t)j * d * d + row * d + col] : 0.0;
                cacc[t][ct][r] = 0.0;
            }
    const double pl = lane < d ? P[(size_t)j * de + lane] : 0.0;
    const double wj = w ? w[j] : 0.0, vj = v ? v[j] : 0.0;
    double a0 = 0.0, r1 = 0.0, r2 = 0.0, acc1 = 0.0;
    int r0 = chunk * rows_per_chunk, rend = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; rend = chunktab[2 * chunk + 1]; }   // chunks that end at pattern boundaries
    for (int i = r0; i < rend; ++i) {
        const double ph = Phi[(size_t)i * ld + j];
        double dp;
        if (rowscal) {
            const double *rs = rowscal + (size_t)i * 4;
            dp = (-rs[0] * T[(size_t)i * ld + j] - rs[1] * wj + rs[2] * vj) * ph;      // GPz.m:72,90,106,113
            r1 = fma(ph, rs[1], r1);
            r2 = fma(ph, rs[2], r2);
        } else {
            dp = T[(size_t)i * ld + j];
        }
        const unsigned char *ob = MISS ? pat + (size_t)gid[i] * d : nullptr;
        const double *ps = Psi3 + (size_t)i * d * d;
        d4_t a[NT][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + lr + 4 * r, col = 16 * ct + lc;
                    bool kp = inb[t][ct][r];
                    if (MISS && kp) kp = ob[row] && ob[col];
                    a[t][ct][r] = kp ? sg[t][ct][r] + ps[col + d * row] : ((row == col) ? 1.0 : 0.0);   // Sigma + Psi_i   GPz.m:170
                }
        bool obl = lane < d;
        if (MISS) obl = obl && ob[lane];
        double dl = obl ? Xr[(size_t)i * de + lane] - pl : 0.0;
        double quad, logdet;
        cpsi_eliminate<NT, true>(a, dl, d, lds, lane, &quad, &logdet);                 // a = -M^-1, dl = (M^-1 Delta)[lane]
        ubuf[lane] = dl;
        cpsi_sync();
        double uc[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) uc[ct] = ubuf[16 * ct + lc];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ur = ubuf[16 * t + lr + 4 * r];
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) cacc[t][ct][r] = fma(dp, fma(ur, uc[ct], a[t][ct][r]), cacc[t][ct][r]);   // GPz.m:174
            }
        acc1 = fma(dp, dl, acc1);                                                      // GPz.m:172
        a0 += dp;
        cpsi_sync();
    }
    double *rec = slab + ((size_t)chunk * m + j) * nrec;
    if (lane == 0) {
        rec[0] = a0;
        rec[1 + d + d * d] = r1;
        rec[2 + d + d * d] = r2;
    }
    if (lane < d) rec[1 + lane] = acc1;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + lr + 4 * r, col = 16 * ct + lc;
                if (inb[t][ct][r]) rec[1 + d + row * d + col] = cacc[t][ct][r];
            }
}

bool cpsi_available(int d) {
    return !gpz_opts().cpsi_off && d > 10 && d <= 64;   // (developer switch: back to the general kernels of k_gen.hip)
}

#define CPSI_CASES(MACRO)                   \
    do {                                    \
        if (d <= 16) { MACRO(1); }          \
        else if (d <= 32) { MACRO(2); }     \
        else if (d <= 48) { MACRO(3); }     \
        else { MACRO(4); }                  \
    } while (0)

int launch_cpsi_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                    const double *lnS, double *Phi, int ld, const unsigned char *pat) {
    if (!cpsi_available(d)) return -1;
    if (r.n <= 0) return 0;
#define PHI_CASE(NT)                                                                                                         \
    do {                                                                                                                     \
        if (pat)                                                                                                             \
            hipLaunchKernelGGL((k_cpsi_phi<NT, true>), dim3((r.n + 3) / 4), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, d, P,  \
                               Sig, lnS, Phi, ld, r.gid, pat);                                                              \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_cpsi_phi<NT, false>), dim3((r.n + 3) / 4), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, d, P, \
                               Sig, lnS, Phi, ld, nullptr, nullptr);                                                        \
    } while (0)
    CPSI_CASES(PHI_CASE);
#undef PHI_CASE
    return 0;
}

int launch_cpsi_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                        const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                        int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                        const int *chunktab) {
    if (!cpsi_available(d)) return -1;
    if (nchunk <= 0) return 0;
#define MOM_CASE(NT)                                                                                                         \
    do {                                                                                                                     \
        if (pat)                                                                                                             \
            hipLaunchKernelGGL((k_cpsi_moments<NT, true>), dim3(nchunk, (m + 3) / 4), dim3(256), 0, st, Phi, T, ld, rowscal, w, \
                               v, r.Xr, de, r.Psi3, r.n, m, d, P, Sig, rows_per_chunk, slab, nrec, r.gid, pat, chunktab);      \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_cpsi_moments<NT, false>), dim3(nchunk, (m + 3) / 4), dim3(256), 0, st, Phi, T, ld, rowscal,  \
                               w, v, r.Xr, de, r.Psi3, r.n, m, d, P, Sig, rows_per_chunk, slab, nrec, nullptr, nullptr,       \
                               chunktab);                                                                                    \
    } while (0)
    CPSI_CASES(MOM_CASE);
#undef MOM_CASE
    return 0;
}
