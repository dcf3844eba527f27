// Prediction for inputs with missing dimensions, covariance kinds (GC/VC):
//   predictMissing       predictCov.m:134-229   (no input noise)
//   predictNoisyMissing  predictCov.m:231-337   (input noise Psi, d x d x n)
// for one group of rows that share a NaN pattern (o = observed dimensions, u = missing ones).
//
// The reference conditions every basis function on the observed dimensions (X_hat, Psi_hat) and then, for every
// basis pair (i,j), sums a d-dimensional Gaussian over all m conditioned components: O(n m^3 d^3) interpreted work
// (a d x d factorisation per (row, pair, component) once there is input noise).  Three routes for the sums: for d <= 10 the
// register-resident record sums further down (k_pmc_sum_*); for 10 < d <= 32 the same sums as MFMA sweeps (k_pmc4.hip); else
// (more than 8 outputs) the first, scratch-resident kernels (runtime d <= 32, one thread per (row, basis) or one wave per
// (row, pair chunk)).  Sums over pairs are ordered (chunk slabs + fixed-order sum).
//
// Kept quirk (predictCov.m:266-268): with input noise the block T*Psi_oo*T' (in [o u] order) is ASSIGNED through
// `unshuffle`, the inverse of the permutation [find(o) find(~o)] — the intended placement only when that permutation
// is its own inverse.
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

// This file is compiled twice: as it stands (GDM = 32: every route), and from k_pmiss_cov64.hip with GDM = 64 and PMC_WIDE defined -
// the scratch-resident kernels only, under other names, for 32 < d <= 64 (launch_pmc_wide).  There a thread's d x d temporaries are
// 32 KB each in scratch memory: correct and slow (DESIGN.md section 7 has the cost line); the sums of the narrow routes do not apply.
#ifndef GDM
#define GDM 32   // widest input of this file's scratch-resident kernels
#endif
#ifdef PMC_WIDE
#define k_pmc_prep k_pmc_prep_w
#define k_pmc_rows k_pmc_rows_w
#define k_pmc_phi k_pmc_phi_w
#define k_pmc_pairs k_pmc_pairs_w
#define k_pmc_accum k_pmc_accum_w
#define PmcPat PmcPatW
#define pmc_chol pmc_chol_w
#define pmc_lognorm pmc_lognorm_w
#define pmc_inv pmc_inv_w
#endif

__device__ inline void pmc_chol(double *M, int n) {        // lower Cholesky in place, leading dimension GDM
    for (int c = 0; c < n; ++c) {
        double p = M[c * GDM + c];
        for (int q = 0; q < c; ++q) p = fma(-M[c * GDM + q], M[c * GDM + q], p);
        const double dd = sqrt(p);
        M[c * GDM + c] = dd;
        for (int r = c + 1; r < n; ++r) {
            double s = M[r * GDM + c];
            for (int q = 0; q < c; ++q) s = fma(-M[r * GDM + q], M[c * GDM + q], s);
            M[r * GDM + c] = s / dd;
        }
    }
}
// -1/2 dl' S^-1 dl - 1/2 ln|S| for SPD S (destroyed): the exponent of every density in predictCov.m
__device__ inline double pmc_lognorm(double *S, const double *dl, int n) {
    pmc_chol(S, n);
    double quad = 0.0, hl = 0.0, y[GDM];
    for (int r = 0; r < n; ++r) {
        double s = dl[r];
        for (int c = 0; c < r; ++c) s = fma(-S[r * GDM + c], y[c], s);
        y[r] = s / S[r * GDM + r];
        quad = fma(y[r], y[r], quad);
        hl += log(S[r * GDM + r]);
    }
    return -0.5 * quad - hl;
}
// Ai = inv(A) for SPD A (n x n, leading dimension GDM); returns ln|A|.  A is destroyed.
__device__ inline double pmc_inv(double *A, int n, double *Ai) {
    pmc_chol(A, n);
    double W[GDM * GDM], ld = 0.0;
    for (int c = 0; c < n; ++c) {
        W[c * GDM + c] = 1.0 / A[c * GDM + c];
        ld += log(A[c * GDM + c]);
        for (int r = c + 1; r < n; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(A[r * GDM + q], W[q * GDM + c], s);
            W[r * GDM + c] = -s / A[r * GDM + r];
        }
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int q = a; q < n; ++q) s = fma(W[q * GDM + a], W[q * GDM + b], s);
            Ai[a * GDM + b] = s;
            Ai[b * GDM + a] = s;
        }
    return 2.0 * ld;
}

struct PmcPat {          // the group's pattern: observed / missing dimension lists and inv = unshuffle
    int d, no, nu;
    int o[GDM], u[GDM], inv[GDM];
};

// Per basis i (predictCov.m:158-176 / :255-262): lnz, inv(Sigma_oo), ln|Sigma_oo|, R = Sigma_oo \ Sigma_ou,
// CU = Sigma_uu - Sigma_uo R.   rec[i] = [lnz | lnSoo | SooInv (no*no) | R (no*nu) | CU (nu*nu)]
__global__ void k_pmc_prep(PmcPat pt, int m, const double *__restrict__ Sig, const double *__restrict__ iSig,
                           double *__restrict__ rec, int nrec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int d = pt.d, no = pt.no, nu = pt.nu;
    const double *S = Sig + (size_t)i * d * d;
    double A[GDM * GDM], Ai[GDM * GDM];
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) A[a * GDM + b] = iSig[(size_t)i * d * d + a * d + b];
    pmc_chol(A, d);
    double l = 0.0;
    for (int a = 0; a < d; ++a) l += log(A[a * GDM + a]);
    double *r = rec + (size_t)i * nrec;
    r[0] = -l;                                                              // lnz = -1/2 ln|iSigma|   (:165)
    for (int a = 0; a < no; ++a)
        for (int b = 0; b < no; ++b) A[a * GDM + b] = S[pt.o[a] * d + pt.o[b]];
    r[1] = pmc_inv(A, no, Ai);                                              // ln|Sigma_oo|
    double *si = r + 2, *R = si + no * no, *CU = R + no * nu;
    for (int a = 0; a < no; ++a)
        for (int b = 0; b < no; ++b) si[a * no + b] = Ai[a * GDM + b];
    for (int a = 0; a < no; ++a)
        for (int c = 0; c < nu; ++c) {
            double s = 0.0;
            for (int q = 0; q < no; ++q) s = fma(Ai[a * GDM + q], S[pt.o[q] * d + pt.u[c]], s);
            R[a * nu + c] = s;                                              // Sigma(o,o) \ Sigma(o,~o)   (:172)
        }
    for (int a = 0; a < nu; ++a)
        for (int c = 0; c < nu; ++c) {
            double s = S[pt.u[a] * d + pt.u[c]];
            for (int q = 0; q < no; ++q) s = fma(-S[pt.u[a] * d + pt.o[q]], R[q * nu + c], s);
            CU[a * nu + c] = s;                                             // Sigma(~o,~o) - Sigma(~o,o) R   (:174)
        }
}

// Per (row, basis): Ex (without the prior), X_hat, and with input noise Psi_hat.   (:167-176 / :260-274)
// tl != 0: transposed layouts for the register-resident kernels below — X_hat as [row][d][m], Psi_hat as packed lower
// triangles [row][d(d+1)/2][m] (component index fastest: the lanes of a wave run along the components).
#define PLT(r, c) ((r) * ((r) + 1) / 2 + (c))
__global__ void k_pmc_rows(PmcPat pt, int row0, int nrows, int m, int ld, const double *__restrict__ Xr, int de,
                           const double *__restrict__ Psi3, const double *__restrict__ P, const double *__restrict__ Sig,
                           const double *__restrict__ rec, int nrec, double *__restrict__ Ex, double *__restrict__ Xhat,
                           double *__restrict__ Phat, int tl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rr = blockIdx.y;
    if (i >= m || rr >= nrows) return;
    const int row = row0 + rr;
    const int d = pt.d, no = pt.no, nu = pt.nu;
    const double *r = rec + (size_t)i * nrec;
    const double *si = r + 2, *R = si + no * no, *CU = R + no * nu;
    double dl[GDM];
    for (int a = 0; a < no; ++a) dl[a] = Xr[(size_t)row * de + pt.o[a]] - P[(size_t)i * de + pt.o[a]];
    double lp;
    if (!Psi3) {
        double quad = 0.0;
        for (int a = 0; a < no; ++a) {
            double s = 0.0;
            for (int b = 0; b < no; ++b) s = fma(si[a * no + b], dl[b], s);
            quad = fma(dl[a], s, quad);
        }
        lp = -0.5 * quad - 0.5 * r[1];
    } else {
        double M[GDM * GDM];
        const double *ps = Psi3 + (size_t)row * d * d;
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b)
                M[a * GDM + b] = Sig[(size_t)i * d * d + pt.o[a] * d + pt.o[b]] + ps[pt.o[a] + d * pt.o[b]];
        lp = pmc_lognorm(M, dl, no);
    }
    Ex[(size_t)rr * ld + i] = exp(lp);
    double *xh = tl ? Xhat + (size_t)rr * d * m + i : Xhat + ((size_t)rr * m + i) * d;
    const size_t xs = tl ? (size_t)m : 1;
    for (int a = 0; a < no; ++a) xh[pt.o[a] * xs] = Xr[(size_t)row * de + pt.o[a]];
    for (int c = 0; c < nu; ++c) {
        double s = P[(size_t)i * de + pt.u[c]];
        for (int a = 0; a < no; ++a) s = fma(dl[a], R[a * nu + c], s);
        xh[pt.u[c] * xs] = s;
    }
    if (Phat) {
        // B = T Psi_oo T' in [o u] order, T = [I; R'];  Psi_hat(unshuffle, unshuffle) = B;  Psi_hat(u,u) += CU
        const double *ps = Psi3 + (size_t)row * d * d;
        double PR[GDM * GDM];                                   // Psi_oo R  (no x nu)
        for (int a = 0; a < no; ++a)
            for (int c = 0; c < nu; ++c) {
                double s = 0.0;
                for (int q = 0; q < no; ++q) s = fma(ps[pt.o[a] + d * pt.o[q]], R[q * nu + c], s);
                PR[a * GDM + c] = s;
            }
        double *ph = tl ? Phat + (size_t)rr * (d * (d + 1) / 2) * m + i : Phat + ((size_t)rr * m + i) * d * d;
        for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) {
                double v;
                if (a < no && b < no) v = ps[pt.o[a] + d * pt.o[b]];
                else if (a < no) v = PR[a * GDM + (b - no)];
                else if (b < no) v = PR[b * GDM + (a - no)];
                else {
                    v = 0.0;
                    for (int q = 0; q < no; ++q) v = fma(R[q * nu + (a - no)], PR[q * GDM + (b - no)], v);
                }
                if (!tl) ph[pt.inv[a] * d + pt.inv[b]] = v;
                else if (pt.inv[a] >= pt.inv[b]) ph[(size_t)PLT(pt.inv[a], pt.inv[b]) * m] = v;
            }
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nu; ++c) {
                if (!tl) ph[pt.u[a] * d + pt.u[c]] += CU[a * nu + c];
                else if (pt.u[a] >= pt.u[c]) ph[(size_t)PLT(pt.u[a], pt.u[c]) * m] += CU[a * nu + c];
            }
    }
}

// S = base (d x d, row-major stride d) + Psi_hat of component l for this row: without input noise Psi_hat_l is CU_l on
// the (u,u) block, zero elsewhere.
__device__ inline void pmc_add_phat(double *S, const double *base, PmcPat &pt, const double *Phat_row_l,
                                    const double *rec_l) {
    const int d = pt.d, no = pt.no, nu = pt.nu;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) S[a * GDM + b] = base[a * d + b] + (Phat_row_l ? Phat_row_l[a * d + b] : 0.0);
    if (!Phat_row_l) {
        const double *CU = rec_l + 2 + no * no + no * nu;
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nu; ++c) S[pt.u[a] * GDM + pt.u[c]] += CU[a * nu + c];
    }
}

// PHI(row,i) = exp(lnz_i) * sum_j N(X_hat(row,j) - P_i ; Sigma_i + Psi_hat_j(row)) * Pio(row,j)   (:178-207 / :276-318:
// the pair loop adds both orders of every pair and takes the doubled (i,i) term out again)
__global__ void k_pmc_phi(PmcPat pt, int nrows, int m, int ld, int de, const double *__restrict__ P,
                          const double *__restrict__ Sig, const double *__restrict__ rec, int nrec,
                          const double *__restrict__ Pio, const double *__restrict__ Xhat, const double *__restrict__ Phat,
                          double *__restrict__ Phi, int row0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rr = blockIdx.y;
    if (i >= m || rr >= nrows) return;
    const int d = pt.d;
    double acc = 0.0, S[GDM * GDM], dl[GDM];
    for (int j = 0; j < m; ++j) {
        pmc_add_phat(S, Sig + (size_t)i * d * d, pt, Phat ? Phat + ((size_t)rr * m + j) * d * d : nullptr,
                     rec + (size_t)j * nrec);
        const double *xh = Xhat + ((size_t)rr * m + j) * d;
        for (int a = 0; a < d; ++a) dl[a] = xh[a] - P[(size_t)i * de + a];
        acc += exp(pmc_lognorm(S, dl, d)) * Pio[(size_t)rr * ld + j];
    }
    Phi[(size_t)(row0 + rr) * ld + i] = exp(rec[(size_t)i * nrec]) * acc;
}

// Per pair q = i(i+1)/2 + j:  tab[q] = [Cij (d*d) | cij (d) | lnZ | c2 w_i w_j (k) | c2 v_i v_j (k) | c2 iSigma_w(i,j,:) (k)]
//   Cij = inv(iSigma_i + iSigma_j), cij = (P_i iSigma_i + P_j iSigma_j) Cij,
//   lnZ = lnz_i + lnz_j - 1/2 dP (Sigma_i+Sigma_j)^-1 dP' - 1/2 ln|Sigma_i+Sigma_j|      (:180-182,196-197 / :278-280,307-308)
__global__ void k_pmc_pairs(PmcPat pt, int m, int de, int k, const double *__restrict__ P, const double *__restrict__ Sig,
                            const double *__restrict__ iSig, const double *__restrict__ rec, int nrec,
                            const double *__restrict__ w, const double *__restrict__ v, const double *__restrict__ iS,
                            double *__restrict__ tab, int ntab) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long npairs = (long)m * (m + 1) / 2;
    if (q >= npairs) return;
    int i = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((long)(i + 1) * (i + 2) / 2 <= q) ++i;
    while ((long)i * (i + 1) / 2 > q) --i;
    const int j = (int)(q - (long)i * (i + 1) / 2);
    const int d = pt.d;
    double A[GDM * GDM], Ai[GDM * GDM], dl[GDM];
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) A[a * GDM + b] = iSig[(size_t)i * d * d + a * d + b] + iSig[(size_t)j * d * d + a * d + b];
    (void)pmc_inv(A, d, Ai);
    double *t = tab + (size_t)q * ntab;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) t[a * d + b] = Ai[a * GDM + b];
    for (int b = 0; b < d; ++b) {
        double s = 0.0;
        for (int a = 0; a < d; ++a) {
            double pa = 0.0;
            for (int c = 0; c < d; ++c)
                pa += P[(size_t)i * de + c] * iSig[(size_t)i * d * d + c * d + a] + P[(size_t)j * de + c] * iSig[(size_t)j * d * d + c * d + a];
            s = fma(pa, Ai[a * GDM + b], s);
        }
        t[d * d + b] = s;
    }
    for (int a = 0; a < d; ++a) {
        dl[a] = P[(size_t)i * de + a] - P[(size_t)j * de + a];
        for (int b = 0; b < d; ++b) A[a * GDM + b] = Sig[(size_t)i * d * d + a * d + b] + Sig[(size_t)j * d * d + a * d + b];
    }
    t[d * d + d] = rec[(size_t)i * nrec] + rec[(size_t)j * nrec] + pmc_lognorm(A, dl, d);
    const double c2 = (j < i) ? 2.0 : 1.0;
    for (int o = 0; o < k; ++o) {
        t[d * d + d + 1 + o] = c2 * w[i + (size_t)m * o] * w[j + (size_t)m * o];
        t[d * d + d + 1 + k + o] = v ? c2 * v[i + (size_t)m * o] * v[j + (size_t)m * o] : 0.0;
        t[d * d + d + 1 + 2 * k + o] = c2 * iS[i + (size_t)m * j + (size_t)m * m * o];
    }
}

// part[chunk][3k][ldx]: sums over the pairs of a chunk of  Z_q(row) * weights,
//   Z = exp(lnZ) * sum_l N(X_hat(row,l) - cij ; Cij + Psi_hat_l(row)) Pio(row,l)     (:190-201 / :300-313)
// One WAVE per (row, pair chunk), lanes along the m components l of the inner sum (reduced over the wave per pair): a
// NaN-pattern group is often a handful of rows, and with one thread per row a launch kept ten lanes busy for seconds.
__global__ __launch_bounds__(64) void k_pmc_accum(PmcPat pt, int row0, int nrows, int m, int ld, int k, long npairs,
                                                  long pairs_per_chunk, const double *__restrict__ rec, int nrec,
                                                  const double *__restrict__ tab, int ntab, const double *__restrict__ Pio,
                                                  const double *__restrict__ Xhat, const double *__restrict__ Phat, long ldx,
                                                  double *__restrict__ part) {
    const int rr = blockIdx.x, chunk = blockIdx.y, lane = threadIdx.x;
    const int d = pt.d;
    double S[GDM * GDM], dl[GDM];
    const long q0 = (long)chunk * pairs_per_chunk, q1 = min(npairs, q0 + pairs_per_chunk);
    for (int e0 = 0; e0 < 3 * k; e0 += 24) {   // the 3k sums in blocks of 24 (k <= 8: one pass)
        double acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.0;
        for (long q = q0; q < q1; ++q) {
            const double *t = tab + (size_t)q * ntab;
            double ec = 0.0;
            for (int l = lane; l < m; l += 64) {
                pmc_add_phat(S, t, pt, Phat ? Phat + ((size_t)rr * m + l) * d * d : nullptr, rec + (size_t)l * nrec);
                const double *xh = Xhat + ((size_t)rr * m + l) * d;
                for (int a = 0; a < d; ++a) dl[a] = xh[a] - t[d * d + a];
                ec += exp(pmc_lognorm(S, dl, d)) * Pio[(size_t)rr * ld + l];
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ec += __shfl_xor(ec, off, 64);
            const double Z = exp(t[d * d + d]) * ec;
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e0 + e < 3 * k) acc[e] = fma(Z, t[d * d + d + 1 + e0 + e], acc[e]);
        }
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e0 + e < 3 * k) part[((size_t)chunk * 3 * k + e0 + e) * ldx + row0 + rr] = acc[e];
        }
    }
}

#ifndef PMC_WIDE
// ---------------------------------------------------------------------------------------------------------------
// Register-resident form of the two sums above for 2 <= d <= 10 (template D, packed triangles, division-free Cholesky):
//     out(row, r) = exp(lnZ_r) * sum_l N(X_hat(row,l) - c_r ; C_r + Psi_hat_l(row)) * Pio(row,l)
// over a table of records r = [C_r (d x d) | c_r (d) | lnZ_r | weights (nw)]:  the basis pairs with their 3k weights
// (k_pmc_pairs' table -> gamma / VlnS / nu partial sums, predictCov.m:190-201 / :300-313), or the basis functions themselves
// with C = Sigma_i, c = P_i, lnZ = lnz_i and no weights (PHI, predictCov.m:178-207 / :276-318).
// Lanes run along the components l.  The density is exp(-1/2 q) * prod 1/L_cc: no log, no sqrt, no divide per triple.
//
// Without input noise Psi_hat_l is CU_l on the (u,u) block whatever the row is, so the d x d factorisation is shared by all
// rows of the NaN-pattern group (k_pmc_sum_s: one wave per chunk of records, the rows looped inside, per-lane partial sums
// in LDS, lanes re-dealt along the rows for the sum over l).  With input noise it depends on (row, record, component) — the
// reference's O(n m^3 d^3) — and a wave takes one (row, chunk of records) (k_pmc_sum_n).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pmc_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}
template <int D>
__device__ __forceinline__ void pmc_chol_rd(double (&M)[D * (D + 1) / 2], double (&rd)[D], double *prod_rd) {
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double p = M[PLT(c, c)];
#pragma unroll
        for (int q = 0; q < c; ++q) p = fma(-M[PLT(c, q)], M[PLT(c, q)], p);
        const double inv = pmc_rsqrt(p);
        rd[c] = inv;
        prod *= inv;
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double s = M[PLT(r, c)];
#pragma unroll
            for (int q = 0; q < c; ++q) s = fma(-M[PLT(r, q)], M[PLT(c, q)], s);
            M[PLT(r, c)] = s * inv;
        }
    }
    *prod_rd = prod;
}

// CUT[e][l]: Psi_hat_l without input noise as a packed triangle (CU_l on the (u,u) block, zero elsewhere)
__global__ void k_pmc_cut(PmcPat pt, int m, const double *__restrict__ rec, int nrec, double *__restrict__ CUT) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= m) return;
    const int d = pt.d, no = pt.no, nu = pt.nu, np = d * (d + 1) / 2;
    for (int e = 0; e < np; ++e) CUT[(size_t)e * m + l] = 0.0;
    const double *CU = rec + (size_t)l * nrec + 2 + no * no + no * nu;
    for (int a = 0; a < nu; ++a)
        for (int c = 0; c < nu; ++c)
            if (pt.u[a] >= pt.u[c]) CUT[(size_t)PLT(pt.u[a], pt.u[c]) * m + l] = CU[a * nu + c];
}
// record table of the PHI sum: [Sigma_i (d x d) | P_i (d) | lnz_i]
__global__ void k_pmc_phitab(int m, int d, int de, const double *__restrict__ P, const double *__restrict__ Sig,
                             const double *__restrict__ rec, int nrec, double *__restrict__ ptab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double *t = ptab + (size_t)i * (d * d + d + 1);
    for (int e = 0; e < d * d; ++e) t[e] = Sig[(size_t)i * d * d + e];
    for (int a = 0; a < d; ++a) t[d * d + a] = P[(size_t)i * de + a];
    t[d * d + d] = rec[(size_t)i * nrec];
}

// k_pmc_prep for d <= 10 in registers: ONE Cholesky of Sigma_i permuted to [observed | missing] order gives every block of the record -
//   L = [L_oo 0; L_uo L_uu]:  ln|Sigma_oo| = 2 sum_{c<no} ln L_cc,  inv(Sigma_oo) = W_oo' W_oo (W = inv(L), W_oo its leading block),
//   R = Sigma_oo \ Sigma_ou = W_oo' L_uo',  CU = Sigma_uu - Sigma_uo R = L_uu L_uu' (the Schur complement),  lnz = 1/2 ln|Sigma_i| = sum_c ln L_cc
// - with compile-time loops over D and the split point `no` as a predicate, so nothing is indexed at run time (the scratch-resident
// version above runs its four d x d temporaries through scratch memory: 157 us per launch at d = 10, m = 200, a fifth of a
// predict() call over 39 NaN patterns).
template <int D>
__global__ void k_pmc_prep_t(PmcPat pt, int m, const double *__restrict__ Sig, double *__restrict__ rec, int nrec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int no = pt.no, nu = pt.nu;
    const double *S = Sig + (size_t)i * D * D;
    int perm[D];
#pragma unroll
    for (int a = 0; a < D; ++a) perm[a] = a < no ? pt.o[a < no ? a : 0] : pt.u[a >= no ? a - no : 0];
    double L[D * (D + 1) / 2], W[D * (D + 1) / 2], rd[D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) L[PLT(a, b)] = S[perm[a] * D + perm[b]];
    double lz = 0.0, lo = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {                        // Cholesky, packed lower triangle
        double p = L[PLT(c, c)];
#pragma unroll
        for (int q = 0; q < c; ++q) p = fma(-L[PLT(c, q)], L[PLT(c, q)], p);
        const double dd = sqrt(p);
        L[PLT(c, c)] = dd;
        rd[c] = 1.0 / dd;
        const double lg = log(dd);
        lz += lg;
        if (c < no) lo += lg;
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double s = L[PLT(r, c)];
#pragma unroll
            for (int q = 0; q < c; ++q) s = fma(-L[PLT(r, q)], L[PLT(c, q)], s);
            L[PLT(r, c)] = s * rd[c];
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {                        // W = inv(L)
        W[PLT(c, c)] = rd[c];
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double s = 0.0;
#pragma unroll
            for (int q = c; q < r; ++q) s = fma(L[PLT(r, q)], W[PLT(q, c)], s);
            W[PLT(r, c)] = -s * rd[r];
        }
    }
    double *r = rec + (size_t)i * nrec;
    r[0] = lz;                                           // lnz = -1/2 ln|iSigma| = 1/2 ln|Sigma|   (:165)
    r[1] = 2.0 * lo;                                     // ln|Sigma_oo|
    double *si = r + 2, *R = si + no * no, *CU = R + no * nu;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;                              // inv(Sigma_oo)[a][b] = sum_{q = a .. no-1} W[q][a] W[q][b]
#pragma unroll
            for (int q = a; q < D; ++q) s = fma(q < no ? W[PLT(q, a)] : 0.0, W[PLT(q, b)], s);
            if (a < no) { si[a * no + b] = s; si[b * no + a] = s; }
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int ru = 0; ru < D; ++ru) {
            double s = 0.0;                              // R[a][ru - no] = sum_{q = a .. no-1} W[q][a] L[ru][q]      (:172)
#pragma unroll
            for (int q = a; q < D; ++q)
                if (q <= ru) s = fma(q < no ? W[PLT(q, a)] : 0.0, L[PLT(ru > q ? ru : q, ru > q ? q : ru)], s);
            if (a < no && ru >= no) R[a * nu + (ru - no)] = s;
        }
#pragma unroll
    for (int ra = 0; ra < D; ++ra)
#pragma unroll
        for (int rc = 0; rc <= ra; ++rc) {
            double s = 0.0;                              // CU = L_uu L_uu'                                           (:174)
#pragma unroll
            for (int q = 0; q <= rc; ++q) s = fma(q >= no ? L[PLT(ra, q)] : 0.0, L[PLT(rc, q)], s);
            if (rc >= no) { CU[(ra - no) * nu + (rc - no)] = s; CU[(rc - no) * nu + (ra - no)] = s; }
        }
}

template <int D>
__global__ __launch_bounds__(64) void k_pmc_sum_s(int nrows, int row0, int m, int ld, long R, long rec_per_chunk,
                                                   const double *__restrict__ tab, int ntab, int nw,
                                                   const double *__restrict__ Pio, const double *__restrict__ XhT,
                                                   const double *__restrict__ CUT, double *__restrict__ Phi, long ldx,
                                                   double *__restrict__ part) {
    constexpr int NP = D * (D + 1) / 2;
    __shared__ double ecs[64 * 65];
    const int lane = threadIdx.x, chunk = blockIdx.x;
    const long r0 = (long)chunk * rec_per_chunk, r1 = min(R, r0 + rec_per_chunk);
    double acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.0;
    for (long r = r0; r < r1; ++r) {
        const double *t = tab + (size_t)r * ntab;
        for (int rr = 0; rr < nrows; ++rr) ecs[rr * 65 + lane] = 0.0;
        for (int l0 = 0; l0 < m; l0 += 64) {
            const int l = l0 + lane, lc = min(l, m - 1);
            double M[NP], rd[D], prd;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) M[PLT(a, b)] = t[a * D + b] + CUT[(size_t)PLT(a, b) * m + lc];
            pmc_chol_rd<D>(M, rd, &prd);
            for (int rr = 0; rr < nrows; ++rr) {
                double q = 0.0, y[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double s = XhT[((size_t)rr * D + a) * m + lc] - t[D * D + a];
#pragma unroll
                    for (int c = 0; c < a; ++c) s = fma(-M[PLT(a, c)], y[c], s);
                    y[a] = s * rd[a];
                    q = fma(y[a], y[a], q);
                }
                if (l < m) ecs[rr * 65 + lane] += exp(-0.5 * q) * prd * Pio[(size_t)rr * ld + l];
            }
        }
        __syncthreads();
        if (lane < nrows) {                      // lanes re-dealt along the rows: the sum over the components, in lane order
            double ec = 0.0;
            for (int j = 0; j < 64; ++j) ec += ecs[lane * 65 + j];
            const double Z = exp(t[D * D + D]) * ec;
            if (nw == 0) Phi[(size_t)(row0 + lane) * ld + r] = Z;
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e < nw) acc[e] = fma(Z, t[D * D + D + 1 + e], acc[e]);
        }
        __syncthreads();
    }
    if (nw > 0 && lane < nrows) {
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) part[((size_t)chunk * nw + e) * ldx + row0 + lane] = acc[e];
    }
}

template <int D>
__global__ __launch_bounds__(64) void k_pmc_sum_n(int nrows, int row0, int m, int ld, long R, long rec_per_chunk,
                                                   const double *__restrict__ tab, int ntab, int nw,
                                                   const double *__restrict__ Pio, const double *__restrict__ XhT,
                                                   const double *__restrict__ PhT, double *__restrict__ Phi, long ldx,
                                                   double *__restrict__ part) {
    constexpr int NP = D * (D + 1) / 2;
    const int lane = threadIdx.x, rr = blockIdx.x, chunk = blockIdx.y;
    const long r0 = (long)chunk * rec_per_chunk, r1 = min(R, r0 + rec_per_chunk);
    double acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.0;
    for (long r = r0; r < r1; ++r) {
        const double *t = tab + (size_t)r * ntab;
        double ec = 0.0;
        for (int l0 = 0; l0 < m; l0 += 64) {
            const int l = l0 + lane, lc = min(l, m - 1);
            double M[NP], rd[D], prd;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) M[PLT(a, b)] = t[a * D + b] + PhT[((size_t)rr * NP + PLT(a, b)) * m + lc];
            pmc_chol_rd<D>(M, rd, &prd);
            double q = 0.0, y[D];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                double s = XhT[((size_t)rr * D + a) * m + lc] - t[D * D + a];
#pragma unroll
                for (int c = 0; c < a; ++c) s = fma(-M[PLT(a, c)], y[c], s);
                y[a] = s * rd[a];
                q = fma(y[a], y[a], q);
            }
            if (l < m) ec += exp(-0.5 * q) * prd * Pio[(size_t)rr * ld + l];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ec += __shfl_xor(ec, off, 64);
        const double Z = exp(t[D * D + D]) * ec;
        if (nw == 0 && lane == 0) Phi[(size_t)(row0 + rr) * ld + r] = Z;
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) acc[e] = fma(Z, t[D * D + D + 1 + e], acc[e]);
    }
    if (nw > 0 && lane == 0) {
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) part[((size_t)chunk * nw + e) * ldx + row0 + rr] = acc[e];
    }
}

// out = the weighted sums (nw > 0, part) or PHI (nw == 0) over the record table; false when d is outside 2..10
static bool pmc_sum(hipStream_t st, int d, bool noisy, int nrows, int row0, int m, int ld, long R, int nchunk, const double *tab,
                    int ntab, int nw, const double *Pio, const double *XhT, const double *PsT, double *Phi, long ldx,
                    double *part) {
    const long rpc = (R + nchunk - 1) / nchunk;
    const int nch = (int)((R + rpc - 1) / rpc);
#define PMC_CASE(DD)                                                                                                      \
    case DD:                                                                                                              \
        if (noisy)                                                                                                        \
            hipLaunchKernelGGL((k_pmc_sum_n<DD>), dim3(nrows, nch), dim3(64), 0, st, nrows, row0, m, ld, R, rpc, tab, ntab, nw, \
                               Pio, XhT, PsT, Phi, ldx, part);                                                            \
        else                                                                                                              \
            hipLaunchKernelGGL((k_pmc_sum_s<DD>), dim3(nch), dim3(64), 0, st, nrows, row0, m, ld, R, rpc, tab, ntab, nw, Pio,   \
                               XhT, PsT, Phi, ldx, part);                                                                 \
        return true;
    switch (d) {
        PMC_CASE(2) PMC_CASE(3) PMC_CASE(4) PMC_CASE(5) PMC_CASE(6) PMC_CASE(7) PMC_CASE(8) PMC_CASE(9) PMC_CASE(10)
        default:   // 10 < d <= 32: four components per wave on 4 x 4 MFMA tiles (k_pmc4.hip)
            return launch_pmc4_sum(st, d, noisy, nrows, row0, m, ld, R, nchunk, tab, ntab, nw, Pio, XhT, PsT, Phi, ldx, part);
    }
#undef PMC_CASE
}

#endif   // !PMC_WIDE

// ---- host side -------------------------------------------------------------------------------------
// One NaN-pattern group.  obs: bit c set = dimension c observed.  Sig/iSig: m x d*d (k_gen_prep).  Work buffers are
// allocated by the caller: rec (m*nrec), tab (npairs*ntab), Ex/Pio (rows_blk*ld each), Xhat (rows_blk*m*d),
// Phat (rows_blk*m*d*d, only with Psi3), part (nchunk*3k*ldx).  Writes PHI rows [0,n) and part; the caller sums part.
#ifndef PMC_WIDE
int pmc_rec_len(int d, unsigned long long obs) {
    int no = 0;
    for (int c = 0; c < d; ++c) no += (obs >> c) & 1ull;
    const int nu = d - no;
    return 2 + no * no + no * nu + nu * nu;
}
bool pmc_fast(int d, int k) { return d >= 2 && (d <= 10 || pmc4_available(d)) && k <= 8; }   // register-resident kernels (rows_blk <= 64 then); more outputs: the scratch kernels, 24 sums per pass
#endif
// work2: m * (d(d+1)/2 + d*d + d + 1) doubles, used by the register-resident route
#ifdef PMC_WIDE
void launch_pmc_wide(hipStream_t st, unsigned long long obs, int n, long ldx, int m, int ld, int d, int de, int k, const double *Xr,
                const double *Psi3, const double *P, const double *Sig, const double *iSig, const double *priors,
                const double *w, const double *v, const double *iS, int rows_blk, double *rec, double *tab, double *Ex,
                double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                double *work2, bool tab_ready) 
#else
void launch_pmc(hipStream_t st, unsigned long long obs, int n, long ldx, int m, int ld, int d, int de, int k, const double *Xr,
                const double *Psi3, const double *P, const double *Sig, const double *iSig, const double *priors,
                const double *w, const double *v, const double *iS, int rows_blk, double *rec, double *tab, double *Ex,
                double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                double *work2, bool tab_ready) 
#endif
{
    PmcPat pt;
    pt.d = d; pt.no = 0; pt.nu = 0;
    for (int c = 0; c < d; ++c) {
        if ((obs >> c) & 1ull) pt.o[pt.no++] = c;
        else pt.u[pt.nu++] = c;
    }
    int perm[GDM];
    for (int a = 0; a < pt.no; ++a) perm[a] = pt.o[a];
    for (int a = 0; a < pt.nu; ++a) perm[pt.no + a] = pt.u[a];
    for (int a = 0; a < d; ++a) pt.inv[perm[a]] = a;                        // [~,unshuffle] = sort([find(o) find(~o)])
    const int nrec = pmc_rec_len(d, obs), ntab = d * d + d + 1 + 3 * k;
    const long npairs = (long)m * (m + 1) / 2;
    // GPZ_PMC_SCRATCH=1 (developer switch): keep the scratch-resident kernels, to compare the two routes on one input
#ifdef PMC_WIDE
    const bool fast = false;
#else
    const bool fast = pmc_fast(d, k) && work2 != nullptr && rows_blk <= 64 && !gpz_opts().pmc_scratch;
#endif
#ifndef PMC_WIDE
#define PREP_CASE(DD) case DD: hipLaunchKernelGGL(k_pmc_prep_t<DD>, dim3((m + 63) / 64), dim3(64), 0, st, pt, m, Sig, rec, nrec); break;
    switch (gpz_opts().pmc_prep_scratch ? 0 : d) {
        PREP_CASE(2) PREP_CASE(3) PREP_CASE(4) PREP_CASE(5) PREP_CASE(6) PREP_CASE(7) PREP_CASE(8) PREP_CASE(9) PREP_CASE(10)
        default: hipLaunchKernelGGL(k_pmc_prep, dim3((m + 63) / 64), dim3(64), 0, st, pt, m, Sig, iSig, rec, nrec); break;
    }
#undef PREP_CASE
#else
    hipLaunchKernelGGL(k_pmc_prep, dim3((m + 63) / 64), dim3(64), 0, st, pt, m, Sig, iSig, rec, nrec);
#endif
    if (!tab_ready)   // the pair table depends on theta, w, iSigma_w only: predict.m calls once per NaN-pattern group with the same model
        hipLaunchKernelGGL(k_pmc_pairs, dim3((unsigned)((npairs + 63) / 64)), dim3(64), 0, st, pt, m, de, k, P, Sig, iSig,
                           (const double *)rec, nrec, w, v, iS, tab, ntab);
    double *CUT = work2, *ptab = work2 ? work2 + (size_t)m * (d * (d + 1) / 2) : nullptr;
#ifndef PMC_WIDE
    if (fast) {
        if (!Psi3) hipLaunchKernelGGL(k_pmc_cut, dim3((m + 63) / 64), dim3(64), 0, st, pt, m, (const double *)rec, nrec, CUT);
        hipLaunchKernelGGL(k_pmc_phitab, dim3((m + 63) / 64), dim3(64), 0, st, m, d, de, P, Sig, (const double *)rec, nrec, ptab);
    }
#endif
    for (int row0 = 0; row0 < n; row0 += rows_blk) {
        const int nr = (n - row0 < rows_blk) ? n - row0 : rows_blk;
        hipLaunchKernelGGL(k_pmc_rows, dim3((m + 63) / 64, nr), dim3(64), 0, st, pt, row0, nr, m, ld, Xr, de, Psi3, P, Sig,
                           (const double *)rec, nrec, Ex, Xhat, Psi3 ? Phat : nullptr, fast ? 1 : 0);
        launch_pm_pio(st, Ex, ld, nr, m, priors, Pio);
#ifndef PMC_WIDE
        if (fast) {
            const double *PsT = Psi3 ? Phat : CUT;
            pmc_sum(st, d, Psi3 != nullptr, nr, row0, m, ld, (long)m, m < 256 ? m : 256, ptab, d * d + d + 1, 0, Pio, Xhat, PsT,
                    Phi, ldx, nullptr);
            pmc_sum(st, d, Psi3 != nullptr, nr, row0, m, ld, npairs, nchunk, tab, ntab, 3 * k, Pio, Xhat, PsT, nullptr, ldx,
                    part);
            continue;
        }
#endif
        hipLaunchKernelGGL(k_pmc_phi, dim3((m + 63) / 64, nr), dim3(64), 0, st, pt, nr, m, ld, de, P, Sig, (const double *)rec,
                           nrec, (const double *)Pio, (const double *)Xhat, (const double *)(Psi3 ? Phat : nullptr), Phi, row0);
        hipLaunchKernelGGL(k_pmc_accum, dim3(nr, nchunk), dim3(64), 0, st, pt, row0, nr, m, ld, k, npairs,
                           pairs_per_chunk, (const double *)rec, nrec, (const double *)tab, ntab, (const double *)Pio,
                           (const double *)Xhat, (const double *)(Psi3 ? Phat : nullptr), ldx, part);
    }
}
