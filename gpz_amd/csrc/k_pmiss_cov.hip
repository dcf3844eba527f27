// Prediction for inputs with missing dimensions, covariance kinds (GC/VC):
//   predictMissing       predictCov.m:134-229   (no input noise)
//   predictNoisyMissing  predictCov.m:231-337   (input noise Psi, d x d x n)
// for one group of rows that share a NaN pattern (o = observed dimensions, u = missing ones).
//
// The reference conditions every basis function on the observed dimensions (X_hat, Psi_hat) and then, for every
// basis pair (i,j), sums a d-dimensional Gaussian over all m conditioned components: O(n m^3 d^3) interpreted work
// (a d x d factorisation per (row, pair, component) once there is input noise).  This is a correctness path:
// runtime d <= 20, the small matrices live in per-thread scratch, one thread per (row, basis) or per (row, pair chunk);
// sums over pairs are ordered (chunk slabs + fixed-order sum).
//
// Kept quirk (predictCov.m:266-268): with input noise the block T*Psi_oo*T' (in [o u] order) is ASSIGNED through
// `unshuffle`, the inverse of the permutation [find(o) find(~o)] — the intended placement only when that permutation
// is its own inverse.
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define GDM 20

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
__global__ void k_pmc_rows(PmcPat pt, int row0, int nrows, int m, int ld, const double *__restrict__ Xr, int de,
                           const double *__restrict__ Psi3, const double *__restrict__ P, const double *__restrict__ Sig,
                           const double *__restrict__ rec, int nrec, double *__restrict__ Ex, double *__restrict__ Xhat,
                           double *__restrict__ Phat) {
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
    double *xh = Xhat + ((size_t)rr * m + i) * d;
    for (int a = 0; a < no; ++a) xh[pt.o[a]] = Xr[(size_t)row * de + pt.o[a]];
    for (int c = 0; c < nu; ++c) {
        double s = P[(size_t)i * de + pt.u[c]];
        for (int a = 0; a < no; ++a) s = fma(dl[a], R[a * nu + c], s);
        xh[pt.u[c]] = s;
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
        double *ph = Phat + ((size_t)rr * m + i) * d * d;
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
                ph[pt.inv[a] * d + pt.inv[b]] = v;
            }
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nu; ++c) ph[pt.u[a] * d + pt.u[c]] += CU[a * nu + c];
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
    double acc[24], S[GDM * GDM], dl[GDM];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.0;
    const long q0 = (long)chunk * pairs_per_chunk, q1 = min(npairs, q0 + pairs_per_chunk);
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
            if (e < 3 * k) acc[e] = fma(Z, t[d * d + d + 1 + e], acc[e]);
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < 3 * k) part[((size_t)chunk * 3 * k + e) * ldx + row0 + rr] = acc[e];
    }
}

// ---- host side -------------------------------------------------------------------------------------
// One NaN-pattern group.  obs: bit c set = dimension c observed.  Sig/iSig: m x d*d (k_gen_prep).  Work buffers are
// allocated by the caller: rec (m*nrec), tab (npairs*ntab), Ex/Pio (rows_blk*ld each), Xhat (rows_blk*m*d),
// Phat (rows_blk*m*d*d, only with Psi3), part (nchunk*3k*ldx).  Writes PHI rows [0,n) and part; the caller sums part.
int pmc_rec_len(int d, unsigned obs) {
    int no = 0;
    for (int c = 0; c < d; ++c) no += (obs >> c) & 1u;
    const int nu = d - no;
    return 2 + no * no + no * nu + nu * nu;
}
void launch_pmc(hipStream_t st, unsigned obs, int n, long ldx, int m, int ld, int d, int de, int k, const double *Xr,
                const double *Psi3, const double *P, const double *Sig, const double *iSig, const double *priors,
                const double *w, const double *v, const double *iS, int rows_blk, double *rec, double *tab, double *Ex,
                double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi) {
    PmcPat pt;
    pt.d = d; pt.no = 0; pt.nu = 0;
    for (int c = 0; c < d; ++c) {
        if ((obs >> c) & 1u) pt.o[pt.no++] = c;
        else pt.u[pt.nu++] = c;
    }
    int perm[GDM];
    for (int a = 0; a < pt.no; ++a) perm[a] = pt.o[a];
    for (int a = 0; a < pt.nu; ++a) perm[pt.no + a] = pt.u[a];
    for (int a = 0; a < d; ++a) pt.inv[perm[a]] = a;                        // [~,unshuffle] = sort([find(o) find(~o)])
    const int nrec = pmc_rec_len(d, obs), ntab = d * d + d + 1 + 3 * k;
    const long npairs = (long)m * (m + 1) / 2;
    hipLaunchKernelGGL(k_pmc_prep, dim3((m + 63) / 64), dim3(64), 0, st, pt, m, Sig, iSig, rec, nrec);
    hipLaunchKernelGGL(k_pmc_pairs, dim3((unsigned)((npairs + 63) / 64)), dim3(64), 0, st, pt, m, de, k, P, Sig, iSig,
                       (const double *)rec, nrec, w, v, iS, tab, ntab);
    for (int row0 = 0; row0 < n; row0 += rows_blk) {
        const int nr = (n - row0 < rows_blk) ? n - row0 : rows_blk;
        hipLaunchKernelGGL(k_pmc_rows, dim3((m + 63) / 64, nr), dim3(64), 0, st, pt, row0, nr, m, ld, Xr, de, Psi3, P, Sig,
                           (const double *)rec, nrec, Ex, Xhat, Psi3 ? Phat : nullptr);
        launch_pm_pio(st, Ex, ld, nr, m, priors, Pio);
        hipLaunchKernelGGL(k_pmc_phi, dim3((m + 63) / 64, nr), dim3(64), 0, st, pt, nr, m, ld, de, P, Sig, (const double *)rec,
                           nrec, (const double *)Pio, (const double *)Xhat, (const double *)(Psi3 ? Phat : nullptr), Phi, row0);
        hipLaunchKernelGGL(k_pmc_accum, dim3(nr, nchunk), dim3(64), 0, st, pt, row0, nr, m, ld, k, npairs,
                           pairs_per_chunk, (const double *)rec, nrec, (const double *)tab, ntab, (const double *)Pio,
                           (const double *)Xhat, (const double *)(Psi3 ? Phat : nullptr), ldx, part);
    }
}
