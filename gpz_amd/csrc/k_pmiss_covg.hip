// Prediction for inputs with missing dimensions, covariance kinds (GC/VC), inputs of ANY width (used for d > 64):
//   predictMissing       predictCov.m:134-229   (no input noise)
//   predictNoisyMissing  predictCov.m:231-337   (input noise Psi, d x d x n)
// The reference is generic in d.  k_pmiss_cov.hip keeps a thread's d x d temporaries in registers, MFMA tiles or (32 < d <= 64)
// compiler-managed scratch; here they live in a workspace in device memory, `pmg_ws_per_thread(d)` doubles per thread, and every
// launch covers at most `ws_threads` items.  One thread per item as in the scratch-resident kernels, same formulas, same order of
// the sums over pairs (chunk slabs + fixed-order sum): the slow, correct route - a d x d factorisation per (row, pair, component)
// is O(n m^3 d^3) work in the reference as well.
// Kept quirk (predictCov.m:266-268): see k_pmiss_cov.hip.
#include "gpz_dev.h"
#include "gpz_kernels.h"

struct PmgPat {          // the group's pattern: observed / missing dimension lists and inv = unshuffle (device memory)
    int d, no, nu;
    const int *o, *u, *inv;
};

size_t pmg_ws_per_thread(int d) { return (size_t)3 * d * d + 2 * (size_t)d; }

// lower Cholesky in place, row-major, leading dimension ld
__device__ inline void pmg_chol(double *M, int n, int ld) {
    for (int c = 0; c < n; ++c) {
        double p = M[(size_t)c * ld + c];
        for (int q = 0; q < c; ++q) p = fma(-M[(size_t)c * ld + q], M[(size_t)c * ld + q], p);
        const double dd = sqrt(p);
        M[(size_t)c * ld + c] = dd;
        for (int r = c + 1; r < n; ++r) {
            double s = M[(size_t)r * ld + c];
            for (int q = 0; q < c; ++q) s = fma(-M[(size_t)r * ld + q], M[(size_t)c * ld + q], s);
            M[(size_t)r * ld + c] = s / dd;
        }
    }
}
// -1/2 dl' S^-1 dl - 1/2 ln|S| for SPD S (destroyed); y: n doubles of workspace
__device__ inline double pmg_lognorm(double *S, const double *dl, int n, int ld, double *y) {
    pmg_chol(S, n, ld);
    double quad = 0.0, hl = 0.0;
    for (int r = 0; r < n; ++r) {
        double s = dl[r];
        for (int c = 0; c < r; ++c) s = fma(-S[(size_t)r * ld + c], y[c], s);
        y[r] = s / S[(size_t)r * ld + r];
        quad = fma(y[r], y[r], quad);
        hl += log(S[(size_t)r * ld + r]);
    }
    return -0.5 * quad - hl;
}
// Ai = inv(A) for SPD A (destroyed); W: n*ld doubles of workspace; returns ln|A|
__device__ inline double pmg_inv(double *A, int n, int ld, double *Ai, double *W) {
    pmg_chol(A, n, ld);
    double lg = 0.0;
    for (int c = 0; c < n; ++c) {
        W[(size_t)c * ld + c] = 1.0 / A[(size_t)c * ld + c];
        lg += log(A[(size_t)c * ld + c]);
        for (int r = c + 1; r < n; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(A[(size_t)r * ld + q], W[(size_t)q * ld + c], s);
            W[(size_t)r * ld + c] = -s / A[(size_t)r * ld + r];
        }
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int q = a; q < n; ++q) s = fma(W[(size_t)q * ld + a], W[(size_t)q * ld + b], s);
            Ai[(size_t)a * ld + b] = s;
            Ai[(size_t)b * ld + a] = s;
        }
    return 2.0 * lg;
}

// Per basis i (predictCov.m:158-176 / :255-262):  rec[i] = [lnz | lnSoo | SooInv (no*no) | R (no*nu) | CU (nu*nu)]
__global__ void k_pmg_prep(PmgPat pt, int m, const double *__restrict__ Sig, const double *__restrict__ iSig, double *__restrict__ rec,
                           int nrec, int base, int count, double *__restrict__ ws, size_t wsp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int i = base + t;
    const int d = pt.d, no = pt.no, nu = pt.nu;
    double *A = ws + (size_t)t * wsp, *Ai = A + (size_t)d * d, *W = Ai + (size_t)d * d;
    const double *S = Sig + (size_t)i * d * d;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) A[(size_t)a * d + b] = iSig[(size_t)i * d * d + a * d + b];
    pmg_chol(A, d, d);
    double l = 0.0;
    for (int a = 0; a < d; ++a) l += log(A[(size_t)a * d + a]);
    double *r = rec + (size_t)i * nrec;
    r[0] = -l;                                                              // lnz = -1/2 ln|iSigma|   (:165)
    for (int a = 0; a < no; ++a)
        for (int b = 0; b < no; ++b) A[(size_t)a * d + b] = S[(size_t)pt.o[a] * d + pt.o[b]];
    r[1] = pmg_inv(A, no, d, Ai, W);                                        // ln|Sigma_oo|
    double *si = r + 2, *R = si + (size_t)no * no, *CU = R + (size_t)no * nu;
    for (int a = 0; a < no; ++a)
        for (int b = 0; b < no; ++b) si[(size_t)a * no + b] = Ai[(size_t)a * d + b];
    for (int a = 0; a < no; ++a)
        for (int c = 0; c < nu; ++c) {
            double s = 0.0;
            for (int q = 0; q < no; ++q) s = fma(Ai[(size_t)a * d + q], S[(size_t)pt.o[q] * d + pt.u[c]], s);
            R[(size_t)a * nu + c] = s;                                      // Sigma(o,o) \ Sigma(o,~o)   (:172)
        }
    for (int a = 0; a < nu; ++a)
        for (int c = 0; c < nu; ++c) {
            double s = S[(size_t)pt.u[a] * d + pt.u[c]];
            for (int q = 0; q < no; ++q) s = fma(-S[(size_t)pt.u[a] * d + pt.o[q]], R[(size_t)q * nu + c], s);
            CU[(size_t)a * nu + c] = s;                                     // Sigma(~o,~o) - Sigma(~o,o) R   (:174)
        }
}

// Per (row, basis): Ex (without the prior), X_hat, and with input noise Psi_hat   (:167-176 / :260-274); rr = row within the block
__global__ void k_pmg_rows(PmgPat pt, int row, int rr, int m, int ld, const double *__restrict__ Xr, int de,
                           const double *__restrict__ Psi3, const double *__restrict__ P, const double *__restrict__ Sig,
                           const double *__restrict__ rec, int nrec, double *__restrict__ Ex, double *__restrict__ Xhat,
                           double *__restrict__ Phat, int base, int count, double *__restrict__ ws, size_t wsp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int i = base + t;
    const int d = pt.d, no = pt.no, nu = pt.nu;
    double *M = ws + (size_t)t * wsp, *PR = M + (size_t)d * d, *dl = PR + (size_t)d * d, *y = dl + d;
    const double *r = rec + (size_t)i * nrec;
    const double *si = r + 2, *R = si + (size_t)no * no, *CU = R + (size_t)no * nu;
    for (int a = 0; a < no; ++a) dl[a] = Xr[(size_t)row * de + pt.o[a]] - P[(size_t)i * de + pt.o[a]];
    double lp;
    if (!Psi3) {
        double quad = 0.0;
        for (int a = 0; a < no; ++a) {
            double s = 0.0;
            for (int b = 0; b < no; ++b) s = fma(si[(size_t)a * no + b], dl[b], s);
            quad = fma(dl[a], s, quad);
        }
        lp = -0.5 * quad - 0.5 * r[1];
    } else {
        const double *ps = Psi3 + (size_t)row * d * d;
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b)
                M[(size_t)a * d + b] = Sig[(size_t)i * d * d + (size_t)pt.o[a] * d + pt.o[b]] + ps[pt.o[a] + (size_t)d * pt.o[b]];
        lp = pmg_lognorm(M, dl, no, d, y);
    }
    Ex[(size_t)rr * ld + i] = exp(lp);
    double *xh = Xhat + ((size_t)rr * m + i) * d;
    for (int a = 0; a < no; ++a) xh[pt.o[a]] = Xr[(size_t)row * de + pt.o[a]];
    for (int c = 0; c < nu; ++c) {
        double s = P[(size_t)i * de + pt.u[c]];
        for (int a = 0; a < no; ++a) s = fma(dl[a], R[(size_t)a * nu + c], s);
        xh[pt.u[c]] = s;
    }
    if (Phat) {
        // B = T Psi_oo T' in [o u] order, T = [I; R'];  Psi_hat(unshuffle, unshuffle) = B;  Psi_hat(u,u) += CU
        const double *ps = Psi3 + (size_t)row * d * d;
        for (int a = 0; a < no; ++a)
            for (int c = 0; c < nu; ++c) {
                double s = 0.0;
                for (int q = 0; q < no; ++q) s = fma(ps[pt.o[a] + (size_t)d * pt.o[q]], R[(size_t)q * nu + c], s);
                PR[(size_t)a * d + c] = s;                                  // Psi_oo R  (no x nu)
            }
        double *ph = Phat + ((size_t)rr * m + i) * d * d;
        for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) {
                double v;
                if (a < no && b < no) v = ps[pt.o[a] + (size_t)d * pt.o[b]];
                else if (a < no) v = PR[(size_t)a * d + (b - no)];
                else if (b < no) v = PR[(size_t)b * d + (a - no)];
                else {
                    v = 0.0;
                    for (int q = 0; q < no; ++q) v = fma(R[(size_t)q * nu + (a - no)], PR[(size_t)q * d + (b - no)], v);
                }
                ph[(size_t)pt.inv[a] * d + pt.inv[b]] = v;
            }
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nu; ++c) ph[(size_t)pt.u[a] * d + pt.u[c]] += CU[(size_t)a * nu + c];
    }
}

// S = base (d x d) + Psi_hat of component l for this row: without input noise Psi_hat_l is CU_l on the (u,u) block, zero elsewhere
__device__ inline void pmg_add_phat(double *S, const double *base, const PmgPat &pt, const double *Phat_row_l, const double *rec_l) {
    const int d = pt.d, no = pt.no, nu = pt.nu;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) S[(size_t)a * d + b] = base[(size_t)a * d + b] + (Phat_row_l ? Phat_row_l[(size_t)a * d + b] : 0.0);
    if (!Phat_row_l) {
        const double *CU = rec_l + 2 + (size_t)no * no + (size_t)no * nu;
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nu; ++c) S[(size_t)pt.u[a] * d + pt.u[c]] += CU[(size_t)a * nu + c];
    }
}

// PHI(row,i) = exp(lnz_i) * sum_j N(X_hat(row,j) - P_i ; Sigma_i + Psi_hat_j(row)) * Pio(row,j)   (:178-207 / :276-318)
__global__ void k_pmg_phi(PmgPat pt, int row, int rr, int m, int ld, int de, const double *__restrict__ P,
                          const double *__restrict__ Sig, const double *__restrict__ rec, int nrec, const double *__restrict__ Pio,
                          const double *__restrict__ Xhat, const double *__restrict__ Phat, double *__restrict__ Phi, int base,
                          int count, double *__restrict__ ws, size_t wsp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int i = base + t;
    const int d = pt.d;
    double *S = ws + (size_t)t * wsp, *dl = S + (size_t)d * d, *y = dl + d;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) {
        pmg_add_phat(S, Sig + (size_t)i * d * d, pt, Phat ? Phat + ((size_t)rr * m + j) * d * d : nullptr, rec + (size_t)j * nrec);
        const double *xh = Xhat + ((size_t)rr * m + j) * d;
        for (int a = 0; a < d; ++a) dl[a] = xh[a] - P[(size_t)i * de + a];
        acc += exp(pmg_lognorm(S, dl, d, d, y)) * Pio[(size_t)rr * ld + j];
    }
    Phi[(size_t)row * ld + i] = exp(rec[(size_t)i * nrec]) * acc;
}

// Per pair q = i(i+1)/2 + j:  tab[q] = [Cij (d*d) | cij (d) | lnZ | c2 w_i w_j (k) | c2 v_i v_j (k) | c2 iSigma_w(i,j,:) (k)]   (:180-182,196-197)
__global__ void k_pmg_pairs(PmgPat pt, int m, int de, int k, const double *__restrict__ P, const double *__restrict__ Sig,
                            const double *__restrict__ iSig, const double *__restrict__ rec, int nrec, const double *__restrict__ w,
                            const double *__restrict__ v, const double *__restrict__ iS, double *__restrict__ tab, int ntab, long base,
                            int count, double *__restrict__ ws, size_t wsp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const long q = base + t;
    int i = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((long)(i + 1) * (i + 2) / 2 <= q) ++i;
    while ((long)i * (i + 1) / 2 > q) --i;
    const int j = (int)(q - (long)i * (i + 1) / 2);
    const int d = pt.d;
    double *A = ws + (size_t)t * wsp, *Ai = A + (size_t)d * d, *W = Ai + (size_t)d * d, *dl = W + (size_t)d * d, *y = dl + d;
    const double *Ii = iSig + (size_t)i * d * d, *Ij = iSig + (size_t)j * d * d;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) A[(size_t)a * d + b] = Ii[(size_t)a * d + b] + Ij[(size_t)a * d + b];
    (void)pmg_inv(A, d, d, Ai, W);
    double *tq = tab + (size_t)q * ntab;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) tq[(size_t)a * d + b] = Ai[(size_t)a * d + b];
    // pa = P_i iSigma_i + P_j iSigma_j (kept in W's first row: W is free again)
    for (int a = 0; a < d; ++a) {
        double pa = 0.0;
        for (int c = 0; c < d; ++c)
            pa += P[(size_t)i * de + c] * Ii[(size_t)c * d + a] + P[(size_t)j * de + c] * Ij[(size_t)c * d + a];
        W[a] = pa;
    }
    for (int b = 0; b < d; ++b) {
        double s = 0.0;
        for (int a = 0; a < d; ++a) s = fma(W[a], Ai[(size_t)a * d + b], s);
        tq[(size_t)d * d + b] = s;
    }
    for (int a = 0; a < d; ++a) {
        dl[a] = P[(size_t)i * de + a] - P[(size_t)j * de + a];
        for (int b = 0; b < d; ++b) A[(size_t)a * d + b] = Sig[(size_t)i * d * d + (size_t)a * d + b] + Sig[(size_t)j * d * d + (size_t)a * d + b];
    }
    tq[(size_t)d * d + d] = rec[(size_t)i * nrec] + rec[(size_t)j * nrec] + pmg_lognorm(A, dl, d, d, y);
    const double c2 = (j < i) ? 2.0 : 1.0;
    for (int o = 0; o < k; ++o) {
        tq[(size_t)d * d + d + 1 + o] = c2 * w[i + (size_t)m * o] * w[j + (size_t)m * o];
        tq[(size_t)d * d + d + 1 + k + o] = v ? c2 * v[i + (size_t)m * o] * v[j + (size_t)m * o] : 0.0;
        tq[(size_t)d * d + d + 1 + 2 * k + o] = c2 * iS[i + (size_t)m * j + (size_t)m * m * o];
    }
}

// part[chunk][3k][ldx] of ONE row: sums over the pairs of a chunk of  Z_q(row) * weights,
//   Z = exp(lnZ) * sum_l N(X_hat(row,l) - cij ; Cij + Psi_hat_l(row)) Pio(row,l)     (:190-201 / :300-313)
// One wave per pair chunk, lanes along the m components l of the inner sum (reduced over the wave per pair).
__global__ __launch_bounds__(64) void k_pmg_accum(PmgPat pt, int row, int rr, int m, int ld, int k, long npairs, long pairs_per_chunk,
                                                  const double *__restrict__ rec, int nrec, const double *__restrict__ tab, int ntab,
                                                  const double *__restrict__ Pio, const double *__restrict__ Xhat,
                                                  const double *__restrict__ Phat, long ldx, double *__restrict__ part, int chunk0,
                                                  double *__restrict__ ws, size_t wsp) {
    const int chunk = chunk0 + blockIdx.x, lane = threadIdx.x;
    const int d = pt.d;
    double *S = ws + ((size_t)blockIdx.x * 64 + lane) * wsp, *dl = S + (size_t)d * d, *y = dl + d;
    const long q0 = (long)chunk * pairs_per_chunk, q1 = min(npairs, q0 + pairs_per_chunk);
    for (int e0 = 0; e0 < 3 * k; e0 += 24) {   // the 3k sums in blocks of 24 (k <= 8: one pass)
        double acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.0;
        for (long q = q0; q < q1; ++q) {
            const double *tq = tab + (size_t)q * ntab;
            double ec = 0.0;
            for (int l = lane; l < m; l += 64) {
                pmg_add_phat(S, tq, pt, Phat ? Phat + ((size_t)rr * m + l) * d * d : nullptr, rec + (size_t)l * nrec);
                const double *xh = Xhat + ((size_t)rr * m + l) * d;
                for (int a = 0; a < d; ++a) dl[a] = xh[a] - tq[(size_t)d * d + a];
                ec += exp(pmg_lognorm(S, dl, d, d, y)) * Pio[(size_t)rr * ld + l];
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ec += __shfl_xor(ec, off, 64);
            const double Z = exp(tq[(size_t)d * d + d]) * ec;
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e0 + e < 3 * k) acc[e] = fma(Z, tq[(size_t)d * d + d + 1 + e0 + e], acc[e]);
        }
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e0 + e < 3 * k) part[((size_t)chunk * 3 * k + e0 + e) * ldx + row] = acc[e];
        }
    }
}

// ---- host side: one NaN-pattern group, any d -----------------------------------------------------------------------
void launch_pmc_generic(hipStream_t st, const unsigned char *obs_host, int n, long ldx, int m, int ld, int d, int de, int k,
                        const double *Xr, const double *Psi3, const double *P, const double *Sig, const double *iSig,
                        const double *priors, const double *w, const double *v, const double *iS, double *rec, double *tab, double *Ex,
                        double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                        bool tab_ready, int *pat_dev, double *ws, long ws_threads) {
    // pattern lists -> device (3 d ints); the copy is ordered on the stream, the host vector must outlive it: synchronous copy
    int *hp = (int *)malloc((size_t)3 * d * sizeof(int));
    int no = 0, nu = 0;
    for (int c = 0; c < d; ++c) {
        if (obs_host[c]) hp[no++] = c;
        else hp[d + nu++] = c;
    }
    for (int a = 0; a < no; ++a) hp[2 * d + hp[a]] = a;                       // [~,unshuffle] = sort([find(o) find(~o)])
    for (int a = 0; a < nu; ++a) hp[2 * d + hp[d + a]] = no + a;
    (void)hipMemcpyAsync(pat_dev, hp, (size_t)3 * d * sizeof(int), hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);
    free(hp);
    PmgPat pt{d, no, nu, pat_dev, pat_dev + d, pat_dev + 2 * d};
    const int nrec = 2 + no * no + no * nu + nu * nu, ntab = d * d + d + 1 + 3 * k;
    const long npairs = (long)m * (m + 1) / 2;
    const size_t wsp = pmg_ws_per_thread(d);
    const int T = (int)(ws_threads < (1L << 30) ? ws_threads : (1L << 30));
    auto blocks = [](int cnt) { return dim3((unsigned)((cnt + 63) / 64)); };
    for (int b0 = 0; b0 < m; b0 += T) {
        const int cnt = m - b0 < T ? m - b0 : T;
        hipLaunchKernelGGL(k_pmg_prep, blocks(cnt), dim3(64), 0, st, pt, m, Sig, iSig, rec, nrec, b0, cnt, ws, wsp);
    }
    if (!tab_ready)
        for (long q0 = 0; q0 < npairs; q0 += T) {
            const int cnt = (int)(npairs - q0 < T ? npairs - q0 : T);
            hipLaunchKernelGGL(k_pmg_pairs, blocks(cnt), dim3(64), 0, st, pt, m, de, k, P, Sig, iSig, (const double *)rec, nrec, w, v, iS,
                               tab, ntab, q0, cnt, ws, wsp);
        }
    const int cpl = (int)(T / 64 < 1 ? 1 : T / 64);                           // pair chunks (waves) per accumulation launch
    for (int row = 0; row < n; ++row) {   // row block of one row: Ex / Pio / Xhat / Phat hold one row's tables (rr = 0)
        for (int b0 = 0; b0 < m; b0 += T) {
            const int cnt = m - b0 < T ? m - b0 : T;
            hipLaunchKernelGGL(k_pmg_rows, blocks(cnt), dim3(64), 0, st, pt, row, 0, m, ld, Xr, de, Psi3, P, Sig, (const double *)rec,
                               nrec, Ex, Xhat, Psi3 ? Phat : nullptr, b0, cnt, ws, wsp);
        }
        launch_pm_pio(st, Ex, ld, 1, m, priors, Pio);
        for (int b0 = 0; b0 < m; b0 += T) {
            const int cnt = m - b0 < T ? m - b0 : T;
            hipLaunchKernelGGL(k_pmg_phi, blocks(cnt), dim3(64), 0, st, pt, row, 0, m, ld, de, P, Sig, (const double *)rec, nrec,
                               (const double *)Pio, (const double *)Xhat, (const double *)(Psi3 ? Phat : nullptr), Phi, b0, cnt, ws, wsp);
        }
        for (int c0 = 0; c0 < nchunk; c0 += cpl) {
            const int nc = nchunk - c0 < cpl ? nchunk - c0 : cpl;
            hipLaunchKernelGGL(k_pmg_accum, dim3(nc), dim3(64), 0, st, pt, row, 0, m, ld, k, npairs, pairs_per_chunk, (const double *)rec,
                               nrec, (const double *)tab, ntab, (const double *)Pio, (const double *)Xhat,
                               (const double *)(Psi3 ? Phat : nullptr), ldx, part, c0, ws, wsp);
        }
    }
}
