// GC/VC with input noise, no missing dimensions, d <= 10: register-resident per-pair factorisations.
//
//   getPHI.m:78-89   ln PHI_ij = -1/2 Delta' M^-1 Delta + 1/2 ln|Sigma_j| - 1/2 ln|M|,   M = Psi_i + Sigma_j
//   GPz.m:164-185    moments of M^-1 Delta and (M^-1 Delta)(M^-1 Delta)' - M^-1 (chained to dP/dGamma by k_gen_finish)
//
// Same mathematics as the general kernels of k_gen.hip, but the dimension is a template parameter, every small
// matrix is a packed lower triangle with compile-time indices (registers, no scratch) and the loops are fully
// unrolled.  The d x d Cholesky / inverse per (sample, basis) pair is the reference's own algorithm
// (n*m interpreted iterations there); it is VALU work, not MFMA-shaped.
#include "gpz_dev.h"
#include "gpz_kernels.h"

// packed lower triangle, row-major: element (r, c), c <= r, at r(r+1)/2 + c
#define LT(r, c) ((r) * ((r) + 1) / 2 + (c))

// 1/sqrt(p) in fp64 without the library's sqrt + divide (~50 instructions): v_rsq_f64 seed and two Newton steps.
__device__ __forceinline__ double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

// In-place Cholesky of a packed lower triangle, division-free: rd[c] = 1 / L_cc (the diagonal slots of M hold L_cc for the
// callers that read them), *prod_rd = prod_c rd[c] = |M|^-1/2.  A non-positive pivot gives NaN as sqrt() would.
template <int D>
__device__ __forceinline__ void chol_packed_rd(double (&M)[D * (D + 1) / 2], double (&rd)[D], double *prod_rd) {
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double p = M[LT(c, c)];
#pragma unroll
        for (int q = 0; q < c; ++q) p = fma(-M[LT(c, q)], M[LT(c, q)], p);
        const double inv = rsqrt_nr(p);
        rd[c] = inv;
        M[LT(c, c)] = p * inv;
        prod *= inv;
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double s = M[LT(r, c)];
#pragma unroll
            for (int q = 0; q < c; ++q) s = fma(-M[LT(r, q)], M[LT(c, q)], s);
            M[LT(r, c)] = s * inv;
        }
    }
    *prod_rd = prod;
}
// returns sum(log(diag(L))) = 1/2 ln|M| through *half_logdet.
template <int D>
__device__ __forceinline__ void chol_packed(double (&M)[D * (D + 1) / 2], double (&rd)[D], double *half_logdet) {
    double pr;
    chol_packed_rd<D>(M, rd, &pr);
    *half_logdet = -log(pr);
}

// W = inv(L) (packed lower), Mi = W' W (packed lower, symmetric)
template <int D>
__device__ __forceinline__ void inv_packed(const double (&L)[D * (D + 1) / 2], const double (&rd)[D],
                                           double (&W)[D * (D + 1) / 2], double (&Mi)[D * (D + 1) / 2]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        W[LT(c, c)] = rd[c];
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            double s = 0.0;
#pragma unroll
            for (int q = c; q < r; ++q) s = fma(L[LT(r, q)], W[LT(q, c)], s);
            W[LT(r, c)] = -s * rd[r];
        }
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
#pragma unroll
            for (int q = a; q < D; ++q) s = fma(W[LT(q, a)], W[LT(q, b)], s);
            Mi[LT(a, b)] = s;
        }
}

// PHI, one thread per row; Sigma_j / p_j / ln|Sigma_j| staged through LDS for JB basis functions at a time.
// Xr: n_pad x de, Psi3: n_pad x d*d (column-major d x d per row), Sig: m x d*d, lnS: m (pattern 0).
// MISS: rows may have missing dimensions (gid = pattern per row, pat = observed flags [G][D], lnS = ln|Sigma_j,oo| as
// [G][m]).  A missing dimension is marginalised by making M block diagonal with an identity block on it and zeroing
// its Delta: ln|M| = ln|M_oo| and Delta' M^-1 Delta = Delta_o' M_oo^-1 Delta_o exactly, so the full-width register
// kernel serves every pattern (getPHI.m:80-87 with o = observed dimensions).
template <int D, bool MISS>
__global__ __launch_bounds__(256) void k_psi_phi(const double *__restrict__ Xr, int de, const double *__restrict__ Psi3,
                                                  int n, int m, const double *__restrict__ P,
                                                  const double *__restrict__ Sig, const double *__restrict__ lnS,
                                                  double *__restrict__ Phi, int ld, const int *__restrict__ gid,
                                                  const unsigned char *__restrict__ pat) {
    constexpr int NP = D * (D + 1) / 2;
    constexpr int JB = 8;
    constexpr int REC = NP + D + 1;          // [Sigma_j packed | p_j | ln|Sigma_j|]
    __shared__ double prm[JB * REC];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n;
    const int ic = act ? i : 0;
    double x[D], psi[NP], sel[MISS ? D : 1];
    const int g = MISS ? gid[ic] : 0;
    double cmiss = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = Xr[(size_t)ic * de + c];
    if (MISS) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            sel[c] = pat[g * D + c] ? 1.0 : 0.0;
            cmiss -= 0.5 * GPZ_LOG2 * (1.0 - sel[c]);                           // -1/2 |u| ln 2   (getPHI.m:87)
        }
    }
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const double v = Psi3[(size_t)ic * D * D + r + D * c];
            if (MISS) psi[LT(r, c)] = (sel[r] * sel[c] != 0.0) ? v : (r == c ? 1.0 : 0.0);   // Psi of a missing dimension is never read
            else psi[LT(r, c)] = v;
        }
    for (int j0 = 0; j0 < m; j0 += JB) {
        __syncthreads();
        for (int e = threadIdx.x; e < JB * REC; e += 256) {
            const int jj = e / REC, q = e % REC, j = min(j0 + jj, m - 1);
            double v;
            if (q < NP) {
                // q -> (r, c) of the packed triangle
                int r = 0;
                while ((r + 1) * (r + 2) / 2 <= q) ++r;
                const int c = q - r * (r + 1) / 2;
                v = Sig[(size_t)j * D * D + r * D + c];
            } else if (q < NP + D) {
                v = P[(size_t)j * de + (q - NP)];
            } else {
                v = lnS[j];
            }
            prm[e] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            if (j >= m) break;
            const double *t = prm + jj * REC;
            double M[NP], y[D];
            if (MISS) {
#pragma unroll
                for (int r = 0; r < D; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) M[LT(r, c)] = fma(sel[r] * sel[c], t[LT(r, c)], psi[LT(r, c)]);
            } else {
#pragma unroll
                for (int e = 0; e < NP; ++e) M[e] = psi[e] + t[e];             // Psi(o,o,i) + Sigma(o,o)      getPHI.m:84
            }
            double hl, rd[D];
            chol_packed<D>(M, rd, &hl);
            double quad = 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r) {                                      // y = L^-1 Delta
                double s = x[r] - t[NP + r];
                if (MISS) s *= sel[r];
#pragma unroll
                for (int c = 0; c < r; ++c) s = fma(-M[LT(r, c)], y[c], s);
                y[r] = s * rd[r];
                quad = fma(y[r], y[r], quad);
            }
            const double lns = MISS ? lnS[(size_t)g * m + j] : t[NP + D];
            const double lp = -0.5 * quad + 0.5 * lns - hl + cmiss;            // getPHI.m:86
            if (act) Phi[(size_t)i * ld + j] = exp(lp);
        }
    }
}

// Moment records for k_gen_finish (pattern 0 = all dimensions observed): thread per basis j, rows walked with x_i and
// Psi_i wave-uniform.  rec = [A0 | Acc1 (d) | Cacc (d*d) | r1 | r2]  (see k_gen_moments).
template <int D, bool MISS>
__global__ __launch_bounds__(64) void k_psi_moments(const double *__restrict__ Phi, const double *__restrict__ T, int ld,
                                                     const double *__restrict__ rowscal, const double *__restrict__ w,
                                                     const double *__restrict__ v, const double *__restrict__ Xr, int de,
                                                     const double *__restrict__ Psi3, int n, int m,
                                                     const double *__restrict__ P, const double *__restrict__ Sig,
                                                     int rows_per_chunk, double *__restrict__ slab, int nrec,
                                                     const int *__restrict__ gid, const unsigned char *__restrict__ pat,
                                                     const int *__restrict__ chunktab) {
    constexpr int NP = D * (D + 1) / 2;
    const int j = blockIdx.y * 64 + threadIdx.x;
    const bool act = j < m;
    const int jc = act ? j : 0;
    const int chunk = blockIdx.x;
    double sg[NP], p[D], acc1[D], cacc[NP];
#pragma unroll
    for (int r = 0; r < D; ++r) {
        p[r] = P[(size_t)jc * de + r];
        acc1[r] = 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c) { sg[LT(r, c)] = Sig[(size_t)jc * D * D + r * D + c]; cacc[LT(r, c)] = 0.0; }
    }
    const double wj = w ? w[jc] : 0.0, vj = v ? v[jc] : 0.0;
    double a0 = 0.0, r1 = 0.0, r2 = 0.0;
    int r0 = chunk * rows_per_chunk, rend = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; rend = chunktab[2 * chunk + 1]; }   // chunks that end at pattern boundaries
    for (int i = r0; i < rend; ++i) {
        const double ph = Phi[(size_t)i * ld + jc];
        double dp;
        if (rowscal) {
            const double *rs = rowscal + (size_t)i * 4;
            dp = (-rs[0] * T[(size_t)i * ld + jc] - rs[1] * wj + rs[2] * vj) * ph;      // GPz.m:72,90,106,113
            r1 = fma(ph, rs[1], r1);
            r2 = fma(ph, rs[2], r2);
        } else {
            dp = T[(size_t)i * ld + jc];
        }
        double L[NP], W[NP], Mi[NP], dl[D], u[D];
        const double *ps = Psi3 + (size_t)i * D * D;
        if (MISS) {   // a missing dimension: identity block in M, zero Delta (see k_psi_phi); wave-uniform flags
            const unsigned char *ob = pat + (size_t)gid[i] * D;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                dl[r] = ob[r] ? Xr[(size_t)i * de + r] - p[r] : 0.0;
#pragma unroll
                for (int c = 0; c <= r; ++c)
                    L[LT(r, c)] = (ob[r] && ob[c]) ? sg[LT(r, c)] + ps[r + D * c] : (r == c ? 1.0 : 0.0);
            }
        } else {
#pragma unroll
        for (int r = 0; r < D; ++r) {
            dl[r] = Xr[(size_t)i * de + r] - p[r];
#pragma unroll
            for (int c = 0; c <= r; ++c) L[LT(r, c)] = sg[LT(r, c)] + ps[r + D * c];     // Sigma + Psi_i    GPz.m:170
        }
        }
        double hl, rd[D];
        chol_packed<D>(L, rd, &hl);
        inv_packed<D>(L, rd, W, Mi);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) s = fma(b <= a ? Mi[LT(a, b)] : Mi[LT(b, a)], dl[b], s);
            u[a] = s;
        }
        a0 += dp;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            acc1[a] = fma(dp, u[a], acc1[a]);                                            // GPz.m:172
#pragma unroll
            for (int b = 0; b <= a; ++b) cacc[LT(a, b)] = fma(dp, u[a] * u[b] - Mi[LT(a, b)], cacc[LT(a, b)]);   // :174
        }
    }
    if (act) {
        double *rec = slab + ((size_t)chunk * m + j) * nrec;
        rec[0] = a0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            rec[1 + a] = acc1[a];
#pragma unroll
            for (int b = 0; b < D; ++b) rec[1 + D + a * D + b] = b <= a ? cacc[LT(a, b)] : cacc[LT(b, a)];
        }
        rec[1 + D + D * D] = r1;
        rec[2 + D + D * D] = r2;
    }
}

#define PSI_CASES(MACRO) \
    switch (d) {         \
        case 2: MACRO(2); break;   \
        case 3: MACRO(3); break;   \
        case 4: MACRO(4); break;   \
        case 5: MACRO(5); break;   \
        case 6: MACRO(6); break;   \
        case 7: MACRO(7); break;   \
        case 8: MACRO(8); break;   \
        case 9: MACRO(9); break;   \
        case 10: MACRO(10); break; \
        default: return -1;        \
    }

// returns -1 when d is outside the instantiated range (caller falls back to the general kernels).
// pat != nullptr: rows carry missing dimensions (r.gid, lnS = [G][m]); nullptr: one pattern, lnS = [m].
int launch_psi_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                   const double *lnS, double *Phi, int ld, const unsigned char *pat, bool shared) {
    if (cpsi4_available(d)) return launch_cpsi4_phi(st, r, m, d, de, P, Sig, lnS, Phi, ld, pat, shared);   // four pairs per wave on 4 x 4 tiles (k_cpsi4.hip)
    if (!pat && cpsi4w_available(d)) return launch_cpsi4w_phi(st, r, m, d, de, P, Sig, lnS, Phi, ld);   // the same up to d = 48 (k_cpsi4w.hip)
    if (d > 10) return launch_cpsi_phi(st, r, m, d, de, P, Sig, lnS, Phi, ld, pat);   // wave-per-pair MFMA elimination (k_cpsi.hip)
    if (r.n <= 0) return (d >= 2 && d <= 10) ? 0 : -1;   // a rank of a sharded run may hold no row of this set
#define PHI_CASE(DD)                                                                                                       \
    do {                                                                                                                   \
        if (pat)                                                                                                           \
            hipLaunchKernelGGL((k_psi_phi<DD, true>), dim3((r.n + 255) / 256), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, P, \
                               Sig, lnS, Phi, ld, r.gid, pat);                                                            \
        else                                                                                                               \
            hipLaunchKernelGGL((k_psi_phi<DD, false>), dim3((r.n + 255) / 256), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, P, \
                               Sig, lnS, Phi, ld, nullptr, nullptr);                                                      \
    } while (0)
    PSI_CASES(PHI_CASE)
#undef PHI_CASE
    return 0;
}

// chunktab (optional): {first row, end row} per chunk instead of chunk * rows_per_chunk
int launch_psi_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                       const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                       int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                       const int *chunktab, const double *gc_minv) {
    if (cpsi4_available(d))
        return launch_cpsi4_moments(st, Phi, T, ld, rowscal, w, v, r, m, d, de, P, Sig, nchunk, rows_per_chunk, slab, nrec, pat,
                                    chunktab, gc_minv);
    if (!pat && cpsi4w_available(d))
        return launch_cpsi4w_moments(st, Phi, T, ld, rowscal, w, v, r, m, d, de, P, Sig, nchunk, rows_per_chunk, slab, nrec, chunktab);
    if (d > 10)
        return launch_cpsi_moments(st, Phi, T, ld, rowscal, w, v, r, m, d, de, P, Sig, nchunk, rows_per_chunk, slab, nrec, pat,
                                   chunktab);
    if (nchunk <= 0) return (d >= 2 && d <= 10) ? 0 : -1;
#define MOM_CASE(DD)                                                                                                       \
    do {                                                                                                                   \
        if (pat)                                                                                                           \
            hipLaunchKernelGGL((k_psi_moments<DD, true>), dim3(nchunk, (m + 63) / 64), dim3(64), 0, st, Phi, T, ld, rowscal, \
                               w, v, r.Xr, de, r.Psi3, r.n, m, P, Sig, rows_per_chunk, slab, nrec, r.gid, pat, chunktab);   \
        else                                                                                                               \
            hipLaunchKernelGGL((k_psi_moments<DD, false>), dim3(nchunk, (m + 63) / 64), dim3(64), 0, st, Phi, T, ld, rowscal, \
                               w, v, r.Xr, de, r.Psi3, r.n, m, P, Sig, rows_per_chunk, slab, nrec, nullptr, nullptr,       \
                               chunktab);                                                                                  \
    } while (0)
    PSI_CASES(MOM_CASE)
#undef MOM_CASE
    return 0;
}

// predictNoisy, covariance kinds (predictCov.m:70-132): one thread per sample, the pairs [p0, p1) of this chunk; partial sums
// part[chunk][3][k][ldx] — the register-resident twin of k_predict_noisy's covariance branch (k_gen.hip): Psi_i and the pair
// matrix Cij + Psi_i are packed triangles with compile-time indices, the pair record (uniform over the wave) is read through
// the scalar cache.  tab record: [lnz | cij (d) | Cij (d x d)].
//   DIAGPSI  every Psi_i of the call is diagonal (what fixPsi.m builds from per-dimension variances): d instead of d(d+1)/2
//            registers for it, two waves per SIMD instead of one
//   SHARED   GC: every basis function has the same covariance, so Cij = Sigma/2 for every pair and the sample's pair matrix
//            Sigma/2 + Psi_i is factorised ONCE; a pair then costs the forward substitution and one exp
//   KM       outputs carried in registers (1 or 8)
// N(x; cij, Cij + Psi_i) = exp(-1/2 q) * prod_c (1 / L_cc): no logarithm, no division, no sqrt per pair.
template <int D, bool DIAGPSI, bool SHARED, int KM>
__global__ __launch_bounds__(64) void k_predict_noisy_cov(int n, long ldx, int m, int de, int k, const double *__restrict__ Xr,
                                                           const double *__restrict__ Psi3, const double *__restrict__ tab,
                                                           int rec, const double *__restrict__ w, const double *__restrict__ v,
                                                           const double *__restrict__ iS, long pairs_per_chunk,
                                                           double *__restrict__ part) {
    constexpr int NP = D * (D + 1) / 2;
    constexpr int NPS = DIAGPSI ? D : NP;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n;
    const int ic = act ? i : n - 1;
    const long npair = (long)m * (m + 1) / 2;
    const long p0 = (long)blockIdx.y * pairs_per_chunk, p1 = min(npair, p0 + pairs_per_chunk);
    double x[D], ps[NPS];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = Xr[(size_t)ic * de + c];
#pragma unroll
    for (int r = 0; r < D; ++r) {
        if (DIAGPSI) ps[r] = Psi3[(size_t)ic * D * D + r + D * r];
        else {
#pragma unroll
            for (int c = 0; c <= r; ++c) ps[DIAGPSI ? 0 : LT(r, c)] = Psi3[(size_t)ic * D * D + r + D * c];
        }
    }
    double ga[KM], vl[KM], nu[KM];
#pragma unroll
    for (int o = 0; o < KM; ++o) { ga[o] = 0.0; vl[o] = 0.0; nu[o] = 0.0; }
    long a = (long)((sqrt(8.0 * (double)p0 + 1.0) - 1.0) * 0.5);   // (a, b) of the first pair, then walk
    while (a * (a + 1) / 2 > p0) --a;
    while ((a + 1) * (a + 2) / 2 <= p0) ++a;
    long b = p0 - a * (a + 1) / 2;
    double M[NP], rd[D], prd = 1.0;
    auto build = [&](const double *t) {
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                if (DIAGPSI) M[LT(r, c)] = (r == c) ? t[1 + D + r * D + c] + ps[r] : t[1 + D + r * D + c];
                else M[LT(r, c)] = t[1 + D + r * D + c] + ps[DIAGPSI ? 0 : LT(r, c)];   // Cij + Psi        predictCov.m:109
            }
        chol_packed_rd<D>(M, rd, &prd);
    };
    if (SHARED && p0 < p1) build(tab + (size_t)p0 * rec);
#pragma unroll 1
    for (long e = p0; e < p1; ++e) {
        const double *t = tab + (size_t)e * rec;
        if (!SHARED) build(t);
        double q = 0.0, y[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double s = x[r] - t[1 + r];
#pragma unroll
            for (int c = 0; c < r; ++c) s = fma(-M[LT(r, c)], y[c], s);
            y[r] = s * rd[r];
            q = fma(y[r], y[r], q);
        }
        const double z = ((a == b) ? 1.0 : 2.0) * exp(t[0] - 0.5 * q) * prd;                   // :111, 2x in the loop (:113-119)
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                ga[o] = fma(z, w[a + (size_t)m * o] * w[b + (size_t)m * o], ga[o]);
                vl[o] = fma(z, v ? v[a + (size_t)m * o] * v[b + (size_t)m * o] : 0.0, vl[o]);
                nu[o] = fma(z, iS[a + (size_t)m * b + (size_t)m * m * o], nu[o]);
            }
        if (++b > a) { ++a; b = 0; }
    }
    if (act) {
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                part[(((size_t)blockIdx.y * 3 + 0) * k + o) * ldx + i] = ga[o];
                part[(((size_t)blockIdx.y * 3 + 1) * k + o) * ldx + i] = vl[o];
                part[(((size_t)blockIdx.y * 3 + 2) * k + o) * ldx + i] = nu[o];
            }
    }
}

// flags: bit 0 = every Psi_i is diagonal, bit 1 = all basis functions share one covariance (GC)
int launch_predict_noisy_cov(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                             const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                             long pairs_per_chunk, double *part, int flags) {
    if (cpsi4_available(d) && k <= 8)   // 10 < d <= 32: four (sample, pair) units per wave on 4 x 4 MFMA tiles (k_cpsi4.hip)
        return launch_cpsi4_predict_noisy(st, n, ldx, m, d, de, k, Xr, Psi3, tab, rec, w, v, iS, nchunk, pairs_per_chunk, part,
                                          (flags & 2) != 0);
    if (cpsi4w_available(d) && k == 1)   // 32 < d <= 48, one output (k_cpsi4wp.hip)
        return launch_cpsi4w_predict_noisy(st, n, ldx, m, d, de, k, Xr, Psi3, tab, rec, w, v, iS, nchunk, pairs_per_chunk, part,
                                           (flags & 2) != 0);
    if (n <= 0) return (d >= 2 && d <= 10) ? 0 : -1;
    const bool dg = flags & 1, sh = flags & 2;
#define PN_LAUNCH(DD, DG, SH, KM)                                                                                          \
    hipLaunchKernelGGL((k_predict_noisy_cov<DD, DG, SH, KM>), dim3((n + 63) / 64, nchunk), dim3(64), 0, st, n, ldx, m, de, k, Xr, \
                       Psi3, tab, rec, w, v, iS, pairs_per_chunk, part)
#define PN_CASE(DD)                                                                                                        \
    do {                                                                                                                   \
        if (k == 1) {                                                                                                      \
            if (dg && sh) PN_LAUNCH(DD, true, true, 1); else if (dg) PN_LAUNCH(DD, true, false, 1);                        \
            else if (sh) PN_LAUNCH(DD, false, true, 1); else PN_LAUNCH(DD, false, false, 1);                               \
        } else {                                                                                                           \
            if (dg && sh) PN_LAUNCH(DD, true, true, 8); else if (dg) PN_LAUNCH(DD, true, false, 8);                        \
            else if (sh) PN_LAUNCH(DD, false, true, 8); else PN_LAUNCH(DD, false, false, 8);                               \
        }                                                                                                                  \
    } while (0)
    PSI_CASES(PN_CASE)
#undef PN_CASE
#undef PN_LAUNCH
    return 0;
}

// predictNoisy, diagonal kinds (predictDiag.m:75-125): one thread per sample, the pairs [p0, p1) of this chunk.  tab record:
// [lnZ | cij (d) | Cij (d)].  Per dimension one reciprocal square root r = (Cij + psi)^-1/2 serves both the quadratic form
// (Delta^2 r^2) and the normalisation (prod r): no divide and no logarithm per pair (the scratch-resident branch of
// k_predict_noisy spent two thirds of its instructions in d divides and a log per pair).
template <int D, int KM>
__global__ __launch_bounds__(64) void k_predict_noisy_diag(int n, long ldx, int m, int d, int de, int k,
                                                            const double *__restrict__ Xr, const double *__restrict__ Psir,
                                                            const double *__restrict__ tab, int rec,
                                                            const double *__restrict__ w, const double *__restrict__ v,
                                                            const double *__restrict__ iS, long pairs_per_chunk,
                                                            double *__restrict__ part) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n;
    const int ic = act ? i : n - 1;
    const long npair = (long)m * (m + 1) / 2;
    const long p0 = (long)blockIdx.y * pairs_per_chunk, p1 = min(npair, p0 + pairs_per_chunk);
    double x[D], ps[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        x[c] = (c < d) ? Xr[(size_t)ic * de + c] : 0.0;
        ps[c] = (c < d) ? Psir[(size_t)ic * de + c] : 0.0;
    }
    double ga[KM], vl[KM], nu[KM];
#pragma unroll
    for (int o = 0; o < KM; ++o) { ga[o] = 0.0; vl[o] = 0.0; nu[o] = 0.0; }
    long a = (long)((sqrt(8.0 * (double)p0 + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > p0) --a;
    while ((a + 1) * (a + 2) / 2 <= p0) ++a;
    long b = p0 - a * (a + 1) / 2;
#pragma unroll 1
    for (long e = p0; e < p1; ++e) {
        const double *t = tab + (size_t)e * rec;
        double q = 0.0, pr = 1.0;
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < d) {
                const double r = rsqrt_nr(t[1 + d + c] + ps[c]);               // (Cij + Psi)^-1/2          predictDiag.m:105
                const double dl = (x[c] - t[1 + c]) * r;
                q = fma(dl, dl, q);
                pr *= r;
            }
        const double z = ((a == b) ? 1.0 : 2.0) * exp(t[0] - 0.5 * q) * pr;    // :107, 2x in the loop (:113-119)
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                ga[o] = fma(z, w[a + (size_t)m * o] * w[b + (size_t)m * o], ga[o]);
                vl[o] = fma(z, v ? v[a + (size_t)m * o] * v[b + (size_t)m * o] : 0.0, vl[o]);
                nu[o] = fma(z, iS[a + (size_t)m * b + (size_t)m * m * o], nu[o]);
            }
        if (++b > a) { ++a; b = 0; }
    }
    if (act) {
#pragma unroll
        for (int o = 0; o < KM; ++o)
            if (o < k) {
                part[(((size_t)blockIdx.y * 3 + 0) * k + o) * ldx + i] = ga[o];
                part[(((size_t)blockIdx.y * 3 + 1) * k + o) * ldx + i] = vl[o];
                part[(((size_t)blockIdx.y * 3 + 2) * k + o) * ldx + i] = nu[o];
            }
    }
}

// d <= 20, k <= 8: register-resident; else -1 (the caller keeps the runtime-d kernel)
int launch_predict_noisy_diag(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psir,
                              const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                              long pairs_per_chunk, double *part) {
    if (d > 20 || k > 8) return -1;
    if (n <= 0) return 0;
#define PND(DD)                                                                                                              \
    do {                                                                                                                     \
        if (k == 1)                                                                                                          \
            hipLaunchKernelGGL((k_predict_noisy_diag<DD, 1>), dim3((n + 63) / 64, nchunk), dim3(64), 0, st, n, ldx, m, d, de, k, Xr, \
                               Psir, tab, rec, w, v, iS, pairs_per_chunk, part);                                             \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_predict_noisy_diag<DD, 8>), dim3((n + 63) / 64, nchunk), dim3(64), 0, st, n, ldx, m, d, de, k, Xr, \
                               Psir, tab, rec, w, v, iS, pairs_per_chunk, part);                                             \
    } while (0)
    if (d <= 4) PND(4); else if (d <= 8) PND(8); else if (d <= 12) PND(12); else if (d <= 16) PND(16); else PND(20);
#undef PND
    return 0;
}

bool psi_fast_path_available(int d) { return (d >= 2 && d <= 10) || cpsi_available(d); }
