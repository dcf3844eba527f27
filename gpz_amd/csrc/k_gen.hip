// General covariance-kind path: input noise Psi (d x d per sample) and/or missing input dimensions.
//
//   getPHI.m:73-89   ln PHI_ij = -1/2 Delta_o' M^-1 Delta_o + 1/2 ln|Sigma_oo| - 1/2 ln|M| - 1/2 |u| ln 2,
//                    M = Psi_i,oo + Sigma_j,oo,  Sigma_j = inv(Gamma_j' Gamma_j)       (without Psi the two log-dets cancel)
//   GPz.m:146-185    dP(j,o) += dPHI_ij Delta_o' M^-1;  dSoo = 1/2 (Sigma_oo^-1 - M^-1 + M^-1 Delta Delta' M^-1);
//                    diSoo = -Sigma_oo dSoo Sigma_oo;  dGo = 2 (Gamma(:,o) - Gamma(:,u) GuuGuo) diSoo;
//                    dGamma(:,o,j) += dPHI_ij dGo;  dGamma(:,u,j) -= dPHI_ij dGo GuuGuo'   (GuuGuo = iSigma_uu^-1 iSigma_uo).
//   With Psi = 0 this reduces to the no-noise branch GPz.m:151-159 (M^-1 Delta = Sigma_oo^-1 Delta, diSoo = -1/2 Delta Delta').
//
// These are per-(sample, basis) d x d factorisations (SURVEY.md §8a rows a7, a22 and the missing-value branches of
// a5, a21): not MFMA-shaped.  A straightforward correctness path — runtime d (any: see GEN_ARR below), the small
// matrices live in per-thread scratch — used only when Psi is given or X has NaNs with GC/VC; everything around it
// (SYRK, factorisation, T-GEMM, row scalars) is the regular pipeline.
#include "gpz_dev.h"
#include "gpz_kernels.h"

// Small-matrix storage of the general path.  CAP > 0: compile-time capacity, the matrices live in per-thread scratch with
// leading dimension CAP (d <= 20, the layout the path was measured with).  CAP == 0: runtime d of any size — leading
// dimension d, the matrices live in a global workspace of gen_ws_per_thread(d) doubles per thread and the kernels run as
// grid-stride loops over gen_rt_threads(d) threads, so the workspace does not grow with the problem.
#define GCAP 20
// CAP == 0: element e of a thread's array sits at ws[(offset + e) * nthreads + thread] — the lanes of a wave touch consecutive
// doubles (per-thread contiguous blocks made every access 64 separate cache lines: 25 x slower at d = 24).  ES is the element
// stride (1 for the scratch arrays of CAP > 0); every workspace array is indexed as name[(i) * ES].
#define GEN_ARR(name, cap_count, rt_count)                       \
    double name##_loc[CAP ? (cap_count) : 1];                    \
    double *name = name##_loc;                                   \
    if (!CAP) { name = wsp; wsp += (size_t)(rt_count) * nth_; }
#define GEN_IARR(name, cap_count, rt_count)                      \
    int name##_loc[CAP ? (cap_count) : 1];                       \
    int *name = name##_loc;                                      \
    if (!CAP) { name = (int *)wsp; wsp += (size_t)(rt_count) * nth_; }   /* one 8-byte slot per int: the same stride */
#define GEN_SETUP()                                                                                   \
    const long gt_ = (long)blockIdx.x * blockDim.x + threadIdx.x, nth_ = (long)gridDim.x * blockDim.x; \
    const int GDM = CAP ? CAP : d;                                                                    \
    const long ES = CAP ? 1 : nth_;                                                                   \
    const long ESI = CAP ? 1 : 2 * nth_;   /* int arrays: stride in ints */                           \
    double *wsp = ws ? ws + gt_ : nullptr;                                                            \
    (void)wsp; (void)ES; (void)ESI; (void)ws_stride

size_t gen_ws_per_thread(int d) { return (size_t)9 * d * d + (size_t)10 * d + 16; }
// threads of the runtime-d pool: as many as 2 GiB of workspace hold, 4096 .. 65536 (whole waves)
int gen_rt_threads(int d) {
    long t = (long)((2048UL << 20) / (gen_ws_per_thread(d) * sizeof(double)));
    t = t < 1024 ? 1024 : (t > 65536 ? 65536 : t);   // (at least 16 waves; a floor of 4096 threads made the workspace 26 GB at d = 300)
    return (int)(t / 64 * 64);
}

// Cholesky of the leading no x no block of M (row-major, leading dimension ld), lower factor in place.
// 1/sqrt(p) without the library's sqrt and divide (v_rsq_f64 seed + two Newton steps); NaN for p < 0 as sqrt() gives
__device__ __forceinline__ double gen_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ bool chol_small(double *M, int no, int ld, long es) {
    bool ok = true;
    for (int c = 0; c < no; ++c) {
        double p = M[(c * ld + c) * es];
        for (int q = 0; q < c; ++q) p = fma(-M[(c * ld + q) * es], M[(c * ld + q) * es], p);
        if (!(p > 0.0)) ok = false;
        const double inv = gen_rsqrt(p);
        M[(c * ld + c) * es] = p * inv;
        for (int r = c + 1; r < no; ++r) {
            double s = M[(r * ld + c) * es];
            for (int q = 0; q < c; ++q) s = fma(-M[(r * ld + q) * es], M[(c * ld + q) * es], s);
            M[(r * ld + c) * es] = s * inv;
        }
    }
    return ok;
}

// W = inv(L) (lower), then Minv = W' W; L in M (lower), result symmetric full in Minv.
__device__ __forceinline__ void inv_from_chol(const double *L, int no, double *W, double *Minv, int ld, long es) {
    for (int c = 0; c < no; ++c) W[(c * ld + c) * es] = 1.0 / L[(c * ld + c) * es];     // the d reciprocals first: the substitution multiplies
    for (int c = 0; c < no; ++c) {
        for (int r = c + 1; r < no; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(L[(r * ld + q) * es], W[(q * ld + c) * es], s);
            W[(r * ld + c) * es] = -s * W[(r * ld + r) * es];
        }
    }
    for (int a = 0; a < no; ++a)
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int q = a; q < no; ++q) s = fma(W[(q * ld + a) * es], W[(q * ld + b) * es], s);
            Minv[(a * ld + b) * es] = s;
            Minv[(b * ld + a) * es] = s;
        }
}

// Sigma_j = inv(Gamma_j' Gamma_j), iSigma_j = Gamma_j' Gamma_j   (getPHI.m:73, GPz.m:146-147); d x d row-major, stride d*d.
template <int CAP>
__global__ void k_gen_prep(const double *__restrict__ G, int m, int d, int de, double *__restrict__ Sig,
                           double *__restrict__ iSig, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_ARR(A, GCAP * GCAP, d * d) GEN_ARR(W, GCAP * GCAP, d * d) GEN_ARR(Ai, GCAP * GCAP, d * d)
    for (long j = gt_; j < m; j += nth_) {
        const double *Gj = G + (size_t)j * de * de;
        for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) {
                double s = 0.0;
                for (int q = 0; q < d; ++q) s = fma(Gj[q * de + a], Gj[q * de + b], s);
                A[(a * GDM + b) * ES] = s;
                iSig[(size_t)j * d * d + a * d + b] = s;
            }
        chol_small(A, d, GDM, ES);
        inv_from_chol(A, d, W, Ai, GDM, ES);
        for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) Sig[(size_t)j * d * d + a * d + b] = Ai[(a * GDM + b) * ES];
    }
}

// ln|Sigma_j,oo| for every (pattern g, basis j):  lnS[g*m + j]
template <int CAP>
__global__ void k_gen_lndet(const double *__restrict__ Sig, const unsigned char *__restrict__ pat, int G, int m, int d,
                            double *__restrict__ lnS, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_IARR(o, GCAP, d) GEN_ARR(M, GCAP * GCAP, d * d)
    for (long e = gt_; e < (long)G * m; e += nth_) {
        const int g = (int)(e / m), j = (int)(e % m);
        int no = 0;
        for (int c = 0; c < d; ++c)
            if (pat[g * d + c]) o[(no++) * ESI] = c;
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b) M[(a * GDM + b) * ES] = Sig[(size_t)j * d * d + o[(a) * ESI] * d + o[(b) * ESI]];
        chol_small(M, no, GDM, ES);
        double s = 0.0;
        for (int a = 0; a < no; ++a) s += log(M[(a * GDM + a) * ES]);
        lnS[e] = 2.0 * s;
    }
}

// PHI build, one thread per row, loop over basis functions.  Xr: n_pad x de (0 at missing), gid: pattern per row,
// Psi3: n_pad x d*d (nullptr without input noise).  Writes columns [0, m) of PHI.
template <int CAP>
__global__ __launch_bounds__(64) void k_gen_phi(const double *__restrict__ Xr, int de, const int *__restrict__ gid,
                                                 const unsigned char *__restrict__ pat, const double *__restrict__ Psi3,
                                                 int n, int m, int d, const double *__restrict__ P,
                                                 const double *__restrict__ Sig, const double *__restrict__ lnS,
                                                 double *__restrict__ Phi, int ld, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_IARR(o, GCAP, d) GEN_ARR(x, GCAP, d) GEN_ARR(M, GCAP * GCAP, d * d) GEN_ARR(y, GCAP, d)
    for (long i = gt_; i < n; i += nth_) {
        const int g = gid[i];
        int no = 0;
        for (int c = 0; c < d; ++c)
            if (pat[g * d + c]) o[(no++) * ESI] = c;
        const double cmiss = -0.5 * (double)(d - no) * GPZ_LOG2;               // -1/2 |u| ln 2
        for (int a = 0; a < no; ++a) x[(a) * ES] = Xr[(size_t)i * de + o[(a) * ESI]];
        for (int j = 0; j < m; ++j) {
            const double *Sj = Sig + (size_t)j * d * d;
            for (int a = 0; a < no; ++a)
                for (int b = 0; b <= a; ++b) {
                    double v = Sj[o[(a) * ESI] * d + o[(b) * ESI]];
                    if (Psi3) v += Psi3[(size_t)i * d * d + o[(a) * ESI] + d * o[(b) * ESI]];   // Psi(o,o,i)   getPHI.m:84
                    M[(a * GDM + b) * ES] = v;
                }
            chol_small(M, no, GDM, ES);
            double quad = 0.0, ldM = 0.0;
            for (int a = 0; a < no; ++a) {                                      // y = L^-1 Delta_o
                double s = x[(a) * ES] - P[(size_t)j * de + o[(a) * ESI]];
                for (int q = 0; q < a; ++q) s = fma(-M[(a * GDM + q) * ES], y[(q) * ES], s);
                y[(a) * ES] = s / M[(a * GDM + a) * ES];
                quad = fma(y[(a) * ES], y[(a) * ES], quad);
                ldM += log(M[(a * GDM + a) * ES]);
            }
            double lp = -0.5 * quad + cmiss;                                    // getPHI.m:76
            if (Psi3) lp += 0.5 * lnS[(size_t)g * m + j] - ldM;                 // getPHI.m:86  (+1/2 ln|S_oo| - 1/2 ln|M|)
            Phi[(size_t)i * ld + j] = exp(lp);
        }
    }
}

// Columns m..mp-1 of PHI: y in m..m+k-1, zero after; rows >= n zero everywhere.
__global__ void k_gen_fill(double *__restrict__ Phi, int ld, int n, int n_pad, int m, int mp, int k,
                           const double *__restrict__ Y, long ldx) {
    const size_t gs = (size_t)blockDim.x * gridDim.x;
    const size_t tot = (size_t)n_pad * mp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gs) {
        const size_t i = e / mp;
        const int j = (int)(e % mp);
        if ((int)i >= n) Phi[i * ld + j] = 0.0;
        else if (j >= m) Phi[i * ld + j] = (Y && j - m < k) ? Y[(size_t)(j - m) * ldx + i] : 0.0;
    }
}

// lnbeta = b + PHI v, omega*beta, PHI w: one wave per row over an existing PHI (getPHI.m:116-125, GPz.m:43-48).
__global__ __launch_bounds__(256) void k_gen_rowdot(const double *__restrict__ Phi, int ld, int n, long ldx, int m, int k,
                                                     const double *__restrict__ v, const double *__restrict__ bvec,
                                                     const double *__restrict__ omega, long om_ld, const double *__restrict__ wv,
                                                     double *__restrict__ lnbeta, double *__restrict__ wbeta,
                                                     double *__restrict__ phiw) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= ldx) return;
    for (int o = 0; o < k; ++o) {
        double sv = 0.0, sw = 0.0;
        if (i < n) {
            for (int j = lane; j < m; j += 64) {
                const double ph = Phi[(size_t)i * ld + j];
                if (v) sv = fma(ph, v[j + (size_t)m * o], sv);
                if (wv) sw = fma(ph, wv[j + (size_t)m * o], sw);
            }
        }
        sv = wave_sum(sv);
        sw = wave_sum(sw);
        if (lane == 0) {
            const bool valid = i < n;
            const double lb = bvec[o] + sv;
            lnbeta[(size_t)o * ldx + i] = valid ? lb : 0.0;
            if (wbeta) wbeta[(size_t)o * ldx + i] = valid ? (omega ? omega[(size_t)o * om_ld + i] : 1.0) * exp(-lb) : 0.0;
            if (phiw) phiw[(size_t)o * ldx + i] = sw;
        }
    }
}

// Moment accumulation for one pattern group: thread per (chunk, basis j), rows of the group given by an index list.
//   rec[j] = [A0 = sum dp | Acc1 (d) = sum dp * embed(M^-1 Delta) | Cacc (d*d) = sum dp * embed(-M^-1 + u u') | r1 | r2]
// dp = dPHI_ij from T, PHI and the row scalars (single-output fused form) or read from dPhi (multi-output).
template <int CAP>
__global__ __launch_bounds__(64) void k_gen_moments(const double *__restrict__ Phi, const double *__restrict__ T, int ld,
                                                     const double *__restrict__ rowscal, const double *__restrict__ w,
                                                     const double *__restrict__ v, const double *__restrict__ Xr, int de,
                                                     const int *__restrict__ rows, int nrows,
                                                     const unsigned char *__restrict__ pat, int g,
                                                     const double *__restrict__ Psi3, int m, int d,
                                                     const double *__restrict__ P, const double *__restrict__ Sig,
                                                     int nchunk, int rows_per_chunk, double *__restrict__ slab, int nrec,
                                                     double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_IARR(o, GCAP, d)
    GEN_ARR(M, GCAP * GCAP, d * d) GEN_ARR(W, GCAP * GCAP, d * d) GEN_ARR(Mi, GCAP * GCAP, d * d) GEN_ARR(u, GCAP, d)
    GEN_ARR(dl, GCAP, d) GEN_ARR(acc1, GCAP, d) GEN_ARR(cacc, GCAP * GCAP, d * d)
    const int mj = (m + 63) / 64 * 64;                  // basis index fastest, padded to whole waves (as the 2-D grid was)
    for (long it = gt_; it < (long)nchunk * mj; it += nth_) {
        const int chunk = (int)(it / mj), j = (int)(it % mj);
        if (j >= m) continue;
        int no = 0;
        for (int c = 0; c < d; ++c)
            if (pat[g * d + c]) o[(no++) * ESI] = c;
        double a0 = 0.0, r1 = 0.0, r2 = 0.0;
        for (int a = 0; a < no; ++a) acc1[(a) * ES] = 0.0;
        for (int a = 0; a < no * GDM; ++a) cacc[(a) * ES] = 0.0;
        const double wj = w ? w[j] : 0.0, vj = v ? v[j] : 0.0;
        const double *Sj = Sig + (size_t)j * d * d;
        const int r0 = chunk * rows_per_chunk, rend = min(nrows, r0 + rows_per_chunk);
        if (!Psi3) {   // without input noise M = Sigma_j,oo is the same for every row of the pattern: invert once
            for (int a = 0; a < no; ++a)
                for (int b = 0; b <= a; ++b) M[(a * GDM + b) * ES] = Sj[o[(a) * ESI] * d + o[(b) * ESI]];
            chol_small(M, no, GDM, ES);
            inv_from_chol(M, no, W, Mi, GDM, ES);
        }
        for (int rr = r0; rr < rend; ++rr) {
            const int i = rows[rr];
            const double ph = Phi[(size_t)i * ld + j];
            double dp;
            if (rowscal) {
                const double *rs = rowscal + (size_t)i * 4;
                dp = (-rs[0] * T[(size_t)i * ld + j] - rs[1] * wj + rs[2] * vj) * ph;   // GPz.m:72,90,106,113
                r1 = fma(ph, rs[1], r1);
                r2 = fma(ph, rs[2], r2);
            } else {
                dp = T[(size_t)i * ld + j];                                     // dPHI already formed (k > 1)
            }
            for (int a = 0; a < no; ++a) dl[(a) * ES] = Xr[(size_t)i * de + o[(a) * ESI]] - P[(size_t)j * de + o[(a) * ESI]];
            if (Psi3) {
                for (int a = 0; a < no; ++a)
                    for (int b = 0; b <= a; ++b)
                        M[(a * GDM + b) * ES] = Sj[o[(a) * ESI] * d + o[(b) * ESI]] + Psi3[(size_t)i * d * d + o[(a) * ESI] + d * o[(b) * ESI]];
                chol_small(M, no, GDM, ES);
                inv_from_chol(M, no, W, Mi, GDM, ES);                               // iPSoo   GPz.m:170
            }
            for (int a = 0; a < no; ++a) {
                double s = 0.0;
                for (int b = 0; b < no; ++b) s = fma(Mi[(a * GDM + b) * ES], dl[(b) * ES], s);
                u[(a) * ES] = s;
            }
            a0 += dp;
            for (int a = 0; a < no; ++a) {
                acc1[(a) * ES] = fma(dp, u[(a) * ES], acc1[(a) * ES]);                               // GPz.m:152,172
                for (int b = 0; b < no; ++b) cacc[(a * GDM + b) * ES] = fma(dp, u[(a) * ES] * u[(b) * ES] - Mi[(a * GDM + b) * ES], cacc[(a * GDM + b) * ES]);
            }
        }
        double *rec = slab + ((size_t)chunk * m + j) * nrec;
        for (int q = 0; q < nrec; ++q) rec[q] = 0.0;
        rec[0] = a0;
        for (int a = 0; a < no; ++a) {
            rec[1 + o[(a) * ESI]] = acc1[(a) * ES];
            for (int b = 0; b < no; ++b) rec[1 + d + o[(a) * ESI] * d + o[(b) * ESI]] = cacc[(a * GDM + b) * ES];
        }
        rec[1 + d + d * d] = r1;
        rec[2 + d + d * d] = r2;
    }
}

// Chain the per-(pattern, basis) records to dP, dGamma (GPz.m:151-159,174-181) and the column sums.
// recs: [G][m][nrec] reduced over chunks.  Writes grad dP (scaled), dGamma into grad (VC) or dGfull (GC), cols[2][mp].
// One thread per (basis, pattern): the pattern's contribution [dP (d) | dGamma (d x d) | r1 | r2] goes to
// part[(g*m + j)*(d + d*d + 2)]; k_gen_finish_sum adds the patterns up in fixed order and writes the gradient blocks.
template <int CAP>
__global__ __launch_bounds__(64) void k_gen_finish(const double *__restrict__ recs, int G,
                                                    const unsigned char *__restrict__ pat, int m, int d, int de,
                                                    const double *__restrict__ Gam, const double *__restrict__ Sig,
                                                    const double *__restrict__ iSig, double *__restrict__ part, int nrec,
                                                    int raw, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_ARR(dP, GCAP, d) GEN_ARR(dG, GCAP * GCAP, d * d)
    GEN_ARR(Soo, GCAP * GCAP, d * d) GEN_ARR(W, GCAP * GCAP, d * d) GEN_ARR(Sinv, GCAP * GCAP, d * d)
    GEN_ARR(dS, GCAP * GCAP, d * d) GEN_ARR(tmp, GCAP * GCAP, d * d) GEN_ARR(Kuo, GCAP * GCAP, d * d)
    GEN_ARR(Aeff, GCAP * GCAP, d * d) GEN_ARR(y, GCAP, d) GEN_ARR(dgo, GCAP, d)
    GEN_IARR(o, GCAP, d) GEN_IARR(uix, GCAP, d)
    const int mj = (m + 63) / 64 * 64;
    for (long it = gt_; it < (long)G * mj; it += nth_) {
    const int g = (int)(it / mj), j = (int)(it % mj);
    if (j >= m) continue;
    for (int c = 0; c < d; ++c) dP[(c) * ES] = 0.0;
    for (int e = 0; e < d * GDM; ++e) dG[(e) * ES] = 0.0;
    double r1 = 0.0, r2 = 0.0;
    const double *Sj = Sig + (size_t)j * d * d, *iSj = iSig + (size_t)j * d * d;
    const double *Gj = Gam + (size_t)j * de * de;
    {
        const double *rec = recs + ((size_t)g * m + j) * nrec;
        int no = 0, nu = 0;
        for (int c = 0; c < d; ++c) {
            if (pat[g * d + c]) o[(no++) * ESI] = c; else uix[(nu++) * ESI] = c;
        }
        r1 += rec[1 + d + d * d];
        r2 += rec[2 + d + d * d];
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b) Soo[(a * GDM + b) * ES] = Sj[o[(a) * ESI] * d + o[(b) * ESI]];
        for (int e = 0; e < no * GDM; ++e) tmp[(e) * ES] = Soo[(e) * ES];
        chol_small(tmp, no, GDM, ES);
        if (raw) {
            // No input noise: the records are the plain sums M1 = sum dPHI Delta_o and S = sum dPHI Delta_o Delta_o'.
            // dP_o = Sigma_oo^-1 M1 (GPz.m:153) by two triangular solves, and since dSoo = 1/2 Sigma_oo^-1 S Sigma_oo^-1
            // (GPz.m:154, the A0 terms of :174 cancel), diSoo = -Soo dSoo Soo = -1/2 S: no inverse, no products.
            for (int a = 0; a < no; ++a) {
                double t = rec[1 + o[(a) * ESI]];
                for (int q = 0; q < a; ++q) t = fma(-tmp[(a * GDM + q) * ES], y[(q) * ES], t);
                y[(a) * ES] = t / tmp[(a * GDM + a) * ES];
            }
            for (int a = no - 1; a >= 0; --a) {
                double t = y[(a) * ES];
                for (int q = a + 1; q < no; ++q) t = fma(-tmp[(q * GDM + a) * ES], y[(q) * ES], t);
                y[(a) * ES] = t / tmp[(a * GDM + a) * ES];
                dP[(o[(a) * ESI]) * ES] += y[(a) * ES];
            }
            for (int a = 0; a < no; ++a)
                for (int b = 0; b < no; ++b) dS[(a * GDM + b) * ES] = -0.5 * rec[1 + d + o[(a) * ESI] * d + o[(b) * ESI]];
        } else {
        for (int a = 0; a < no; ++a) dP[(o[(a) * ESI]) * ES] += rec[1 + o[(a) * ESI]];
        // Sigma_oo^-1
        inv_from_chol(tmp, no, W, Sinv, GDM, ES);
        // dSoo = 1/2 (A0 Sigma_oo^-1 + Cacc)       GPz.m:174
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b)
                dS[(a * GDM + b) * ES] = 0.5 * (rec[0] * Sinv[(a * GDM + b) * ES] + rec[1 + d + o[(a) * ESI] * d + o[(b) * ESI]]);
        // diSoo = -Soo dSoo Soo                    GPz.m:176
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b) {
                double s = 0.0;
                for (int q = 0; q < no; ++q) s = fma(Soo[(a * GDM + q) * ES], dS[(q * GDM + b) * ES], s);
                tmp[(a * GDM + b) * ES] = s;
            }
        for (int a = 0; a < no; ++a)
            for (int b = 0; b < no; ++b) {
                double s = 0.0;
                for (int q = 0; q < no; ++q) s = fma(tmp[(a * GDM + q) * ES], Soo[(q * GDM + b) * ES], s);
                dS[(a * GDM + b) * ES] = -s;                                       // now diSoo
            }
        }
        // GuuGuo = iSigma_uu^-1 iSigma_uo  (nu x no)   GPz.m:156,178
        if (nu > 0) {
            for (int a = 0; a < nu; ++a)
                for (int b = 0; b < nu; ++b) tmp[(a * GDM + b) * ES] = iSj[uix[(a) * ESI] * d + uix[(b) * ESI]];
            chol_small(tmp, nu, GDM, ES);
            inv_from_chol(tmp, nu, W, Sinv, GDM, ES);                           // Sinv = iSigma_uu^-1
            for (int a = 0; a < nu; ++a)
                for (int b = 0; b < no; ++b) {
                    double s = 0.0;
                    for (int q = 0; q < nu; ++q) s = fma(Sinv[(a * GDM + q) * ES], iSj[uix[(q) * ESI] * d + o[(b) * ESI]], s);
                    Kuo[(a * GDM + b) * ES] = s;
                }
        }
        // Aeff = Gamma(:,o) - Gamma(:,u) GuuGuo  (d x no);  dGo = 2 Aeff diSoo     GPz.m:157,179
        for (int r = 0; r < d; ++r)
            for (int b = 0; b < no; ++b) {
                double s = Gj[r * de + o[(b) * ESI]];
                for (int q = 0; q < nu; ++q) s = fma(-Gj[r * de + uix[(q) * ESI]], Kuo[(q * GDM + b) * ES], s);
                Aeff[(r * GDM + b) * ES] = s;
            }
        for (int r = 0; r < d; ++r) {
            for (int b = 0; b < no; ++b) {
                double s = 0.0;
                for (int q = 0; q < no; ++q) s = fma(Aeff[(r * GDM + q) * ES], dS[(q * GDM + b) * ES], s);
                dgo[(b) * ES] = 2.0 * s;
                dG[(r * GDM + o[(b) * ESI]) * ES] += dgo[(b) * ES];                               // dGamma(:,o,j) += dGo     GPz.m:158,180
            }
            for (int a = 0; a < nu; ++a) {
                double s = 0.0;
                for (int b = 0; b < no; ++b) s = fma(dgo[(b) * ES], Kuo[(a * GDM + b) * ES], s);
                dG[(r * GDM + uix[(a) * ESI]) * ES] -= s;                                  // dGamma(:,u,j) -= dGo GuuGuo'   GPz.m:159,181
            }
        }
    }
    double *op = part + ((size_t)g * m + j) * (d + d * d + 2);
    for (int c = 0; c < d; ++c) op[c] = dP[(c) * ES];
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) op[d + a * d + b] = dG[(a * GDM + b) * ES];
    op[d + d * d] = r1;
    op[d + d * d + 1] = r2;
    }
}

// One thread per (basis, entry): threads along the entries so the loads of a pattern's block are contiguous.
__global__ void k_gen_finish_sum(const double *__restrict__ part, int G, int m, int d, int method_id,
                                 const double *__restrict__ sums1, int k, double *__restrict__ grad,
                                 double *__restrict__ dGfull, double *__restrict__ cols, int mp) {
    const int md = m * d, np = d + d * d + 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)m * np) return;
    const int j = (int)(t / np), e = (int)(t % np);
    const double nk = sums1[10] * (double)k;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += part[((size_t)g * m + j) * np + e];
    if (e < d) grad[j + m * e] = -s / nk;
    else if (e < d + d * d) {
        const int a = (e - d) / d, b = (e - d) % d;
        if (method_id == 5) grad[md + a + d * b + d * d * j] = -s / nk;
        else dGfull[(size_t)j * d * d + a * d + b] = s;
    } else if (cols) {
        cols[(e == d + d * d ? 0 : mp) + j] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Prediction with input noise, no missing values (predictDiag.m:75-125, predictCov.m:70-132):
//   gamma_i = sum_ab w_a w_b E[phi_a phi_b] - mu_i^2,  nu_i = sum_ab iSigma_w(a,b) E[phi_a phi_b],
//   VlnS_i = sum_ab v_a v_b E[phi_a phi_b] - (ElnS_i - b)^2,  E[phi_a phi_b](x_i) = Z_ab N(x_i | c_ab, C_ab + Psi_i).
// Pair table: one record per (a >= b): [lnZ_ab | c_ab (d) | C_ab (d diag or d*d)].
// ---------------------------------------------------------------------------------------------
template <int CAP>
__global__ void k_pair_table(int kind, int m, int d, int de, const double *__restrict__ P, const double *__restrict__ G,
                             const double *__restrict__ Sig, const double *__restrict__ iSig, double *__restrict__ tab,
                             int rec, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_ARR(A, GCAP * GCAP, d * d) GEN_ARR(W, GCAP * GCAP, d * d) GEN_ARR(Ci, GCAP * GCAP, d * d)
    GEN_ARR(S2, GCAP * GCAP, d * d) GEN_ARR(S2i, GCAP * GCAP, d * d) GEN_ARR(rhs, GCAP, d)
    const long npair = (long)m * (m + 1) / 2;
    for (long e = gt_; e < npair; e += nth_) {
    // e -> (a >= b): a = floor((sqrt(8e+1)-1)/2)
    long a = (long)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > e) --a;
    while ((a + 1) * (a + 2) / 2 <= e) ++a;
    const long b = e - a * (a + 1) / 2;
    double *ot = tab + (size_t)e * rec;
    if (kind == GPZ_KIND_DIAG) {
        double lnz = 0.0;
        for (int c = 0; c < d; ++c) {
            const double ga = G[a * de + c], gb = G[b * de + c];
            const double isa = ga * ga, isb = gb * gb;            // iSigma = Gamma.^2      predictDiag.m:89
            const double sa = 1.0 / isa, sb = 1.0 / isb;          // Sigma = Gamma.^-2
            const double icij = isa + isb;
            const double pa = P[a * de + c], pb = P[b * de + c];
            ot[1 + c] = (pa * isa + pb * isb) / icij;              // cij                    :98
            ot[1 + d + c] = 1.0 / icij;                            // Cij                    :97
            lnz += -0.5 * log(isa) - 0.5 * log(isb) - 0.5 * (pa - pb) * (pa - pb) / (sa + sb) - 0.5 * log(sa + sb);   // :101
        }
        ot[0] = lnz;
    } else {
        const double *iSa = iSig + (size_t)a * d * d, *iSb = iSig + (size_t)b * d * d;
        const double *Sa = Sig + (size_t)a * d * d, *Sb = Sig + (size_t)b * d * d;
        double lnz = 0.0;
        // lnz(a) = -1/2 ln|iSigma_a| = +1/2 ln|Sigma_a|           predictCov.m:91
        for (int pass = 0; pass < 2; ++pass) {
            const double *S = pass ? Sb : Sa;
            for (int r = 0; r < d; ++r)
                for (int c = 0; c <= r; ++c) A[(r * GDM + c) * ES] = S[r * d + c];
            chol_small(A, d, GDM, ES);
            for (int r = 0; r < d; ++r) lnz += log(A[(r * GDM + r) * ES]);          // 1/2 ln|Sigma| = sum ln L_rr
        }
        for (int r = 0; r < d; ++r)
            for (int c = 0; c <= r; ++c) A[(r * GDM + c) * ES] = iSa[r * d + c] + iSb[r * d + c];    // iCij          :100
        chol_small(A, d, GDM, ES);
        inv_from_chol(A, d, W, Ci, GDM, ES);                                                       // Cij = inv(iCij)
        for (int c = 0; c < d; ++c) {
            double s = 0.0;
            for (int r = 0; r < d; ++r) s += P[a * de + r] * iSa[r * d + c] + P[b * de + r] * iSb[r * d + c];
            rhs[(c) * ES] = s;
        }
        for (int c = 0; c < d; ++c) {                                                          // cij = rhs * Cij   :102
            double s = 0.0;
            for (int r = 0; r < d; ++r) s = fma(rhs[(r) * ES], Ci[(r * GDM + c) * ES], s);
            ot[1 + c] = s;
        }
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) ot[1 + d + r * d + c] = Ci[(r * GDM + c) * ES];
        for (int r = 0; r < d; ++r)
            for (int c = 0; c <= r; ++c) S2[(r * GDM + c) * ES] = Sa[r * d + c] + Sb[r * d + c];
        chol_small(S2, d, GDM, ES);
        double ld = 0.0;
        for (int r = 0; r < d; ++r) ld += log(S2[(r * GDM + r) * ES]);
        inv_from_chol(S2, d, W, S2i, GDM, ES);
        double q = 0.0;
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) q += (P[a * de + r] - P[b * de + r]) * S2i[(r * GDM + c) * ES] * (P[a * de + c] - P[b * de + c]);
        ot[0] = lnz - 0.5 * q - ld;                                                             // :105  (-1/2 ln|Sa+Sb| = -ld)
    }
    }
}

// One thread per sample, pairs [p0, p1) of this chunk; partial sums part[chunk][3][k][n_pad].
template <int CAP>
__global__ __launch_bounds__(64) void k_predict_noisy(int kind, int n, long ldx, int m, int d, int de, int k,
                                                       const double *__restrict__ Xr, const double *__restrict__ Psir,
                                                       const double *__restrict__ Psi3, const double *__restrict__ tab,
                                                       int rec, const double *__restrict__ w, const double *__restrict__ v,
                                                       const double *__restrict__ iS, long pairs_per_chunk,
                                                       double *__restrict__ part, int nchunk, double *ws, size_t ws_stride) {
    extern __shared__ double sums_lds[];               // [3][k][64]: gamma, VlnS, nu partial sums of this lane (any k)
    GEN_SETUP();
    GEN_ARR(x, GCAP, d) GEN_ARR(ps, GCAP, d) GEN_ARR(M, GCAP * GCAP, d * d) GEN_ARR(y, GCAP, d)
    double *ga = sums_lds + threadIdx.x, *vl = ga + (size_t)k * 64, *nu = vl + (size_t)k * 64;   // element o at [o * 64]
    const long nrb = ((long)n + 63) / 64 * 64;         // rows padded to whole waves; item = chunk * nrb + row
    for (long it = gt_; it < (long)nchunk * nrb; it += nth_) {
    const int i = (int)(it % nrb), chunk = (int)(it / nrb);
    if (i >= n) continue;
    const long npair = (long)m * (m + 1) / 2;
    const long p0 = (long)chunk * pairs_per_chunk, p1 = min(npair, p0 + pairs_per_chunk);
    for (int o = 0; o < k; ++o) { ga[o * 64] = 0.0; vl[o * 64] = 0.0; nu[o * 64] = 0.0; }
    for (int c = 0; c < d; ++c) {
        x[(c) * ES] = Xr[(size_t)i * de + c];
        if (kind == GPZ_KIND_DIAG) ps[(c) * ES] = Psir[(size_t)i * de + c];
    }
    // recover (a, b) of the first pair, then walk
    long a = (long)((sqrt(8.0 * (double)p0 + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > p0) --a;
    while ((a + 1) * (a + 2) / 2 <= p0) ++a;
    long b = p0 - a * (a + 1) / 2;
    for (long e = p0; e < p1; ++e) {
        const double *t = tab + (size_t)e * rec;
        double ln;
        if (kind == GPZ_KIND_DIAG) {
            double q = 0.0, pr = 1.0;
            for (int c = 0; c < d; ++c) {
                const double cp = t[1 + d + c] + ps[(c) * ES];                     // Cij + Psi          :105
                const double dl = x[(c) * ES] - t[1 + c];
                q = fma(dl * dl, 1.0 / cp, q);
                pr *= cp;
            }
            ln = -0.5 * q - 0.5 * log(pr);                                  // :107
        } else {
            for (int r = 0; r < d; ++r)
                for (int c = 0; c <= r; ++c) M[(r * GDM + c) * ES] = t[1 + d + r * d + c] + Psi3[(size_t)i * d * d + r + d * c];
            chol_small(M, d, GDM, ES);
            double q = 0.0, ld = 0.0;
            for (int r = 0; r < d; ++r) {
                double s = x[(r) * ES] - t[1 + r];
                for (int c = 0; c < r; ++c) s = fma(-M[(r * GDM + c) * ES], y[(c) * ES], s);
                y[(r) * ES] = s / M[(r * GDM + r) * ES];
                q = fma(y[(r) * ES], y[(r) * ES], q);
                ld += log(M[(r * GDM + r) * ES]);
            }
            ln = -0.5 * q - ld;                                             // predictCov.m:111
        }
        const double z = ((a == b) ? 1.0 : 2.0) * exp(t[0] + ln);           // 2x in the loop, -1x for a == b  (:113-119)
        for (int o = 0; o < k; ++o) {
            ga[o * 64] = fma(z, w[a + (size_t)m * o] * w[b + (size_t)m * o], ga[o * 64]);
            vl[o * 64] = fma(z, v ? v[a + (size_t)m * o] * v[b + (size_t)m * o] : 0.0, vl[o * 64]);
            nu[o * 64] = fma(z, iS[a + (size_t)m * b + (size_t)m * m * o], nu[o * 64]);
        }
        if (++b > a) { ++a; b = 0; }
    }
    for (int o = 0; o < k; ++o) {
        part[(((size_t)chunk * 3 + 0) * k + o) * ldx + i] = ga[o * 64];
        part[(((size_t)chunk * 3 + 1) * k + o) * ldx + i] = vl[o * 64];
        part[(((size_t)chunk * 3 + 2) * k + o) * ldx + i] = nu[o * 64];
    }
    }
}

// sums[3][k][ldx] (gamma, VlnS, nu raw) -> gamma, nu, beta_i   (predictDiag.m:121-125)
__global__ void k_predict_noisy_final(const double *__restrict__ sums, long ldx, int n, int k,
                                      const double *__restrict__ mu, const double *__restrict__ lnbeta,
                                      const double *__restrict__ bvec, double *__restrict__ gamma,
                                      double *__restrict__ nu, double *__restrict__ beta_i) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int o = 0; o < k; ++o) {
        const double muv = mu[(size_t)o * ldx + i], els = lnbeta[(size_t)o * ldx + i];
        const double vl = sums[((size_t)1 * k + o) * ldx + i] - (els - bvec[o]) * (els - bvec[o]);
        gamma[(size_t)o * ldx + i] = sums[((size_t)0 * k + o) * ldx + i] - muv * muv;
        nu[(size_t)o * ldx + i] = sums[((size_t)2 * k + o) * ldx + i];
        beta_i[(size_t)o * ldx + i] = exp(els) * (1.0 + 0.5 * vl);
    }
}

// getPrior.m:7-20: one fixed-point iteration  prior <- mean_i( N_i. .* prior / sum_j N_ij prior_j ).
// Workgroup per row (grid-stride), lanes along basis functions; per-workgroup column sums in colslab[wg][mp].
template <int NJ>
__global__ __launch_bounds__(256) void k_prior_iter(const double *__restrict__ N, int ld, int n, int m,
                                                     const double *__restrict__ prior, double *__restrict__ colslab) {
    __shared__ double sh4[4];
    const int tid = threadIdx.x;
    double pq[NJ], acc[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = tid + 256 * q;
        pq[q] = (j < m) ? prior[j] : 0.0;
        acc[q] = 0.0;
    }
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        double wv[NJ], part = 0.0;
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int j = tid + 256 * q;
            wv[q] = (j < m) ? N[(size_t)i * ld + j] * pq[q] : 0.0;
            part += wv[q];
        }
        const double tot = block_sum_256(part, sh4);
#pragma unroll
        for (int q = 0; q < NJ; ++q) acc[q] += wv[q] / tot;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = tid + 256 * q;
        if (j < m) colslab[(size_t)blockIdx.x * m + j] = acc[q];
    }
}

// The same iteration for m <= 256 with a WAVE per row (lane l holds columns l, l + 64, l + 128, l + 192): the row sum is a wave
// reduction - no LDS, no barrier in the row loop (the workgroup-per-row form above pays two barriers per row: 88 us a pass at
// n = 1e5, m = 200, where the 160 MB of N are 40 us of HBM time; getPrior runs up to 100 passes, twice per train()).
__global__ __launch_bounds__(256) void k_prior_iter_wave(const double *__restrict__ N, int ld, int n, int m,
                                                          const double *__restrict__ prior, double *__restrict__ colslab) {
    __shared__ double sacc[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double pq[4], acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        pq[q] = (j < m) ? prior[j] : 0.0;
        acc[q] = 0.0;
    }
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        double wv[4], part = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            wv[q] = (j < m) ? N[(size_t)i * ld + j] * pq[q] : 0.0;
            part += wv[q];
        }
        const double tot = wave_sum(part);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += wv[q] / tot;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sacc[wave][lane + 64 * q] = acc[q];
    __syncthreads();
    const int j = threadIdx.x;
    if (j < m) colslab[(size_t)blockIdx.x * m + j] = ((sacc[0][j] + sacc[1][j]) + sacc[2][j]) + sacc[3][j];
}

// prior <- colsum / n (getPrior.m:15: mean(w)), a copy kept in `keep` for the host's convergence test
__global__ void k_prior_update(const double *__restrict__ colsum, double ns, int m, double *__restrict__ prior, double *__restrict__ keep) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const double v = colsum[j] / ns;
    prior[j] = v;
    keep[j] = v;
}
void launch_prior_update(hipStream_t st, const double *colsum, double ns, int m, double *prior, double *keep) {
    hipLaunchKernelGGL(k_prior_update, dim3((m + 255) / 256), dim3(256), 0, st, colsum, ns, m, prior, keep);
}

void launch_prior_iter(hipStream_t st, const double *N, int ld, int n, int m, const double *prior, double *colslab,
                       int nwg) {
    const int nj = (m + 255) / 256;
    dim3 g(nwg), b(256);
    if (m <= 256) {
        hipLaunchKernelGGL(k_prior_iter_wave, g, b, 0, st, N, ld, n, m, prior, colslab);
        return;
    }
    if (nj <= 1) hipLaunchKernelGGL(k_prior_iter<1>, g, b, 0, st, N, ld, n, m, prior, colslab);
    else if (nj <= 2) hipLaunchKernelGGL(k_prior_iter<2>, g, b, 0, st, N, ld, n, m, prior, colslab);
    else if (nj <= 4) hipLaunchKernelGGL(k_prior_iter<4>, g, b, 0, st, N, ld, n, m, prior, colslab);
    else if (nj <= 8) hipLaunchKernelGGL(k_prior_iter<8>, g, b, 0, st, N, ld, n, m, prior, colslab);
    else hipLaunchKernelGGL(k_prior_iter<16>, g, b, 0, st, N, ld, n, m, prior, colslab);
}

// grid of a general-path kernel: one thread per item in 64-thread workgroups (d <= 20), or the fixed grid-stride pool
// over the runtime-d workspace
#define GEN_LAUNCH(KERNEL, items, lds, ...)                                                                            \
    do {                                                                                                               \
        if (d <= GCAP)                                                                                                 \
            hipLaunchKernelGGL((KERNEL<GCAP>), dim3((unsigned)(((items) + 63) / 64)), dim3(64), lds, st, __VA_ARGS__,   \
                               (double *)nullptr, (size_t)0);                                                          \
        else                                                                                                           \
            hipLaunchKernelGGL((KERNEL<0>), dim3(gen_rt_threads(d) / 64), dim3(64), lds, st, __VA_ARGS__, ws,          \
                               gen_ws_per_thread(d));                                                                  \
    } while (0)

void launch_pair_table(hipStream_t st, int kind, int m, int d, int de, const double *P, const double *G, const double *Sig,
                       const double *iSig, double *tab, int rec, double *ws) {
    const long npair = (long)m * (m + 1) / 2;
    GEN_LAUNCH(k_pair_table, npair, 0, kind, m, d, de, P, G, Sig, iSig, tab, rec);
}

void launch_predict_noisy(hipStream_t st, int kind, int n, long ldx, int m, int d, int de, int k, const double *Xr,
                          const double *Psir, const double *Psi3, const double *tab, int rec, const double *w,
                          const double *v, const double *iS, int nchunk, long pairs_per_chunk, double *part, double *ws,
                          int flags) {
    // covariance kinds, d <= 10: register-resident per-pair factorisations (k_psi.hip) instead of the scratch-resident branch
    if (kind != GPZ_KIND_DIAG && k <= 8 &&
        launch_predict_noisy_cov(st, n, ldx, m, d, de, k, Xr, Psi3, tab, rec, w, v, iS, nchunk, pairs_per_chunk, part, flags) == 0)
        return;
    if (kind == GPZ_KIND_DIAG &&
        launch_predict_noisy_diag(st, n, ldx, m, d, de, k, Xr, Psir, tab, rec, w, v, iS, nchunk, pairs_per_chunk, part) == 0)
        return;
    const long items = (((long)n + 63) / 64 * 64) * nchunk;
    const size_t lds = (size_t)3 * k * 64 * sizeof(double);
    GEN_LAUNCH(k_predict_noisy, items, lds, kind, n, ldx, m, d, de, k, Xr, Psir, Psi3, tab, rec, w, v, iS, pairs_per_chunk,
               part, nchunk);
}

void launch_predict_noisy_final(hipStream_t st, const double *sums, long ldx, int n, int k, const double *mu,
                                const double *lnbeta, const double *b, double *gamma, double *nu, double *beta_i) {
    hipLaunchKernelGGL(k_predict_noisy_final, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sums, ldx, n, k, mu, lnbeta,
                       b, gamma, nu, beta_i);
}

// N_ij = PHI_ij * exp(cn), cn = -1/2 ln|Sigma_j,oo| - 1/2 |o| ln 2pi + 1/2 |u| ln 2      (getPHI.m:77,87,98,105)
__global__ void k_phi_norm(NormArgs a) {
    const size_t gs = (size_t)blockDim.x * gridDim.x;
    const size_t tot = (size_t)a.n * a.m;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gs) {
        const size_t i = e / a.m;
        const int j = (int)(e % a.m);
        double cn;
        if (a.kind == GPZ_KIND_DIAG) {
            // -1/2 sum_o ln sigma = sum_o ln|gamma|
            double s = 0.0, nu = a.ucnt ? a.ucnt[i] : 0.0;
            for (int c = 0; c < a.d; ++c) {
                const double mk = a.Mr ? a.Mr[i * a.de + c] : 1.0;
                if (mk != 0.0) s += log(fabs(a.G[(size_t)j * a.de + c]));
            }
            cn = s - 0.5 * ((double)a.d - nu) * GPZ_LOG2PI + 0.5 * nu * GPZ_LOG2;
        } else if (a.gen) {
            const int g = a.gid[i];
            int no = 0;
            for (int c = 0; c < a.d; ++c) no += a.pat[g * a.d + c] ? 1 : 0;
            cn = -0.5 * a.lnS[(size_t)g * a.m + j] - 0.5 * no * GPZ_LOG2PI + 0.5 * (a.d - no) * GPZ_LOG2;
        } else {
            // -1/2 ln|Sigma_j| = ln|det Gamma_j| = sum ln|R_aa|
            const int nt = a.de * (a.de + 1) / 2;
            const double *rj = a.Rc + (size_t)j * (nt + a.de);
            double s = 0.0;
            for (int q = 0; q < a.d; ++q) s += log(fabs(rj[q * a.de - q * (q - 1) / 2]));
            cn = s - 0.5 * a.d * GPZ_LOG2PI;
        }
        a.N[i * a.ld + j] = a.Phi[i * a.ld + j] * exp(cn);
    }
}

// The same without missing values on the tuned routes: cn depends on the basis function alone, so a thread keeps ONE column and walks
// rows - d logarithms and an exponential once per thread instead of once per element (0.73 ms -> the time of the copy at n = 1e5, m = 200).
__global__ __launch_bounds__(256) void k_phi_norm_cols(NormArgs a) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= a.m) return;
    double s = 0.0;
    if (a.kind == GPZ_KIND_DIAG) {
        for (int c = 0; c < a.d; ++c) s += log(fabs(a.G[(size_t)j * a.de + c]));
    } else {
        const int nt = a.de * (a.de + 1) / 2;
        const double *rj = a.Rc + (size_t)j * (nt + a.de);
        for (int q = 0; q < a.d; ++q) s += log(fabs(rj[q * a.de - q * (q - 1) / 2]));
    }
    const double ecn = exp(s - 0.5 * (double)a.d * GPZ_LOG2PI);
    for (size_t i = blockIdx.x; i < (size_t)a.n; i += gridDim.x) a.N[i * a.ld + j] = a.Phi[i * a.ld + j] * ecn;
}

void launch_phi_norm(hipStream_t st, const NormArgs &a) {
    if (!a.gen && !a.Mr && !a.ucnt) {
        const int nb = a.n < 2048 ? (a.n > 0 ? a.n : 1) : 2048;
        hipLaunchKernelGGL(k_phi_norm_cols, dim3(nb, (a.m + 255) / 256), dim3(256), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(k_phi_norm, dim3(1024), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// Missing dimensions WITHOUT input noise: M = Sigma_j,oo does not depend on the row, so each (pattern, basis) pair has
// one triangular factor and the rows of a pattern (stored contiguously) run through the tuned kernels of k_phi.hip /
// k_rows.hip with a pattern-specific parameter block:
//     ln PHI_ij = -1/2 Delta_o' Sigma_oo^-1 Delta_o - 1/2 |u| ln 2 = -1/2 |R~ x - c~|^2          (getPHI.m:73-76)
// R~ = upper Cholesky factor of Sigma_oo^-1 scattered into d x d (zero rows / columns at the missing dimensions),
// c~ = R~ p_j, and the constant |u| ln 2 rides on the first missing dimension's (otherwise empty) row: c~[u0] = sqrt(|u| ln 2).
// Output in the layout of k_prep_cov: [R~ packed upper, row a at a*de - a(a-1)/2 | c~ (de)] per basis function.
template <int CAP>
__global__ void k_gen_pattern_params(const double *__restrict__ Sig, const double *__restrict__ P,
                                     const unsigned char *__restrict__ pat, int G, int m, int d, int de,
                                     double *__restrict__ RcAll, double *ws, size_t ws_stride) {
    GEN_SETUP();
    GEN_IARR(o, GCAP, d) GEN_ARR(A, GCAP * GCAP, d * d) GEN_ARR(W, GCAP * GCAP, d * d) GEN_ARR(Ki, GCAP * GCAP, d * d)
    const int mj = (m + 63) / 64 * 64;
    for (long it = gt_; it < (long)G * mj; it += nth_) {
    const int j = (int)(it % mj), g = (int)(it / mj);  // one parameter block of m*(nt+de) doubles per pattern
    if (j >= m) continue;
    double *Rc = RcAll + (size_t)g * m * (de * (de + 1) / 2 + de);
    int no = 0, u0 = -1;
    for (int c = 0; c < d; ++c) {
        if (pat[g * d + c]) o[(no++) * ESI] = c;
        else if (u0 < 0) u0 = c;
    }
    for (int a = 0; a < no; ++a)
        for (int b = 0; b < no; ++b) A[(a * GDM + b) * ES] = Sig[(size_t)j * d * d + o[(a) * ESI] * d + o[(b) * ESI]];
    chol_small(A, no, GDM, ES);
    inv_from_chol(A, no, W, Ki, GDM, ES);                  // Ki = Sigma_oo^-1
    chol_small(Ki, no, GDM, ES);                           // Ki = L L'  ->  R~ = L' (upper)
    const int nt = de * (de + 1) / 2;
    double *out = Rc + (size_t)j * (nt + de);
    for (int e = 0; e < nt + de; ++e) out[e] = 0.0;
    for (int a = 0; a < no; ++a) {
        double cs = 0.0;
        for (int b = a; b < no; ++b) {
            const double r = Ki[(b * GDM + a) * ES];          // R~[o_a][o_b] = L[b][a]
            out[o[(a) * ESI] * de - o[(a) * ESI] * (o[(a) * ESI] - 1) / 2 + (o[(b) * ESI] - o[(a) * ESI])] = r;
            cs = fma(r, P[(size_t)j * de + o[(b) * ESI]], cs);
        }
        out[nt + o[(a) * ESI]] = cs;
    }
    if (u0 >= 0) out[nt + u0] = sqrt((double)(d - no) * GPZ_LOG2);
    }
}

// Tuned moment sums of one pattern, [m][nmt (+2)] = [M1 (de) | S packed upper (de(de+1)/2) | r1, r2], scattered into the
// record layout of k_gen_moments ([a0 = 0 | M1 (d) | S (d x d) | r1 | r2], zeros on missing dimensions) for k_gen_finish's
// `raw` mode.  The sums stay linear in the rows, so they can be all-reduced across ranks as they are.
__global__ void k_gen_convert_moments(const double *__restrict__ frecAll, int stride, int has_r,
                                      const unsigned char *__restrict__ pat, int m, int d,
                                      int de, double *__restrict__ recsAll, int nrec) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;                          // [G][m][stride] in, [G][m][nrec] out
    if (j >= m) return;
    const double *f = frecAll + ((size_t)g * m + j) * stride;
    double *rec = recsAll + ((size_t)g * m + j) * nrec;
    const unsigned char *pg = pat + (size_t)g * d;
    rec[0] = 0.0;
    for (int a = 0; a < d; ++a) {
        rec[1 + a] = pg[a] ? f[a] : 0.0;
        for (int b = 0; b < d; ++b) {
            const int lo = a < b ? a : b, hi = a < b ? b : a;   // packed upper of the tuned kernels, symmetric
            rec[1 + d + a * d + b] = (pg[a] && pg[b]) ? f[de + lo * de - lo * (lo - 1) / 2 + (hi - lo)] : 0.0;
        }
    }
    rec[1 + d + d * d] = has_r ? f[stride - 2] : 0.0;
    rec[2 + d + d * d] = has_r ? f[stride - 1] : 0.0;
}

void launch_gen_pattern_params(hipStream_t st, const double *Sig, const double *P, const unsigned char *pat, int G, int m,
                               int d, int de, double *RcAll, double *ws) {
    GEN_LAUNCH(k_gen_pattern_params, (long)G * ((m + 63) / 64 * 64), 0, Sig, P, pat, G, m, d, de, RcAll);
}
void launch_gen_convert_moments(hipStream_t st, const double *frecAll, int stride, int has_r, const double *Sig,
                                const unsigned char *pat, int G, int m, int d, int de, double *recsAll, int nrec) {
    (void)Sig;
    hipLaunchKernelGGL(k_gen_convert_moments, dim3((m + 63) / 64, G), dim3(64), 0, st, frecAll, stride, has_r, pat, m, d, de,
                       recsAll, nrec);
}

void launch_gen_prep(hipStream_t st, const double *G, int m, int d, int de, double *Sig, double *iSig,
                     const unsigned char *pat, int ngroups, double *lnS, double *ws) {
    GEN_LAUNCH(k_gen_prep, (long)m, 0, G, m, d, de, Sig, iSig);
    GEN_LAUNCH(k_gen_lndet, (long)ngroups * m, 0, (const double *)Sig, pat, ngroups, m, d, lnS);
}

void launch_gen_phi(hipStream_t st, const GenRows &r, int m, int mp, int d, int de, int k, const double *P,
                    const double *Sig, const double *lnS, const unsigned char *pat, double *Phi, const double *Y, double *ws) {
    if (r.n > 0)   // a rank of a sharded run may hold no row of this set
        GEN_LAUNCH(k_gen_phi, (long)r.n, 0, r.Xr, de, r.gid, pat, r.Psi3, r.n, m, d, P, Sig, lnS, Phi, mp);
    hipLaunchKernelGGL(k_gen_fill, dim3(1024), dim3(256), 0, st, Phi, mp, r.n, r.n_pad, m, mp, k, Y, (long)r.n_pad);
}

void launch_gen_fill(hipStream_t st, double *Phi, int ld, int n, int n_pad, int m, int mp, int k, const double *Y) {
    hipLaunchKernelGGL(k_gen_fill, dim3(1024), dim3(256), 0, st, Phi, ld, n, n_pad, m, mp, k, Y, (long)n_pad);
}

void launch_gen_rowdot(hipStream_t st, const double *Phi, int ld, int n, long ldx, int m, int k, const double *v,
                       const double *b, const double *omega, const double *w, double *lnbeta, double *wbeta,
                       double *phiw, long om_ld) {
    hipLaunchKernelGGL(k_gen_rowdot, dim3((unsigned)((ldx + 3) / 4)), dim3(256), 0, st, Phi, ld, n, ldx, m, k, v, b, omega, om_ld, w,
                       lnbeta, wbeta, phiw);
}

void launch_gen_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                        const double *v, const GenRows &r, int g, int row_begin, int nrows, const unsigned char *pat, int m,
                        int d, int de, const double *P, const double *Sig, int nchunk, int rows_per_chunk, double *slab,
                        int nrec, double *ws) {
    if (nrows <= 0) return;
    GEN_LAUNCH(k_gen_moments, (long)nchunk * ((m + 63) / 64 * 64), 0, Phi, T, ld, rowscal, w, v, r.Xr, de,
               r.rows_by_group + row_begin, nrows, pat, g, r.Psi3, m, d, P, Sig, nchunk, rows_per_chunk, slab, nrec);
}

void launch_gen_finish(hipStream_t st, const double *recs, int G, const unsigned char *pat, int m, int d, int de,
                       const double *Gam, const double *Sig, const double *iSig, int method_id, const double *sums1, int k,
                       double *grad, double *dGfull, double *cols, int mp, int nrec, double *part, int raw, double *ws) {
    GEN_LAUNCH(k_gen_finish, (long)G * ((m + 63) / 64 * 64), 0, recs, G, pat, m, d, de, Gam, Sig, iSig, part, nrec, raw);
    const long nt = (long)m * (d + d * d + 2);
    hipLaunchKernelGGL(k_gen_finish_sum, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, (const double *)part, G, m, d,
                       method_id, sums1, k, grad, dGfull, cols, mp);
}
