// Runtime-dimension kernels: the PHI build, the dP/dGamma moment sums and the QR preparation for inputs WIDER than the
// instantiated kernels (d > 20) or with more than 8 outputs.  The reference is generic in d and k (getPHI.m:60-110,
// GPz.m:133-213 are loops over columns); the tuned kernels of k_phi.hip / k_rows.hip keep a row of X and the d + d(d+1)/2
// accumulators of a basis function in registers and are instantiated for d <= 20 only.  Here the same arithmetic runs
// with the row data in LDS and the accumulators cut into compile-time blocks of 8 dimensions — same argument structs,
// same output layouts (so the by-pattern drivers, the slab sums and k_finish_* are shared), about an order of magnitude
// below the tuned kernels' throughput (DESIGN.md section 7 has the measured cost).
//
//   k_phi_wide      lanes along rows (one wave = 64 rows per workgroup), X / Psi / mask tiles in LDS, per-basis parameters
//                   wave-uniform (scalar loads), 8 basis functions per pass so a lane writes 64-byte runs of PHI; the
//                   ln beta / PHI w running sums live in LDS (any k).
//   k_moments_wide  lanes along basis functions, block z of the grid = one 8-wide block of dimensions (diag kinds) or one
//                   8 x 8 block of the upper triangle of S_j (cov kinds); dPHI is read from T.
//   k_form_dphi     single-output fused route: T <- dPHI in place + the column sums PHI'c, PHI'dbeta (GPz.m:72,89,104).
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define WB 8   // block width (dimensions per accumulator block, basis functions per PHI pass)

// ---------------------------------------------------------------------------------------------
// PHI build (getPHI.m:60-125), any d, any k.  Argument meaning as in PhiArgs / k_phi_diag / k_phi_cov.
// ---------------------------------------------------------------------------------------------
// TILES = false: the row tiles would not fit a CU's LDS (d beyond ~100 ... 300, by the number of tiles); x / psi / mask are then read
// from the column-major arrays where they are used (lanes along rows: coalesced, served by the L1 / L2 after the first pass).
template <int KIND, bool PSI, bool MASK, bool TILES>
__global__ __launch_bounds__(64) void k_phi_wide(PhiArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, d = a.d, k = a.k, m = a.m, mp = a.mp;
    const long ldx = a.ldx;
    const size_t td = TILES ? (size_t)d : 0;
    double *xs = lds;                                   // [d][64]
    double *pss = xs + td * 64;                         // [d][64] when PSI
    double *mks = pss + (PSI ? td * 64 : 0);            // [d][64] when MASK
    double *svs = mks + (MASK ? td * 64 : 0);           // [k][64]
    double *sws = svs + (size_t)k * 64;                 // [k][64]
    long row0, rend;
    const double *Gp = a.G;
    const int nt = d * (d + 1) / 2;
    if (a.wgtab) {                                      // rows sorted by NaN pattern: this workgroup's range and parameter block
        const int *t = a.wgtab + 4 * blockIdx.x;
        row0 = t[0];
        rend = t[1];
        Gp = a.G + (size_t)t[2] * m * (nt + d);
    } else {
        row0 = (long)blockIdx.x * 64;
        rend = a.n;
    }
    const long i = row0 + lane;
    const bool valid = i < rend;
    const bool inb = a.wgtab ? valid : (i < a.n_pad);   // rows this lane may write
    const long il = inb ? i : row0;                     // clamped row for loads
    if (TILES)
        for (int c = 0; c < d; ++c) {
            xs[c * 64 + lane] = a.Xc[c * ldx + il];
            if (PSI) pss[c * 64 + lane] = a.Psic[c * ldx + il];
            if (MASK) mks[c * 64 + lane] = a.Mc[c * ldx + il];
        }
    auto XS = [&](int c) { return TILES ? xs[c * 64 + lane] : a.Xc[c * ldx + il]; };
    auto PS = [&](int c) { return TILES ? pss[c * 64 + lane] : a.Psic[c * ldx + il]; };
    auto MK = [&](int c) { return TILES ? mks[c * 64 + lane] : a.Mc[c * ldx + il]; };
    for (int o = 0; o < k; ++o) { svs[o * 64 + lane] = 0.0; sws[o * 64 + lane] = 0.0; }
    const double q0 = (MASK && a.ucnt) ? a.ucnt[il] * GPZ_LOG2 : 0.0;   // |u_i| ln 2

    for (int j0 = 0; j0 < mp; j0 += WB) {
        double ph[WB];
        if (j0 < m) {
            double q[WB];
            int jc[WB];
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) { jc[jj] = min(j0 + jj, m - 1); q[jj] = q0; }
            if (KIND == GPZ_KIND_DIAG) {
                double lg[WB], pr[WB];
#pragma unroll
                for (int jj = 0; jj < WB; ++jj) { lg[jj] = 0.0; pr[jj] = 1.0; }
                for (int c = 0; c < d; ++c) {
                    const double x = XS(c);
                    const double mk = MASK ? MK(c) : 1.0;
                    const double ps = PSI ? PS(c) : 0.0;
#pragma unroll
                    for (int jj = 0; jj < WB; ++jj) {
                        const double pc = a.P[(size_t)jc[jj] * d + c], gc = Gp[(size_t)jc[jj] * d + c];   // G = gamma^2
                        const double dl = (x - pc) * mk;
                        if (PSI) {
                            const double u = fma(ps, gc, 1.0);                 // 1 + psi/sigma   (getPHI.m:104)
                            q[jj] = fma(dl * dl, gc * gpz_rcp(u), q[jj]);
                            pr[jj] *= u;
                        } else {
                            q[jj] = fma(dl * dl, gc, q[jj]);                   // getPHI.m:97
                        }
                    }
                    if (PSI && (c & 7) == 7) {                                 // keep the running product in range for wide inputs
#pragma unroll
                        for (int jj = 0; jj < WB; ++jj) { lg[jj] += log(pr[jj]); pr[jj] = 1.0; }
                    }
                }
                if (PSI) {
#pragma unroll
                    for (int jj = 0; jj < WB; ++jj) q[jj] += lg[jj] + log(pr[jj]);
                }
            } else {
                // |R_j x - c_j|^2, R_j packed upper row-major, c_j behind it (k_prep_cov / k_gen_pattern_params layout)
                const double *rj[WB];
#pragma unroll
                for (int jj = 0; jj < WB; ++jj) rj[jj] = Gp + (size_t)jc[jj] * (nt + d);
                for (int aa = 0; aa < d; ++aa) {
                    const int roff = aa * d - aa * (aa - 1) / 2 - aa;          // + b
                    double s[WB];
#pragma unroll
                    for (int jj = 0; jj < WB; ++jj) s[jj] = -rj[jj][nt + aa];
                    for (int b = aa; b < d; ++b) {
                        const double x = XS(b);
#pragma unroll
                        for (int jj = 0; jj < WB; ++jj) s[jj] = fma(rj[jj][roff + b], x, s[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < WB; ++jj) q[jj] = fma(s[jj], s[jj], q[jj]);   // getPHI.m:73,76
                }
            }
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) {
                const int j = j0 + jj;
                if (j < m) {
                    ph[jj] = valid ? exp(-0.5 * q[jj]) : 0.0;                  // getPHI.m:113
                    for (int o = 0; o < k; ++o) {
                        if (a.v) svs[o * 64 + lane] = fma(ph[jj], a.v[j + (size_t)m * o], svs[o * 64 + lane]);   // getPHI.m:124
                        if (a.w) sws[o * 64 + lane] = fma(ph[jj], a.w[j + (size_t)m * o], sws[o * 64 + lane]);
                    }
                } else {
                    ph[jj] = (a.Y != nullptr && (j - m) < k && valid) ? a.Y[(size_t)(j - m) * ldx + il] : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) {
                const int j = j0 + jj;
                ph[jj] = (a.Y != nullptr && (j - m) < k && valid) ? a.Y[(size_t)(j - m) * ldx + il] : 0.0;
            }
        }
        if (a.Phi && inb) {
            double *o = a.Phi + (size_t)i * mp + j0;
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) o[jj] = ph[jj];
        }
    }
    if (inb) {
        for (int o = 0; o < k; ++o) {
            const double lb = a.b[o] + svs[o * 64 + lane];                     // getPHI.m:119,124
            a.lnbeta[(size_t)o * ldx + i] = valid ? lb : 0.0;
            if (a.wbeta) {
                const double om = a.omega ? a.omega[(size_t)o * a.om_ld + i] : 1.0;
                a.wbeta[(size_t)o * ldx + i] = valid ? om * exp(-lb) : 0.0;    // GPz.m:43,48
            }
            if (a.phiw) a.phiw[(size_t)o * ldx + i] = valid ? sws[o * 64 + lane] : 0.0;
        }
    }
}

int phi_wide_rows_per_wg() { return 64; }

int launch_phi_wide(hipStream_t st, const PhiArgs &a) {
    const bool psi = a.kind == GPZ_KIND_DIAG && a.Psic, mask = a.kind == GPZ_KIND_DIAG && a.Mc;
    const size_t ntile = 1 + (psi ? 1 : 0) + (mask ? 1 : 0);
    const size_t sums = 2 * (size_t)a.k * 64 * sizeof(double);
    size_t lds = ntile * a.d * 64 * sizeof(double) + sums;
    const bool tiles = lds <= 160 * 1024;
    if (!tiles) lds = sums;
    if (lds > 160 * 1024) return -1;                    // more than 160 outputs
    const int nwg = a.wgtab ? a.nwg_tab : (a.n_pad + 63) / 64;
    if (nwg <= 0) return 0;
#define PW1(KIND, PS, MK, TL)                                                                                         \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void *)k_phi_wide<KIND, PS, MK, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                      \
        hipLaunchKernelGGL((k_phi_wide<KIND, PS, MK, TL>), dim3(nwg), dim3(64), lds, st, a);                          \
    } while (0)
#define PW(KIND, PS, MK)                                                                                              \
    do {                                                                                                              \
        if (tiles) PW1(KIND, PS, MK, true);                                                                           \
        else PW1(KIND, PS, MK, false);                                                                                \
    } while (0)
    if (a.kind == GPZ_KIND_COV) PW(GPZ_KIND_COV, false, false);
    else if (psi && mask) PW(GPZ_KIND_DIAG, true, true);
    else if (psi) PW(GPZ_KIND_DIAG, true, false);
    else if (mask) PW(GPZ_KIND_DIAG, false, true);
    else PW(GPZ_KIND_DIAG, false, false);
#undef PW
#undef PW1
    return 0;
}

// ---------------------------------------------------------------------------------------------
// R_j = triangular factor of Gamma_j (Householder QR), c_j = R_j p_j: one wave per basis function, the matrix in LDS,
// lanes along the columns.  Output layout of k_prep_cov.
// ---------------------------------------------------------------------------------------------
// WS: the matrix does not fit the LDS (de > 142): it lives in `ws` (de*de + de doubles per basis function, device memory; the barriers
// between the phases order the wave's own stores and loads).
template <bool WS>
__global__ __launch_bounds__(64) void k_prep_cov_wide(const double *__restrict__ G, const double *__restrict__ P, int m, int de,
                                                       double *__restrict__ Rc, double *__restrict__ ws) {
    extern __shared__ double lds[];
    double *A = WS ? ws + (size_t)blockIdx.x * ((size_t)de * de + de) : lds;   // [de][de] row-major
    double *v = A + (size_t)de * de;
    const int j = blockIdx.x, lane = threadIdx.x;
    const double *Gj = G + (size_t)j * de * de;
    for (int e = lane; e < de * de; e += 64) A[e] = Gj[e];
    __syncthreads();
    for (int c = 0; c < de; ++c) {
        double n2 = 0.0;
        for (int r = c; r < de; ++r) n2 = fma(A[r * de + c], A[r * de + c], n2);    // every lane: the same broadcast reads
        if (n2 == 0.0) continue;                                                   // wave-uniform
        const double nrm = sqrt(n2), acc = A[c * de + c];
        const double alpha = (acc > 0.0) ? -nrm : nrm;
        const double vc = acc - alpha;
        const double vn2 = n2 - acc * acc + vc * vc;
        __syncthreads();
        for (int r = c + lane; r < de; r += 64) v[r] = (r == c) ? vc : A[r * de + c];
        __syncthreads();
        for (int cc = c + 1 + lane; cc < de; cc += 64) {                            // lane = column: conflict-free row reads
            double dot = 0.0;
            for (int r = c; r < de; ++r) dot = fma(v[r], A[r * de + cc], dot);
            const double f = 2.0 * dot / vn2;
            for (int r = c; r < de; ++r) A[r * de + cc] = fma(-f, v[r], A[r * de + cc]);
        }
        __syncthreads();
        for (int r = c + lane; r < de; r += 64) A[r * de + c] = (r == c) ? alpha : 0.0;
        __syncthreads();
    }
    const int nt = de * (de + 1) / 2;
    double *o = Rc + (size_t)j * (nt + de);
    const double *pj = P + (size_t)j * de;
    for (int aa = lane; aa < de; aa += 64) {
        const int roff = aa * de - aa * (aa - 1) / 2;
        double s = 0.0;
        for (int b = aa; b < de; ++b) {
            const double r = A[aa * de + b];
            o[roff + (b - aa)] = r;
            s = fma(r, pj[b], s);
        }
        o[nt + aa] = s;
    }
}

size_t prep_cov_ws_len(int m, int de) {   // doubles of workspace launch_prep_cov_wide needs (0: the QR runs in LDS)
    const size_t per = (size_t)de * de + de;
    return per * sizeof(double) > 160 * 1024 ? per * (size_t)m : 0;
}
int launch_prep_cov_wide(hipStream_t st, const double *G, const double *P, int m, int de, double *Rc, double *ws) {
    const size_t lds = ((size_t)de * de + de) * sizeof(double);
    if (lds > 160 * 1024) {
        if (!ws) return -1;
        hipLaunchKernelGGL(k_prep_cov_wide<true>, dim3(m), dim3(64), 0, st, G, P, m, de, Rc, ws);
        return 0;
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_prep_cov_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_prep_cov_wide<false>, dim3(m), dim3(64), lds, st, G, P, m, de, Rc, (double *)nullptr);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Moments from a formed dPHI (GPz.m:152-154,192-194,202-206).  Output records as k_moments_cov / k_moments_diag write
// them, record stride `stride` (nm, or nm + 2 on the fused route whose last two entries k_form_dphi fills).
// ---------------------------------------------------------------------------------------------
template <int KIND, bool PSI>
__global__ __launch_bounds__(256) void k_moments_wide(const double *__restrict__ dPhi, int ld, const double *__restrict__ Xr,
                                                       int n, int m, int d, const double *__restrict__ P, int rows_per_chunk,
                                                       double *__restrict__ slab, int stride, const double *__restrict__ Psir,
                                                       const double *__restrict__ Mr, const double *__restrict__ G2,
                                                       const int *__restrict__ chunktab, int ncg) {
    const int j = (int)(blockIdx.x % ncg) * 256 + threadIdx.x;
    const int chunk = blockIdx.x / ncg;
    const bool act = j < m;
    const int jc = act ? j : 0;
    int r0 = chunk * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; r1 = chunktab[2 * chunk + 1]; }
    double *o = slab + ((size_t)chunk * m + jc) * stride;
    if (KIND == GPZ_KIND_DIAG) {
        const int c0 = blockIdx.y * WB;
        double p[WB], g2[WB], A1[WB], A2[WB], A3[WB];
        int cc[WB];
#pragma unroll
        for (int e = 0; e < WB; ++e) {
            cc[e] = min(c0 + e, d - 1);
            p[e] = P[(size_t)jc * d + cc[e]];
            g2[e] = PSI ? G2[(size_t)jc * d + cc[e]] : 0.0;
            A1[e] = 0.0; A2[e] = 0.0; A3[e] = 0.0;
        }
        for (int i = r0; i < r1; ++i) {
            const double dp = act ? dPhi[(size_t)i * ld + j] : 0.0;
#pragma unroll
            for (int e = 0; e < WB; ++e) {
                const double mk = Mr ? Mr[(size_t)i * d + cc[e]] : 1.0;
                const double dl = (Xr[(size_t)i * d + cc[e]] - p[e]) * mk;
                if (PSI) {
                    const double psi = Psir[(size_t)i * d + cc[e]];
                    const double iu = gpz_rcp(fma(psi, g2[e], 1.0));
                    const double dr = dl * iu;
                    A1[e] = fma(dp * dl, g2[e] * iu, A1[e]);                   // GPz.m:202
                    A2[e] = fma(dp * dr, dr, A2[e]);                           // GPz.m:204
                    A3[e] = fma(dp, -psi * iu, A3[e]);                         // GPz.m:206
                } else {
                    const double t = dp * dl;
                    A1[e] += t;                                                // GPz.m:192
                    A2[e] = fma(t, dl, A2[e]);                                 // GPz.m:194
                }
            }
        }
        if (act) {
#pragma unroll
            for (int e = 0; e < WB; ++e)
                if (c0 + e < d) {
                    o[c0 + e] = A1[e];
                    o[d + c0 + e] = A2[e];
                    if (PSI) o[2 * d + c0 + e] = A3[e];
                }
        }
    } else {
        // block z -> (ab, bb), bb >= ab, of the upper triangle in WB x WB blocks
        const int nb = (d + WB - 1) / WB;
        int z = blockIdx.y, ab = 0;
        while (z >= nb - ab) { z -= nb - ab; ++ab; }
        const int bb = ab + z, a0 = ab * WB, b0 = bb * WB;
        double pa[WB], pb[WB], S[WB][WB], M1[WB];
        int ca[WB], cb[WB];
#pragma unroll
        for (int e = 0; e < WB; ++e) {
            ca[e] = min(a0 + e, d - 1); cb[e] = min(b0 + e, d - 1);
            pa[e] = P[(size_t)jc * d + ca[e]]; pb[e] = P[(size_t)jc * d + cb[e]];
            M1[e] = 0.0;
#pragma unroll
            for (int f = 0; f < WB; ++f) S[e][f] = 0.0;
        }
        for (int i = r0; i < r1; ++i) {
            const double dp = act ? dPhi[(size_t)i * ld + j] : 0.0;
            const double *xi = Xr + (size_t)i * d;
            double dlb[WB];
#pragma unroll
            for (int f = 0; f < WB; ++f) dlb[f] = xi[cb[f]] - pb[f];
            if (ab == 0) {
#pragma unroll
                for (int f = 0; f < WB; ++f) M1[f] = fma(dp, dlb[f], M1[f]);   // GPz.m:152
            }
#pragma unroll
            for (int e = 0; e < WB; ++e) {
                const double t = dp * (xi[ca[e]] - pa[e]);
#pragma unroll
                for (int f = 0; f < WB; ++f) S[e][f] = fma(t, dlb[f], S[e][f]);   // GPz.m:154
            }
        }
        if (act) {
            if (ab == 0) {
#pragma unroll
                for (int f = 0; f < WB; ++f)
                    if (b0 + f < d) o[b0 + f] = M1[f];
            }
#pragma unroll
            for (int e = 0; e < WB; ++e) {
                const int aa = a0 + e;
                if (aa >= d) continue;
                const int roff = d + aa * d - aa * (aa - 1) / 2;
#pragma unroll
                for (int f = 0; f < WB; ++f) {
                    const int b = b0 + f;
                    if (b >= aa && b < d) o[roff + (b - aa)] = S[e][f];
                }
            }
        }
    }
}

static void moments_wide(hipStream_t st, const double *dPhi, int ld, const double *Xr, int n, int m, int d, int kind,
                         const double *P, int nchunk, int rows_per_chunk, double *slab, int stride, const double *Psir,
                         const double *Mr, const double *G2, const int *chunktab) {
    const int ncg = (m + 255) / 256, nb = (d + WB - 1) / WB;
    if (kind == GPZ_KIND_COV) {
        dim3 g((unsigned)ncg * (unsigned)nchunk, nb * (nb + 1) / 2);
        hipLaunchKernelGGL((k_moments_wide<GPZ_KIND_COV, false>), g, dim3(256), 0, st, dPhi, ld, Xr, n, m, d, P, rows_per_chunk,
                           slab, stride, Psir, Mr, G2, chunktab, ncg);
    } else {
        dim3 g((unsigned)ncg * (unsigned)nchunk, nb);
        if (Psir)
            hipLaunchKernelGGL((k_moments_wide<GPZ_KIND_DIAG, true>), g, dim3(256), 0, st, dPhi, ld, Xr, n, m, d, P,
                               rows_per_chunk, slab, stride, Psir, Mr, G2, chunktab, ncg);
        else
            hipLaunchKernelGGL((k_moments_wide<GPZ_KIND_DIAG, false>), g, dim3(256), 0, st, dPhi, ld, Xr, n, m, d, P,
                               rows_per_chunk, slab, stride, Psir, Mr, G2, chunktab, ncg);
    }
}

int launch_moments_wide(hipStream_t st, const MomentArgs &a) {
    moments_wide(st, a.dPhi, a.ld, a.Xr, a.n, a.m, a.d, a.kind, a.P, a.nchunk, a.rows_per_chunk, a.slab, a.nm, a.Psir, a.Mr,
                 a.G2, a.chunktab);
    return 0;
}

// T_ij <- dPHI_ij = (-omega beta_i T_ij - c_i w_j + dbeta_i v_j) PHI_ij (GPz.m:72,90,106,113) in place, and the per-chunk
// column sums PHI'c, PHI'dbeta (GPz.m:89,104) into slab[chunk][j][nm], [nm + 1].
__global__ __launch_bounds__(256) void k_form_dphi(const double *__restrict__ Phi, double *__restrict__ T, int ld,
                                                    const double *__restrict__ rowscal, int n, int m,
                                                    const double *__restrict__ w, const double *__restrict__ v,
                                                    int rows_per_chunk, double *__restrict__ slab, int nm,
                                                    const int *__restrict__ chunktab, int ncg) {
    const int j = (int)(blockIdx.x % ncg) * 256 + threadIdx.x;
    const int chunk = blockIdx.x / ncg;
    if (j >= m) return;
    int r0 = chunk * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; r1 = chunktab[2 * chunk + 1]; }
    const double wj = w[j], vj = v ? v[j] : 0.0;
    double s1 = 0.0, s2 = 0.0;
    for (int i = r0; i < r1; ++i) {
        const double *rs = rowscal + (size_t)i * 4;
        const double ph = Phi[(size_t)i * ld + j];
        T[(size_t)i * ld + j] = (-rs[0] * T[(size_t)i * ld + j] - rs[1] * wj + rs[2] * vj) * ph;
        s1 = fma(ph, rs[1], s1);
        s2 = fma(ph, rs[2], s2);
    }
    double *o = slab + ((size_t)chunk * m + j) * (nm + 2);
    o[nm] = s1;
    o[nm + 1] = s2;
}

int launch_moments_fused_wide(hipStream_t st, const FusedMomentArgs &a) {
    const int ncg = (a.m + 255) / 256;
    double *T = const_cast<double *>(a.T);   // the single-output route does not read T again after the moments
    hipLaunchKernelGGL(k_form_dphi, dim3((unsigned)ncg * (unsigned)a.nchunk), dim3(256), 0, st, a.Phi, T, a.ld, a.rowscal, a.n,
                       a.m, a.w, a.v, a.rows_per_chunk, a.slab, a.nm, a.chunktab, ncg);
    moments_wide(st, T, a.ld, a.Xr, a.n, a.m, a.d, a.kind, a.P, a.nchunk, a.rows_per_chunk, a.slab, a.nm + 2, a.Psir, a.Mr,
                 a.G2, a.chunktab);
    return 0;
}
