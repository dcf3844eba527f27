// Row-local gradient epilogue, dP/dGamma moment accumulation, final packing (GPz.m:77-237).
#include "gpz_dev.h"
#include "gpz_kernels.h"

// ---------------------------------------------------------------------------------------------
// Row epilogue: one workgroup per row (grid-stride), lanes along basis functions.
//   nu_i    = sum_j PHI_ij T_ij                                   GPz.m:69
//   delta_i = (PHI w)_i - y_i        ((PHI w)_i = T_i,m+out)      GPz.m:77
//   dbeta_i = 0.5*(-beta)*(1/beta - (delta^2+nu))*omega           GPz.m:93
//   dlnPHI  = -omega beta T - (omega beta delta) w' + dbeta v'    GPz.m:72,90,106
//   dPHI    = dlnPHI .* PHI                                       GPz.m:113
// plus the column sums PHI'(omega beta delta), PHI'dbeta (GPz.m:89,104) kept in registers and the
// scalar sums of GPz.m:81,94,236,237.
// ---------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void k_row_epilogue(RowArgs a) {
    __shared__ double sh4[4];
    __shared__ double sh_phiw;
    const int tid = threadIdx.x;
    double wq[NJ], vq[NJ], r1[NJ], r2[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = tid + 256 * q;
        wq[q] = (j < a.m) ? a.w[j] : 0.0;
        vq[q] = (j < a.m && a.v) ? a.v[j] : 0.0;
        r1[q] = 0.0;
        r2[q] = 0.0;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *yo = a.y + (size_t)a.out * a.ldx;
    const double *lbo = a.lnbeta + (size_t)a.out * a.ldx;
    const double *wbo = a.wbeta + (size_t)a.out * a.ldx;

    for (int i = blockIdx.x; i < a.n; i += gridDim.x) {
        const double *prow = a.Phi + (size_t)i * a.ld;
        double *trow = a.T + (size_t)i * a.ld;
        double ph[NJ], tt[NJ];
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int j = tid + 256 * q;
            ph[q] = (j < a.mp) ? prow[j] : 0.0;
            tt[q] = (j < a.mp) ? trow[j] : 0.0;
            if (j < a.m) part = fma(ph[q], tt[q], part);
            if (j == a.m + a.out) sh_phiw = tt[q];
        }
        const double nu = block_sum_256(part, sh4);        // barriers inside also publish sh_phiw
        const double delta = sh_phiw - yo[i];
        const double lb = lbo[i];
        const double beta = exp(-lb);                                      // GPz.m:43
        const double om1 = a.omega ? a.omega[i] : 1.0;                     // omega(training): the first column  GPz.m:236
        const double om = a.omega ? a.omega[(size_t)a.out * a.om_ld + i] : 1.0;
        const double ob = wbo[i];                                          // omega*beta, GPz.m:48
        const double dbeta = 0.5 * (-beta) * (1.0 / beta - (delta * delta + nu)) * om;   // GPz.m:93
        const double c = ob * delta;                                       // GPz.m:79
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int j = tid + 256 * q;
            if (j < a.mp) {
                double dl = 0.0;
                if (j < a.m) {
                    dl = -ob * tt[q] - c * wq[q] + dbeta * vq[q];
                    r1[q] = fma(ph[q], c, r1[q]);
                    r2[q] = fma(ph[q], dbeta, r2[q]);
                }
                if (a.dL) {
                    double *dp = a.dL + (size_t)i * a.ld + j;
                    *dp = (a.out == 0) ? dl : (*dp + dl);
                } else {
                    trow[j] = dl * ph[q];
                }
            }
        }
        s0 = fma(c, delta, s0);
        s1 = fma(om1, delta * delta, s1);
        s2 += om * (-0.5 * beta * delta * delta + 0.5 * (-lb));            // GPz.m:237  log(beta) = -lnBeta_i
        s3 += dbeta;
        __syncthreads();
    }
    double *cs = a.colslab + (size_t)blockIdx.x * 2 * a.mp;
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = tid + 256 * q;
        if (j < a.mp) {
            cs[j] = r1[q];
            cs[a.mp + j] = r2[q];
        }
    }
    if (tid == 0) {
        double *sc = a.scal + (size_t)blockIdx.x * 4;
        sc[0] = s0; sc[1] = s1; sc[2] = s2; sc[3] = s3;
    }
}

void launch_row_epilogue(hipStream_t st, const RowArgs &a) {
    const int nj = (a.mp + 255) / 256;
    dim3 g(a.nwg), b(256);
    if (nj <= 1) hipLaunchKernelGGL(k_row_epilogue<1>, g, b, 0, st, a);
    else if (nj <= 2) hipLaunchKernelGGL(k_row_epilogue<2>, g, b, 0, st, a);
    else if (nj <= 4) hipLaunchKernelGGL(k_row_epilogue<4>, g, b, 0, st, a);
    else if (nj <= 8) hipLaunchKernelGGL(k_row_epilogue<8>, g, b, 0, st, a);
    else hipLaunchKernelGGL(k_row_epilogue<16>, g, b, 0, st, a);
}

// out[e] = sum_s slab[s*count + e].  16 elements x 16 slab lanes per workgroup: lane q adds slabs q, q+16, ... and the
// 16 partial sums are combined in a fixed order through LDS, so the result is deterministic but the chain of
// dependent loads per thread is nslab/16 instead of nslab (many small slabs used to make this latency-bound).
__global__ __launch_bounds__(256) void k_slab_sum(const double *__restrict__ slab, int nslab, size_t count,
                                                   double *__restrict__ out) {
    __shared__ double part[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (size_t e0 = (size_t)blockIdx.x * 16; e0 < count; e0 += (size_t)gridDim.x * 16) {
        const size_t e = e0 + el;
        double s = 0.0;
        if (e < count) {
            // four independent chains keep four loads in flight per lane (the kernel is latency-bound: one or two WGs per CU)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = sl;
            for (; k + 48 < nslab; k += 64) {
                s += slab[(size_t)k * count + e];
                s1 += slab[(size_t)(k + 16) * count + e];
                s2 += slab[(size_t)(k + 32) * count + e];
                s3 += slab[(size_t)(k + 48) * count + e];
            }
            for (; k < nslab; k += 16) s += slab[(size_t)k * count + e];
            s = (s + s1) + (s2 + s3);
        }
        part[sl][el] = s;
        __syncthreads();
        if (sl == 0 && e < count) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][el];
            out[e] = t;
        }
        __syncthreads();
    }
}

// out[g] = sum of the slabs seg[g] .. seg[g+1]-1 (a segment per NaN pattern; an empty segment yields zeros)
__global__ __launch_bounds__(256) void k_slab_sum_seg(const double *__restrict__ slab, const int *__restrict__ seg,
                                                       size_t count, double *__restrict__ out) {
    __shared__ double part[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int k0 = seg[blockIdx.y], k1 = seg[blockIdx.y + 1];
    double *o = out + (size_t)blockIdx.y * count;
    for (size_t e0 = (size_t)blockIdx.x * 16; e0 < count; e0 += (size_t)gridDim.x * 16) {
        const size_t e = e0 + el;
        double s = 0.0;
        if (e < count)
            for (int k = k0 + sl; k < k1; k += 16) s += slab[(size_t)k * count + e];
        part[sl][el] = s;
        __syncthreads();
        if (sl == 0 && e < count) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][el];
            o[e] = t;
        }
        __syncthreads();
    }
}

void launch_slab_sum_seg(hipStream_t st, const double *slab, const int *seg, int nseg, size_t count, double *out) {
    size_t nb = (count + 15) / 16;
    if (nb > 1024) nb = 1024;
    if (nb == 0 || nseg <= 0) return;
    hipLaunchKernelGGL(k_slab_sum_seg, dim3((unsigned)nb, (unsigned)nseg), dim3(256), 0, st, slab, seg, count, out);
}

void launch_slab_sum(hipStream_t st, const double *slab, int nslab, size_t count, double *out) {
    size_t nb = (count + 15) / 16;
    if (nb > 8192) nb = 8192;
    if (nb == 0) return;
    hipLaunchKernelGGL(k_slab_sum, dim3((unsigned)nb), dim3(256), 0, st, slab, nslab, count, out);
}

void launch_colslab_reduce(hipStream_t st, const double *colslab, const double *scal, int nwg, int mp, double *out_cols,
                           double *out_scal) {
    launch_slab_sum(st, colslab, nwg, (size_t)2 * mp, out_cols);
    launch_slab_sum(st, scal, nwg, 4, out_scal);
}

__global__ void k_mul_phi(const double *__restrict__ dL, const double *__restrict__ Phi, double *__restrict__ T,
                          size_t count) {
    const size_t gs = (size_t)blockDim.x * gridDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gs) T[e] = dL[e] * Phi[e];
}

void launch_mul_phi(hipStream_t st, const double *dL, const double *Phi, double *T, size_t count) {
    hipLaunchKernelGGL(k_mul_phi, dim3(4096), dim3(256), 0, st, dL, Phi, T, count);
}

// ---------------------------------------------------------------------------------------------
// dP / dGamma moments: lanes along basis functions, rows walked sequentially with x_i wave-uniform.
//   cov kinds  (GPz.m:152-154):  M1_j[a] = sum_i dPHI_ij Delta_a ;  S_j[a][b] = sum_i dPHI_ij Delta_a Delta_b (a<=b)
//   diag kinds (GPz.m:192-194):  A1_j[c] = sum_i dPHI_ij Delta_c ;  A2_j[c]   = sum_i dPHI_ij Delta_c^2
// Rows [A0,A1) of S are handled per launch so the accumulators stay in registers for large d.
// ---------------------------------------------------------------------------------------------
template <int D, int A0, int A1>
__global__ __launch_bounds__(256) void k_moments_cov(const double *__restrict__ dPhi, int ld,
                                                      const double *__restrict__ Xr, int n, int m,
                                                      const double *__restrict__ P, int rows_per_chunk,
                                                      double *__restrict__ slab, int nm,
                                                      const int *__restrict__ chunktab) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    const int chunk = blockIdx.x;
    const bool act = j < m;
    double p[D];
#pragma unroll
    for (int c = 0; c < D; ++c) p[c] = act ? P[(size_t)j * D + c] : 0.0;
    constexpr int NS = (A1 - A0) * D - (A1 * (A1 - 1) / 2 - A0 * (A0 - 1) / 2);   // sum_{a=A0}^{A1-1} (D-a)
    double M1[D], S[NS];
#pragma unroll
    for (int c = 0; c < D; ++c) M1[c] = 0.0;
#pragma unroll
    for (int e = 0; e < NS; ++e) S[e] = 0.0;
    int r0 = chunk * rows_per_chunk;
    int r1 = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; r1 = chunktab[2 * chunk + 1]; }   // chunks that respect NaN-pattern boundaries
    for (int i = r0; i < r1; ++i) {
        const double dp = act ? dPhi[(size_t)i * ld + j] : 0.0;
        const double *xi = Xr + (size_t)i * D;
        double dl[D];
#pragma unroll
        for (int c = 0; c < D; ++c) dl[c] = xi[c] - p[c];
        if (A0 == 0) {
#pragma unroll
            for (int c = 0; c < D; ++c) M1[c] = fma(dp, dl[c], M1[c]);
        }
        int e = 0;
#pragma unroll
        for (int aa = A0; aa < A1; ++aa) {
            const double t = dp * dl[aa];
#pragma unroll
            for (int bb = aa; bb < D; ++bb) { S[e] = fma(t, dl[bb], S[e]); ++e; }
        }
    }
    if (act) {
        double *o = slab + ((size_t)chunk * m + j) * nm;
        if (A0 == 0) {
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] = M1[c];
        }
        // S stored upper row-major after M1: offset of row aa is D + sum_{q<aa}(D-q)
        int e = 0;
#pragma unroll
        for (int aa = A0; aa < A1; ++aa) {
            const int roff = D + aa * D - aa * (aa - 1) / 2;
#pragma unroll
            for (int bb = aa; bb < D; ++bb) { o[roff + (bb - aa)] = S[e]; ++e; }
        }
    }
}

// PSI (GPz.m:198-206): A1 = sum dPHI Delta/(psi+sigma), A2 = sum dPHI (Delta r)^2, A3 = sum dPHI (r sigma - sigma),
// r = sigma/(sigma+psi) = 1/u, u = 1 + psi gamma^2;  note r sigma - sigma = -psi/u.
template <int D, bool PSI>
__global__ __launch_bounds__(256) void k_moments_diag(const double *__restrict__ dPhi, int ld,
                                                       const double *__restrict__ Xr, int n, int m,
                                                       const double *__restrict__ P, int rows_per_chunk,
                                                       double *__restrict__ slab, int nm,
                                                       const double *__restrict__ Psir, const double *__restrict__ Mr,
                                                       const double *__restrict__ G2) {
    const int j = blockIdx.y * blockDim.x + threadIdx.x;   // (blockDim.x = 64 .. 256: few basis functions do not leave lanes idle)
    const int chunk = blockIdx.x;
    const bool act = j < m;
    const int jc = act ? j : 0;
    double p[D], g2[PSI ? D : 1], A1v[D], A2v[D], A3v[PSI ? D : 1];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        p[c] = P[(size_t)jc * D + c]; A1v[c] = 0.0; A2v[c] = 0.0;
        if (PSI) { g2[c] = G2[(size_t)jc * D + c]; A3v[c] = 0.0; }
    }
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(n, r0 + rows_per_chunk);
    for (int i = r0; i < r1; ++i) {
        const double dp = act ? dPhi[(size_t)i * ld + j] : 0.0;
        const double *xi = Xr + (size_t)i * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double mk = Mr ? Mr[(size_t)i * D + c] : 1.0;
            const double dl = (xi[c] - p[c]) * mk;
            if (PSI) {
                const double psi = Psir[(size_t)i * D + c];
                const double iu = gpz_rcp1(fma(psi, g2[c], 1.0));
                const double dr = dl * iu;
                A1v[c] = fma(dp * dl, g2[c] * iu, A1v[c]);
                A2v[c] = fma(dp * dr, dr, A2v[c]);
                A3v[c] = fma(dp, -psi * iu, A3v[c]);
            } else {
                const double t = dp * dl;
                A1v[c] += t;
                A2v[c] = fma(t, dl, A2v[c]);
            }
        }
    }
    if (act) {
        double *o = slab + ((size_t)chunk * m + j) * nm;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            o[c] = A1v[c]; o[D + c] = A2v[c];
            if (PSI) o[2 * D + c] = A3v[c];
        }
    }
}

#define MOM_COV(D, A0, A1) \
    hipLaunchKernelGGL((k_moments_cov<D, A0, A1>), g, b, 0, st, a.dPhi, a.ld, a.Xr, a.n, a.m, a.P, a.rows_per_chunk, a.slab, a.nm, a.chunktab)
#define MOM_DIAG(D) \
    do { \
        if (a.Psir) hipLaunchKernelGGL((k_moments_diag<D, true>), gd, bd, 0, st, a.dPhi, a.ld, a.Xr, a.n, a.m, a.P, \
                                       a.rows_per_chunk, a.slab, a.nm, a.Psir, a.Mr, a.G2); \
        else hipLaunchKernelGGL((k_moments_diag<D, false>), gd, bd, 0, st, a.dPhi, a.ld, a.Xr, a.n, a.m, a.P, \
                                a.rows_per_chunk, a.slab, a.nm, a.Psir, a.Mr, a.G2); \
    } while (0)

int launch_moments(hipStream_t st, const MomentArgs &a) {
    dim3 g(a.nchunk, (a.m + 255) / 256), b(256);
    // diagonal kinds: lanes run along basis functions - with m = 64 or 128 a 256-thread workgroup left 75 / 50 % of them idle
    const int bs = a.m >= 256 ? 256 : (a.m + 63) / 64 * 64;
    dim3 gd(a.nchunk, (a.m + bs - 1) / bs), bd(bs);
    if (a.kind == GPZ_KIND_COV) {
        switch (a.d) {
            case 1: MOM_COV(1, 0, 1); break;
            case 2: MOM_COV(2, 0, 2); break;
            case 3: MOM_COV(3, 0, 3); break;
            case 4: MOM_COV(4, 0, 4); break;
            case 5: MOM_COV(5, 0, 5); break;
            case 6: MOM_COV(6, 0, 6); break;
            case 8: MOM_COV(8, 0, 8); break;
            case 10: MOM_COV(10, 0, 10); break;
            case 12: MOM_COV(12, 0, 5); MOM_COV(12, 5, 12); break;
            case 16: MOM_COV(16, 0, 4); MOM_COV(16, 4, 9); MOM_COV(16, 9, 16); break;
            case 20: MOM_COV(20, 0, 3); MOM_COV(20, 3, 7); MOM_COV(20, 7, 12); MOM_COV(20, 12, 20); break;
            default: return launch_moments_wide(st, a);
        }
    } else {
        switch (a.d) {
            case 1: MOM_DIAG(1); break;
            case 2: MOM_DIAG(2); break;
            case 3: MOM_DIAG(3); break;
            case 4: MOM_DIAG(4); break;
            case 5: MOM_DIAG(5); break;
            case 6: MOM_DIAG(6); break;
            case 8: MOM_DIAG(8); break;
            case 10: MOM_DIAG(10); break;
            case 12: MOM_DIAG(12); break;
            case 16: MOM_DIAG(16); break;
            case 20: MOM_DIAG(20); break;
            default: return launch_moments_wide(st, a);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Fused single-output path (k == 1): no dPHI matrix is ever written.
// ---------------------------------------------------------------------------------------------
// Row scalars from the T-GEMM epilogue's partial sums (GPz.m:69,77-79,93) + the scalar sums of GPz.m:81,94,236-237.
__global__ __launch_bounds__(256) void k_row_scalars(const double *__restrict__ nupart, int nslots,
                                                      const double *__restrict__ phiw, const double *__restrict__ y,
                                                      const double *__restrict__ omega, long om_off,
                                                      const double *__restrict__ lnbeta,
                                                      const double *__restrict__ wbeta, long n_pad, int n,
                                                      double *__restrict__ rowscal, double *__restrict__ partial) {
    __shared__ double sh4[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // the slots are added in their order, their loads issued eight at a time (one load per dependent add left the kernel
        // waiting nslots memory latencies per row: 31 us for the 125 000 rows of a c4 shard)
        double nu = 0.0;
        int q = 0;
        for (; q + 8 <= nslots; q += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = nupart[(size_t)(q + u) * n_pad + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) nu += v[u];
        }
        for (; q < nslots; ++q) nu += nupart[(size_t)q * n_pad + i];
        const double delta = phiw[i] - y[i];
        const double lb = lnbeta[i];
        const double beta = exp(-lb);
        const double om = omega ? omega[om_off + i] : 1.0;                  // this output's column of an n x k omega
        const double om1 = omega ? omega[i] : 1.0;                          // omega(training): the first column  GPz.m:236
        const double ob = wbeta[i];
        const double dbeta = 0.5 * (-beta) * (1.0 / beta - (delta * delta + nu)) * om;   // GPz.m:93
        const double c = ob * delta;
        double *rs = rowscal + (size_t)i * 4;
        rs[0] = ob; rs[1] = c; rs[2] = dbeta; rs[3] = 0.0;
        s0 = fma(c, delta, s0);
        s1 = fma(om1, delta * delta, s1);
        s2 += om * (-0.5 * beta * delta * delta + 0.5 * (-lb));
        s3 += dbeta;
    }
    s0 = block_sum_256(s0, sh4); __syncthreads();
    s1 = block_sum_256(s1, sh4); __syncthreads();
    s2 = block_sum_256(s2, sh4); __syncthreads();
    s3 = block_sum_256(s3, sh4);
    if (threadIdx.x == 0) {
        double *pw = partial + (size_t)blockIdx.x * GPZ_NS;
        pw[0] = s0; pw[1] = s1; pw[2] = s2; pw[3] = s3;
        for (int q = 4; q < GPZ_NS; ++q) pw[q] = 0.0;
    }
}

void launch_row_scalars(hipStream_t st, const double *nupart, int nslots, const double *phiw, const double *y,
                        const double *omega, const double *lnbeta, const double *wbeta, long n_pad, int n,
                        double *rowscal, double *partial, long om_off) {
    hipLaunchKernelGGL(k_row_scalars, dim3(row_scalars_nwg(n)), dim3(256), 0, st, nupart, nslots, phiw, y, omega, om_off, lnbeta, wbeta,
                       n_pad, n, rowscal, partial);
}

// Moments with dPHI_ij = (-omega beta_i T_ij - c_i w_j + dbeta_i v_j) * PHI_ij formed on the fly (GPz.m:72,90,106,113),
// plus the column sums PHI'c and PHI'dbeta (GPz.m:89,104).  Lanes along basis functions; row data (x_i, row scalars)
// are wave-uniform.  UR rows are in flight per thread to cover the HBM latency.
template <int KIND, int D, int A0, int A1, int UR, bool PSI>
__global__ __launch_bounds__(256) void k_moments_fused(const double *__restrict__ Phi, const double *__restrict__ T,
                                                        int ld, const double *__restrict__ Xr,
                                                        const double *__restrict__ rowscal, int n, int m,
                                                        const double *__restrict__ P, const double *__restrict__ w,
                                                        const double *__restrict__ v, int rows_per_chunk,
                                                        double *__restrict__ slab, int nm,
                                                        const double *__restrict__ Psir, const double *__restrict__ Mr,
                                                        const double *__restrict__ G2, const int *__restrict__ chunktab) {
    // Column group fastest in a 1-D grid: the workgroups that read the same rows of PHI / T are dispatched together
    // and visit those DRAM pages at about the same time (c4: 4.66 -> 4.55 ms against chunk-fastest order)
    const int ncg = (m + 255) >> 8;
    const int j = (int)(blockIdx.x % ncg) * 256 + threadIdx.x;
    const int chunk = blockIdx.x / ncg;
    const bool act = j < m;
    const int jc = act ? j : 0;
    double p[D];
#pragma unroll
    for (int c = 0; c < D; ++c) p[c] = P[(size_t)jc * D + c];
    const double wj = w[jc], vj = v ? v[jc] : 0.0;
    constexpr int NS = (KIND == GPZ_KIND_COV) ? ((A1 - A0) * D - (A1 * (A1 - 1) / 2 - A0 * (A0 - 1) / 2)) : D;
    double M1[D], S[NS], S3[PSI ? D : 1], g2[PSI ? D : 1];
    double r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        M1[c] = 0.0;
        if (PSI) { S3[c] = 0.0; g2[c] = G2[(size_t)jc * D + c]; }
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) S[e] = 0.0;
    int r0 = chunk * rows_per_chunk;
    int rend = min(n, r0 + rows_per_chunk);
    if (chunktab) { r0 = chunktab[2 * chunk]; rend = chunktab[2 * chunk + 1]; }   // chunks that respect NaN-pattern boundaries
    // One row of work.  sc = the wave-uniform row data [omega*beta, c, dbeta | x_i (D)] (scalar registers).
    auto consume = [&](int i, const double phu, const double ttu, const double (&sc)[3 + D]) {
        const double ob = sc[0], cc = sc[1], db = sc[2];
        const double dp = (-ob * ttu - cc * wj + db * vj) * phu;
        if (A0 == 0) {
            r1 = fma(phu, cc, r1);
            r2 = fma(phu, db, r2);
        }
        if (KIND == GPZ_KIND_COV) {
            double dl[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dl[c] = sc[3 + c] - p[c];
            if (A0 == 0) {
#pragma unroll
                for (int c = 0; c < D; ++c) M1[c] = fma(dp, dl[c], M1[c]);
            }
            int e = 0;
#pragma unroll
            for (int aa = A0; aa < A1; ++aa) {
                const double t = dp * dl[aa];
#pragma unroll
                for (int bb = aa; bb < D; ++bb) { S[e] = fma(t, dl[bb], S[e]); ++e; }
            }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double mk = Mr ? Mr[(size_t)i * D + c] : 1.0;
                const double dl = (sc[3 + c] - p[c]) * mk;
                if (PSI) {
                    const double psi = Psir[(size_t)i * D + c];
                    const double iu = gpz_rcp1(fma(psi, g2[c], 1.0));
                    const double q = dp * iu;                       // dPHI / u: three multiplies and three multiply-adds per (i, j, c)
                    const double t = dl * q;                        // (five and three with dPHI Delta, gamma^2 / u and psi / u formed apart)
                    M1[c] = fma(g2[c], t, M1[c]);                   // sum dPHI Delta gamma^2 / u
                    S[c] = fma(t, dl * iu, S[c]);                   // sum dPHI (Delta / u)^2
                    S3[c] = fma(-psi, q, S3[c]);                    // sum dPHI (-psi / u)
                } else {
                    const double t = dp * dl;
                    M1[c] += t;
                    S[c] = fma(t, dl, S[c]);
                }
            }
        }
    };
    auto sload = [&](int i, double (&sc)[3 + D]) {     // scalar loads (uniform address)
        const double *rs = rowscal + (size_t)i * 4;
        sc[0] = rs[0]; sc[1] = rs[1]; sc[2] = rs[2];
        const double *xi = Xr + (size_t)i * D;
#pragma unroll
        for (int c = 0; c < D; ++c) sc[3 + c] = xi[c];
    };
    // Two register sets of UR rows each, the loop unrolled over both so that no set is ever copied: while set A is
    // consumed the loads of set B are in flight and vice versa (a copy would make the compiler wait for the loads
    // it copies).  The uniform data of row i+1 is fetched while row i is worked on.  No branches in the main loop.
    double phA[UR], ttA[UR], phB[UR], ttB[UR];
    const int last = max(rend - 1, 0);
    auto vload = [&](int i0, double (&a)[UR], double (&b)[UR]) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int i = min(i0 + u, last);
            a[u] = Phi[(size_t)i * ld + jc];
            b[u] = T[(size_t)i * ld + jc];
        }
    };
    // the uniform row data alternates between two scalar register sets (no copies): row i+1 is fetched into the other
    // set while row i is worked on
    double scA[3 + D], scB[3 + D];
    static_assert(UR % 2 == 0, "UR must be even");
    auto batch = [&](int i0, const double (&a)[UR], const double (&b)[UR]) {
#pragma unroll
        for (int u = 0; u < UR; u += 2) {
            sload(min(i0 + u + 1, last), scB);
            consume(i0 + u, a[u], b[u], scA);
            sload(min(i0 + u + 2, last), scA);
            consume(i0 + u + 1, a[u + 1], b[u + 1], scB);
        }
    };
    vload(r0, phA, ttA);
    sload(min(r0, last), scA);
    int ib = r0;
    for (; ib + 2 * UR <= rend; ib += 2 * UR) {
        vload(ib + UR, phB, ttB);
        batch(ib, phA, ttA);
        vload(ib + 2 * UR, phA, ttA);
        batch(ib + UR, phB, ttB);
    }
    // remainder (< 2 UR rows): set A holds rows ib .. ib+UR-1
#pragma unroll 1
    for (int i = ib; i < rend; ++i) {
        sload(i, scA);
        consume(i, Phi[(size_t)i * ld + jc], T[(size_t)i * ld + jc], scA);
    }
    if (act) {
        double *o = slab + ((size_t)chunk * m + j) * (nm + 2);
        if (A0 == 0) {
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] = M1[c];
            o[nm] = r1;
            o[nm + 1] = r2;
        }
        if (KIND == GPZ_KIND_COV) {
            int e = 0;
#pragma unroll
            for (int aa = A0; aa < A1; ++aa) {
                const int roff = D + aa * D - aa * (aa - 1) / 2;
#pragma unroll
                for (int bb = aa; bb < D; ++bb) { o[roff + (bb - aa)] = S[e]; ++e; }
            }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                o[D + c] = S[c];
                if (PSI) o[2 * D + c] = S3[c];
            }
        }
    }
}

#ifndef GPZ_MOM_UR
#define GPZ_MOM_UR 4    // rows per register set (two sets; 6 or 8 drop the kernel to one wave per SIMD: 8.1 ms vs 4.8 ms at c4)
#endif
#ifndef GPZ_MOM_UR_DIAG
#define GPZ_MOM_UR_DIAG 2   // diagonal kinds do ~4d flops per row: short row chunks, so a short pipeline (c2: 0.27 -> 0.20 ms)
#endif
#define MOMF_(KIND, D, A0, A1, PS) \
    hipLaunchKernelGGL((k_moments_fused<KIND, D, A0, A1, (KIND == GPZ_KIND_COV ? GPZ_MOM_UR : GPZ_MOM_UR_DIAG), PS>), g, b, 0, st, a.Phi, a.T, a.ld, a.Xr, a.rowscal, a.n, \
                       a.m, a.P, a.w, a.v, a.rows_per_chunk, a.slab, a.nm, a.Psir, a.Mr, a.G2, a.chunktab)
#define MOMF(KIND, D, A0, A1) \
    do { \
        if (KIND == GPZ_KIND_DIAG && a.Psir) MOMF_(KIND, D, A0, A1, true); \
        else MOMF_(KIND, D, A0, A1, false); \
    } while (0)

int launch_moments_fused(hipStream_t st, const FusedMomentArgs &a) {
    dim3 g((unsigned)((a.m + 255) / 256) * (unsigned)a.nchunk), b(256);
    if (a.kind == GPZ_KIND_COV) {
        switch (a.d) {
            case 1: MOMF(GPZ_KIND_COV, 1, 0, 1); break;
            case 2: MOMF(GPZ_KIND_COV, 2, 0, 2); break;
            case 3: MOMF(GPZ_KIND_COV, 3, 0, 3); break;
            case 4: MOMF(GPZ_KIND_COV, 4, 0, 4); break;
            case 5: MOMF(GPZ_KIND_COV, 5, 0, 5); break;
            case 6: MOMF(GPZ_KIND_COV, 6, 0, 6); break;
            case 8: MOMF(GPZ_KIND_COV, 8, 0, 8); break;
            case 10: MOMF(GPZ_KIND_COV, 10, 0, 10); break;
            case 12: MOMF(GPZ_KIND_COV, 12, 0, 5); MOMF(GPZ_KIND_COV, 12, 5, 12); break;
            case 16: MOMF(GPZ_KIND_COV, 16, 0, 4); MOMF(GPZ_KIND_COV, 16, 4, 9); MOMF(GPZ_KIND_COV, 16, 9, 16); break;
            case 20: MOMF(GPZ_KIND_COV, 20, 0, 3); MOMF(GPZ_KIND_COV, 20, 3, 7); MOMF(GPZ_KIND_COV, 20, 7, 12);
                     MOMF(GPZ_KIND_COV, 20, 12, 20); break;
            default: return launch_moments_fused_wide(st, a);
        }
    } else {
        switch (a.d) {
            case 1: MOMF(GPZ_KIND_DIAG, 1, 0, 1); break;
            case 2: MOMF(GPZ_KIND_DIAG, 2, 0, 2); break;
            case 3: MOMF(GPZ_KIND_DIAG, 3, 0, 3); break;
            case 4: MOMF(GPZ_KIND_DIAG, 4, 0, 4); break;
            case 5: MOMF(GPZ_KIND_DIAG, 5, 0, 5); break;
            case 6: MOMF(GPZ_KIND_DIAG, 6, 0, 6); break;
            case 8: MOMF(GPZ_KIND_DIAG, 8, 0, 8); break;
            case 10: MOMF(GPZ_KIND_DIAG, 10, 0, 10); break;
            case 12: MOMF(GPZ_KIND_DIAG, 12, 0, 12); break;
            case 16: MOMF(GPZ_KIND_DIAG, 16, 0, 16); break;
            case 20: MOMF(GPZ_KIND_DIAG, 20, 0, 20); break;
            default: return launch_moments_fused_wide(st, a);
        }
    }
    return 0;
}

// accumulate: the moments of this output are added to those of the outputs before it (dPHI is a sum over outputs, GPz.m:113)
__global__ void k_split_fused(const double *__restrict__ rec, int m, int nm, int mp, double *__restrict__ mom,
                              double *__restrict__ cols, int accumulate) {
    const int gs = blockDim.x * gridDim.x;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m * (nm + 2); e += gs) {
        const int j = e / (nm + 2), q = e % (nm + 2);
        const double v = rec[e];
        if (q < nm) mom[(size_t)j * nm + q] = accumulate ? mom[(size_t)j * nm + q] + v : v;
        else cols[(size_t)(q - nm) * mp + j] = v;
    }
}

void launch_split_fused(hipStream_t st, const double *rec, int m, int nm, int mp, double *mom, double *cols, int accumulate) {
    hipLaunchKernelGGL(k_split_fused, dim3(256), dim3(256), 0, st, rec, m, nm, mp, mom, cols, accumulate);
}

// ---------------------------------------------------------------------------------------------
// Finish: chain the moments to dP/dGamma (GPz.m:146-159,189-194), method reduction (:215-225),
// the m-sized gradient blocks (:73,89,94,104,105), the objective (:81-82,103,110,233) and the
// statistics (:236-237,258-259).  p_e = padded dimension of the parameter block.
// ---------------------------------------------------------------------------------------------
__global__ void k_finish_a(FinishArgs a, int de) {
    const int gs = blockDim.x * gridDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = a.m, d = a.d, k = a.k;
    const double nk = a.sums1[10] * (double)k;
    double *grad = a.out + 1;
    const int md = m * d;
    if (a.kind == GPZ_KIND_COV && a.gen) {
        // dP / dGamma were chained by k_gen_finish (general path)
    } else if (a.kind == GPZ_KIND_COV) {
        // dP(j,:) = M1_j * (Gamma_j' Gamma_j)      (GPz.m:146,151-152)
        for (int e = t0; e < m * d; e += gs) {
            const int j = e / d, c = e % d;
            const double *Gj = a.pr.G + (size_t)j * de * de;
            const double *M1 = a.mom + (size_t)j * a.nm;
            double s = 0.0;
            for (int aa = 0; aa < d; ++aa) {
                double is = 0.0;
                for (int q = 0; q < d; ++q) is = fma(Gj[q * de + aa], Gj[q * de + c], is);
                s = fma(M1[aa], is, s);
            }
            grad[j + m * c] = -s / nk;
        }
        // dGamma_j = 2*Gamma_j*(-0.5*S_j) = -Gamma_j S_j     (GPz.m:154,157-158)
        for (int e = t0; e < m * d * d; e += gs) {
            const int j = e / (d * d), aa = (e / d) % d, bb = e % d;
            const double *Gj = a.pr.G + (size_t)j * de * de;
            const double *Sj = a.mom + (size_t)j * a.nm + de;
            double s = 0.0;
            for (int q = 0; q < d; ++q) {
                const int lo = q < bb ? q : bb, hi = q < bb ? bb : q;
                const double sv = Sj[lo * de - lo * (lo - 1) / 2 + (hi - lo)];
                s = fma(Gj[aa * de + q], sv, s);
            }
            const double val = -s;
            if (a.method_id == 5) grad[md + aa + d * bb + d * d * j] = -val / nk;   // VC
            else a.dGfull[e] = val;                                                   // GC: summed in k_finish_b
        }
    } else {
        for (int e = t0; e < m * d; e += gs) {
            const int j = e / d, c = e % d;
            const double *mo = a.mom + (size_t)j * a.nm;
            const double g = a.pr.G[(size_t)j * de + c];
            double dpv, dg;
            if (a.psi) {
                dpv = mo[c];                                               // GPz.m:202  (1/(psi+sigma) applied per pair)
                dg = -g * (mo[de + c] - mo[2 * de + c]);                   // GPz.m:206
            } else {
                dpv = mo[c] * (g * g);                                     // GPz.m:192  ./Sigma, Sigma = gamma^-2
                dg = -g * mo[de + c];                                      // GPz.m:194
            }
            grad[j + m * c] = -dpv / nk;
            if (a.method_id == 3) grad[md + j + m * c] = -dg / nk;         // VD
            else a.dGfull[e] = dg;
        }
    }
    const int off = md + a.g_dim;
    for (int e = t0; e < m * k; e += gs) {
        const int j = e % m, o = e / m;
        const double al = a.pr.alpha[e], w = a.w[e], dw = a.dwda[e];
        const double r1 = a.cols[(size_t)o * 2 * a.nmp + j];
        const double r2 = a.cols[(size_t)o * 2 * a.nmp + a.nmp + j];
        // GPz.m:73,89
        const double dla = -0.5 * a.dgi[e] * al - r1 * dw - al * w * dw - 0.5 * al * w * w + 0.5;
        grad[off + e] = -dla / nk;
        if (a.hetero) {
            const double v = a.pr.v[e], tau = a.pr.tau[e];
            grad[off + m * k + k + e] = -(r2 - v * tau) / nk;              // GPz.m:104
            grad[off + m * k + k + m * k + e] = -(-0.5 * tau * v * v + 0.5) / nk;   // GPz.m:105
        }
    }
    for (int o = t0; o < k; o += gs) grad[off + m * k + o] = -a.scal[o * 4 + 3] / nk;   // db, GPz.m:94
}

__global__ __launch_bounds__(256) void k_finish_b(FinishArgs a, int de) {
    __shared__ double sh4[4];
    const int tid = threadIdx.x;
    const int m = a.m, d = a.d, k = a.k;
    const double n = a.sums1[10];
    const double nk = n * (double)k;
    double *grad = a.out + 1;
    const int md = m * d;
    // method reductions of dGamma (GPz.m:215-225)
    if (a.method_id == 0) {
        double s = 0.0;
        for (int e = tid; e < m * d; e += 256) s += a.dGfull[e];
        s = block_sum_256(s, sh4);
        if (tid == 0) grad[md] = -s / nk;
    } else if (a.method_id == 1) {
        for (int j = tid; j < m; j += 256) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) s += a.dGfull[j * d + c];
            grad[md + j] = -s / nk;
        }
    } else if (a.method_id == 2) {
        for (int c = tid; c < d; c += 256) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += a.dGfull[j * d + c];
            grad[md + c] = -s / nk;
        }
    } else if (a.method_id == 4) {
        for (int e = tid; e < d * d; e += 256) {
            const int aa = e / d, bb = e % d;
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += a.dGfull[(size_t)j * d * d + e];
            grad[md + aa + d * bb] = -s / nk;
        }
    }
    // objective (GPz.m:81-82,103,110,233)
    double L = 0.0;
    for (int o = 0; o < k; ++o) {
        double part = 0.0;
        for (int j = tid; j < m; j += 256) {
            const int e = j + m * o;
            const double w = a.w[e];
            part += -0.5 * a.pr.alpha[e] * w * w + 0.5 * a.pr.lnAlpha[e];
            if (a.hetero) {
                const double v = a.pr.v[e];
                part += -0.5 * v * v * a.pr.tau[e] + 0.5 * a.pr.lnTau[e];
            }
        }
        part = block_sum_256(part, sh4);
        double Lo = part - 0.5 * a.scal[o * 4 + 0] - 0.5 * a.logdet[o] + 0.5 * (-a.sums1[gpz_ns_idx(1, o)]);
        if (a.hetero) Lo -= 0.5 * (double)m * (double)k * GPZ_LOG2PI;
        L += Lo;
        __syncthreads();
    }
    if (tid == 0) {
        L -= 0.5 * GPZ_LOG2PI * a.sums1[0];
        const bool bad = (*a.info != 0);
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        double s1 = 0.0, s2 = 0.0;
        for (int o = 0; o < k; ++o) { s1 += a.scal[o * 4 + 1]; s2 += a.scal[o * 4 + 2]; }
        a.out[0] = bad ? nanv : -L / nk;
        double *st = a.out + 1 + a.p;
        st[0] = sqrt(s1 / nk);                                             // GPz.m:236
        st[1] = s2 / nk - 0.5 * GPZ_LOG2PI;                                // GPz.m:237
        if (a.vsums) {
            const double nvk = a.vsums[10] * (double)k;
            st[2] = sqrt(a.vsums[0] / nvk);                                // GPz.m:258
            st[3] = a.vsums[1] / nvk - 0.5 * GPZ_LOG2PI;                   // GPz.m:259
        }
        st[4] = (double)(*a.info);
        st[5] = n;
        st[6] = a.logdet[0];
        st[7] = (double)a.info[1];                                         // k_cond_flag: the SVD route is wanted
    }
    if (*a.info != 0) {
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        __syncthreads();
        for (int e = tid; e < a.p; e += 256) grad[e] = nanv;
    }
}

void launch_finish(hipStream_t st, const FinishArgs &a) {
    hipLaunchKernelGGL(k_finish_a, dim3(256), dim3(256), 0, st, a, a.de);
    hipLaunchKernelGGL(k_finish_b, dim3(1), dim3(256), 0, st, a, a.de);
}

// ---------------------------------------------------------------------------------------------
// small row reductions
// ---------------------------------------------------------------------------------------------
// partial[wg][0] = sum omega*delta^2 (all outputs), [1] = sum omega*(-0.5 beta delta^2 + 0.5 ln beta),
// [2+o] = sum omega*beta*delta^2 for output o, with delta = phiw - y.   (GPz.m:81,236-237,258-259)
__global__ __launch_bounds__(256) void k_row_stats(const double *__restrict__ phiw, const double *__restrict__ y,
                                                    const double *__restrict__ omega, long om_ld,
                                                    const double *__restrict__ lnbeta, long ldx, int n, int k,
                                                    double *__restrict__ partial) {
    __shared__ double sh4[4];
    double s0 = 0.0, s1 = 0.0, cnt = 0.0;
    double so[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double om1 = omega ? omega[i] : 1.0;   // omega(training) of GPz.m:236,258: the first column of an n x k omega
        cnt += 1.0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (o < k) {
                const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
                const double delta = phiw[(size_t)o * ldx + i] - y[(size_t)o * ldx + i];
                const double lb = lnbeta[(size_t)o * ldx + i];
                const double beta = exp(-lb);
                s0 = fma(om1, delta * delta, s0);
                s1 += om * (-0.5 * beta * delta * delta + 0.5 * (-lb));
                so[o] = fma(om * beta, delta * delta, so[o]);
            }
        }
    }
    s0 = block_sum_256(s0, sh4);
    __syncthreads();
    s1 = block_sum_256(s1, sh4);
    __syncthreads();
    cnt = block_sum_256(cnt, sh4);
    const int ns = gpz_ns(k);
    double *pw = partial + (size_t)blockIdx.x * ns;
    if (threadIdx.x == 0) { pw[0] = s0; pw[1] = s1; pw[10] = cnt; pw[11] = 0.0; }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        __syncthreads();
        const double t = block_sum_256(so[o], sh4);
        if (threadIdx.x == 0) pw[2 + o] = t;
    }
    for (int o = 8; o < k; ++o) {   // more than 8 outputs: one more pass over the rows per extra output (records grow by k - 8)
        double t = 0.0, t0 = 0.0, t1 = 0.0;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const double om1 = omega ? omega[i] : 1.0, om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
            const double delta = phiw[(size_t)o * ldx + i] - y[(size_t)o * ldx + i];
            const double lb = lnbeta[(size_t)o * ldx + i];
            const double beta = exp(-lb);
            t0 = fma(om1, delta * delta, t0);
            t1 += om * (-0.5 * beta * delta * delta + 0.5 * (-lb));
            t = fma(om * beta, delta * delta, t);
        }
        __syncthreads();
        t = block_sum_256(t, sh4);
        __syncthreads();
        t0 = block_sum_256(t0, sh4);
        __syncthreads();
        t1 = block_sum_256(t1, sh4);
        if (threadIdx.x == 0) { pw[gpz_ns_idx(2, o)] = t; pw[0] += t0; pw[1] += t1; }
    }
}

// partial[wg][0] = sum omega, [1+o] = sum_i omega_i lnbeta_io      (GPz.m:82,110)
__global__ __launch_bounds__(256) void k_sums1(const double *__restrict__ omega, long om_ld, const double *__restrict__ lnbeta,
                                                long ldx, int n, int k, double *__restrict__ partial) {
    __shared__ double sh4[4];
    double s0 = 0.0, cnt = 0.0;
    double so[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // sum(sum(omega(training,:))) of GPz.m:110: an n x 1 omega counts once, an n x k omega every column
        if (!om_ld) s0 += omega ? omega[i] : 1.0;
        cnt += 1.0;
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < k) {
                const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
                if (om_ld) s0 += om;
                so[o] = fma(om, lnbeta[(size_t)o * ldx + i], so[o]);
            }
    }
    s0 = block_sum_256(s0, sh4);
    __syncthreads();
    cnt = block_sum_256(cnt, sh4);
    const int ns = gpz_ns(k);
    double *pw = partial + (size_t)blockIdx.x * ns;
    if (threadIdx.x == 0) { pw[0] = s0; pw[9] = 0.0; pw[10] = cnt; pw[11] = 0.0; }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        __syncthreads();
        const double t = block_sum_256(so[o], sh4);
        if (threadIdx.x == 0) pw[1 + o] = t;
    }
    for (int o = 8; o < k; ++o) {
        double t = 0.0, ts = 0.0;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
            if (om_ld) ts += om;
            t = fma(om, lnbeta[(size_t)o * ldx + i], t);
        }
        __syncthreads();
        t = block_sum_256(t, sh4);
        __syncthreads();
        ts = block_sum_256(ts, sh4);
        if (threadIdx.x == 0) { pw[gpz_ns_idx(1, o)] = t; pw[0] += ts; }
    }
}

// nlogML(o) of GPz.m:81-82 (solve-only mode returns it un-normalised)
__global__ __launch_bounds__(256) void k_solve_partial(GpzParams pr, const double *__restrict__ w,
                                                        const double *__restrict__ logdet,
                                                        const double *__restrict__ sums1,
                                                        const double *__restrict__ rstats, int m, int k,
                                                        double *__restrict__ out) {
    __shared__ double sh4[4];
    for (int o = 0; o < k; ++o) {
        double part = 0.0;
        for (int j = threadIdx.x; j < m; j += 256) {
            const int e = j + m * o;
            part += -0.5 * pr.alpha[e] * w[e] * w[e] + 0.5 * pr.lnAlpha[e];
        }
        part = block_sum_256(part, sh4);
        if (threadIdx.x == 0) out[o] = part - 0.5 * rstats[gpz_ns_idx(2, o)] - 0.5 * logdet[o] + 0.5 * (-sums1[gpz_ns_idx(1, o)]);
        __syncthreads();
    }
}

void launch_solve_partial(hipStream_t st, GpzParams pr, const double *w, const double *logdet, const double *sums1,
                          const double *rstats, int m, int k, double *out) {
    hipLaunchKernelGGL(k_solve_partial, dim3(1), dim3(256), 0, st, pr, w, logdet, sums1, rstats, m, k, out);
}

#define SMALL_NWG GPZ_SMALL_NWG
void launch_row_stats(hipStream_t st, const double *phiw, const double *y, const double *omega, long om_ld, const double *lnbeta,
                      long ldx, int n, int k, double *partial) {
    hipLaunchKernelGGL(k_row_stats, dim3(SMALL_NWG), dim3(256), 0, st, phiw, y, omega, om_ld, lnbeta, ldx, n, k, partial);
}
void launch_sums1(hipStream_t st, const double *omega, long om_ld, const double *lnbeta, long ldx, int n, int k, double *partial) {
    hipLaunchKernelGGL(k_sums1, dim3(SMALL_NWG), dim3(256), 0, st, omega, om_ld, lnbeta, ldx, n, k, partial);
}

// ---------------------------------------------------------------------------------------------
// misc entry points
// ---------------------------------------------------------------------------------------------
// D = | |x|^2 + |y|^2 - 2 x.y |   (Dxy.m:3-7); X nx x d, Y ny x d, D nx x ny, all column-major.
__global__ void k_dxy(const double *__restrict__ X, long nx, const double *__restrict__ Y, long ny, int d,
                      double *__restrict__ D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j = blockIdx.y;
    if (i >= nx) return;
    double xx = 0.0, yy = 0.0, xy = 0.0;
    for (int c = 0; c < d; ++c) {
        const double xv = X[c * nx + i], yv = Y[c * ny + j];
        xx = fma(xv, xv, xx);
        yy = fma(yv, yv, yy);
        xy = fma(xv, yv, xy);
    }
    D[j * nx + i] = fabs(fabs(yy + (xx - 2.0 * xy)));
}

void launch_dxy(hipStream_t st, const double *X, long nx, const double *Y, long ny, int d, double *D) {
    hipLaunchKernelGGL(k_dxy, dim3((unsigned)((nx + 255) / 256), (unsigned)ny), dim3(256), 0, st, X, nx, Y, ny, d, D);
}

// dst (n x m column-major) = src (row-major, leading dimension ld), tiled through LDS.
// perm (optional): source row i goes to destination row perm[i].
__global__ __launch_bounds__(256) void k_transpose_out(const double *__restrict__ src, int ld, long n, int m,
                                                        double *__restrict__ dst, const int *__restrict__ perm) {
    __shared__ double t[32][33];
    const long i0 = (long)blockIdx.x * 32;
    const int j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long i = i0 + r;
        const int j = j0 + tx;
        t[r][tx] = (i < n && j < m) ? src[(size_t)i * ld + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r;
        const long i = i0 + tx;
        if (i < n && j < m) dst[(size_t)j * n + (perm ? perm[i] : i)] = t[tx][r];
    }
}

void launch_transpose_out(hipStream_t st, const double *src, int ld, long n, int m, double *dst, const int *perm) {
    hipLaunchKernelGGL(k_transpose_out, dim3((unsigned)((n + 31) / 32), (unsigned)((m + 31) / 32)), dim3(256), 0, st, src,
                       ld, n, m, dst, perm);
}

// nu_i = sum_j PHI_ij T_ij, one wave per row (predictDiag.m:69-71).
__global__ __launch_bounds__(256) void k_nu(const double *__restrict__ Phi, const double *__restrict__ T, int ld, int n,
                                             int m, double *__restrict__ nu) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j < m; j += 64) s = fma(Phi[(size_t)i * ld + j], T[(size_t)i * ld + j], s);
    s = wave_sum(s);
    if (lane == 0) nu[i] = s;
}

void launch_nu(hipStream_t st, const double *Phi, const double *T, int ld, int n, int m, double *nu) {
    hipLaunchKernelGGL(k_nu, dim3((n + 3) / 4), dim3(256), 0, st, Phi, T, ld, n, m, nu);
}

// ---------------------------------------------------------------------------------------------
// NaN-pattern grouping (getPHI.m:43-54, duplicated at GPz.m:118-129, predict.m:45-56):
// group id of a row = rank, by first occurrence, of its isnan() bit pattern.  Integer work, bit-exact.
// ---------------------------------------------------------------------------------------------
// Any d (W = ceil(d/64) mask words per row) and any number of distinct patterns:
//   1. k_nan_mask      isnan() bits of every row
//   2. k_nan_insert    open-addressing hash table keyed by the mask words; a slot holds the SMALLEST row index carrying its
//                      pattern (atomicMin: the final table does not depend on the order the rows arrive in)
//   3. k_nan_flag      flag[r] = 1 where row r is the first row of its pattern
//   4. k_scan_*        exclusive prefix sum of the flags = the pattern's rank by first occurrence (unique(...,'stable'))
//   5. k_nan_ids       every row looks its pattern's slot up again and reads the rank of the slot's row
#define NAN_EMPTY (-1LL)   // all-ones: written by one memset; as unsigned it is above every row index, so atomicMin works
__global__ void k_nan_mask(const double *__restrict__ X, long n, int d, int W, unsigned long long *__restrict__ mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int w = 0; w < W; ++w) {
        unsigned long long b = 0ULL;
        for (int c = w * 64; c < d && c < w * 64 + 64; ++c) {
            const double v = X[(size_t)c * n + i];
            if (v != v) b |= (1ULL << (c - w * 64));
        }
        mask[(size_t)i * W + w] = b;
    }
}

__device__ __forceinline__ unsigned long long nan_hash(const unsigned long long *mk, int W) {
    unsigned long long h = 0x9e3779b97f4a7c15ULL;
    for (int w = 0; w < W; ++w) {
        h ^= mk[w];
        h *= 0xff51afd7ed558ccdULL;
        h ^= h >> 33;
    }
    return h;
}
__device__ __forceinline__ bool nan_same(const unsigned long long *a, const unsigned long long *b, int W) {
    for (int w = 0; w < W; ++w)
        if (a[w] != b[w]) return false;
    return true;
}

__global__ void k_nan_insert(const unsigned long long *__restrict__ mask, long n, int W, long long *__restrict__ slot,
                             unsigned long long hmask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long *mk = mask + (size_t)i * W;
    unsigned long long h = nan_hash(mk, W) & hmask;
    while (true) {
        long long cur = atomicAdd((unsigned long long *)&slot[h], 0ULL);
        if (cur == NAN_EMPTY) {
            const long long prev = (long long)atomicCAS((unsigned long long *)&slot[h], (unsigned long long)NAN_EMPTY,
                                                        (unsigned long long)i);
            if (prev == NAN_EMPTY) return;
            cur = prev;
        }
        if (nan_same(mask + (size_t)cur * W, mk, W)) {   // every row a slot ever held carries the slot's pattern
            atomicMin((unsigned long long *)&slot[h], (unsigned long long)i);
            return;
        }
        h = (h + 1) & hmask;
    }
}

__global__ void k_nan_flag(const long long *__restrict__ slot, unsigned long long hsize, int *__restrict__ flag) {
    const unsigned long long h = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= hsize) return;
    const long long r = slot[h];
    if (r != NAN_EMPTY) flag[r] = 1;
}

// exclusive prefix sum over n ints, 1024 per workgroup: block totals, their scan by one workgroup, the final pass
__global__ __launch_bounds__(256) void k_scan_blocks(const int *__restrict__ in, long n, int *__restrict__ out,
                                                      int *__restrict__ btot) {
    __shared__ int sh[256];
    const long base = (long)blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
    for (int q = 0; q < 4; ++q) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - s;                                         // exclusive within the block
    for (int q = 0; q < 4; ++q) {
        if (base + q < n) out[base + q] = run;
        run += v[q];
    }
    if (threadIdx.x == 255) btot[blockIdx.x] = sh[255];
}
__global__ void k_scan_totals(int *__restrict__ btot, long nb, int *__restrict__ total) {   // one thread: nb = n/1024 entries
    int run = 0;
    for (long b = 0; b < nb; ++b) { const int t = btot[b]; btot[b] = run; run += t; }
    *total = run;
}
__global__ void k_nan_ids(const unsigned long long *__restrict__ mask, long n, int W, const long long *__restrict__ slot,
                          unsigned long long hmask, const int *__restrict__ rank, const int *__restrict__ boff,
                          int *__restrict__ gid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long *mk = mask + (size_t)i * W;
    unsigned long long h = nan_hash(mk, W) & hmask;
    while (true) {
        const long long r = slot[h];
        if (nan_same(mask + (size_t)r * W, mk, W)) { gid[i] = rank[r] + boff[r / 1024]; return; }
        h = (h + 1) & hmask;
    }
}

// work: see nan_groups_work_bytes().  Returns 0.
size_t nan_groups_work_bytes(long n, int d) {
    const int W = (d + 63) / 64;
    unsigned long long H = 64;
    while (H < 2ULL * (unsigned long long)n) H <<= 1;
    const long nb = (n + 1023) / 1024;
    return (size_t)n * W * 8 + (size_t)H * 8 + (size_t)n * 4 * 2 + (size_t)nb * 4 + 64;
}
int launch_nan_groups(hipStream_t st, const double *X, long n, int d, void *work, int *n_groups, int *group_id) {
    const int W = (d + 63) / 64;
    unsigned long long H = 64;
    while (H < 2ULL * (unsigned long long)n) H <<= 1;
    const long nb = (n + 1023) / 1024;
    unsigned long long *mask = (unsigned long long *)work;
    long long *slot = (long long *)(mask + (size_t)n * W);
    int *flag = (int *)(slot + H);
    int *rank = flag + n;
    int *btot = rank + n;
    const unsigned nblk = (unsigned)((n + 255) / 256);
    (void)hipMemsetAsync(flag, 0, (size_t)n * sizeof(int), st);
    (void)hipMemsetAsync(slot, 0xff, (size_t)H * sizeof(long long), st);             // NAN_EMPTY everywhere
    hipLaunchKernelGGL(k_nan_mask, dim3(nblk), dim3(256), 0, st, X, n, d, W, mask);
    hipLaunchKernelGGL(k_nan_insert, dim3(nblk), dim3(256), 0, st, (const unsigned long long *)mask, n, W, slot, H - 1);
    hipLaunchKernelGGL(k_nan_flag, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, (const long long *)slot, H, flag);
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(256), 0, st, (const int *)flag, n, rank, btot);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1), 0, st, btot, nb, n_groups);
    hipLaunchKernelGGL(k_nan_ids, dim3(nblk), dim3(256), 0, st, (const unsigned long long *)mask, n, W,
                       (const long long *)slot, H - 1, (const int *)rank, (const int *)btot, group_id);
    return 0;
}
