// theta unpacking and the PHI build  (getPHI.m:24-40, 60-125), no-Psi / no-missing branch.
//
// Thread mapping: lanes run along ROWS (samples).  Each thread keeps R rows of X in registers and
// walks over all basis functions; the per-basis parameters (centre p_j, length-scale matrix Gamma_j)
// are wave-uniform, so they reach the VALU as scalar (SGPR) operands through the scalar cache.
// Every 16 basis functions the wave transposes its R x 64 x 16 block through LDS and writes PHI
// row-major (128 contiguous bytes per row).  The heteroscedastic noise model ln beta_i = b + PHI v
// (getPHI.m:116-125) is a per-thread running sum, so PHI is touched once.
#include "gpz_dev.h"
#include "gpz_kernels.h"

// ---------------------------------------------------------------------------------------------
// theta -> expanded parameter block
// ---------------------------------------------------------------------------------------------
// method_id: 0 GL, 1 VL, 2 GD, 3 VD, 4 GC, 5 VC.  de = padded dimension used by the kernels (>= d);
// padded entries are zero so they drop out of every sum.
__global__ void k_unpack(const double *__restrict__ theta, int method_id, int m, int d, int de, int k, int hetero,
                         GpzParams pr) {
    const int md = m * d;
    int g_dim;
    switch (method_id) {
        case 0: g_dim = 1; break;
        case 1: g_dim = m; break;
        case 2: g_dim = d; break;
        case 3: g_dim = md; break;
        case 4: g_dim = d * d; break;
        default: g_dim = d * d * m; break;
    }
    const int gs = blockDim.x * gridDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int e = t0; e < m * de; e += gs) {
        const int j = e / de, c = e % de;
        pr.P[e] = (c < d) ? theta[j + m * c] : 0.0;                        // getPHI.m:24
    }
    if (method_id <= 3) {
        for (int e = t0; e < m * de; e += gs) {
            const int j = e / de, c = e % de;
            double g = 0.0;
            if (c < d) {
                switch (method_id) {
                    case 0: g = theta[md]; break;                          // :29
                    case 1: g = theta[md + j]; break;                      // :31
                    case 2: g = theta[md + c]; break;                      // :33
                    default: g = theta[md + j + m * c]; break;             // :35
                }
            }
            pr.G[e] = g;
            pr.G2[e] = g * g;
        }
    }
    if (method_id >= 4) {
        const int dd = de * de;
        for (int e = t0; e < m * dd; e += gs) {
            const int j = e / dd, a = (e % dd) / de, b = e % de;
            double g = 0.0;
            if (a < d && b < d) g = (method_id == 4) ? theta[md + a + d * b]                 // :37
                                                      : theta[md + a + d * b + d * d * j];   // :39
            pr.G[e] = g;
        }
    }
    const int off = md + g_dim;
    for (int e = t0; e < m * k; e += gs) {
        const double la = theta[off + e];                                  // GPz.m:32
        pr.lnAlpha[e] = la;
        pr.alpha[e] = exp(la);                                             // GPz.m:50
        if (hetero) {
            pr.v[e] = theta[off + m * k + k + e];                          // GPz.m:98
            const double lt = theta[off + m * k + k + m * k + e];          // GPz.m:100
            pr.lnTau[e] = lt;
            pr.tau[e] = exp(lt);
        } else {
            pr.v[e] = 0.0;
            pr.lnTau[e] = 0.0;
            pr.tau[e] = 1.0;
        }
    }
    for (int e = t0; e < k; e += gs) pr.b[e] = theta[off + m * k + e];    // getPHI.m:117
}

void launch_unpack(hipStream_t st, const double *theta, int method_id, int m, int d, int de, int k, int hetero,
                   GpzParams pr) {
    hipLaunchKernelGGL(k_unpack, dim3(64), dim3(256), 0, st, theta, method_id, m, d, de, k, hetero, pr);
}

// ---------------------------------------------------------------------------------------------
// PHI build
// ---------------------------------------------------------------------------------------------
#define PHI_R 2   // rows per thread
#ifndef PHI_UA
#define PHI_UA 1  // unroll of the Gamma-row loop (keeps the scalar-load working set to one row)
#endif

template <int KIND, int D, bool KGEN>
__global__ __launch_bounds__(256) void k_phi(const double *__restrict__ Xc, long ldx, int n, int m, int mp, int k,
                                              const double *__restrict__ P, const double *__restrict__ G,
                                              const double *__restrict__ v, const double *__restrict__ bvec,
                                              const double *__restrict__ omega, const double *__restrict__ Y,
                                              double *__restrict__ Phi, double *__restrict__ lnbeta,
                                              double *__restrict__ wbeta, const double *__restrict__ wv,
                                              double *__restrict__ phiw) {
    constexpr int KM = KGEN ? 8 : 1;
    __shared__ double tile[4][PHI_R][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = ((long)blockIdx.x * 4 + wave) * (64 * PHI_R);

    double x[PHI_R][D];
    bool valid[PHI_R];
#pragma unroll
    for (int r = 0; r < PHI_R; ++r) {
        const long i = row0 + r * 64 + lane;
        valid[r] = i < n;
#pragma unroll
        for (int c = 0; c < D; ++c) x[r][c] = valid[r] ? Xc[c * ldx + i] : 0.0;
    }
    double sv[PHI_R][KM], sw[PHI_R][KM];
#pragma unroll
    for (int r = 0; r < PHI_R; ++r)
#pragma unroll
        for (int o = 0; o < KM; ++o) { sv[r][o] = 0.0; sw[r][o] = 0.0; }

    for (int j0 = 0; j0 < mp; j0 += 16) {
#pragma unroll 1
        for (int jj = 0; jj < 16; ++jj) {
            const int j = j0 + jj;
            double ph[PHI_R];
            if (j < m) {
                double q[PHI_R];
#pragma unroll
                for (int r = 0; r < PHI_R; ++r) q[r] = 0.0;
                if (KIND == GPZ_KIND_DIAG) {
                    const double *pj = P + (size_t)j * D, *gj = G + (size_t)j * D;   // G = gamma^2
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const double pc = pj[c], gc = gj[c];
#pragma unroll
                        for (int r = 0; r < PHI_R; ++r) {
                            const double dl = x[r][c] - pc;
                            q[r] = fma(dl * dl, gc, q[r]);                 // getPHI.m:97  Delta.^2 ./ Sigma
                        }
                    }
                } else {
                    const double *pj = P + (size_t)j * D, *gj = G + (size_t)j * D * D;
                    double dl[PHI_R][D];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const double pc = pj[c];
#pragma unroll
                        for (int r = 0; r < PHI_R; ++r) dl[r][c] = x[r][c] - pc;
                    }
#pragma unroll 2
                    for (int a = 0; a < D; ++a) {
                        double s[PHI_R];
#pragma unroll
                        for (int r = 0; r < PHI_R; ++r) s[r] = 0.0;
#pragma unroll
                        for (int b = 0; b < D; ++b) {
                            const double g = gj[a * D + b];
#pragma unroll
                            for (int r = 0; r < PHI_R; ++r) s[r] = fma(g, dl[r][b], s[r]);
                        }
#pragma unroll
                        for (int r = 0; r < PHI_R; ++r) q[r] = fma(s[r], s[r], q[r]);   // |Gamma_j Delta'|^2  (getPHI.m:73,76)
                    }
                }
#pragma unroll
                for (int r = 0; r < PHI_R; ++r) {
                    ph[r] = valid[r] ? exp(-0.5 * q[r]) : 0.0;             // getPHI.m:113
#pragma unroll
                    for (int o = 0; o < KM; ++o) {
                        if (o < k) {
                            if (v) sv[r][o] = fma(ph[r], v[j + (size_t)m * o], sv[r][o]);      // getPHI.m:124
                            if (wv) sw[r][o] = fma(ph[r], wv[j + (size_t)m * o], sw[r][o]);
                        }
                    }
                }
            } else {
                // padding columns: y in columns m..m+k-1 (the SYRK then yields PHI' W y for free), zeros after
#pragma unroll
                for (int r = 0; r < PHI_R; ++r) {
                    const long i = row0 + r * 64 + lane;
                    ph[r] = (Y != nullptr && (j - m) < k && valid[r]) ? Y[(size_t)(j - m) * ldx + i] : 0.0;
                }
            }
            if (Phi) {
#pragma unroll
                for (int r = 0; r < PHI_R; ++r) tile[wave][r][lane][jj] = ph[r];
            }
        }
        if (Phi) {
            __syncthreads();
            // wave-uniform base + 32-bit lane offset keeps the 32 stores from each holding a 64-bit address
            double *base = Phi + (size_t)row0 * mp + j0;
            const unsigned loff = (unsigned)(lane >> 4) * (unsigned)mp + (unsigned)(lane & 15);
#pragma unroll 4
            for (int q = 0; q < PHI_R * 16; ++q) {
                const int r = q >> 4, it = q & 15;
                base[loff + (unsigned)(r * 64 + it * 4) * (unsigned)mp] = tile[wave][r][it * 4 + (lane >> 4)][lane & 15];
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int r = 0; r < PHI_R; ++r) {
        const long i = row0 + r * 64 + lane;
#pragma unroll
        for (int o = 0; o < KM; ++o) {
            if (o < k && i < ldx) {
                const double lb = bvec[o] + sv[r][o];                      // getPHI.m:119,124
                lnbeta[(size_t)o * ldx + i] = valid[r] ? lb : 0.0;
                if (wbeta) {
                    const double om = omega ? (valid[r] ? omega[i] : 0.0) : 1.0;
                    wbeta[(size_t)o * ldx + i] = valid[r] ? om * exp(-lb) : 0.0;   // GPz.m:43,48
                }
                if (phiw) phiw[(size_t)o * ldx + i] = sw[r][o];
            }
        }
    }
}

template <int KIND, int D>
static void launch_phi_kd(hipStream_t st, const PhiArgs &a) {
    const int rows_per_wg = 4 * 64 * PHI_R;
    const int nwg = (a.n_pad + rows_per_wg - 1) / rows_per_wg;
    if (a.k == 1)
        hipLaunchKernelGGL((k_phi<KIND, D, false>), dim3(nwg), dim3(256), 0, st, a.Xc, a.ldx, a.n, a.m, a.mp, a.k, a.P,
                           a.G, a.v, a.b, a.omega, a.Y, a.Phi, a.lnbeta, a.wbeta, a.w, a.phiw);
    else
        hipLaunchKernelGGL((k_phi<KIND, D, true>), dim3(nwg), dim3(256), 0, st, a.Xc, a.ldx, a.n, a.m, a.mp, a.k, a.P,
                           a.G, a.v, a.b, a.omega, a.Y, a.Phi, a.lnbeta, a.wbeta, a.w, a.phiw);
}

template <int KIND>
static int launch_phi_k(hipStream_t st, const PhiArgs &a) {
    switch (a.d) {
        case 1: launch_phi_kd<KIND, 1>(st, a); break;
        case 2: launch_phi_kd<KIND, 2>(st, a); break;
        case 3: launch_phi_kd<KIND, 3>(st, a); break;
        case 4: launch_phi_kd<KIND, 4>(st, a); break;
        case 5: launch_phi_kd<KIND, 5>(st, a); break;
        case 6: launch_phi_kd<KIND, 6>(st, a); break;
        case 8: launch_phi_kd<KIND, 8>(st, a); break;
        case 10: launch_phi_kd<KIND, 10>(st, a); break;
        case 12: launch_phi_kd<KIND, 12>(st, a); break;
        case 16: launch_phi_kd<KIND, 16>(st, a); break;
        case 20: launch_phi_kd<KIND, 20>(st, a); break;
        default: return -1;
    }
    return 0;
}

// a.d must be one of the padded dimensions returned by gpz_pad_dim().
int launch_phi(hipStream_t st, const PhiArgs &a) {
    if (a.k > 8) return -1;
    return (a.kind == GPZ_KIND_DIAG) ? launch_phi_k<GPZ_KIND_DIAG>(st, a) : launch_phi_k<GPZ_KIND_COV>(st, a);
}
