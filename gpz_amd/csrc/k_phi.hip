// theta unpacking and the PHI build  (getPHI.m:24-40, 60-125), no-Psi / no-missing branch (+ diagonal Psi, + missing
// dimensions for the diagonal kinds).
//
// Thread mapping: lanes run along ROWS (samples).  Each thread keeps R rows of X in registers and walks over all basis
// functions.  Diagonal kinds: the per-basis parameters (centre p_j, squared length scales) are wave-uniform and reach
// the VALU as scalar (SGPR) operands.  Covariance kinds: [R_j | R_j p_j] of EIGHT basis functions (Gamma_j = Q_j R_j,
// k_prep_cov) are staged in LDS per workgroup and read back as broadcast ds_reads through a VGPR base + immediates, each
// read feeding the thread's R rows, software-pipelined row by row of R_j under sched_barrier.  Every 8 (cov) / 16 (diag)
// basis functions the wave transposes its block through LDS and writes PHI row-major.  The heteroscedastic noise model
// ln beta_i = b + PHI v (getPHI.m:116-125) is a per-thread running sum, so PHI is touched once.
#include <type_traits>
#include "gpz_dev.h"
#include "gpz_kernels.h"

// ---------------------------------------------------------------------------------------------
// theta -> expanded parameter block
// ---------------------------------------------------------------------------------------------
// method_id: 0 GL, 1 VL, 2 GD, 3 VD, 4 GC, 5 VC.  de = padded dimension used by the kernels (>= d);
// padded entries are zero so they drop out of every sum.
__global__ void k_unpack(const double *__restrict__ theta, int method_id, int m, int d, int de, int k, int hetero,
                         GpzParams pr, int *__restrict__ clear2, double *__restrict__ zero_p, int zero_n) {
    if (clear2 && blockIdx.x == 0 && threadIdx.x < 2) clear2[threadIdx.x] = 0;   // the evaluation's status words (was a memset node of its own)
    if (zero_p && blockIdx.x == 1)                                               // ... and a few doubles a later stage only adds to or skips
        for (int e = threadIdx.x; e < zero_n; e += blockDim.x) zero_p[e] = 0.0;
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
                   GpzParams pr, int *clear2, double *zero_p, int zero_n) {
    hipLaunchKernelGGL(k_unpack, dim3(64), dim3(256), 0, st, theta, method_id, m, d, de, k, hetero, pr, clear2, zero_p, zero_n);
}

// ---------------------------------------------------------------------------------------------
// Covariance kinds: R_j = triangular factor of Gamma_j (Householder QR), c_j = R_j p_j.
// |Gamma_j (x - p_j)|^2 = |R_j x - c_j|^2 exactly (orthogonal invariance), with d(d+1)/2 instead of d^2
// multiply-adds per (sample, basis) pair and no loss of conditioning (no Gamma'Gamma is formed).
// Rc[j] = [R packed upper row-major: row a holds b = a..de-1 | c (de)], stride de(de+1)/2 + de.
// ---------------------------------------------------------------------------------------------
// DE > 0: compile-time dimension, the matrix lives in registers (fully unrolled); DE == 0: runtime de, scratch.
template <int DE>
__global__ __launch_bounds__(64) void k_prep_cov(const double *__restrict__ G, const double *__restrict__ P, int m, int de_rt,
                           double *__restrict__ Rc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    constexpr int CAP = DE > 0 ? DE : 20;
    const int de = DE > 0 ? DE : de_rt;
    double A[CAP * CAP], v[CAP];
    const double *Gj = G + (size_t)j * de * de;
#pragma unroll
    for (int r = 0; r < CAP; ++r)
#pragma unroll
        for (int c = 0; c < CAP; ++c)
            if (r < de && c < de) A[r * CAP + c] = Gj[r * de + c];          // A[r][c] = Gamma_j(r, c)
#pragma unroll
    for (int c = 0; c < CAP; ++c) {
        if (c >= de) break;
        double n2 = 0.0;
#pragma unroll
        for (int r = 0; r < CAP; ++r)
            if (r >= c && r < de) n2 = fma(A[r * CAP + c], A[r * CAP + c], n2);
        if (n2 == 0.0) continue;
        const double nrm = sqrt(n2);
        const double acc = A[c * CAP + c];
        const double alpha = (acc > 0.0) ? -nrm : nrm;
#pragma unroll
        for (int r = 0; r < CAP; ++r)
            if (r >= c && r < de) v[r] = A[r * CAP + c];
        v[c] -= alpha;
        const double vn2 = n2 - acc * acc + v[c] * v[c];
#pragma unroll
        for (int cc = 0; cc < CAP; ++cc) {
            if (cc <= c || cc >= de) continue;
            double dot = 0.0;
#pragma unroll
            for (int r = 0; r < CAP; ++r)
                if (r >= c && r < de) dot = fma(v[r], A[r * CAP + cc], dot);
            const double f = 2.0 * dot / vn2;
#pragma unroll
            for (int r = 0; r < CAP; ++r)
                if (r >= c && r < de) A[r * CAP + cc] = fma(-f, v[r], A[r * CAP + cc]);
        }
        A[c * CAP + c] = alpha;
#pragma unroll
        for (int r = 0; r < CAP; ++r)
            if (r > c && r < de) A[r * CAP + c] = 0.0;
    }
    const int nt = de * (de + 1) / 2;
    double *o = Rc + (size_t)j * (nt + de);
    const double *pj = P + (size_t)j * de;
#pragma unroll
    for (int a = 0; a < CAP; ++a) {
        if (a >= de) break;
        double s = 0.0;
        const int roff = a * de - a * (a - 1) / 2;
#pragma unroll
        for (int b = 0; b < CAP; ++b)
            if (b >= a && b < de) {
                const double r = A[a * CAP + b];
                o[roff + (b - a)] = r;
                s = fma(r, pj[b], s);
            }
        o[nt + a] = s;
    }
}

void launch_prep_cov(hipStream_t st, const double *G, const double *P, int m, int de, double *Rc, double *ws) {
    dim3 g((m + 63) / 64), b(64);
    switch (de) {
        case 1: hipLaunchKernelGGL(k_prep_cov<1>, g, b, 0, st, G, P, m, de, Rc); break;
        case 2: hipLaunchKernelGGL(k_prep_cov<2>, g, b, 0, st, G, P, m, de, Rc); break;
        case 3: hipLaunchKernelGGL(k_prep_cov<3>, g, b, 0, st, G, P, m, de, Rc); break;
        case 4: hipLaunchKernelGGL(k_prep_cov<4>, g, b, 0, st, G, P, m, de, Rc); break;
        case 5: hipLaunchKernelGGL(k_prep_cov<5>, g, b, 0, st, G, P, m, de, Rc); break;
        case 6: hipLaunchKernelGGL(k_prep_cov<6>, g, b, 0, st, G, P, m, de, Rc); break;
        case 8: hipLaunchKernelGGL(k_prep_cov<8>, g, b, 0, st, G, P, m, de, Rc); break;
        case 10: hipLaunchKernelGGL(k_prep_cov<10>, g, b, 0, st, G, P, m, de, Rc); break;
        default:
            if (de > 20) (void)launch_prep_cov_wide(st, G, P, m, de, Rc, ws);   // runtime-d QR in LDS / in `ws` (k_wide.hip)
            else hipLaunchKernelGGL(k_prep_cov<0>, g, b, 0, st, G, P, m, de, Rc);
            break;
    }
}

// ---------------------------------------------------------------------------------------------
// PHI build
// ---------------------------------------------------------------------------------------------
// Diagonal kinds (GL, VL, GD, VD).  R rows per thread, JB basis functions per LDS transposition block.
//   no Psi  (getPHI.m:97):   ln PHI = -1/2 sum_c Delta_c^2 gamma_c^2
//   Psi     (getPHI.m:104):  ln PHI = -1/2 sum_c Delta_c^2/(psi_c + sigma_c) - 1/2 sum_c ln(1 + psi_c/sigma_c),  sigma = gamma^-2
//   missing dimensions (NaN in X; getPHI.m:64-69,97,104): dropped from the sums, minus 1/2 |u_i| ln 2.
// Mc (optional) is the observed-mask (1.0 / 0.0) in the layout of Xc; X and Psi hold 0 at missing entries.
template <int D, bool KGEN, bool PSI, int R, int JB>
__global__ __launch_bounds__(256) void k_phi_diag(const double *__restrict__ Xc, long ldx, int n, int m, int mp, int k,
                                                   const double *__restrict__ P, const double *__restrict__ G,
                                                   const double *__restrict__ v, const double *__restrict__ bvec,
                                                   const double *__restrict__ omega, long om_ld, const double *__restrict__ Y,
                                                   double *__restrict__ Phi, double *__restrict__ lnbeta,
                                                   double *__restrict__ wbeta, const double *__restrict__ wv,
                                                   double *__restrict__ phiw, const double *__restrict__ Psic,
                                                   const double *__restrict__ Mc, const double *__restrict__ ucnt, int jgroup,
                                                   double *__restrict__ part, long ldp) {
    constexpr int KM = KGEN ? 8 : 1;
    __shared__ double tile[4][R][64][JB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = ((long)blockIdx.x * 4 + wave) * (64 * R);
    // column group of this workgroup (few rows: the basis functions are split as well so that the grid fills the chip - lanes run along
    // rows, and 1e5 rows are 1.5 waves per SIMD; the per-row sums of the groups are then combined by k_phi_finalize)
    const int jlo = part ? (int)blockIdx.y * jgroup : 0, jhi = part ? min(mp, jlo + jgroup) : mp;

    double x[R][D], ps[PSI ? R : 1][PSI ? D : 1], mk[R][D], q0[R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = row0 + r * 64 + lane;
        valid[r] = i < n;
        q0[r] = ucnt ? ucnt[i] * GPZ_LOG2 : 0.0;                          // |u_i| ln 2
#pragma unroll
        for (int c = 0; c < D; ++c) {
            x[r][c] = Xc[c * ldx + i];                                     // rows >= n are zero-padded
            mk[r][c] = Mc ? Mc[c * ldx + i] : 1.0;
            if (PSI) ps[r][c] = Psic[c * ldx + i];
        }
    }
    double sv[R][KM], sw[R][KM];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < KM; ++o) { sv[r][o] = 0.0; sw[r][o] = 0.0; }

    for (int j0 = jlo; j0 < jhi; j0 += JB) {
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            double ph[R];
            if (j < m) {
                double q[R], pr[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { q[r] = q0[r]; pr[r] = 1.0; }
                const double *pj = P + (size_t)j * D, *gj = G + (size_t)j * D;   // G = gamma^2 = 1/sigma
                // two copies of the sum, with and without the observed-mask factor: complete inputs do not pay a multiply per dimension
                auto quad = [&](auto masked) {
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const double pc = pj[c], gc = gj[c];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            double dl = x[r][c] - pc;
                            if (decltype(masked)::value) dl *= mk[r][c];
                            if (PSI) {
                                const double u = fma(ps[r][c], gc, 1.0);   // 1 + psi/sigma  (psi = 0 where missing)
                                q[r] = fma(dl * dl, gc * gpz_rcp1(u), q[r]);   // Delta^2/(psi+sigma)
                                pr[r] *= u;
                            } else {
                                q[r] = fma(dl * dl, gc, q[r]);             // getPHI.m:97  Delta.^2 ./ Sigma
                            }
                        }
                    }
                };
                if (Mc) quad(std::true_type{}); else quad(std::false_type{});
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (PSI) q[r] += log(pr[r]);                           // sum_c ln(1 + psi/sigma)
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
                for (int r = 0; r < R; ++r) {
                    const long i = row0 + r * 64 + lane;
                    ph[r] = (Y != nullptr && (j - m) < k) ? Y[(size_t)(j - m) * ldx + i] : 0.0;   // zero-padded rows
                }
            }
            if (Phi) {
#pragma unroll
                for (int r = 0; r < R; ++r) tile[wave][r][lane][jj] = ph[r];
            }
        }
        if (Phi) {
            // tile[wave] belongs to this wave alone and a wave's LDS operations complete in order: the transposition needs no
            // workgroup barrier (two per 16 columns kept the four waves in lockstep), only the program order of its writes and reads
            __builtin_amdgcn_wave_barrier();
            // wave-uniform base + 32-bit lane offset; each store instruction covers 64/JB rows x JB columns
            constexpr int RPI = 64 / JB;                    // rows per store instruction
            double *base = Phi + (size_t)row0 * mp + j0;
            const int lr = lane / JB, lc = lane % JB;
            const unsigned loff = (unsigned)lr * (unsigned)mp + (unsigned)lc;
#pragma unroll 4
            for (int qq = 0; qq < R * JB; ++qq) {
                const int r = qq / JB, it = qq % JB;
                base[loff + (unsigned)(r * 64 + it * RPI) * (unsigned)mp] = tile[wave][r][it * RPI + lr][lc];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (part) {   // partial row sums of this column group (layout of k_phi_cov / k_phi_finalize)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long i = row0 + r * 64 + lane;
#pragma unroll
            for (int o = 0; o < KM; ++o)
                if (o < k && i < ldp) {
                    part[(((size_t)blockIdx.y * 2 + 0) * k + o) * ldp + i] = sv[r][o];
                    part[(((size_t)blockIdx.y * 2 + 1) * k + o) * ldp + i] = sw[r][o];
                }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = row0 + r * 64 + lane;
#pragma unroll
        for (int o = 0; o < KM; ++o) {
            if (o < k && i < ldx) {
                const double lb = bvec[o] + sv[r][o];                      // getPHI.m:119,124
                lnbeta[(size_t)o * ldx + i] = valid[r] ? lb : 0.0;
                if (wbeta) {
                    const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
                    wbeta[(size_t)o * ldx + i] = valid[r] ? om * exp(-lb) : 0.0;   // GPz.m:43,48
                }
                if (phiw) phiw[(size_t)o * ldx + i] = sw[r][o];
            }
        }
    }
}

// |R_j x - c_j|^2 for R rows per lane, wide inputs (D > 10): the packed factor is streamed from LDS in chunks of 8
// operands, one chunk ahead of the multiply-adds.  Holding two whole rows of R_j (as the D <= 10 path does) needs
// 4(D+1) VGPRs on top of the 2*R*D of x and spilled 0.5-1.5 KB per lane at D = 12..20.
template <int D, int R>
__device__ __forceinline__ void cov_quad_chunked(const double *__restrict__ rj, const double (&x)[R][D], double (&q)[R]) {
    constexpr int NT = D * (D + 1) / 2;
    constexpr int CH = 8;
    double cur[CH + 1], nxt[CH + 1];                 // [CH]: c_a, fetched with the first chunk of row a
    auto load = [&](int a, int b0, double (&dst)[CH + 1]) {
        const int off = a * D - a * (a - 1) / 2 + (b0 - a);
#pragma unroll
        for (int e = 0; e < CH; ++e)
            if (b0 + e < D) dst[e] = rj[off + e];
        if (b0 == a) dst[CH] = rj[NT + a];
    };
    double s[R];
    load(0, 0, cur);
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
        for (int b0 = a; b0 < D; b0 += CH) {
            const bool last = b0 + CH >= D;
            const int na = last ? a + 1 : a, nb0 = last ? a + 1 : b0 + CH;
            if (na < D) load(na, nb0, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (b0 == a) {
#pragma unroll
                for (int r = 0; r < R; ++r) s[r] = -cur[CH];
            }
#pragma unroll
            for (int e = 0; e < CH; ++e)
                if (b0 + e < D) {
#pragma unroll
                    for (int r = 0; r < R; ++r) s[r] = fma(cur[e], x[r][b0 + e], s[r]);
                }
            if (last) {
#pragma unroll
                for (int r = 0; r < R; ++r) q[r] = fma(s[r], s[r], q[r]);           // getPHI.m:73,76
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e <= CH; ++e) cur[e] = nxt[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PHI build, covariance kinds.  Same thread mapping (lanes along rows, R rows per thread) but the per-basis
// parameters [R_j | c_j] are staged once per workgroup into LDS for JB basis functions at a time and read back
// as wave-uniform (broadcast) ds_reads: d(d+1)/2 + d doubles per basis do not fit the SGPR file, and each
// LDS operand feeds R = 4 multiply-adds, so the LDS pipe stays well below the VALU time.
// ---------------------------------------------------------------------------------------------
template <int D, bool KGEN, int R, int JB, bool TAB>
__global__ __launch_bounds__(256, 2) void k_phi_cov(const double *__restrict__ Xc, long ldx, int n, int m, int mp, int k,
                                                  const double *__restrict__ Rc,
                                                  const double *__restrict__ v, const double *__restrict__ bvec,
                                                  const double *__restrict__ omega, long om_ld, const double *__restrict__ Y,
                                                  double *__restrict__ Phi, double *__restrict__ lnbeta,
                                                  double *__restrict__ wbeta, const double *__restrict__ wv,
                                                  double *__restrict__ phiw, int jgroup, double *__restrict__ part, long ldp,
                                                  const int *__restrict__ wgtab) {
    constexpr int KM = KGEN ? 8 : 1;
    constexpr int NT = D * (D + 1) / 2;
    constexpr int NP = NT + D;                       // doubles per basis function
    constexpr int NPB = JB * NP;                     // doubles per parameter block
    constexpr int NLD = (NPB + 255) / 256;           // staging loads per thread
    __shared__ double tile[4][R][64][JB + 1];
    __shared__ double prm[NPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // TAB: rows are sorted by NaN pattern and this workgroup serves the rows [tab.x, tab.y) of pattern tab.z, whose
    // parameter block follows the others in Rc; rows at or past tab.y belong to another workgroup and are not written.
    long row0 = ((long)blockIdx.x * 4 + wave) * (64 * R);
    long row_end = n;
    if (TAB) {
        const int *t = wgtab + 4 * (size_t)blockIdx.x;
        row0 = (long)t[0] + wave * (64 * R);
        row_end = t[1];
        Rc += (size_t)t[2] * m * (NT + D);
    }
    const long wr_end = TAB ? row_end : ldx;

    double x[R][D];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = row0 + r * 64 + lane;
        valid[r] = i < row_end;
#pragma unroll
        for (int c = 0; c < D; ++c) x[r][c] = Xc[c * ldx + i];   // rows >= n are zero-padded
    }
    double sv[R][KM], sw[R][KM];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < KM; ++o) { sv[r][o] = 0.0; sw[r][o] = 0.0; }

    // parameters of block 0
    double stg[NLD];
    const size_t prm_total = (size_t)m * NP;
    auto pload = [&](int j0) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = tid + 256 * q;
            const size_t ge = (size_t)j0 * NP + e;
            stg[q] = (e < NPB && ge < prm_total) ? Rc[ge] : 0.0;
        }
    };
    auto pstore = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = tid + 256 * q;
            if (e < NPB) prm[e] = stg[q];
        }
    };
    // column group of this workgroup: [jlo, jhi) in steps of JB (small row counts are split over basis functions
    // as well so the grid still fills the chip; the per-row sums are then combined by k_phi_finalize)
    const int jlo = blockIdx.y * jgroup;
    const int jhi = min(mp, jlo + jgroup);
    pload(jlo);
    pstore();
    __syncthreads();

    for (int j0 = jlo; j0 < jhi; j0 += JB) {
        if (j0 + JB < m && j0 + JB < jhi) pload(j0 + JB);   // in flight during the compute phase
        // Two loops instead of one with an `if (j < m)` inside: the optimiser hoists LDS reads out of a conditional
        // (they are speculatable), i.e. above the sched_barriers that bound their live ranges - at D = 20 that held
        // the whole factor in registers and spilled 1.5 KB per lane.
        const int jcnt = min(JB, m - j0);
#pragma unroll 1
        for (int jj = 0; jj < jcnt; ++jj) {
            const int j = j0 + jj;
            double ph[R];
            {
                // v_j, w_j first: a conditional load between the quadratic form and its use would let the optimiser
                // sink all multiply-adds below it, away from the LDS reads that the sched_barriers pair them with
                double vj[KM], wj[KM];
#pragma unroll
                for (int o = 0; o < KM; ++o) {
                    vj[o] = (v && o < k) ? v[j + (size_t)m * o] : 0.0;
                    wj[o] = (wv && o < k) ? wv[j + (size_t)m * o] : 0.0;
                }
                // R_j's LDS address through a vector register: as a scalar the compiler rebuilds every broadcast-read address
                // with s_add + v_mov (39 VALU moves per basis and 4 rows at d = 10); a VGPR base takes immediates
                unsigned rjo = (unsigned)(jj * NP * sizeof(double));
                asm volatile("" : "+v"(rjo));
                const double *rj = reinterpret_cast<const double *>(reinterpret_cast<const char *>(prm) + rjo);
                double q[R];
#pragma unroll
                for (int r = 0; r < R; ++r) q[r] = 0.0;
                // software-pipelined over the rows of R_j: the LDS (broadcast) reads of row a+1 are issued before
                // the multiply-adds of row a, and sched_barrier keeps the compiler from hoisting every read of
                // the factor to the top (which needs d(d+1) VGPRs and spills)
                if constexpr (D > 10) {
                    cov_quad_chunked<D, R>(rj, x, q);
                } else {
                double gc[D + 1], gn[D + 1];
#pragma unroll
                for (int b = 0; b < D; ++b) gc[b] = rj[b];
                gc[D] = rj[NT];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    if (a + 1 < D) {
#pragma unroll
                        for (int b = a + 1; b < D; ++b) gn[b] = rj[(a + 1) * D - (a + 1) * a / 2 + (b - a - 1)];
                        gn[D] = rj[NT + a + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double s[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) s[r] = -gc[D];
#pragma unroll
                    for (int b = a; b < D; ++b) {
#pragma unroll
                        for (int r = 0; r < R; ++r) s[r] = fma(gc[b], x[r][b], s[r]);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) q[r] = fma(s[r], s[r], q[r]);       // |R_j x - c_j|^2  (getPHI.m:73,76)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = a + 1; b <= D; ++b) gc[b] = gn[b];
                }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double e = exp(-0.5 * q[r]);                             // getPHI.m:113
                    ph[r] = valid[r] ? e : 0.0;
#pragma unroll
                    for (int o = 0; o < KM; ++o) {
                        sv[r][o] = fma(ph[r], vj[o], sv[r][o]);                    // getPHI.m:124
                        sw[r][o] = fma(ph[r], wj[o], sw[r][o]);
                    }
                }
            }
            if (Phi) {
#pragma unroll
                for (int r = 0; r < R; ++r) tile[wave][r][lane][jj] = ph[r];
            }
        }
        if (Phi) {
#pragma unroll 1
            for (int jj = max(jcnt, 0); jj < JB; ++jj) {                                  // padding columns: [Y | 0]
                const int j = j0 + jj;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const long i = row0 + r * 64 + lane;
                    tile[wave][r][lane][jj] = (Y != nullptr && (j - m) < k) ? Y[(size_t)(j - m) * ldx + i] : 0.0;   // zero-padded rows
                }
            }
        }
        __syncthreads();                             // tile complete; every wave is done with prm
        if (Phi) {
            constexpr int RPI = 64 / JB;
            double *base = Phi + (size_t)row0 * mp + j0;
            const int lr = lane / JB, lc = lane % JB;
            const unsigned loff = (unsigned)lr * (unsigned)mp + (unsigned)lc;
#pragma unroll 4
            for (int qq = 0; qq < R * JB; ++qq) {
                const int r = qq / JB, it = qq % JB;
                if (!TAB || row0 + r * 64 + it * RPI + lr < row_end)
                    base[loff + (unsigned)(r * 64 + it * RPI) * (unsigned)mp] = tile[wave][r][it * RPI + lr][lc];
            }
        }
        if (j0 + JB < m && j0 + JB < jhi) pstore();
        __syncthreads();
    }

    if (part) {   // column-split launch: partial sums [group][2][k][ldp] (ldp = the rows of this launch); k_phi_finalize combines them in fixed order
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long i = row0 + r * 64 + lane;
#pragma unroll
            for (int o = 0; o < KM; ++o)
                if (o < k && i < wr_end && i < ldp) {
                    part[(((size_t)blockIdx.y * 2 + 0) * k + o) * ldp + i] = sv[r][o];
                    part[(((size_t)blockIdx.y * 2 + 1) * k + o) * ldp + i] = sw[r][o];
                }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = row0 + r * 64 + lane;
#pragma unroll
        for (int o = 0; o < KM; ++o) {
            if (o < k && i < wr_end) {
                const double lb = bvec[o] + sv[r][o];                      // getPHI.m:119,124
                lnbeta[(size_t)o * ldx + i] = valid[r] ? lb : 0.0;
                if (wbeta) {
                    const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
                    wbeta[(size_t)o * ldx + i] = valid[r] ? om * exp(-lb) : 0.0;   // GPz.m:43,48
                }
                if (phiw) phiw[(size_t)o * ldx + i] = sw[r][o];
            }
        }
    }
}

// lnbeta = b + sum_g part_v[g], omega*beta, PHI*w from the column-group partial sums (fixed order: repeatable)
__global__ void k_phi_finalize(const double *__restrict__ part, long ldp, int ngroup, long ldx, long rows, int n, int k,
                               const double *__restrict__ bvec, const double *__restrict__ omega, long om_ld,
                               double *__restrict__ lnbeta, double *__restrict__ wbeta, double *__restrict__ phiw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    for (int o = 0; o < k; ++o) {
        double sv = 0.0, sw = 0.0;
        for (int g = 0; g < ngroup; ++g) {
            sv += part[(((size_t)g * 2 + 0) * k + o) * ldp + i];
            sw += part[(((size_t)g * 2 + 1) * k + o) * ldp + i];
        }
        const bool valid = i < n;
        const double lb = bvec[o] + sv;
        lnbeta[(size_t)o * ldx + i] = valid ? lb : 0.0;
        if (wbeta) {
            const double om = omega ? omega[(size_t)o * om_ld + i] : 1.0;
            wbeta[(size_t)o * ldx + i] = valid ? om * exp(-lb) : 0.0;
        }
        if (phiw) phiw[(size_t)o * ldx + i] = valid ? sw : 0.0;   // rows past the data carry no partial sums
    }
}

// Column groups of a PHI build with few rows.  Lanes run along rows, so `nwg` workgroups may not fill the chip; the basis functions
// are then split into groups (multiples of JB columns) whose per-row sums k_phi_finalize combines.  Both kernels are bound by the
// vector ALU their co-resident workgroups share, so a launch takes (workgroups on the busiest CU) x (columns per workgroup + its
// prologue in columns' worth), with at least `min_wgs` workgroups' worth of time per CU (fewer cannot hide the parameter loads).
// c2 (391 row blocks, 208 columns): g = 1 .. 13 measured 136, 84, 83, 93, 78, -, 75, ..., 74 us and the model ranks them the same
// ("about 1024 workgroups" chose g = 3).  Among choices within 1 % the largest wins (c4: two groups 4.34 ms, one 4.44).
static int phi_pick_groups(int nwg, int mp, int JB, int min_wgs, int prologue_cols, int max_groups) {
    const int ncu = gpz_cu_count();
    constexpr int MAXG = 64;                                     // (the callers' part_groups is 16: phipart holds that many partial sums per row)
    long cost[MAXG + 1];
    long best_cost = -1;
    if (max_groups > MAXG) max_groups = MAXG;
    if (max_groups < 1) max_groups = 1;
    for (int g = 1; g <= max_groups; ++g) {
        const int jg = ((mp + g - 1) / g + JB - 1) / JB * JB, ge = (mp + jg - 1) / jg;
        cost[g] = -1;
        if (ge != g) continue;                                   // this count collapses to a smaller one
        long per_cu = ((long)nwg * g + ncu - 1) / ncu;
        if (per_cu < min_wgs) per_cu = min_wgs;
        cost[g] = per_cu * (jg + prologue_cols);
        if (best_cost < 0 || cost[g] < best_cost) best_cost = cost[g];
    }
    int best = 1;
    for (int g = 1; g <= max_groups; ++g)
        if (cost[g] >= 0 && cost[g] * 100 <= best_cost * 101) best = g;
    return best;
}

// 4 rows per thread and 8-wide blocks while [tile | params] fits twice in a CU's LDS; else 2 rows
template <int D>
struct PhiCovShape {
    static constexpr int NP = D * (D + 1) / 2 + D;
    static constexpr bool BIG = (4 * 4 * 64 * 9 + 8 * NP) * 8 * 2 <= 160 * 1024;
#ifdef GPZ_PHI_COV_R   // measured at c4: two rows per thread (124 VGPRs, three workgroups per CU) 5.08 ms against 4.16 ms with four
    static constexpr int R = GPZ_PHI_COV_R;
#else
    static constexpr int R = BIG ? 4 : 2;
#endif
};

template <int D>
static void launch_phi_cov_d(hipStream_t st, const PhiArgs &a) {
    constexpr int R = PhiCovShape<D>::R;
    constexpr int JB = 8;
    const int rows_per_wg = 4 * 64 * R;
    const int nwg = a.wgtab ? a.nwg_tab : (a.n_pad + rows_per_wg - 1) / rows_per_wg;
    if (nwg <= 0) return;
    // few rows: split the basis functions into groups as well
    int ngroup = 1;
    if (a.part && nwg < 1024) {
        int maxg = a.mp / (4 * JB);                                                      // at least 32 columns per group
        if (maxg > a.part_groups) maxg = a.part_groups;
        ngroup = phi_pick_groups(nwg, a.mp, JB, 2, 8, maxg < 1 ? 1 : maxg);   // two workgroups of this kernel are resident per CU
    }
    int jgroup = ((a.mp + ngroup - 1) / ngroup + JB - 1) / JB * JB;
    ngroup = (a.mp + jgroup - 1) / jgroup;
    double *part = ngroup > 1 ? a.part : nullptr;
    dim3 grid(nwg, ngroup);
#define PHI_COV(KG, TB) \
    hipLaunchKernelGGL((k_phi_cov<D, KG, R, JB, TB>), grid, dim3(256), 0, st, a.Xc, a.ldx, a.n, a.m, a.mp, a.k, a.G, a.v, \
                       a.b, a.omega, a.om_ld, a.Y, a.Phi, a.lnbeta, a.wbeta, a.w, a.phiw, jgroup, part, (long)a.n_pad, a.wgtab)
    if (a.wgtab) { if (a.k == 1) PHI_COV(false, true); else PHI_COV(true, true); }
    else { if (a.k == 1) PHI_COV(false, false); else PHI_COV(true, false); }
#undef PHI_COV
    if (part)
        hipLaunchKernelGGL(k_phi_finalize, dim3((unsigned)((a.n_pad + 255) / 256)), dim3(256), 0, st, (const double *)part,
                           (long)a.n_pad, ngroup, a.ldx, (long)a.n_pad, a.n, a.k, a.b, a.omega, a.om_ld, a.lnbeta, a.wbeta, a.phiw);
}

#ifndef GPZ_PHI_DIAG_RP
#define GPZ_PHI_DIAG_RP(D) ((D) == 16 ? 1 : 2)   // measured: d=16 0.69 -> 0.52 ms, d=20 0.93 -> 1.05 ms (n=1e5, m=256)
#endif
template <int KIND, int D>
static void launch_phi_kd(hipStream_t st, const PhiArgs &a) {
    if (KIND == GPZ_KIND_COV) {
        launch_phi_cov_d<D>(st, a);
        return;
    }
    constexpr int R = 2, JB = 16;                    // diagonal kinds are store-bound
    // with input noise a thread holds x, the mask and Psi of its rows (3 R D doubles): one row per thread where that measured faster
    constexpr int RP = GPZ_PHI_DIAG_RP(D);
    // few rows (fewer than two workgroups per CU at two rows per thread): one row per thread doubles the workgroups
    const bool small = (a.n_pad + 511) / 512 < 512;
    const int rr_eff = a.Psic ? RP : (small ? 1 : R);
    const int nwg = (a.n_pad + 256 * rr_eff - 1) / (256 * rr_eff);
    int ngroup = 1;
    if (a.part && nwg > 0 && nwg < 1024 && !gpz_opts().phi_diag_no_split) {
        int maxg = a.mp / JB;                                                            // down to one transposition block per group
        if (maxg > a.part_groups) maxg = a.part_groups;
        ngroup = phi_pick_groups(nwg, a.mp, JB, 3, 3, maxg < 1 ? 1 : maxg);   // three workgroups per CU hide the parameter loads
    }
    int jgroup = ((a.mp + ngroup - 1) / ngroup + JB - 1) / JB * JB;
    ngroup = (a.mp + jgroup - 1) / jgroup;
    double *part = ngroup > 1 ? a.part : nullptr;
#define PHI_DIAG_R(KG, PS, RR) \
    hipLaunchKernelGGL((k_phi_diag<D, KG, PS, RR, JB>), dim3((a.n_pad + 256 * RR - 1) / (256 * RR), ngroup),              \
                       dim3(256), 0, st, a.Xc, a.ldx, a.n, a.m, a.mp, a.k, a.P,                                           \
                       a.G, a.v, a.b, a.omega, a.om_ld, a.Y, a.Phi, a.lnbeta, a.wbeta, a.w, a.phiw, a.Psic, a.Mc, a.ucnt, jgroup, part, (long)a.n_pad)
#define PHI_DIAG(KG, PS) \
    do { if (PS) PHI_DIAG_R(KG, PS, RP); else if (small) PHI_DIAG_R(KG, PS, 1); else PHI_DIAG_R(KG, PS, R); } while (0)
    if (a.k == 1) { if (a.Psic) PHI_DIAG(false, true); else PHI_DIAG(false, false); }
    else { if (a.Psic) PHI_DIAG(true, true); else PHI_DIAG(true, false); }
#undef PHI_DIAG
#undef PHI_DIAG_R
    if (part)
        hipLaunchKernelGGL(k_phi_finalize, dim3((unsigned)((a.n_pad + 255) / 256)), dim3(256), 0, st, (const double *)part,
                           (long)a.n_pad, ngroup, a.ldx, (long)a.n_pad, a.n, a.k, a.b, a.omega, a.om_ld, a.lnbeta, a.wbeta, a.phiw);
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
bool phi_is_wide(int de, int k) { return de > 20 || k > 8; }   // no instantiated kernel: runtime-d route (k_wide.hip)
int phi_cov_rows_per_wg(int de, int k) {
    if (phi_is_wide(de, k)) return phi_wide_rows_per_wg();
    switch (de) {
        case 12: return 256 * PhiCovShape<12>::R;
        case 16: return 256 * PhiCovShape<16>::R;
        case 20: return 256 * PhiCovShape<20>::R;
        default: return 256 * PhiCovShape<10>::R;    // every width up to 10 takes the 4-row shape
    }
}

int launch_phi(hipStream_t st, const PhiArgs &a) {
    if (phi_is_wide(a.d, a.k)) return launch_phi_wide(st, a);
    return (a.kind == GPZ_KIND_DIAG) ? launch_phi_k<GPZ_KIND_DIAG>(st, a) : launch_phi_k<GPZ_KIND_COV>(st, a);
}
