// Rank-truncating pseudo-inverse of SIGMA — the branch of inv_logdet.m that the Cholesky route cannot mimic.
//
// inv_logdet.m:3-15 takes an SVD, drops the singular values s <= m*eps(max s) and returns V*diag(1/s)*U' and
// sum(log(s)) over the kept ones.  While SIGMA is comfortably positive definite nothing is dropped and the
// Cholesky inverse of k_chol.hip is the same matrix to cond*eps.  When SIGMA is numerically singular (duplicate
// basis functions with vanishing alpha, a failed pivot) the reference silently truncates; this file reproduces
// that on the device with a one-sided (Hestenes) Jacobi SVD:
//     G <- SIGMA, V <- I;  rotate column pairs of G (and of V) until all columns of G are orthogonal;
//     then G = U*diag(s), s_j = ||g_j||, and  pinv = sum_{s_j > tol} v_j g_j' / s_j^2.
// Columns are stored as rows (SIGMA is symmetric, so G starts as a plain copy).  One launch per round of a
// round-robin tournament (m/2 independent pairs, one workgroup per pair); the host reads one convergence word
// per sweep.  Used only when k_cond_flag asks for it, so speed is secondary to fidelity.
#include <math.h>
#include <string.h>

#include "gpz_dev.h"
#include "gpz_kernels.h"

__device__ __forceinline__ double eps_of(double x) {   // MATLAB eps(x) for finite x > 0: the spacing of doubles at x
    return ldexp(1.0, ilogb(x) - 52);
}

// flag[0] (info[1]) |= 1 when the reference might truncate: 1/||inv||_F <= 4 m eps(||SIGMA||_F), or the Cholesky failed.
// (s_min >= 1/||inv||_F and tol <= m eps(||SIGMA||_F): outside that region inv_logdet.m keeps every singular value.)
// A non-finite SIGMA is left alone: svd() raises in the reference and the evaluation returns NaN here.
#define COND_NWG 256   // workgroups at m >= 512 (64 below: the ticket costs more than the rows); 64 at m = 1000 read their rows for 25 us
__global__ __launch_bounds__(256) void k_cond_norms(const double *__restrict__ S, int lds, const double *__restrict__ alpha,
                                                    const double *__restrict__ Sinv, int ldsi, int m,
                                                    double *part, int *info, unsigned *ticket) {
    __shared__ double sh4[4];
    double a = 0.0, b = 0.0;
    const unsigned nwg = gridDim.x;                              // 64 or COND_NWG
    for (int i = blockIdx.x; i < m; i += (int)nwg) {
        const double *sr = S + (size_t)i * lds, *ir = Sinv + (size_t)i * ldsi;
#pragma unroll 4
        for (int j = threadIdx.x; j < m; j += 256) {
            double v = sr[j];
            if (i == j) v += alpha[i];
            a = fma(v, v, a);
            b = fma(ir[j], ir[j], b);
        }
    }
    a = block_sum_256(a, sh4);
    __syncthreads();
    b = block_sum_256(b, sh4);
    // The workgroup that takes the last ticket adds the partials in their fixed order and sets the flag (one launch
    // instead of two: an evaluation of a small problem is a sequence of ~4 us launches); it puts the ticket counter back to zero.
    __shared__ int last;
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
        __threadfence();
        last = (atomicAdd(ticket, 1u) == nwg - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();
    a = 0.0; b = 0.0;
    for (unsigned q = 0; q < nwg / 64; ++q) {
        a += part[2 * (threadIdx.x + 64 * q)];
        b += part[2 * (threadIdx.x + 64 * q) + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        *ticket = 0u;
        const bool finite_s = (a == a) && (a < 1.0e300 * 1.0e300) && a > 0.0;
        if (finite_s) {
            const double ns = sqrt(a), ni = sqrt(b);
            const bool certified = (info[0] == 0) && (ni == ni) && (1.0 / ni > 4.0 * (double)m * eps_of(ns));
            if (!certified) info[1] |= 1;
        }
    }
}

// Gt (m rows of length m, ld) <- SIGMA = S + diag(alpha);  Vt <- I.
__global__ void k_jacobi_init(const double *__restrict__ S, int lds, const double *__restrict__ alpha, int m,
                              double *__restrict__ Gt, double *__restrict__ Vt, int ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= m) return;
    double v = S[(size_t)i * lds + j];
    if (alpha && i == j) v += alpha[i];
    Gt[(size_t)i * ld + j] = v;
    Vt[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// One tournament round: workgroup i rotates the pair it is assigned in round r (M players, M even, player >= m is a bye).
__global__ __launch_bounds__(256) void k_jacobi_round(double *__restrict__ Gt, double *__restrict__ Vt, int ld, int m,
                                                      int M, int r, double floor_norm,
                                                      unsigned long long *__restrict__ offmax) {
    __shared__ double sh4[4];
    __shared__ double rot[2];
    const int i = blockIdx.x, tid = threadIdx.x;
    int p, q;
    if (i == 0) { p = M - 1; q = r; }
    else { p = (r + i) % (M - 1); q = (r - i + (M - 1)) % (M - 1); }
    if (p > q) { const int t = p; p = q; q = t; }
    if (q >= m) return;
    double *gp = Gt + (size_t)p * ld, *gq = Gt + (size_t)q * ld;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int e = tid; e < m; e += 256) {
        const double x = gp[e], y = gq[e];
        a = fma(x, x, a); b = fma(y, y, b); c = fma(x, y, c);
    }
    a = block_sum_256(a, sh4); __syncthreads();
    b = block_sum_256(b, sh4); __syncthreads();
    c = block_sum_256(c, sh4);
    if (tid == 0) {
        double cs = 1.0, sn = 0.0;
        const double na = sqrt(a), nb = sqrt(b), den = na * nb;
        atomicMax(offmax + 1, (unsigned long long)__double_as_longlong(fmax(na, nb)));
        if (den > 0.0) {
            const double rel = fabs(c) / den;
            if (rel > 2.220446049250313e-16) {
                // columns at the noise floor (below the truncation threshold) are re-randomised by every rotation with
                // a large column: they are still rotated but do not hold up convergence — they are dropped anyway
                if (fmin(na, nb) > floor_norm) atomicMax(offmax, (unsigned long long)__double_as_longlong(rel));
                const double zeta = (b - a) / (2.0 * c);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                cs = 1.0 / sqrt(1.0 + t * t);
                sn = cs * t;
            }
        }
        rot[0] = cs; rot[1] = sn;
    }
    __syncthreads();
    const double cs = rot[0], sn = rot[1];
    if (sn == 0.0) return;
    double *vp = Vt + (size_t)p * ld, *vq = Vt + (size_t)q * ld;
    for (int e = tid; e < m; e += 256) {
        const double x = gp[e], y = gq[e];
        gp[e] = cs * x - sn * y;
        gq[e] = sn * x + cs * y;
        const double u = vp[e], w = vq[e];
        vp[e] = cs * u - sn * w;
        vq[e] = sn * u + cs * w;
    }
}

// s_j = ||g_j||  (one workgroup per column)
__global__ __launch_bounds__(256) void k_jacobi_norms(const double *__restrict__ Gt, int ld, int m, double *__restrict__ s) {
    __shared__ double sh4[4];
    const double *g = Gt + (size_t)blockIdx.x * ld;
    double a = 0.0;
    for (int e = threadIdx.x; e < m; e += 256) a = fma(g[e], g[e], a);
    a = block_sum_256(a, sh4);
    if (threadIdx.x == 0) s[blockIdx.x] = sqrt(a);
}

// tol = m*eps(max s); scale_j = 1/s_j^2 for the kept values, 0 for the dropped ones; logdet = sum ln s over the kept
// (inv_logdet.m:7-15).  out3 = [logdet, rank, max s].
__global__ __launch_bounds__(256) void k_jacobi_truncate(double *__restrict__ s, int m, double *__restrict__ logdet,
                                                          double *__restrict__ out3) {
    __shared__ double sh4[4];
    __shared__ double smax_s;
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int j = tid; j < m; j += 256) mx = fmax(mx, s[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) sh4[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) smax_s = fmax(fmax(sh4[0], sh4[1]), fmax(sh4[2], sh4[3]));
    __syncthreads();
    const double smax = smax_s;
    const double tol = (smax > 0.0) ? (double)m * eps_of(smax) : 0.0;
    double ld = 0.0, rk = 0.0;
    for (int j = tid; j < m; j += 256) {
        const double sj = s[j];
        if (sj > tol) { ld += log(sj); rk += 1.0; s[j] = 1.0 / (sj * sj); }
        else s[j] = 0.0;
    }
    ld = block_sum_256(ld, sh4); __syncthreads();
    rk = block_sum_256(rk, sh4);
    if (tid == 0) {
        *logdet = ld;
        if (out3) { out3[0] = ld; out3[1] = rk; out3[2] = smax; }
    }
}

// Xi[a][b] = sum_j Vt[j][a] * scale[j] * Gt[j][b]   (= V*diag(1/s)*U', inv_logdet.m:14); 64x64 tile per workgroup.
__global__ __launch_bounds__(256) void k_jacobi_pinv(const double *__restrict__ Vt, const double *__restrict__ Gt, int ld,
                                                      const double *__restrict__ scale, int m, double *__restrict__ Xi,
                                                      int ldx) {
    __shared__ double As[16][64 + 1], Bs[16][64 + 1];
    const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64, tid = threadIdx.x;
    const int ta = (tid >> 4) * 4, tb = (tid & 15) * 4;
    double acc[4][4] = {};
    for (int j0 = 0; j0 < m; j0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int jj = e >> 6, cc = e & 63, j = j0 + jj;
            const bool ok = j < m;
            As[jj][cc] = (ok && a0 + cc < m) ? Vt[(size_t)j * ld + a0 + cc] * scale[j] : 0.0;
            Bs[jj][cc] = (ok && b0 + cc < m) ? Gt[(size_t)j * ld + b0 + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = As[jj][ta + u]; bv[u] = Bs[jj][tb + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(av[u], bv[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (a0 + ta + u < m && b0 + tb + v < m) Xi[(size_t)(a0 + ta + u) * ldx + b0 + tb + v] = acc[u][v];
}

void launch_cond_flag(hipStream_t st, const double *S, int lds, const double *alpha, const double *Sinv, int ldsi, int m,
                      double *part, int *info) {
    // info: [pivot failure, truncation flag, ticket counter of this kernel (zero between launches), -]
    hipLaunchKernelGGL(k_cond_norms, dim3(m >= 512 ? COND_NWG : 64), dim3(256), 0, st, S, lds, alpha, Sinv, ldsi, m, part, info, (unsigned *)(info + 2));
}

// Pseudo-inverse of SIGMA = S + diag(alpha) (alpha may be nullptr) into Xi, ln-det of the kept part into *logdet.
// Gt, Vt: m x ld work matrices; sbuf: m doubles; word: two 8-byte device words; out3 (optional, device): [logdet, rank, max s].
// Synchronises the stream once per sweep.  Returns the number of sweeps, or -1 on a HIP error.
int run_jacobi_pinv(hipStream_t st, const double *S, int lds, const double *alpha, int m, double *Gt, double *Vt, int ld,
                    double *sbuf, unsigned long long *word, double *Xi, int ldx, double *logdet, double *out3) {
    hipLaunchKernelGGL(k_jacobi_init, dim3((m + 255) / 256, m), dim3(256), 0, st, S, lds, alpha, m, Gt, Vt, ld);
    const int M = (m + 1) & ~1;
    const double conv = 4.0 * sqrt((double)m) * 2.220446049250313e-16;
    int sweep = 0;
    double floor_norm = 0.0;
    if (m > 1) {
        for (sweep = 1; sweep <= 60; ++sweep) {
            if (hipMemsetAsync(word, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return -1;
            for (int r = 0; r < M - 1; ++r)
                hipLaunchKernelGGL(k_jacobi_round, dim3(M / 2), dim3(256), 0, st, Gt, Vt, ld, m, M, r, floor_norm, word);
            unsigned long long h[2] = {0, 0};
            if (hipMemcpyAsync(h, word, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
            if (hipStreamSynchronize(st) != hipSuccess) return -1;
            double rel, mx;
            memcpy(&rel, &h[0], sizeof rel);
            memcpy(&mx, &h[1], sizeof mx);
            if (mx > 0.0 && mx < 1e300) {   // m*eps(largest column norm): the truncation threshold of inv_logdet.m:7
                int ex;
                (void)frexp(mx, &ex);
                floor_norm = (double)m * ldexp(1.0, ex - 1 - 52);
            }
            if (!(rel > conv)) break;
        }
    }
    hipLaunchKernelGGL(k_jacobi_norms, dim3(m), dim3(256), 0, st, Gt, ld, m, sbuf);
    hipLaunchKernelGGL(k_jacobi_truncate, dim3(1), dim3(256), 0, st, sbuf, m, logdet, out3);
    hipLaunchKernelGGL(k_jacobi_pinv, dim3((m + 63) / 64, (m + 63) / 64), dim3(256), 0, st, Vt, Gt, ld, sbuf, m, Xi, ldx);
    return sweep > 60 ? 60 : sweep;
}
