// m x m stage of the evaluation: SIGMA = PHI'W PHI + diag(alpha), its inverse and log-determinant
// (GPz.m:65-67, inv_logdet.m), then w, dwda and diag(inv) (GPz.m:70-73).
//
// The reference uses an SVD pseudo-inverse.  SIGMA is symmetric positive definite by construction,
// so the device path is a right-looking blocked Cholesky (32-wide panels), a recursive blocked
// triangular inverse and inv(SIGMA) = inv(L)' * inv(L) on the f64 MFMA SYRK kernel; logdet =
// 2*sum(log(diag(L))).  The matrix is padded with an identity block to a multiple of 32 so every
// panel is full.  A non-positive pivot is reported through *info (results are NaN then); the
// rank-truncating branch of inv_logdet.m:7-12 is not reproduced ("deviations").
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define CH_NB GPZ_CH_NB

__global__ void k_build_sigma(const double *__restrict__ S, int lds, const double *__restrict__ alpha, int m, int mq,
                              double *__restrict__ A, int lda) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mq) return;
    double v;
    if (i < m && j < m) {
        v = S[(size_t)i * lds + j];
        if (i == j) v += alpha[i];                                         // GPz.m:65
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    A[(size_t)i * lda + j] = v;
}

// Cholesky of a 32x32 block held one row per lane (lanes 0..31 of one wave), fully unrolled so every index is a
// compile-time register index.  Column c: the pivot and the multipliers l_cc',c travel between lanes through
// v_readlane (wave-uniform broadcasts), no LDS round trips and no barriers.  Returns the first bad pivot (1-based, 0 = ok).
__device__ __forceinline__ int chol32_rows(double (&a)[CH_NB], int lane) {
    int bad = 0;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        const double piv = __shfl(a[c], c, 64);                 // a_cc lives in lane c
        if (!(piv > 0.0) && bad == 0) bad = c + 1;
        const double invd = rsqrt(piv);                         // one reciprocal square root instead of sqrt + divide:
        const double d = piv * invd;                            // both sit on the 32-step serial chain of the panel
        const double l = a[c] * invd;                           // l_rc for this lane's row r (meaningful for r >= c)
        a[c] = (lane == c) ? d : l;
#pragma unroll
        for (int cc = c + 1; cc < CH_NB; ++cc) {
            const double lcc = __shfl(l, cc, 64);               // l_cc,c
            a[cc] = fma(-l, lcc, a[cc]);                        // a_r,cc -= l_rc * l_cc,c   (used for r >= cc)
        }
    }
    return bad;
}

// Factor the 32x32 diagonal block at k0 (wave 0 of every workgroup does it redundantly in registers and publishes
// it through LDS; workgroup 0 writes it back) and solve the panel rows below it:  L21 = A21 * inv(L11)'.
// Reads A, writes the factor to Lm (a separate buffer: the redundant per-workgroup factorisation must
// never observe another workgroup's write-back).
__global__ __launch_bounds__(256) void k_chol_panel(const double *__restrict__ A, double *__restrict__ Lm, int lda,
                                                     int mq, int k0, double *__restrict__ logdet,
                                                     int *__restrict__ info) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double Dinv[CH_NB];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int lane = tid;
        const int r = lane & (CH_NB - 1);   // CH_NB = 32: lanes 32..63 duplicate rows 0..31; CH_NB = 64: one row per lane
        double a[CH_NB];
        const double *ar = A + (size_t)(k0 + r) * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) a[c] = ar[c];
        const int bad = chol32_rows(a, lane);
        if (lane < CH_NB) {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) D[lane][c] = (c <= lane) ? a[c] : 0.0;
            Dinv[lane] = 1.0 / D[lane][lane];                   // the row solves multiply instead of dividing
            if (blockIdx.x == 0) {
                double *lr = Lm + (size_t)(k0 + lane) * lda + k0;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) lr[c] = (c <= lane) ? a[c] : 0.0;
            }
        }
        if (blockIdx.x == 0) {
            // log-determinant contribution: 2*sum(log(diag))   (inv_logdet.m:15, sum of logs)
            double ld = 0.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                const double dcc = __shfl(a[c], c, 64);
                ld += log(dcc);
            }
            if (lane == 0) {
                *logdet += 2.0 * ld;
                if (bad && *info == 0) *info = k0 + bad;
            }
        }
    }
    __syncthreads();
    const int row = k0 + CH_NB + blockIdx.x * 256 + tid;
    if (row < mq) {
        double x[CH_NB];
        const double *ar = A + (size_t)row * lda + k0;
        double *lr = Lm + (size_t)row * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = ar[c];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) {
            double s = x[c];
#pragma unroll
            for (int q = 0; q < c; ++q) s = fma(-x[q], D[c][q], s);
            x[c] = s * Dinv[c];
        }
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) lr[c] = x[c];
    }
}

// W(diag block) = inv(L(diag block)) for every 32x32 diagonal block; one 64-thread workgroup each.
__global__ __launch_bounds__(64) void k_trtri_diag(const double *__restrict__ L, double *__restrict__ W, int ld) {
    __shared__ double Ls[CH_NB][CH_NB + 1];
    __shared__ double Ws[CH_NB][CH_NB + 1];
    const int k0 = blockIdx.x * CH_NB, tid = threadIdx.x;
    for (int e = tid; e < CH_NB * CH_NB; e += 64) {
        const int r = e / CH_NB, c = e % CH_NB;
        Ls[r][c] = L[(size_t)(k0 + r) * ld + k0 + c];
        Ws[r][c] = 0.0;
    }
    __syncthreads();
    if (tid < CH_NB) {
        const int c = tid;
        Ws[c][c] = 1.0 / Ls[c][c];
        for (int r = c + 1; r < CH_NB; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(Ls[r][q], Ws[q][c], s);
            Ws[r][c] = -s / Ls[r][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < CH_NB * CH_NB; e += 64) {
        const int r = e / CH_NB, c = e % CH_NB;
        W[(size_t)(k0 + r) * ld + k0 + c] = Ws[r][c];
    }
}

__global__ void k_zero(double *__restrict__ p, size_t count) {
    const size_t gs = (size_t)blockDim.x * gridDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gs) p[e] = 0.0;
}

// y[i] = scale * sum_j M[i][j] * x[j*xs] * (a ? a[j] : 1), one wave per row.
__global__ __launch_bounds__(256) void k_gemv(const double *__restrict__ M, int ld, int m, const double *__restrict__ x,
                                               long xs, const double *__restrict__ a, double scale,
                                               double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    double s = 0.0;
    for (int j = lane; j < m; j += 64) {
        double xv = x[(size_t)j * xs];
        if (a) xv *= a[j];
        s = fma(M[(size_t)row * ld + j], xv, s);
    }
    s = wave_sum(s);
    if (lane == 0) y[row] = scale * s;
}

// Bext (mp x mp row-major) = [ inv(SIGMA) | w in column m+out | 0 ];  dgi = diag(inv(SIGMA)).
__global__ void k_fill_bext(const double *__restrict__ Sinv, int ldsi, const double *__restrict__ w, int m, int mp,
                            int out, double *__restrict__ Bext, double *__restrict__ dgi) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    double v = 0.0;
    if (i < m) {
        if (j < m) v = Sinv[(size_t)i * ldsi + j];
        else if (j == m + out) v = w[i];
    }
    Bext[(size_t)i * mp + j] = v;
    if (i == j && i < m) dgi[i] = v;
}

void launch_build_sigma(hipStream_t st, const double *S, int lds, const double *alpha, int m, int mq, double *A, int lda) {
    hipLaunchKernelGGL(k_build_sigma, dim3((mq + 255) / 256, mq), dim3(256), 0, st, S, lds, alpha, m, mq, A, lda);
}

void launch_chol_panel(hipStream_t st, const double *A, double *Lm, int lda, int mq, int k0, double *logdet, int *info) {
    const int rows = mq - k0 - CH_NB;
    const int nwg = rows > 0 ? (rows + 255) / 256 : 1;
    hipLaunchKernelGGL(k_chol_panel, dim3(nwg), dim3(256), 0, st, A, Lm, lda, mq, k0, logdet, info);
}

void launch_trtri_diag(hipStream_t st, const double *L, double *W, int ld, int mq) {
    hipLaunchKernelGGL(k_trtri_diag, dim3(mq / CH_NB), dim3(64), 0, st, L, W, ld);
}

void launch_zero(hipStream_t st, double *p, size_t count) {
    if (count == 0) return;
    size_t nb = (count + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)nb), dim3(256), 0, st, p, count);
}

void launch_fill_bext(hipStream_t st, const double *Sinv, int ldsi, const double *w, int m, int mp, int out,
                      double *Bext, double *dgi) {
    hipLaunchKernelGGL(k_fill_bext, dim3((mp + 255) / 256, mp), dim3(256), 0, st, Sinv, ldsi, w, m, mp, out, Bext, dgi);
}

void launch_post_inverse(hipStream_t st, const double *Sinv, int ldsi, const double *S, int lds, const double *alpha,
                         int m, int mp, int out, double *Bext, double *w, double *dwda, double *dgi, int *info,
                         double *logdet) {
    (void)info; (void)logdet;
    const int nwg = (m + 3) / 4;
    // w = inv(SIGMA) * (PHI' (omega beta y))   (GPz.m:70); the right-hand side is column m+out of S
    hipLaunchKernelGGL(k_gemv, dim3(nwg), dim3(256), 0, st, Sinv, ldsi, m, S + m + out, (long)lds, (const double *)nullptr,
                       1.0, w);
    // dwda = -inv(SIGMA) * (alpha .* w)          (GPz.m:71)
    hipLaunchKernelGGL(k_gemv, dim3(nwg), dim3(256), 0, st, Sinv, ldsi, m, (const double *)w, 1L, alpha, -1.0, dwda);
    hipLaunchKernelGGL(k_fill_bext, dim3((mp + 255) / 256, mp), dim3(256), 0, st, Sinv, ldsi, (const double *)w, m, mp,
                       out, Bext, dgi);
}
