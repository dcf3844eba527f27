// m x m stage of the evaluation: SIGMA = PHI'W PHI + diag(alpha), its inverse and log-determinant
// (GPz.m:65-67, inv_logdet.m), then w, dwda and diag(inv) (GPz.m:70-73).
//
// The reference uses an SVD pseudo-inverse.  SIGMA is symmetric positive definite by construction,
// so the device path is a right-looking blocked Cholesky (32-wide panels), a recursive blocked
// triangular inverse and inv(SIGMA) = inv(L)' * inv(L) on the f64 MFMA SYRK kernel; logdet =
// 2*sum(log(diag(L))).  The matrix is padded with an identity block to a multiple of 32 so every
// panel is full.  A non-positive pivot is reported through *info (results are NaN then); the
// rank-truncating branch of inv_logdet.m:7-12 is not reproduced ("deviations").
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define CH_NB GPZ_CH_NB

// Also clears what the chain behind it accumulates into (two launches of k_zero less per output: an evaluation of a small problem is
// a sequence of ~4 us launches): Wz (mq x mq, the inverse factor's workspace) and *logdet, when given.
__global__ void k_build_sigma(const double *__restrict__ S, int lds, const double *__restrict__ alpha, int m, int mq,
                              double *__restrict__ A, int lda, double *__restrict__ Wz, double *__restrict__ logdet) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mq) return;
    if (Wz) Wz[(size_t)i * mq + j] = 0.0;
    if (logdet && i == 0 && j == 0) *logdet = 0.0;
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

// One whole step of the right-looking factorisation in a single launch (CH_NB == 32): every workgroup owns one
// 64x64 tile (tm >= tn) of the trailing matrix.  Wave 0 factors the diagonal block at k0 in registers (redundantly
// per workgroup: it must never observe another workgroup's write-back, hence the separate factor buffer Lm) while waves 1 and 2 already hold the panel rows of row blocks tm and tn in
// registers; after the barrier they solve their rows against L11, park the result in LDS, and all four waves apply
// the rank-32 update to the tile on the f64 MFMA.  The row solves are repeated by every tile of a block row/column
// (about half the tile's own flops) in exchange for half the launches of the panel + trailing pair: the step is
// bound by launch-to-launch latency, not by arithmetic.  Reads of this step touch only columns < k0 + 32 of A,
// writes only columns >= k0 + 32, so the update is safely in place.
__global__ __launch_bounds__(256) void k_chol_step(double *__restrict__ A, double *__restrict__ Lm, int lda, int mq,
                                                    int k0, double *__restrict__ logdet, int *__restrict__ info) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double Dinv[CH_NB];
    __shared__ double Xs[2][64][CH_NB + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int t0 = k0 + CH_NB;
    // tile index -> (tm, tn), tm >= tn
    int tm = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= (int)blockIdx.x) ++tm;
    while (tm * (tm + 1) / 2 > (int)blockIdx.x) --tm;
    const int tn = (int)blockIdx.x - tm * (tm + 1) / 2;
    // this wave's part of the trailing tile is fetched first, so its latency hides behind the factorisation
    const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
    double cold[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = t0 + tm * 64 + wr * 32 + a * 16 + lk + 4 * r;
                const int gc = t0 + tn * 64 + wc * 32 + b * 16 + li;
                cold[a][b][r] = (gr < mq && gc < mq) ? A[(size_t)gr * lda + gc] : 0.0;
            }
    double x[CH_NB];
    int row = -1;
    if (wave == 0) {
        const int r = lane & (CH_NB - 1);
        const double *ar = A + (size_t)(k0 + r) * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = ar[c];
        const int bad = chol32_rows(x, lane);
        if (lane < CH_NB) {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) D[lane][c] = (c <= lane) ? x[c] : 0.0;
            Dinv[lane] = 1.0 / D[lane][lane];
            if (blockIdx.x == 0) {
                double *lr = Lm + (size_t)(k0 + lane) * lda + k0;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) lr[c] = (c <= lane) ? x[c] : 0.0;
            }
        }
        if (blockIdx.x == 0) {
            // one logarithm per lane (lane c holds l_cc) and a wave sum, instead of 32 logarithms in sequence on the
            // critical path of the workgroup that also owns tile (0, 0)
            double dg = 1.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) dg = (lane == c) ? x[c] : dg;
            double ld = log(dg);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ld += __shfl_xor(ld, off, 64);
            if (lane == 0) {
                *logdet += 2.0 * ld;                                       // inv_logdet.m:15
                if (bad && *info == 0) *info = k0 + bad;
            }
        }
    } else if (wave <= 2) {
        const int blk = wave == 1 ? tm : tn;
        row = t0 + blk * 64 + lane;
        if (row < mq && !(wave == 2 && tm == tn)) {
            const double *ar = A + (size_t)row * lda + k0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) x[c] = ar[c];
        } else {
            row = -1;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) x[c] = 0.0;
        }
    }
    __syncthreads();
    if (wave == 1 || wave == 2) {
        if (row >= 0) {
            // column-oriented substitution: the updates of one column are independent of each other (a row-oriented
            // dot product is one dependent chain of up to 31 multiply-adds per entry)
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                x[c] *= Dinv[c];
#pragma unroll
                for (int q = c + 1; q < CH_NB; ++q) x[q] = fma(-x[c], D[q][c], x[q]);
            }
            if (tn == 0 && wave == 1) {
                double *lr = Lm + (size_t)row * lda + k0;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) lr[c] = x[c];
            }
        }
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) Xs[wave - 1][lane][c] = x[c];
    }
    __syncthreads();
    if (t0 + tm * 64 >= mq) return;                                        // last step: nothing below the diagonal block
    const double (*Xm)[CH_NB + 1] = Xs[0];
    const double (*Xn)[CH_NB + 1] = Xs[tm == tn ? 0 : 1];
    d4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < CH_NB / 4; ++kk) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = Xm[wr * 32 + a * 16 + li][4 * kk + lk];
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[b] = Xn[wc * 32 + b * 16 + li][4 * kk + lk];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = MFMA_F64(av[a], bv[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = t0 + tm * 64 + wr * 32 + a * 16 + lk + 4 * r;
                const int gc = t0 + tn * 64 + wc * 32 + b * 16 + li;
                if (gr < mq && gc < mq) A[(size_t)gr * lda + gc] = cold[a][b][r] - acc[a][b][r];
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
        // Column c of W in registers, every loop with compile-time bounds: entries above the diagonal are kept as zeros, so the sums run
        // over q < r for every lane (the same values and order as sums from q = c: the leading terms are exact zeros).  The version with
        // the column in LDS and lane-dependent bounds paid an LDS round trip per term: 20 us per launch, 2 us of arithmetic.
        const int c = tid;
        double w[CH_NB];
#pragma unroll
        for (int r = 0; r < CH_NB; ++r) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < r; ++q) s = fma(Ls[r][q], w[q], s);
            const double rd = 1.0 / Ls[r][r];                     // one division per row (wave-uniform)
            w[r] = (r == c) ? rd : (r > c ? -s * rd : 0.0);
        }
#pragma unroll
        for (int r = 0; r < CH_NB; ++r) Ws[r][c] = w[r];
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
                            int out, double *__restrict__ Bext, double *__restrict__ dgi, int round32) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    double v = 0.0;
    if (i < m) {
        if (j < m) v = Sinv[(size_t)i * ldsi + j];
        else if (j == m + out) v = w[i];
    }
    // round32: experiment switch (tools/f32_operand_experiment.py) - the B operand of the T-GEMM as an fp32-operand MFMA would see it
    Bext[(size_t)i * mp + j] = round32 ? (double)(float)v : v;
    if (i == j && i < m) dgi[i] = v;
}
// The same with dwda = -inv(SIGMA) (alpha .* w) (GPz.m:71) formed on the way: one wave per row reads the row of inv(SIGMA) once for
// both (the lane-strided sum and its wave reduction are k_gemv's, so dwda has k_gemv's bits) - one launch less per output.
__global__ __launch_bounds__(256) void k_fill_bext_dwda(const double *__restrict__ Sinv, int ldsi, const double *__restrict__ w,
                                                         const double *__restrict__ alpha, int m, int mp, int out,
                                                         double *__restrict__ Bext, double *__restrict__ dgi,
                                                         double *__restrict__ dwda, int round32) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= mp) return;
    double s = 0.0;
    for (int j = lane; j < mp; j += 64) {
        double v = 0.0;
        if (i < m) {
            if (j < m) {
                v = Sinv[(size_t)i * ldsi + j];
                s = fma(v, w[j] * alpha[j], s);
                if (i == j) dgi[i] = v;
            } else if (j == m + out) v = w[i];
        }
        Bext[(size_t)i * mp + j] = round32 ? (double)(float)v : v;
    }
    s = wave_sum(s);
    if (lane == 0 && i < m) dwda[i] = -1.0 * s;
}
static int bext_round32() { return gpz_opts().round_phi32 ? 1 : 0; }

void launch_build_sigma(hipStream_t st, const double *S, int lds, const double *alpha, int m, int mq, double *A, int lda, double *Wz,
                        double *logdet) {
    hipLaunchKernelGGL(k_build_sigma, dim3((mq + 255) / 256, mq), dim3(256), 0, st, S, lds, alpha, m, mq, A, lda, Wz, logdet);
}

void launch_chol_step(hipStream_t st, double *A, double *Lm, int lda, int mq, int k0, double *logdet, int *info) {
    const int M = mq - k0 - CH_NB, nt = (M + 63) / 64, tiles = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(k_chol_step, dim3(tiles > 0 ? tiles : 1), dim3(256), 0, st, A, Lm, lda, mq, k0, logdet, info);
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
    hipLaunchKernelGGL(k_fill_bext, dim3((mp + 255) / 256, mp), dim3(256), 0, st, Sinv, ldsi, w, m, mp, out, Bext, dgi, bext_round32());
}

void launch_post_inverse(hipStream_t st, const double *Sinv, int ldsi, const double *S, int lds, const double *alpha,
                         int m, int mp, int out, double *Bext, double *w, double *dwda, double *dgi, int *info,
                         double *logdet) {
    (void)info; (void)logdet;
    const int nwg = (m + 3) / 4;
    // w = inv(SIGMA) * (PHI' (omega beta y))   (GPz.m:70); the right-hand side is column m+out of S
    hipLaunchKernelGGL(k_gemv, dim3(nwg), dim3(256), 0, st, Sinv, ldsi, m, S + m + out, (long)lds, (const double *)nullptr,
                       1.0, w);
    // dwda = -inv(SIGMA) * (alpha .* w)   (GPz.m:71) and Bext = [inv(SIGMA) | w], diag: one launch
    hipLaunchKernelGGL(k_fill_bext_dwda, dim3((mp + 3) / 4), dim3(256), 0, st, Sinv, ldsi, (const double *)w, alpha, m, mp, out, Bext,
                       dgi, dwda, bext_round32());
}
