// m x m stage of the evaluation: SIGMA = PHI'W PHI + diag(alpha), its inverse and log-determinant
// (GPz.m:65-67, inv_logdet.m), then w, dwda and diag(inv) (GPz.m:70-73).
//
// The reference uses an SVD pseudo-inverse.  SIGMA is symmetric positive definite by construction,
// so the device path is a right-looking blocked Cholesky (32-wide panels), a recursive blocked
// triangular inverse and inv(SIGMA) = inv(L)' * inv(L) on the f64 MFMA SYRK kernel; logdet =
// 2*sum(log(diag(L))).  The matrix is padded with an identity block to a multiple of 32 so every
// panel is full.  A non-positive pivot is reported through *info (results are NaN then); the
// rank-truncating branch of inv_logdet.m:7-12 is k_pinv.hip (taken when the certificate of DESIGN.md section 5 fails).
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define CH_NB GPZ_CH_NB
#ifdef GPZ_CHOL_TRACE   // developer builds only (tools/chol_trace.hip): s_memtime stamps of the phases of one factorisation step
__device__ unsigned long long *g_chol_trace = nullptr;
#define CH_MARK(slot)                                                                                                     \
    do {                                                                                                                  \
        if (g_chol_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 4)                                                    \
            g_chol_trace[(((size_t)(k0 / CH_NB) * 4 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CH_MARK(slot) do { } while (0)
#endif

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

// Wave-uniform broadcast of one lane's double: two v_readlane_b32 into a scalar pair (the lane index is a compile-time constant in
// the unrolled loops below).  __shfl compiles to ds_bpermute - an LDS round trip per value on the serial chain of the panel.
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

#define CH_POST 8   // columns per posting of the diagonal block's factor (posting the last group column by column was measured: slower)
// Cholesky of a 32x32 block held one row per lane (lanes 0..31 of one wave; lanes 32..63 carry copies), fully unrolled so every
// index is a compile-time register index.  Column c: the pivot and the multipliers l_cc',c travel between lanes through v_readlane
// (wave-uniform broadcasts), no LDS round trips and no barriers.  Finished columns are POSTED to LDS (D[.][c], Dinv[c], then
// *posted = c + 1 after every CH_POST columns) so the waves that solve the panel rows follow one group behind instead of waiting
// for the whole block (LDS operations of one wave complete in order: a wave that sees the counter sees the columns).
// A lone wave issues an instruction every ~7 cycles (a v_fma_f64 every ~12.6), so the block's time is its instruction count.  Only the
// updates inside a group of 8 columns go lane-to-lane; a finished group updates the columns behind it as ONE small product on the f64
// MFMA with both operands read back from the posted columns (after column 7: columns 8..15 of every row, K = 8; after column 15:
// the trailing 16 x 16 block, K = 16; after column 23: columns 24..31, K = 8) and an LDS transposition of the result
// (tools/chol_trace.hip, profiles/r05_chol_step_timeline.txt: 22 600 cycles with ds_bpermute shuffles -> 16 100 with v_readlane ->
// 12 000 with the K = 16 product -> 8 000).  The pivot check is one ballot at the end (*badpiv = first bad pivot, 1-based, 0 = none):
// a non-positive pivot leaves NaN on that diagonal entry.
__device__ __forceinline__ void chol32_rows(double (&a)[CH_NB], int lane, double (*D)[CH_NB + 1], double *Dinv, int *posted, int *badpiv,
                                           double (*Ct)[CH_NB / 2 + 1]) {
    constexpr int H = CH_NB / 2, G = CH_POST;
    static_assert(CH_NB == 32 && CH_POST == 8, "the deferred products below are written for 32 = 4 x 8");
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        const double piv = bcast_lane(a[c], c);                 // a_cc lives in lane c
        // 1/sqrt(piv): v_rsq_f64 and one third-order correction (the library's sequence without its special-case selects);
        // one reciprocal square root instead of sqrt + divide: both sit on the 32-step serial chain of the panel
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double e = fma(y0 * -piv, y0, 1.0);
        const double invd = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        const double d = piv * invd;
        const double l = a[c] * invd;                           // l_rc for this lane's row r (meaningful for r >= c)
        a[c] = (lane == c) ? d : l;
        D[lane][c] = a[c];                                      // rows < c: written (every lane stores), with the row's stale a_rc times 1/d - finite
                                                                // for finite input; the solves never read them (D[q][c], q > c), the MFMA products
                                                                // below read them into outputs nobody uses
        // 1/d for the row solves and the inverse's diagonal: invd = 1/sqrt(piv) is up to ~5 ulp away from the reciprocal of the ROUNDED
        // d (d carries its own rounding and twice invd's); one Newton step, off the chain.  Wave-uniform, every lane stores it.
        Dinv[c] = fma(fma(-d, invd, 1.0), invd, invd);
#pragma unroll
        for (int cc = c + 1; cc < (c / G + 1) * G; ++cc) {
            const double lcc = bcast_lane(l, cc);               // l_cc,c
            a[cc] = fma(-l, lcc, a[cc]);                        // a_r,cc -= l_rc * l_cc,c   (used for r >= cc)
        }
        if (c == G - 1) {
            // columns 8..15 of all rows: C = D[:, 0..7] * D[0..15, 0..7]'  (two row blocks; columns 0..7 of C are not used)
            d4_t acc0 = d4_t{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
            for (int kb = 0; kb < G / 4; ++kb) {
                const double v0 = D[li][4 * kb + lk], v1 = D[H + li][4 * kb + lk];
                acc0 = MFMA_F64(v0, v0, acc0);
                acc1 = MFMA_F64(v1, v0, acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { Ct[lk + 4 * r][li] = acc0[r]; Ct[H + lk + 4 * r][li] = acc1[r]; }
#pragma unroll
            for (int j = G; j < H; ++j) a[j] -= Ct[lane & (CH_NB - 1)][j];
        }
        if (c == H - 1) {
            // rows 16..31, columns 16..31: A22 -= L21 L21' with L21 = D[16.., 0..15]; both MFMA operands are the same registers
            d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < H / 4; ++kb) {
                const double v = D[H + li][4 * kb + lk];
                acc = MFMA_F64(v, v, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[lk + 4 * r][li] = acc[r];
#pragma unroll
            for (int j = 0; j < H; ++j) a[H + j] -= Ct[li][j];               // row 16 + i lives in lane 16 + i
        }
        if (c == H + G - 1) {
            // rows 16..31, columns 24..31, K = columns 16..23
            d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < G / 4; ++kb) {
                const double v = D[H + li][H + 4 * kb + lk];
                acc = MFMA_F64(v, v, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[lk + 4 * r][li] = acc[r];
#pragma unroll
            for (int j = G; j < H; ++j) a[H + j] -= Ct[li][j];
        }
        if ((c + 1) % CH_POST == 0) {
            if (c + 1 == CH_NB) {                               // before the last posting: the wave that reports it waits for that
                double dg = 1.0;
#pragma unroll
                for (int q = 0; q < CH_NB; ++q) dg = (lane == q) ? a[q] : dg;
                const unsigned long long nb = __ballot(lane < CH_NB && !(dg > 0.0));
                *badpiv = nb ? __ffsll((long long)nb) : 0;      // first bad pivot, 1-based
            }
            // RELEASE: the plain LDS stores of D / Dinv / badpiv above are complete before the count is visible (an s_waitcnt lgkmcnt(0)
            // in front of the store; the hand-off used to rest on one wave's DS operations completing in order - ADVICE r05)
            __hip_atomic_store(posted, c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// Sum over lanes 0..31 of a wave through DPP (quad swaps, mirrors) and two v_readlane pairs: ~25 instructions without an LDS trip.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_lanes32(double v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror: every lane of a 16-lane row holds the row's sum
    return bcast_lane(v, 0) + bcast_lane(v, 16);
}

// Columns [C0, C0 + NC) of the row solve x * inv(L11)' for one row per lane, applied to the entries before QE (column-oriented:
// the updates of one column are independent of each other; a row-oriented dot product is one dependent chain per entry).
template <int C0, int NC, int QE>
__device__ __forceinline__ void solve_cols(double (&x)[CH_NB], const double (*D)[CH_NB + 1], const double *Dinv, const int *posted) {
    while (__hip_atomic_load(posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < C0 + NC) __builtin_amdgcn_s_sleep(1);   // ACQUIRE: pairs with the poster's release
#pragma unroll
    for (int c = C0; c < C0 + NC; ++c) {
        x[c] *= Dinv[c];
#pragma unroll
        for (int q = c + 1; q < QE; ++q) x[q] = fma(-x[c], D[q][c], x[q]);
    }
    // applied before the next posting is waited for (the optimiser otherwise sinks every multiply-add below the last wait and
    // parks the factor in scratch)
#pragma unroll
    for (int q = 0; q < CH_NB; ++q) asm volatile("" : "+v"(x[q]));
}

// The 64 rows of one wave against the posted factor.  Columns 0..15 and 16..31 are solved lane-wise; between them the update
// X2 -= X1 * L21' (half of the multiply-adds) is one product on the f64 MFMA: X1 goes to its final place in Xw early, L21 is read from
// the posted columns, the result passes through the still unused half of Xw on its way back to one row per lane.
__device__ __forceinline__ void solve_rows(double (&x)[CH_NB], int lane, const double (*D)[CH_NB + 1], const double *Dinv, const int *posted,
                                           double (*Xw)[CH_NB + 1]) {
    constexpr int H = CH_NB / 2;
    static_assert(CH_NB == 4 * CH_POST, "four postings per block");
    const int li = lane & 15, lk = lane >> 4;
    solve_cols<0, CH_POST, H>(x, D, Dinv, posted);
    solve_cols<CH_POST, CH_POST, H>(x, D, Dinv, posted);
#pragma unroll
    for (int c = 0; c < H; ++c) Xw[lane][c] = x[c];
    d4_t acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < H / 4; ++kb) {
        const double bv = D[H + li][4 * kb + lk];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA_F64(Xw[rt * 16 + li][4 * kb + lk], bv, acc[rt]);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[rt * 16 + lk + 4 * r][H + li] = acc[rt][r];
#pragma unroll
    for (int j = 0; j < H; ++j) x[H + j] -= Xw[lane][H + j];
    solve_cols<2 * CH_POST, CH_POST, CH_NB>(x, D, Dinv, posted);
    solve_cols<3 * CH_POST, CH_POST, CH_NB>(x, D, Dinv, posted);
#pragma unroll
    for (int c = H; c < CH_NB; ++c) Xw[lane][c] = x[c];
}

// Block row k = k0 / 32 of the triangular inverse W = inv(L), 16 columns of block j < k, by a workgroup of its own inside the step's launch (mq <= 256):
//   W_kj = - W_kk * S_j,   S_j = sum_{i = j .. k-1} L_ki * W_ij        (row by row: L W = I)
// L_ki are the panel rows earlier steps wrote to Lm, W_ij the block rows earlier steps wrote to Wd; W_kk = inv(L_kk) does not exist
// before THIS step, so the workgroup repeats what every tile of the step does - wave 0 factors the diagonal block, wave 2 solves the
// unit rows against the posted columns (the winv path of tile (0, 0)) - while waves 1 and 3 form S_j on the f64 MFMA, one half of the K
// range each, straight from L2 (32 x K times K x 32, K = 32 (k - j): operands of a few KB, eight K steps of loads in flight).  The
// product with W_kk is 32 x 32 x 32 from LDS.  With it the k_trtri_level launches behind the factorisation (two per level, 6 at mq =
// 224, 37 us of c2's 590) are gone; a step's own time does not change (S_j is done before the factor's last column is posted).
__device__ __forceinline__ void chol_inverse_block(const double *__restrict__ A, const double *__restrict__ Lm, double *__restrict__ Wd, int lda,
                                                   int k0, int e, double (*D)[CH_NB + 1], double *Dinv, double (*Xs)[64][CH_NB + 1],
                                                   double (*Ct)[CH_NB / 2 + 1], int *posted, int *badpiv) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int j = e >> 1, ch = e & 1;                                      // block j of the row, columns 16 ch .. 16 ch + 15 of it
    double x[CH_NB];
    if (wave == 0) {
        const double *ar = A + (size_t)(k0 + (lane & (CH_NB - 1))) * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = ar[c];
    } else if (wave == 2) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = (lane == c) ? 1.0 : 0.0;
    }
    __syncthreads();                                                       // posted = 0 is visible
    if (wave == 0) {
        chol32_rows(x, lane, D, Dinv, posted, badpiv, Ct);
    } else if (wave == 2) {
        solve_rows(x, lane, D, Dinv, posted, Xs[1]);                       // Xs[1][c][r] = W_kk[r][c] (rows c >= 32: zeros)
    } else {
        // S_j's 16 columns: wave 1 the first half of the K steps, wave 3 the second; two 16 x 16 tiles (rows 0..15, 16..31) each - the
        // loop is bound by the wave's own MFMA issue (two products of 64 cycles per K step) and by the latency of its operand loads, which is why a block's two
        // column halves are two workgroups
        const int j0 = j * CH_NB, nst = (k0 - j0) >> 2, half = (nst + 1) >> 1;
        const int s0 = wave == 1 ? 0 : half, s1 = wave == 1 ? half : nst;
        const double *pa = Lm + (size_t)(k0 + li) * lda + j0 + lk;         // A[m][k] = L[k0 + m][j0 + k]
        const double *pb = Wd + (size_t)(j0 + lk) * lda + j0 + 16 * ch + li;   // B[k][n] = W[j0 + k][j0 + 16 ch + n]
        const size_t a16 = (size_t)16 * lda, b4 = (size_t)4 * lda;
        d4_t acc[2];
        acc[0] = d4_t{0.0, 0.0, 0.0, 0.0};
        acc[1] = acc[0];
#pragma unroll 8
        for (int s = s0; s < s1; ++s) {
            const double a0 = pa[4 * s], a1 = pa[a16 + 4 * s];
            const double b0 = pb[s * b4];
            acc[0] = MFMA_F64(a0, b0, acc[0]);
            acc[1] = MFMA_F64(a1, b0, acc[1]);
        }
        double (*Sp)[CH_NB + 1] = Xs[0] + (wave == 1 ? 0 : CH_NB);         // this wave's partial S_j: rows 0..31 / 32..63 of Xs[0]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) Sp[a * 16 + lk + 4 * r][li] = acc[a][r];
    }
    __syncthreads();
    // W_kj = - W_kk S_j: waves 0 and 1 the row tiles
    if (wave > 1) return;
    const int ta = wave;
    d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < CH_NB / 4; ++kk) {
        const double av = Xs[1][4 * kk + lk][ta * 16 + li];               // A[m][k] = W_kk[16 ta + m][k]
        const double bv = Xs[0][4 * kk + lk][li] + Xs[0][CH_NB + 4 * kk + lk][li];
        acc = MFMA_F64(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Wd[(size_t)(k0 + ta * 16 + lk + 4 * r) * lda + j * CH_NB + ch * 16 + li] = -acc[r];
}

// One whole step of the right-looking factorisation in a single launch (CH_NB == 32): every workgroup owns one
// 64x64 tile (tm >= tn) of the trailing matrix.  Wave 0 factors the diagonal block at k0 in registers (redundantly
// per workgroup: it must never observe another workgroup's write-back, hence the separate factor buffer Lm) and posts it column
// by column; waves 1 and 2 hold the panel rows of row blocks tm and tn in registers and apply each column as it is posted
// (the step is one serial chain - tools/chol_trace.hip: of 48 600 cycles per step the factorisation took 22 600 through LDS
// shuffles, the solve another 9 800 behind a barrier; now the solve ends one column after the factorisation); wave 3 of tile (0, 0)
// writes the factor's diagonal block and the log-determinant meanwhile.  The solved rows are parked in LDS, and all four
// waves apply the rank-32 update to the tile on the f64 MFMA.  The row solves are repeated by every tile of a block row/column
// (about half the tile's own flops) in exchange for half the launches of the panel + trailing pair: the step is
// bound by launch-to-launch latency, not by arithmetic.  Reads of this step touch only columns < k0 + 32 of A,
// writes only columns >= k0 + 32, so the update is safely in place.
__global__ __launch_bounds__(256) void k_chol_step(double *__restrict__ A, double *__restrict__ Lm, double *__restrict__ Wd, int lda,
                                                    int mq, int k0, double *__restrict__ logdet, int *__restrict__ info, int ntiles) {
    __shared__ double D[64][CH_NB + 1];                                    // rows 32..63: lanes without a row (stores without a branch)
    __shared__ double Dinv[CH_NB];
    __shared__ double Xs[2][64][CH_NB + 1];
    __shared__ double Ct[CH_NB][CH_NB / 2 + 1];
    __shared__ int posted, badpiv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int t0 = k0 + CH_NB;
    if (tid == 0) posted = 0;
    if ((int)blockIdx.x >= ntiles) {   // workgroups behind the step's tiles: block row k0 / 32 of inv(L) (launch_chol_step: full_inverse)
        chol_inverse_block(A, Lm, Wd, lda, k0, (int)blockIdx.x - ntiles, D, &Dinv[0], Xs, Ct, &posted, &badpiv);
        return;
    }
    // tile index -> (tm, tn), tm >= tn
    int tm = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= (int)blockIdx.x) ++tm;
    while (tm * (tm + 1) / 2 > (int)blockIdx.x) --tm;
    const int tn = (int)blockIdx.x - tm * (tm + 1) / 2;
    CH_MARK(0);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
    double x[CH_NB];
    int row = -1;
    const bool winv = Wd != nullptr && blockIdx.x == 0 && wave == 2;
    if (wave == 0) {
        const int r = lane & (CH_NB - 1);
        const double *ar = A + (size_t)(k0 + r) * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = ar[c];
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
    // this wave's part of the trailing tile: fetched behind the panel rows (loads return in order and the factorisation waits for
    // the rows only), its latency hides behind the factorisation
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
    __syncthreads();                                                       // posted = 0 is visible
    if (wave == 0) {
#ifdef GPZ_CHOL_TRACE
        __builtin_amdgcn_s_waitcnt(0);
#endif
        CH_MARK(1);
        chol32_rows(x, lane, D, Dinv, &posted, &badpiv, Ct);
        CH_MARK(2);
    } else if (wave <= 2) {
        CH_MARK(3);
        // Wave 2 of tile (0, 0) has no rows of its own (a diagonal tile's two row blocks coincide): it produces the diagonal block
        // of the triangular inverse instead.  Lane c starts from the unit row e_c, and the row solve leaves e_c inv(L11)' = column c
        // of inv(L11) in it (exact zeros above the diagonal) - the code path of the panel rows, at no cost in time.
        if (winv) {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) x[c] = (lane == c) ? 1.0 : 0.0;
        }
        solve_rows(x, lane, D, Dinv, &posted, Xs[wave - 1]);
        CH_MARK(4);
    } else {
        if (blockIdx.x == 0) {
            // wave 3 of tile (0, 0): the factor's diagonal block, posting by posting, and the log-determinant, beside the solves
            double *lr = Lm + (size_t)(k0 + (lane & (CH_NB - 1))) * lda + k0;
#pragma unroll
            for (int c0 = 0; c0 < CH_NB; c0 += CH_POST) {
                while (__hip_atomic_load(&posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < c0 + CH_POST) __builtin_amdgcn_s_sleep(2);
                if (lane < CH_NB) {
#pragma unroll
                    for (int c = c0; c < c0 + CH_POST; ++c) lr[c] = (c <= lane) ? D[lane][c] : 0.0;
                }
            }
            const double ld = sum_lanes32(lane < CH_NB ? log(D[lane & (CH_NB - 1)][lane & (CH_NB - 1)]) : 0.0);   // one logarithm per lane
            if (lane == 0) {
                *logdet += 2.0 * ld;                                           // inv_logdet.m:15
                const int bad = __hip_atomic_load(&badpiv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (bad && *info == 0) *info = k0 + bad;
            }
        }
    }
    __syncthreads();
    CH_MARK(5);
    if (t0 + tm * 64 >= mq) {                                              // last step: nothing below the diagonal block
        if (winv && lane < CH_NB) {
#pragma unroll
            for (int r = 0; r < CH_NB; ++r) Wd[(size_t)(k0 + r) * lda + k0 + lane] = x[r];
        }
        return;
    }
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
    CH_MARK(6);
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
    // the inverse's diagonal block and the panel's solved rows (rows of the factor) are issued last: nobody's input in this launch
    if (winv && lane < CH_NB) {
#pragma unroll
        for (int r = 0; r < CH_NB; ++r) Wd[(size_t)(k0 + r) * lda + k0 + lane] = x[r];
    }
    if (wave == 1 && tn == 0 && row >= 0) {
        double *lr = Lm + (size_t)row * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) lr[c] = x[c];
    }
    CH_MARK(7);
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

// full_inverse: the launch also leaves block row k0 / 32 of W = inv(L) (k0 / 32 more workgroups; every step in order yields all of W and
// no k_trtri_level launch is needed).  Otherwise only the diagonal block of that row.
// mq <= 256 (K <= 224).  Measured at mq = 512 (c3): S_j's loads come from other XCDs' writes of the previous launches (~1.5 us per eight
// K steps in flight), a step went 10.9 -> 15.2 us and the 74 us of k_trtri_level launches were only just paid back (2.196 -> 2.195 ms);
// at mq = 224 (c2) a step goes 10.8 -> 11.5 us and 37 us of launches disappear (0.593 -> 0.551 ms)
bool chol_full_inverse_fits(int mq) { return mq <= 256; }
void launch_chol_step(hipStream_t st, double *A, double *Lm, double *W, int lda, int mq, int k0, double *logdet, int *info,
                      bool full_inverse) {
    const int M = mq - k0 - CH_NB, nt = (M + 63) / 64, tiles = nt * (nt + 1) / 2, ntiles = tiles > 0 ? tiles : 1;
    const int extra = (full_inverse && W) ? 2 * (k0 / CH_NB) : 0;   // two workgroups (column halves) per block of the inverse's row
    hipLaunchKernelGGL(k_chol_step, dim3(ntiles + extra), dim3(256), 0, st, A, Lm, W, lda, mq, k0, logdet, info, ntiles);
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
