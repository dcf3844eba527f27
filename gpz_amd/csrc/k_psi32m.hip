// GC/VC with DIAGONAL input noise in fp32 (BASELINE config 5), the moment sums on the matrix pipe: SIXTEEN (sample, basis) pairs per
// wave on v_mfma_f32_4x4x1_16b_f32, one basis function per block of four lanes, the samples of a row chunk one after the other.
//
//   GPz.m:164-185    sum_i dPHI_ij * [1, M^-1 Delta, (M^-1 Delta)(M^-1 Delta)' - M^-1],   M = Psi_i + Sigma_j
//
// in the whitened form of k_psi32.hip (Gamma_j = Q R):  A = I + R Psi_i R',  z = R Delta,  u~ = A^-1 z,  and the sums
//     a~1 = sum_i dp u~,      C~ = sum_i dp (u~ u~' - A^-1)
// are what k_psi32_records / k_psi32_finish consume (same raw layout as k_psi32_moments).
//
// The instruction (measured, tools/mfma_f32_4x4_probe.hip): lane = 4 b + i supplies A[i] and B[i] of block b, register r of lane
// 4 b + c receives D[r][c] += A[r] B[c] - sixteen independent 4 x 4 outer products, K = 1.  A 4 x 4 TILE of a pair's matrix is one
// float4 per lane (register = row, lane = column); a VECTOR spread over the four lanes of the block is an operand as it stands.
// Everything below is arranged so that the vectors the elimination needs are rows of tiles it already holds (a row is one register
// across the four lanes), which takes the transposes out of the algorithm:
//
//   build     A = I + sum_k psi_k r_k r_k',  r_k = column k of R (upper triangular): tile (J, I), J <= I, accumulates
//             mfma(psi_k R[4J + ., k], R[4I + ., k]) for k >= 4I - R_j comes from LDS as "lane = row" registers, psi_ik is wave-uniform
//             (scalar loads): 4 ND(ND+1)(ND+2)/6 instructions (140 at d = 20), one v_mul per operand.
//   eliminate the bordered matrix [A | I] WITHOUT square roots, pivot by pivot (k = 4K + c): its row k is register c of the tiles
//             T(K, I), I >= K, and of the border tiles E(K, Jb), Jb <= K; with d_k the pivot and n = -row/d_k
//                 T(J, I)  += n_J (x) row_I      K <= J <= I          (the trailing matrix, upper tiles only)
//                 E(J, Jb) += n_J (x) e_Jb       J >= K, Jb <= K      (the border: its row k is row k of inv(unit lower factor))
//                 C(Jb,Ib) += (-dp/d_k) e_Jb (x) e_Ib   Jb <= Ib <= K (A^-1 = sum_k e_k' e_k / d_k, straight into the moment sums)
//             - 21 instructions per pivot at d = 20 whatever K is.  Rows of a tile that are already eliminated collect rounding
//             residue and are never read again.  z rides along on the vector pipe (forward substitution with the same n,
//             u~ = sum_k e_k y_k / d_k), 2 (ND + 1) multiply-adds per pivot with the pivot-lane value broadcast by DPP.
//   moments   C(J, I) += (dp u~_J) (x) u~_I, a~1 += dp u~: the accumulators live across the rows of the chunk (fp32, in the MFMA's own
//             accumulator registers) and are added to the fp64 slab record of (chunk, basis) every FLUSH rows - the same 64-row fp32
//             partial sums the wave reduction of k_psi32_moments forms.
//
// Per sample and sixteen pairs: 575 MFMA + about 500 vector instructions at d = 20, against 6000 vector instructions per 64 pairs of the
// lane-per-pair kernel; PHI and T are read once per (sample, basis) as full 128-byte lines, x and psi once per sample and wave.
// What it costs is priced in DESIGN.md section 8 (tools/mfma_f32_4x4_rate.hip): the 4x4x1 instruction issues every ~10 cycles and does
// NOT overlap with vector instructions of the same SIMD.
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

typedef float f4 __attribute__((ext_vector_type(4)));
#define MF4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int C>
__device__ __forceinline__ float qb(float v) {   // lane C of every block of four lanes, to all four
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), C * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float qbc(float v, int c) {   // c is a constant after unrolling
    return c == 0 ? qb<0>(v) : c == 1 ? qb<1>(v) : c == 2 ? qb<2>(v) : qb<3>(v);
}
constexpr int m4_ut(int ND, int J, int I) { return J * ND - J * (J - 1) / 2 + (I - J); }   // upper tile (J <= I)
constexpr int m4_lt(int J, int Jb) { return J * (J + 1) / 2 + Jb; }                         // border tile (Jb <= J)

#ifndef PSI32M_FLUSH
#define PSI32M_FLUSH 64
#endif
#ifndef PSI32M_MINB
#define PSI32M_MINB 2
#endif

template <int ND>
__global__ __launch_bounds__(256, PSI32M_MINB) void k_psi32m_moments(
    const double *__restrict__ Phi, const double *__restrict__ Tm, int ld, const double *__restrict__ rowscal,
    const double *__restrict__ w, const double *__restrict__ v, const double *__restrict__ Xr, int de, int d,
    const float *__restrict__ PsiT, long ldp, int n, int m, const double *__restrict__ P, const double *__restrict__ Rc,
    int nchunk, int rows_per_chunk, double *__restrict__ slab) {
    constexpr int D = 4 * ND, NT = ND * (ND + 1) / 2, NP = D * (D + 1) / 2, NV = 3 + D + NP;
    __shared__ f4 sR[4][NT][64];                       // R_j of the wave's 16 basis functions: group (J, K), lane (b, q): R[4J + q][4K + 0..3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 3;
    // workgroup -> (row chunk, block of 64 basis functions), XCD-aware as in k_psi32_moments: chunk c lives on XCD c % 8
    const int njb = (m + 63) / 64, xcd = blockIdx.x & 7, tseq = blockIdx.x >> 3;
    const int chunk = (tseq / njb) * 8 + xcd, jg = (tseq % njb) * 4 + wave;
    if (chunk >= nchunk || jg * 16 >= m) return;       // no barrier below: every wave stages and reads its own part of sR
    const int j = jg * 16 + (lane >> 2);
    const bool valid = j < m;
    const int jc = valid ? j : m - 1;
    {
        const double *Rj = Rc + (size_t)jc * (de * (de + 1) / 2 + de);
#pragma unroll
        for (int J = 0; J < ND; ++J)
#pragma unroll
            for (int K = J; K < ND; ++K) {
                f4 val;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * J + q, k = 4 * K + c;
                    float x;
                    if (k < r) x = 0.f;
                    else if (r >= d || k >= d) x = (r == k) ? 1.0f : 0.0f;   // padding dimensions: identity
                    else x = (float)Rj[r * de - r * (r - 1) / 2 + (k - r)];
                    val[c] = x;
                }
                sR[wave][m4_ut(ND, J, K)][lane] = val;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double pj[ND];
#pragma unroll
    for (int K = 0; K < ND; ++K) pj[K] = P[(size_t)jc * de + min(4 * K + q, de - 1)];
    const double wj = (rowscal && valid) ? w[jc] : 0.0, vj = (rowscal && v && valid) ? v[jc] : 0.0;
    f4 idn;                                            // the identity tile
#pragma unroll
    for (int r = 0; r < 4; ++r) idn[r] = (q == r) ? 1.0f : 0.0f;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // record of (chunk, j): [a0, r1, r2 | a~1 (D) | C~, packed lower triangle (NP)] as k_psi32_moments; this wave is its only writer
    double *rec = slab + ((size_t)chunk * m + jc) * NV;
    if (valid) {
#pragma unroll
        for (int J = 0; J < ND; ++J) rec[3 + 4 * J + q] = 0.0;
#pragma unroll
        for (int J = 0; J < ND; ++J)
#pragma unroll
            for (int I = J; I < ND; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = 4 * J + r, b = 4 * I + q;
                    if (a <= b) rec[3 + D + b * (b + 1) / 2 + a] = 0.0;
                }
    }
    f4 C[NT];
    float a1[ND];
#pragma unroll
    for (int t = 0; t < NT; ++t) C[t] = zero4;
#pragma unroll
    for (int J = 0; J < ND; ++J) a1[J] = 0.f;
    double a0 = 0.0, r1 = 0.0, r2 = 0.0;
    auto flush = [&]() {
        if (valid) {
#pragma unroll
            for (int J = 0; J < ND; ++J) rec[3 + 4 * J + q] += (double)a1[J];
#pragma unroll
            for (int J = 0; J < ND; ++J)
#pragma unroll
                for (int I = J; I < ND; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 4 * J + r, b = 4 * I + q;
                        if (a <= b) rec[3 + D + b * (b + 1) / 2 + a] += (double)C[m4_ut(ND, J, I)][r];
                    }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) C[t] = zero4;
#pragma unroll
        for (int J = 0; J < ND; ++J) a1[J] = 0.f;
    };
    const int r0 = chunk * rows_per_chunk, rend = min(n, r0 + rows_per_chunk);
    // Per-lane loads of a sample: x[4K + q] (clamped column, masked below), PHI and T of (i, j); wave-uniform loads: psi_i, the row
    // scalars.  Everything of sample i + 1 is requested before sample i is worked on (one sample takes ~10^4 cycles, a load ~10^3).
    int xo[ND];
    float xm[ND];
#pragma unroll
    for (int K = 0; K < ND; ++K) {
        xo[K] = min(4 * K + q, de - 1);
        xm[K] = (4 * K + q < d) ? 1.0f : 0.0f;
    }
    double xn[ND], phn = 0.0, ttn = 0.0, obn = 0.0, ccn = 0.0, dbn = 0.0;
    float psn[D];
    auto request = [&](int i) {
#pragma unroll
        for (int K = 0; K < ND; ++K) xn[K] = Xr[(size_t)i * de + xo[K]];
        phn = Phi[(size_t)i * ld + jc];
        ttn = Tm[(size_t)i * ld + jc];
        if (rowscal) {
            const double *rs = rowscal + (size_t)i * 4;
            obn = rs[0]; ccn = rs[1]; dbn = rs[2];
        }
    };
    auto request_psi = [&](int i) {
#pragma unroll
        for (int k = 0; k < D; ++k) psn[k] = PsiT[(size_t)k * ldp + i];
    };
    if (r0 < rend) {
        request(r0);
        request_psi(r0);
    }
    int since = 0;
#pragma unroll 1
    for (int i = r0; i < rend; ++i) {
        double xc[ND];
#pragma unroll
        for (int K = 0; K < ND; ++K) xc[K] = xn[K];
        const double ph = phn, tt = ttn, ob = obn, cc = ccn, db = dbn;
        request(min(i + 1, rend - 1));
        // ---- this sample: dp (fp64), Delta (formed in fp64) ----
        double dpd;
        if (rowscal) {
            dpd = (-ob * tt - cc * wj + db * vj) * ph;                                 // GPz.m:72,90,106,113
            r1 = fma(ph, cc, r1);
            r2 = fma(ph, db, r2);
        } else {
            dpd = tt;                                                                  // dPHI already formed (k > 1)
        }
        if (!valid) dpd = 0.0;
        a0 += dpd;
        const float dp = (float)dpd;
        float dl[ND];
#pragma unroll
        for (int K = 0; K < ND; ++K) dl[K] = (float)(xc[K] - pj[K]) * xm[K];
        // ---- build A = I + R Psi R' (upper tiles) and z = R Delta ----
        f4 T[NT];
        float z[ND];
#pragma unroll
        for (int J = 0; J < ND; ++J) z[J] = 0.f;
#pragma unroll
        for (int K = 0; K < ND; ++K) {
            f4 rg[ND];
#pragma unroll
            for (int J = 0; J <= K; ++J) rg[J] = sR[wave][m4_ut(ND, J, K)][lane];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float psk = psn[4 * K + c];                                       // wave-uniform
                const float db_ = qbc(dl[K], c);                                       // Delta[4K + c]
#pragma unroll
                for (int J = 0; J <= K; ++J) z[J] = fmaf(rg[J][c], db_, z[J]);
                float g[ND];
#pragma unroll
                for (int J = 0; J <= K; ++J) g[J] = rg[J][c] * psk;
#pragma unroll
                for (int J = 0; J <= K; ++J)
#pragma unroll
                    for (int I = J; I <= K; ++I) {
                        const f4 cin = (I == K && c == 0) ? (I == J ? idn : zero4) : T[m4_ut(ND, J, I)];
                        T[m4_ut(ND, J, I)] = MF4(g[J], rg[I][c], cin);
                    }
            }
        }
        request_psi(min(i + 1, rend - 1));                                             // (the scalar registers of psi_i are free again)
        // ---- eliminate [A | I]; z and u~ on the vector pipe ----
        f4 E[NT];
        float u[ND];
#pragma unroll
        for (int J = 0; J < ND; ++J) u[J] = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int K = k >> 2, c = k & 3;
            float row[ND], e[ND], nA[ND];
#pragma unroll
            for (int I = K; I < ND; ++I) row[I] = T[m4_ut(ND, K, I)][c];
#pragma unroll
            for (int Jb = 0; Jb <= K; ++Jb) e[Jb] = (Jb == K && c == 0) ? idn[0] : E[m4_lt(K, Jb)][c];
            const float piv = qbc(row[K], c);
            const float rp = __builtin_amdgcn_rcpf(piv);
            const float nrp = -rp;
#pragma unroll
            for (int I = K; I < ND; ++I) nA[I] = row[I] * nrp;
            const float yk = qbc(z[K], c);
#pragma unroll
            for (int I = K; I < ND; ++I) z[I] = fmaf(nA[I], yk, z[I]);
            const float t = yk * rp;
#pragma unroll
            for (int Jb = 0; Jb <= K; ++Jb) u[Jb] = fmaf(e[Jb], t, u[Jb]);
            const float cs = dp * nrp;
            // trailing matrix and border first (the next pivot row waits for them), the moment sums last
#pragma unroll
            for (int J = K; J < ND; ++J) {
#pragma unroll
                for (int I = J; I < ND; ++I) T[m4_ut(ND, J, I)] = MF4(nA[J], row[I], T[m4_ut(ND, J, I)]);
#pragma unroll
                for (int Jb = 0; Jb <= K; ++Jb) {
                    // first touch: E(K, K) starts as the identity, E(J, K) for J > K as zero (at c == 0 of block K)
                    const f4 cin = (Jb == K && c == 0) ? (J == K ? idn : zero4) : E[m4_lt(J, Jb)];
                    E[m4_lt(J, Jb)] = MF4(nA[J], e[Jb], cin);
                }
            }
#pragma unroll
            for (int Jb = 0; Jb <= K; ++Jb) {
                const float ce = e[Jb] * cs;
#pragma unroll
                for (int Ib = Jb; Ib <= K; ++Ib) C[m4_ut(ND, Jb, Ib)] = MF4(ce, e[Ib], C[m4_ut(ND, Jb, Ib)]);
            }
        }
        // ---- dp u~ u~' and dp u~ ----
#pragma unroll
        for (int J = 0; J < ND; ++J) {
            const float du = u[J] * dp;
            a1[J] += du;
#pragma unroll
            for (int I = J; I < ND; ++I) C[m4_ut(ND, J, I)] = MF4(du, u[I], C[m4_ut(ND, J, I)]);
        }
        if (++since == PSI32M_FLUSH) {
            flush();
            since = 0;
        }
    }
    if (since) flush();
    if (valid && q == 0) {
        rec[0] = a0;
        rec[1] = r1;
        rec[2] = r2;
    }
}

bool psi32m_available(int d) {
    // Opt-in (GPZ_PSI32_MFMA=1, latched when the context is created): measured 133.6 ms against 125.0 ms of the lane-per-pair kernel
    // on config 5's 250 000-row shard - docs/HISTORY.md has the instruction-issue model behind that.
    return gpz_opts().psi32_mfma && d >= 1 && d <= 20;
}

int launch_psi32m_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                          const double *v, const double *Xr, int de, int d, const float *PsiT, long ldp, int n, int m,
                          const double *P, const double *Rc, int nchunk, int rows_per_chunk, double *slab) {
    if (!psi32m_available(d)) return -1;
    const dim3 grid((nchunk + 7) / 8 * 8 * ((m + 63) / 64));
#define M4_CASE(NDv)                                                                                                            \
    hipLaunchKernelGGL((k_psi32m_moments<NDv>), grid, dim3(256), 0, st, Phi, T, ld, rowscal, w, v, Xr, de, d, PsiT, ldp, n, m, P, Rc, \
                       nchunk, rows_per_chunk, slab)
    switch ((d + 3) / 4) {
        case 1: M4_CASE(1); break;
        case 2: M4_CASE(2); break;
        case 3: M4_CASE(3); break;
        case 4: M4_CASE(4); break;
        case 5: M4_CASE(5); break;
        default: return -1;
    }
#undef M4_CASE
    return 0;
}
