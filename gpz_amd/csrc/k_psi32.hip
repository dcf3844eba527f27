// GC/VC with input noise, fp32 per-pair factorisations (BASELINE config 5: "VC + input-noise (Psi), fp32 path").
//
//   getPHI.m:78-89   ln PHI_ij = -1/2 Delta' M^-1 Delta + 1/2 ln|Sigma_j| - 1/2 ln|M|,   M = Psi_i + Sigma_j
//   GPz.m:164-185    sum_i dPHI_ij * [1, M^-1 Delta, (M^-1 Delta)(M^-1 Delta)' - M^-1]
//
// The reference factorises a d x d matrix for each of the n*m (sample, basis) pairs.  In fp64 a d = 20 triangle is
// 210 doubles = 420 VGPRs and does not fit a lane; in fp32 it does.  With dtype = f32 the pair matrices are built,
// factorised and inverted in fp32 registers (compile-time indices, fully unrolled); Delta = x - p, the exponent
// ln PHI, PHI itself and every sum over rows stay fp64, as do the m x m stage and the MFMA contractions.
// Lanes run along ROWS in both kernels; the moment kernel reduces each of its 3 + d + d(d+1)/2 per-basis sums over
// the 64 rows of a wave with a transposing butterfly (one exchange per value instead of six) and accumulates them in
// fp64 LDS accumulators over the rows of its chunk.
//
// Diagonal Psi_i (what fixPsi.m builds from per-dimension variances, and BASELINE config 5's input noise) takes a
// WHITENED form that never builds Sigma_j = inv(Gamma_j'Gamma_j): with Gamma_j = Q R (the QR factor the tuned PHI
// kernel already uses),  M^-1 = R' A^-1 R,  A = I + R Psi_i R',  ln|M| = ln|Sigma_j| + ln|A|,  so
//     ln PHI_ij = -1/2 z' A^-1 z - 1/2 ln|A|,   z = R Delta,
// and the moment sums are accumulated in the whitened coordinates (u~ = A^-1 z,  u~u~' - A^-1) and mapped back per
// basis function with R at the end.  A has eigenvalues >= 1 and cond(A) <= 1 + max(psi)*|R|^2, so the fp32
// factorisation is safe however ill-conditioned Sigma_j is (cond(Sigma_j) = cond(Gamma_j)^2 reaches 1e6 on the
// benchmark's own theta at d = 20, which breaks an fp32 Cholesky of Sigma_j + Psi_i).  Full Psi_i cubes keep the
// direct form M = Sigma_j + Psi_i (R Psi R' per pair would cost more than everything else) and need
// cond(Sigma_j) * 6e-8 << 1.
//
// d is padded to D in {4, 8, 12, 16, 20} with identity rows (Sigma = I, Psi = 0, Delta = 0 there: no contribution).
// PsiT: packed lower triangle of Psi_i, element-major (PsiT[e * ldp + i]) so the lanes of a wave read consecutive
// floats; DIAG: Psi_i is diagonal for every row (what fixPsi.m builds from per-dimension variances) and PsiT holds
// only the D diagonals.
#include <stdlib.h>
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define LT(r, c) ((r) * ((r) + 1) / 2 + (c))

// ---- packed fp32 arithmetic ------------------------------------------------------------------------------------------
// A plain v_fma_f32 retires 64 FMAs per 4 cycles and SIMD (78.6 TFLOP/s on the chip); the 157 TFLOP/s fp32 vector peak
// is v_pk_fma_f32, two FMAs per lane on an even-aligned register pair, with op_sel picking either half of an operand
// for both results (a free broadcast).  Everything O(D^3) below therefore works on ROW PAIRS of the lower triangle:
// column c holds the pairs t = c/2 .. D/2-1 = rows (2t, 2t+1).  For odd c the first pair starts one row above the
// diagonal; that slot is junk during the factorisation (never read as a row-c value) and is set to zero in the inverse
// factor, where whole columns enter dot products.  D(D+2)/4 pairs instead of D(D+1)/2 scalars: 110 vs 210 at D = 20.
typedef float f2 __attribute__((ext_vector_type(2)));
// A scalar add the SLP vectoriser cannot see: it turns two independent adds of register halves (the horizontal sums of packed dot
// products, the adds behind the lane swaps of reduce32) into THREE moves that build aligned pairs and one v_pk_add_f32 - four issue
// slots instead of two (500 v_mov_b32 per pair and wave in k_psi32_moments<20>).
__device__ __forceinline__ float addf(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 sp(float x) {
    f2 r;
    r.x = x;
    r.y = x;
    return r;
}
// first pair of column c:  c*H - sum_{q<c} floor(q/2)  (closed form: the unroller must see plain arithmetic)
constexpr int p2base(int H, int c) { return c * H - ((c & 1) ? (c / 2) * (c / 2) : (c / 2) * (c / 2 - 1)); }
#define P2(t, c) (p2base(D / 2, (c)) + (t) - (c) / 2)
#define NP2_OF(D) p2base((D) / 2, (D))

// In-place Cholesky, left-looking by columns.  rd[c] = 1 / L_cc (the rsq every column needs anyway), *hl = sum ln L_cc.
template <int D>
__device__ __forceinline__ void chol2(f2 (&M)[NP2_OF(D)], float (&rd)[D], float *hl) {
    constexpr int H = D / 2;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int t = c / 2; t < H; ++t) {
            f2 a = M[P2(t, c)];
#pragma unroll
            for (int q = 0; q < c; ++q) a -= M[P2(t, q)] * sp(M[P2(c / 2, q)][c & 1]);      // L[., q] * L[c][q]
            M[P2(t, c)] = a;
        }
        const float p = M[P2(c / 2, c)][c & 1];
        const float inv = __builtin_amdgcn_rsqf(p);                                         // 1/sqrt(p), 1 ulp
        rd[c] = inv;
        s += __logf(p);
#pragma unroll
        for (int t = c / 2; t < H; ++t) M[P2(t, c)] *= sp(inv);
    }
    *hl = 0.5f * s;
}

// L <- W = inv(L) in place, ascending columns: column c of W is the forward substitution of e_c, done right-looking so
// that every update is a packed column operation; it reads columns >= c of L, which are still untouched.
template <int D>
__device__ __forceinline__ void trinv2(f2 (&L)[NP2_OF(D)], const float (&rd)[D]) {
    constexpr int H = D / 2;
    f2 rd2[H];                                               // (1/L_2t,2t, 1/L_2t+1,2t+1)
#pragma unroll
    for (int t = 0; t < H; ++t) {
        rd2[t].x = rd[2 * t];
        rd2[t].y = rd[2 * t + 1];
    }
    // The diagonal of L is not read again (rd holds its reciprocals): zero it, so that the packed update below leaves the
    // finished row of an even column alone.
#pragma unroll
    for (int c = 0; c < D; ++c) L[P2(c / 2, c)][c & 1] = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        // x = -(partial sums), started at -e_c: then w_q = -x_q / L_qq holds for q = c as well, and the slot above the
        // diagonal of an odd column comes out as zero.  Every q is ONE packed multiply that writes a whole pair (for even
        // q only its lower half is final and is used; the pair of odd q is the finished one) - no half-register inserts.
        f2 x[H], wv[H];
#pragma unroll
        for (int t = c / 2; t < H; ++t) x[t] = sp(0.f);
        x[c / 2][c & 1] = -1.0f;
#pragma unroll
        for (int q = c; q < D; ++q) {
            const f2 wp = -x[q / 2] * rd2[q / 2];
            if (q & 1) wv[q / 2] = wp;
            const float wq = wp[q & 1];
#pragma unroll
            for (int t = (q + 1) / 2; t < H; ++t) x[t] += L[P2(t, q)] * sp(wq);             // rows > q (and q itself, times the zeroed diagonal)
        }
#pragma unroll
        for (int t = c / 2; t < H; ++t) L[P2(t, c)] = wv[t];
    }
}

// Stage the pair-matrix parameters and the centres of JB basis functions into LDS.
//   DIAG: the QR factor R_j by columns, row k of the stage = column k of R as D/2 row pairs (R[2t][k], R[2t+1][k]), zero
//         below the diagonal (Rc: k_prep_cov's records, row a of R at a*de - a(a-1)/2, de = padded dimension of the
//         parameter block);  else: Sigma_j in the row-pair layout above.  Padding dimensions (>= d) are identity.
template <int D, int JB, bool DIAG>
__device__ __forceinline__ void stage_params(int tid, int nt, int j0, int m, int d, int de, const double *__restrict__ Sig,
                                             const double *__restrict__ Rc, const double *__restrict__ P,
                                             f2 (*sS)[D * D / 2], double (*sP)[D]) {
    constexpr int H = D / 2;
    if (DIAG) {
        for (int e = tid; e < JB * D * D; e += nt) {
            const int jj = e / (D * D), q = e % (D * D), k = q / D, r = q % D, j = min(j0 + jj, m - 1);
            float v;
            if (r > k) v = 0.f;
            else if (k >= d) v = (r == k) ? 1.0f : 0.0f;
            else v = (float)Rc[(size_t)j * (de * (de + 1) / 2 + de) + (r * de - r * (r - 1) / 2) + (k - r)];   // R[r][k]
            ((float *)&sS[jj][0])[q] = v;
        }
    } else {
        constexpr int NP2 = NP2_OF(D);
        for (int e = tid; e < JB * NP2 * 2; e += nt) {
            const int jj = e / (NP2 * 2), q = e % (NP2 * 2), pe = q >> 1, j = min(j0 + jj, m - 1);
            int c = 0, b = 0;                                   // column of pair pe: base(c) = c*H - (c even ? a(a-1) : a*a), a = c/2
            for (int cn = 1; cn < D; ++cn) {
                const int a = cn / 2, bn = cn * H - ((cn & 1) ? a * a : a * (a - 1));
                if (bn <= pe) { c = cn; b = bn; }
            }
            const int r = 2 * (pe - b + c / 2) + (q & 1);
            float v;
            if (r < c) v = 0.f;
            else if (r >= d) v = (r == c) ? 1.0f : 0.0f;
            else v = (float)Sig[(size_t)j * d * d + r * d + c];
            ((float *)&sS[jj][0])[q] = v;
        }
    }
    for (int e = tid; e < JB * D; e += nt) {
        const int jj = e / D, c = e % D, j = min(j0 + jj, m - 1);
        sP[jj][c] = (c < d) ? P[(size_t)j * de + c] : 0.0;
    }
}

// The pair matrix in the row-pair layout.  DIAG: A = I + R diag(psi) R' = I + sum_k psi_k r_k r_k' (r_k = column k of R);
// else M = Sigma_j + Psi_i (getPHI.m:84, GPz.m:170).
template <int D, bool DIAG>
__device__ __forceinline__ void pair_matrix2(const f2 *__restrict__ S, const float (&pd)[DIAG ? D : 1],
                                             const float *__restrict__ PsiT, long ldp, unsigned ic, f2 (&M)[NP2_OF(D)]) {
    constexpr int H = D / 2;
    if (DIAG) {
#pragma unroll
        for (int e = 0; e < NP2_OF(D); ++e) M[e] = sp(0.f);
#pragma unroll
        for (int c = 0; c < D; ++c) M[P2(c / 2, c)][c & 1] = 1.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            f2 s[H], g[H];
#pragma unroll
            for (int t = 0; t <= k / 2; ++t) {
                s[t] = S[k * H + t];
                g[t] = s[t] * sp(pd[k]);                                               // psi_k R[., k]
            }
#pragma unroll
            for (int c = 0; c <= k; ++c)
#pragma unroll
                for (int t = c / 2; t <= k / 2; ++t) M[P2(t, c)] += s[t] * sp(g[c / 2][c & 1]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
            for (int t = c / 2; t < H; ++t) {
                f2 ps;
                ps.x = (2 * t >= c) ? PsiT[(size_t)LT(2 * t, c) * ldp + ic] : 0.f;
                ps.y = PsiT[(size_t)LT(2 * t + 1, c) * ldp + ic];
                M[P2(t, c)] = S[P2(t, c)] + ps;
            }
    }
}

// z = Delta (full Psi) or R Delta (whitened), as row pairs; Delta is formed in fp64.
template <int D, bool DIAG>
__device__ __forceinline__ void delta2(const f2 *__restrict__ S, const float (&dl)[D], f2 (&z)[D / 2]) {
    constexpr int H = D / 2;
    if (DIAG) {
#pragma unroll
        for (int t = 0; t < H; ++t) z[t] = sp(0.f);
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int t = 0; t <= k / 2; ++t) z[t] += S[k * H + t] * sp(dl[k]);
    } else {
#pragma unroll
        for (int t = 0; t < H; ++t) {
            z[t].x = dl[2 * t];
            z[t].y = dl[2 * t + 1];
        }
    }
}

// PHI: one thread per row, JB basis functions' parameters staged in LDS at a time; blockIdx.y splits the basis
// functions into groups so that small row counts still fill the chip.
template <int D, bool DIAG>
__global__ __launch_bounds__(256) void k_psi32_phi(const double *__restrict__ Xr, int de, int d,
                                                    const float *__restrict__ PsiT, long ldp, int n, int m,
                                                    const double *__restrict__ P, const double *__restrict__ Sig,
                                                    const double *__restrict__ Rc, const double *__restrict__ lnS,
                                                    double *__restrict__ Phi, int ld, int jgroup, int round32) {
    constexpr int H = D / 2;
    constexpr int JB = 8;
    __shared__ f2 sS[JB][D * D / 2];
    __shared__ double sP[JB][D];
    __shared__ double sL[JB];
    __shared__ double sOut[JB][256];                   // the JB results of every row, written out as one 64-byte run per row
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n;
    const unsigned ic = (unsigned)(act ? i : 0);
    double x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = (c < d) ? Xr[(size_t)ic * de + c] : 0.0;
    float pd[DIAG ? D : 1];
    if (DIAG) {
#pragma unroll
        for (int c = 0; c < D; ++c) pd[c] = PsiT[(size_t)c * ldp + ic];
    }
    const int jlo = blockIdx.y * jgroup, jhi = min(m, jlo + jgroup);
    for (int j0 = jlo; j0 < jhi; j0 += JB) {
        __syncthreads();
        stage_params<D, JB, DIAG>(threadIdx.x, 256, j0, m, d, de, Sig, Rc, P, sS, sP);
        if (threadIdx.x < JB) sL[threadIdx.x] = lnS[min(j0 + (int)threadIdx.x, m - 1)];
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            if (j >= jhi) break;
            f2 M[NP2_OF(D)];
            pair_matrix2<D, DIAG>(sS[jj], pd, PsiT, ldp, ic, M);
            float hl, rd[D];
            chol2<D>(M, rd, &hl);
            float dl[D];
#pragma unroll
            for (int k = 0; k < D; ++k) dl[k] = (float)(x[k] - sP[jj][k]);
            f2 z[H];
            delta2<D, DIAG>(sS[jj], dl, z);
            float quad = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) {                                              // y = L^-1 z, column by column
                const float yc = z[c / 2][c & 1] * rd[c];
                quad = fmaf(yc, yc, quad);
#pragma unroll
                for (int t = (c + 1) / 2; t < H; ++t) z[t] -= M[P2(t, c)] * sp(yc);
            }
            // getPHI.m:86:  -1/2 quad + 1/2 ln|Sigma_j| - 1/2 ln|M|;  whitened: the two log-determinants collapse to -1/2 ln|A|
            const double lp = DIAG ? (-0.5 * (double)quad - (double)hl) : (-0.5 * (double)quad + 0.5 * sL[jj] - (double)hl);
            // round32: experiment switch (tools/f32_operand_experiment.py) - PHI rounded to fp32 as an fp32-operand MFMA would see it
            sOut[jj][threadIdx.x] = round32 ? (double)(float)exp(lp) : exp(lp);
        }
        if (act) {                                      // (own column of sOut: no barrier needed)
            double *dst = Phi + (size_t)i * ld + j0;
            if (j0 + JB <= jhi) {                       // ld is a multiple of 16, j0 of 8: 16-byte aligned
#pragma unroll
                for (int q = 0; q < JB / 2; ++q) {
                    d2_t o;
                    o.x = sOut[2 * q][threadIdx.x];
                    o.y = sOut[2 * q + 1][threadIdx.x];
                    reinterpret_cast<d2_t *>(dst)[q] = o;
                }
            } else {
                for (int jj = 0; j0 + jj < jhi; ++jj) dst[jj] = sOut[jj][threadIdx.x];
            }
        }
    }
}

// Transposing butterfly over the 64 lanes of a wave for NV = 32 values per lane: afterwards lane l holds, in v[0], the
// sum over all lanes of value (l >> 1).  31 exchanges + 1 instead of 6 per value.
// The two widest stages (partner 32 and 16 lanes away, 24 of the 31 exchanges) are gfx950's v_permlane32_swap /
// v_permlane16_swap: one instruction trades the halves (odd rows of the first register with even rows of the second),
// after which a single add leaves "own half + partner's same half" in every lane - no selects, no LDS crossbar.
__device__ __forceinline__ void reduce32(float (&v)[32], int lane) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 16]), false, false);
        v[k] = addf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
        v[k] = addf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int s = 4, bit = 8; s >= 1; s >>= 1, bit >>= 1) {
        // bitwise select (v_bfi_b32) instead of ?: - a select between two array elements is turned into a dynamically
        // indexed load by the optimiser, which would push the whole array into scratch
        const unsigned msk = (lane & bit) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < s; ++k) {
            const unsigned lo = __float_as_uint(v[k]), hi = __float_as_uint(v[k + s]);
            const float keep = __uint_as_float((hi & msk) | (lo & ~msk));
            const float send = __uint_as_float((lo & msk) | (hi & ~msk));
            v[k] = keep + __shfl_xor(send, bit, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 1, 64);
}

// Raw moment sums per (row chunk, basis): [a0, r1, r2 | sum dp*u (D) | sum dp*(uu' - Minv) packed (NP)], whitened when DIAG.
// One wave per workgroup; blockIdx.x = row chunk, blockIdx.y = block of JB basis functions.
template <int D, bool DIAG>
__global__ __launch_bounds__(64) void k_psi32_moments(const double *__restrict__ Phi, const double *__restrict__ T, int ld,
                                                       const double *__restrict__ rowscal, const double *__restrict__ w,
                                                       const double *__restrict__ v, const double *__restrict__ Xr, int de,
                                                       int d, const float *__restrict__ PsiT, long ldp, int n, int m,
                                                       const double *__restrict__ P, const double *__restrict__ Sig,
                                                       const double *__restrict__ Rc, int nchunk, int rows_per_chunk,
                                                       double *__restrict__ slab, int nrec) {
    constexpr int NP = D * (D + 1) / 2;
    constexpr int H = D / 2;
    constexpr int JB = 8;
    constexpr int NV = 3 + D + NP;                     // values reduced per pair: [dp, r1, r2 | dp*u (D) | dp*(uu' - Minv) (NP)]
    constexpr int NG = (NV + 31) / 32;                 // groups of 32 values
    __shared__ f2 sS[JB][D * D / 2];
    __shared__ double sP[JB][D];
    __shared__ double acc[JB][NG * 32];
    __shared__ double sPh[JB][64], sTt[JB][64];        // PHI / T of the wave's 64 rows x JB basis functions
    const int lane = threadIdx.x;
    // Workgroup -> (row chunk, basis block), XCD-aware: consecutive workgroup ids go round the 8 XCDs (each with its own L2), so
    // chunk c belongs to XCD c % 8 and an XCD walks the basis blocks of one chunk before it takes its next chunk.  The chunk's rows of
    // X and Psi then come from HBM once per chunk (not once per basis block: 250 blocks at m = 2000), and the two halves of a 128-byte
    // line of PHI / T (two basis blocks) are read by neighbours in time.  Fabric-side bytes of a launch at config 5's shard: 34 GB
    // with blockIdx.x = chunk, blockIdx.y = block; the algorithmic 8 GB are PHI and T read once.
    const int njb = (m + JB - 1) / JB, xcd = blockIdx.x & 7, tseq = blockIdx.x >> 3;
    const int chunk = (tseq / njb) * 8 + xcd, j0 = (tseq % njb) * JB;
    if (chunk >= nchunk) return;                       // (the grid is padded to a multiple of 8 chunks)
    stage_params<D, JB, DIAG>(lane, 64, j0, m, d, de, Sig, Rc, P, sS, sP);
    for (int e = lane; e < JB * NG * 32; e += 64) (&acc[0][0])[e] = 0.0;
    __syncthreads();
    const int r0 = chunk * rows_per_chunk, rend = min(n, r0 + rows_per_chunk);
    for (int rb = r0; rb < rend; rb += 64) {
        const int i = rb + lane;
        const bool act = i < rend;
        const unsigned ic = (unsigned)(act ? i : rend - 1);
        float pd[DIAG ? D : 1];
        if (DIAG) {
#pragma unroll
            for (int c = 0; c < D; ++c) pd[c] = PsiT[(size_t)c * ldp + ic];
        }
        double ob = 0.0, cc = 0.0, db = 0.0;
        if (rowscal) { const double *rs = rowscal + (size_t)ic * 4; ob = rs[0]; cc = rs[1]; db = rs[2]; }
        {   // this row's PHI / T entries of the JB basis functions: 64 contiguous bytes each, read ONCE as 16-byte loads and
            // parked in LDS (one 8-byte load per basis function touched a different cache line per lane every time:
            // 8.6 x the algorithmic bytes at the fabric).  ld is a multiple of 16 and j0 of 8: aligned, inside the row.
            const d2_t *pr = reinterpret_cast<const d2_t *>(Phi + (size_t)ic * ld + j0);
            const d2_t *tr = reinterpret_cast<const d2_t *>(T + (size_t)ic * ld + j0);
            d2_t pv[JB / 2], tv[JB / 2];
#pragma unroll
            for (int q = 0; q < JB / 2; ++q) { pv[q] = pr[q]; tv[q] = tr[q]; }
#pragma unroll
            for (int q = 0; q < JB / 2; ++q) {
                sPh[2 * q][lane] = pv[q].x; sPh[2 * q + 1][lane] = pv[q].y;
                sTt[2 * q][lane] = tv[q].x; sTt[2 * q + 1][lane] = tv[q].y;
            }
        }
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            if (j >= m) break;
            const double ph = sPh[jj][lane], tt = sTt[jj][lane];   // same lane wrote it: no barrier needed
            double dpd, q1 = 0.0, q2 = 0.0;
            if (rowscal) {
                dpd = (-ob * tt - cc * w[j] + db * (v ? v[j] : 0.0)) * ph;           // GPz.m:72,90,106,113
                q1 = ph * cc;
                q2 = ph * db;
            } else {
                dpd = tt;                                                              // dPHI already formed (k > 1)
            }
            if (!act) { dpd = 0.0; q1 = 0.0; q2 = 0.0; }
            const float dp = (float)dpd;
            f2 L[NP2_OF(D)];
            pair_matrix2<D, DIAG>(sS[jj], pd, PsiT, ldp, ic, L);
            float hl, rd[D];
            chol2<D>(L, rd, &hl);
            trinv2<D>(L, rd);                                                          // L now holds W = inv(L)
            float u[D];
            {
                float dl[D];
#pragma unroll
                for (int c = 0; c < D; ++c) dl[c] = (c < d) ? (float)(Xr[(size_t)ic * de + c] - sP[jj][c]) : 0.f;   // Delta formed in fp64
                f2 z[H], y[H];
                delta2<D, DIAG>(sS[jj], dl, z);                                        // z = Delta, or R Delta (whitened)
#pragma unroll
                for (int t = 0; t < H; ++t) y[t] = sp(0.f);
#pragma unroll
                for (int c = 0; c < D; ++c)                                            // y = W z
#pragma unroll
                    for (int t = c / 2; t < H; ++t) y[t] += L[P2(t, c)] * sp(z[c / 2][c & 1]);
#pragma unroll
                for (int a = 0; a < D; ++a) {                                          // u = W' y = M^-1 Delta
                    f2 s2 = L[P2(a / 2, a)] * y[a / 2];
#pragma unroll
                    for (int t = a / 2 + 1; t < H; ++t) s2 += L[P2(t, a)] * y[t];
                    u[a] = addf(s2.x, s2.y);
                }
            }
            // Values in record order, pushed 32 at a time through the wave reduction and added to the fp64
            // accumulators.  Everything is unrolled, so `cnt` is a compile-time constant at every push.
            float vv[32];
            int cnt = 0;
            auto push = [&](float val) {
                vv[cnt & 31] = val;
                ++cnt;
                if ((cnt & 31) == 0) {
                    reduce32(vv, lane);
                    if ((lane & 1) == 0) acc[jj][cnt - 32 + (lane >> 1)] += (double)vv[0];
                }
            };
            push(dp);
            push((float)q1);
            push((float)q2);
#pragma unroll
            for (int a = 0; a < D; ++a) push(dp * u[a]);                               // GPz.m:172
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    f2 m2 = L[P2(a / 2, a)] * L[P2(a / 2, b)];                         // Minv(a,b) = sum_q W[q][a] W[q][b]
#pragma unroll
                    for (int t = a / 2 + 1; t < H; ++t) m2 += L[P2(t, a)] * L[P2(t, b)];
                    push(dp * (u[a] * u[b] - addf(m2.x, m2.y)));                          // GPz.m:174
                }
#pragma unroll
            for (int e = NV; e < NG * 32; ++e) push(0.f);
        }
    }
    __syncthreads();
    // raw sums [a0, r1, r2 | a~1 (D) | C~ packed (NP)] per (chunk, basis); k_psi32_records expands them after the chunk sum
    for (int jj = 0; jj < JB; ++jj) {
        const int j = j0 + jj;
        if (j >= m) break;
        double *rec = slab + ((size_t)chunk * m + j) * NV;
        for (int e = lane; e < NV; e += 64) rec[e] = acc[jj][e];
    }
}

// Raw sums -> records [a0 | acc1 (d) | C (d x d) | r1 | r2], one workgroup per basis function.
//   diag = 0: the layout k_gen_finish consumes (acc1 = sum dp*u, C = sum dp*(uu' - M^-1), GPz.m:172-174).
//   diag = 1: the sums stay in whitened coordinates for k_psi32_finish:  acc1 = sum dp*u~,
//             C = C~' = sum dp*(u~u~' + I - A^-1) = C~ + a0*I   (the a0*Sigma^-1 term of GPz.m:174 folded in exactly).
__global__ __launch_bounds__(64) void k_psi32_records(const double *__restrict__ raw, int D, int d, int diag,
                                                       double *__restrict__ recs, int nrec) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int NV = 3 + D + D * (D + 1) / 2;
    const double *A = raw + (size_t)j * NV;
    double *rec = recs + (size_t)j * nrec;
    for (int e = lane; e < nrec; e += 64) {
        double val;
        if (e == 0) val = A[0];
        else if (e < 1 + d) val = A[3 + (e - 1)];
        else if (e < 1 + d + d * d) {
            const int t = e - 1 - d, a = t / d, b = t % d;
            val = A[3 + D + (a >= b ? LT(a, b) : LT(b, a))];
            if (diag && a == b) val += A[0];
        } else if (e == 1 + d + d * d) val = A[1];
        else val = A[2];
        rec[e] = val;
    }
}

// Gradient blocks of basis function j from the whitened records (diagonal Psi, all dimensions observed).
// The reference chains  dS = 1/2 (a0 Sigma^-1 + C),  diS = -Sigma dS Sigma,  dGamma = 2 Gamma diS  (GPz.m:174-180) through
// Sigma_j = inv(Gamma_j'Gamma_j) twice and loses cond(Gamma_j'Gamma_j)^1.5 * eps on the way — for the ill-conditioned
// basis functions of the benchmark's own theta (cond 1e7..1e10 at d = 20) nothing is left of the result, in fp64 too.
// With Gamma = Q R and the whitened sums the same expression is
//     a0 Sigma^-1 + C = R' C~' R,     dGamma = -Gamma Sigma (R' C~' R) Sigma = -Q C~' R^-T,     dP = R' a~1,
// one triangular solve with R instead of two products with Sigma: the loss is cond(R) = sqrt(cond(Gamma'Gamma)).
__global__ __launch_bounds__(64) void k_psi32_finish(const double *__restrict__ recs, int m, int d, int de,
                                                      const double *__restrict__ Gam, const double *__restrict__ Rc,
                                                      int method_id, const double *__restrict__ sums1, int k,
                                                      double *__restrict__ grad, double *__restrict__ dGfull,
                                                      double *__restrict__ cols, int mp, int nrec) {
    constexpr int GD = 20;
    __shared__ double Y[GD * GD], Q[GD * GD];
    const int j = blockIdx.x, t = threadIdx.x;                             // one workgroup per basis function
    const double nk = sums1[10] * (double)k;
    const double *rec = recs + (size_t)j * nrec;
    const double *Rj = Rc + (size_t)j * (de * (de + 1) / 2 + de);
    const double *Gj = Gam + (size_t)j * de * de;
    auto R = [&](int a, int b) -> double { return Rj[a * de - a * (a - 1) / 2 + (b - a)]; };   // b >= a
    const int md = m * d;
    if (t < d) {                                                           // dP = R' a~1      (GPz.m:172)
        double s = 0.0;
        for (int q = 0; q <= t; ++q) s = fma(R(q, t), rec[1 + q], s);
        grad[j + m * t] = -s / nk;
    }
    if (t < d) {                                                           // row t of Y = C~' R^-T:  sum_b Y[t][b] R[c][b] = C[t][c]
        for (int c = d - 1; c >= 0; --c) {
            double s = rec[1 + d + t * d + c];
            for (int b = c + 1; b < d; ++b) s = fma(-Y[t * GD + b], R(c, b), s);
            Y[t * GD + c] = s / R(c, c);
        }
    } else if (t >= 32 && t < 32 + d) {                                    // row a of Q = Gamma R^-1:  sum_b Q[a][b] R[b][c] = Gamma[a][c]
        const int a = t - 32;
        for (int c = 0; c < d; ++c) {
            double s = Gj[a * de + c];
            for (int b = 0; b < c; ++b) s = fma(-Q[a * GD + b], R(b, c), s);
            Q[a * GD + c] = s / R(c, c);
        }
    }
    __syncthreads();
    for (int e = t; e < d * d; e += 64) {
        const int a = e / d, b = e % d;
        double s = 0.0;
        for (int q = 0; q < d; ++q) s = fma(Q[a * GD + q], Y[q * GD + b], s);
        const double val = -s;                                             // dGamma_j(a, b)
        if (method_id == 5) grad[md + a + d * b + d * d * j] = -val / nk;
        else dGfull[(size_t)j * d * d + a * d + b] = val;
    }
    if (cols && t == 0) {
        cols[j] = rec[1 + d + d * d];
        cols[mp + j] = rec[2 + d + d * d];
    }
}

#define PSI32_CASES(MACRO)            \
    switch (Dp) {                     \
        case 4: MACRO(4); break;      \
        case 8: MACRO(8); break;      \
        case 12: MACRO(12); break;    \
        case 16: MACRO(16); break;    \
        case 20: MACRO(20); break;    \
        default: return -1;           \
    }

int psi32_raw_len(int d) {
    const int D = psi32_pad_dim(d);
    return 3 + D + D * (D + 1) / 2;
}
void launch_psi32_records(hipStream_t st, const double *raw, int d, int diag, int m, double *recs, int nrec) {
    hipLaunchKernelGGL(k_psi32_records, dim3(m), dim3(64), 0, st, raw, psi32_pad_dim(d), d, diag, recs, nrec);
}
void launch_psi32_finish(hipStream_t st, const double *recs, int m, int d, int de, const double *Gam, const double *Rc,
                         int method_id, const double *sums1, int k, double *grad, double *dGfull, double *cols, int mp,
                         int nrec) {
    hipLaunchKernelGGL(k_psi32_finish, dim3(m), dim3(64), 0, st, recs, m, d, de, Gam, Rc, method_id, sums1, k, grad, dGfull, cols,
                       mp, nrec);
}

int psi32_pad_dim(int d) {
    static const int sup[] = {4, 8, 12, 16, 20};
    for (int s : sup)
        if (d <= s) return s;
    return -1;
}

int launch_psi32_phi(hipStream_t st, const double *Xr, int de, int d, const float *PsiT, long ldp, int diag, int n, int m,
                     const double *P, const double *Sig, const double *Rc, const double *lnS, double *Phi, int ld) {
    const int Dp = psi32_pad_dim(d);
    if (n <= 0) return Dp > 0 ? 0 : -1;   // a rank of a sharded run may hold no row of this set
    // split the basis functions over blockIdx.y until ~1024 workgroups exist (groups are multiples of the staging block)
    const int nrb = (n + 255) / 256;
    int ng = (1024 + nrb - 1) / nrb;
    if (ng > (m + 7) / 8) ng = (m + 7) / 8;
    if (ng < 1) ng = 1;
    const int jgroup = (((m + ng - 1) / ng) + 7) / 8 * 8;
    ng = (m + jgroup - 1) / jgroup;
    const dim3 grid(nrb, ng);
    const int round32 = gpz_opts().round_phi32 ? 1 : 0;   // tools/f32_operand_experiment.py
#define PHI_CASE(DD)                                                                                                     \
    do {                                                                                                                 \
        if (diag)                                                                                                        \
            hipLaunchKernelGGL((k_psi32_phi<DD, true>), grid, dim3(256), 0, st, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, lnS, Phi, \
                               ld, jgroup, round32);                                                                              \
        else                                                                                                             \
            hipLaunchKernelGGL((k_psi32_phi<DD, false>), grid, dim3(256), 0, st, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, lnS, Phi, \
                               ld, jgroup, round32);                                                                              \
    } while (0)
    PSI32_CASES(PHI_CASE)
#undef PHI_CASE
    return 0;
}

int launch_psi32_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                         const double *v, const double *Xr, int de, int d, const float *PsiT, long ldp, int diag, int n, int m,
                         const double *P, const double *Sig, const double *Rc, int nchunk, int rows_per_chunk, double *slab,
                         int nrec) {
    const int Dp = psi32_pad_dim(d);
#define MOM_CASE(DD)                                                                                                    \
    do {                                                                                                                \
        if (diag)                                                                                                       \
            hipLaunchKernelGGL((k_psi32_moments<DD, true>), dim3((nchunk + 7) / 8 * 8 * ((m + 7) / 8)), dim3(64), 0, st, Phi, T, ld, \
                               rowscal, w, v, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, nchunk, rows_per_chunk, slab, nrec); \
        else                                                                                                            \
            hipLaunchKernelGGL((k_psi32_moments<DD, false>), dim3((nchunk + 7) / 8 * 8 * ((m + 7) / 8)), dim3(64), 0, st, Phi, T, ld, \
                               rowscal, w, v, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, nchunk, rows_per_chunk, slab, nrec); \
    } while (0)
    PSI32_CASES(MOM_CASE)
#undef MOM_CASE
    return 0;
}
