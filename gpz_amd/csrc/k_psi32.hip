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
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define LT(r, c) ((r) * ((r) + 1) / 2 + (c))

// In-place Cholesky of a packed lower triangle (column by column, dot-product form); *hl = sum ln L_cc.
template <int D>
__device__ __forceinline__ void chol32(float (&M)[D * (D + 1) / 2], float *hl) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float p = M[LT(c, c)];
#pragma unroll
        for (int q = 0; q < c; ++q) p = fmaf(-M[LT(c, q)], M[LT(c, q)], p);
        const float inv = __builtin_amdgcn_rsqf(p);                 // 1/sqrt(p), 1 ulp
        const float dd = p * inv;
        M[LT(c, c)] = dd;
        s += __logf(dd);
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            float t = M[LT(r, c)];
#pragma unroll
            for (int q = 0; q < c; ++q) t = fmaf(-M[LT(r, q)], M[LT(c, q)], t);
            M[LT(r, c)] = t * inv;
        }
    }
    *hl = s;
}

// L <- inv(L) in place.  First the diagonal is replaced by its reciprocals (every later division becomes a multiply),
// then ascending columns: column c reads only columns >= c of L.
template <int D>
__device__ __forceinline__ void trinv32(float (&L)[D * (D + 1) / 2]) {
#pragma unroll
    for (int c = 0; c < D; ++c) L[LT(c, c)] = __builtin_amdgcn_rcpf(L[LT(c, c)]);
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float wcc = L[LT(c, c)];
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            float s = L[LT(r, c)] * wcc;
#pragma unroll
            for (int q = c + 1; q < r; ++q) s = fmaf(L[LT(r, q)], L[LT(q, c)], s);   // L[q][c] already holds W[q][c]
            L[LT(r, c)] = -s * L[LT(r, r)];
        }
    }
}

// Stage [S | p] of JB basis functions into LDS.  DIAG: S = R_j' packed lower (S(r,c) = R_j[c][r], c <= r) from the QR
// records Rc (k_prep_cov: row a of R at a*de - a(a-1)/2, de = padded dimension of the parameter block); else
// S = Sigma_j packed lower.  Padding dimensions (>= d) are identity.
template <int D, int JB, bool DIAG>
__device__ __forceinline__ void stage_params(int tid, int nt, int j0, int m, int d, int de, const double *__restrict__ Sig,
                                             const double *__restrict__ Rc, const double *__restrict__ P,
                                             float (*sS)[D * (D + 1) / 2], double (*sP)[D]) {
    constexpr int NP = D * (D + 1) / 2;
    for (int e = tid; e < JB * NP; e += nt) {
        const int jj = e / NP, q = e % NP, j = min(j0 + jj, m - 1);
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        const int c = q - r * (r + 1) / 2;
        float v;
        if (r >= d) v = (r == c) ? 1.0f : 0.0f;
        else if (DIAG) v = (float)Rc[(size_t)j * (de * (de + 1) / 2 + de) + (c * de - c * (c - 1) / 2) + (r - c)];   // R[c][r]
        else v = (float)Sig[(size_t)j * d * d + r * d + c];
        (&sS[0][0])[e] = v;
    }
    for (int e = tid; e < JB * D; e += nt) {
        const int jj = e / D, c = e % D, j = min(j0 + jj, m - 1);
        sP[jj][c] = (c < d) ? P[(size_t)j * de + c] : 0.0;
    }
}

// The pair matrix in packed lower form.  DIAG: A = I + R diag(psi) R',  A(r,c) = delta_rc + sum_{k>=r} R[r][k] psi_k R[c][k];
// else M = Sigma_j + Psi_i (getPHI.m:84, GPz.m:170).
template <int D, bool DIAG>
__device__ __forceinline__ void pair_matrix(const float *__restrict__ S, const float (&pd)[DIAG ? D : 1],
                                            const float *__restrict__ PsiT, long ldp, unsigned ic,
                                            float (&M)[D * (D + 1) / 2]) {
    if (DIAG) {
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) M[LT(r, c)] = (r == c) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
#pragma unroll
            for (int r = 0; r <= k; ++r) {
                const float t = S[LT(k, r)] * pd[k];                                   // R[r][k] psi_k
#pragma unroll
                for (int c = 0; c <= r; ++c) M[LT(r, c)] = fmaf(t, S[LT(k, c)], M[LT(r, c)]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) M[LT(r, c)] = S[LT(r, c)] + PsiT[(size_t)LT(r, c) * ldp + ic];
    }
}

// PHI: one thread per row, JB basis functions' parameters staged in LDS at a time; blockIdx.y splits the basis
// functions into groups so that small row counts still fill the chip.
template <int D, bool DIAG>
__global__ __launch_bounds__(256) void k_psi32_phi(const double *__restrict__ Xr, int de, int d,
                                                    const float *__restrict__ PsiT, long ldp, int n, int m,
                                                    const double *__restrict__ P, const double *__restrict__ Sig,
                                                    const double *__restrict__ Rc, const double *__restrict__ lnS,
                                                    double *__restrict__ Phi, int ld, int jgroup) {
    constexpr int NP = D * (D + 1) / 2;
    constexpr int JB = 8;
    __shared__ float sS[JB][NP];
    __shared__ double sP[JB][D];
    __shared__ double sL[JB];
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
            float M[NP];
            pair_matrix<D, DIAG>(sS[jj], pd, PsiT, ldp, ic, M);
            float hl;
            chol32<D>(M, &hl);
            float z[D];
            if (DIAG) {                                                                // z = R Delta
#pragma unroll
                for (int r = 0; r < D; ++r) z[r] = 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float dl = (float)(x[k] - sP[jj][k]);
#pragma unroll
                    for (int r = 0; r <= k; ++r) z[r] = fmaf(sS[jj][LT(k, r)], dl, z[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < D; ++r) z[r] = (float)(x[r] - sP[jj][r]);
            }
            float y[D], quad = 0.f;
#pragma unroll
            for (int r = 0; r < D; ++r) {                                              // y = L^-1 z
                float s = z[r];
#pragma unroll
                for (int c = 0; c < r; ++c) s = fmaf(-M[LT(r, c)], y[c], s);
                y[r] = s * __builtin_amdgcn_rcpf(M[LT(r, r)]);
                quad = fmaf(y[r], y[r], quad);
            }
            // getPHI.m:86:  -1/2 quad + 1/2 ln|Sigma_j| - 1/2 ln|M|;  whitened: the two log-determinants collapse to -1/2 ln|A|
            const double lp = DIAG ? (-0.5 * (double)quad - (double)hl) : (-0.5 * (double)quad + 0.5 * sL[jj] - (double)hl);
            if (act) Phi[(size_t)i * ld + j] = exp(lp);
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
        v[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
        v[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
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
                                                       const double *__restrict__ Rc, int rows_per_chunk,
                                                       double *__restrict__ slab, int nrec) {
    constexpr int NP = D * (D + 1) / 2;
    constexpr int JB = 8;
    constexpr int NV = 3 + D + NP;                     // values reduced per pair: [dp, r1, r2 | dp*u (D) | dp*(uu' - Minv) (NP)]
    constexpr int NG = (NV + 31) / 32;                 // groups of 32 values
    __shared__ float sS[JB][NP];
    __shared__ double sP[JB][D];
    __shared__ double acc[JB][NG * 32];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x, j0 = blockIdx.y * JB;
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
#pragma unroll 1
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            if (j >= m) break;
            const double ph = Phi[(size_t)ic * ld + j], tt = T[(size_t)ic * ld + j];
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
            float L[NP];
            pair_matrix<D, DIAG>(sS[jj], pd, PsiT, ldp, ic, L);
            float hl;
            chol32<D>(L, &hl);
            trinv32<D>(L);                                                             // L now holds W = inv(L)
            float u[D];
            {
                float y[D];
#pragma unroll
                for (int r = 0; r < D; ++r) y[r] = 0.f;
                float z[D];                                                            // z = Delta, or R Delta (whitened)
#pragma unroll
                for (int r = 0; r < D; ++r) z[r] = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) {                                          // Delta formed in fp64
                    const float dl = (c < d) ? (float)(Xr[(size_t)ic * de + c] - sP[jj][c]) : 0.f;
                    if (DIAG) {
#pragma unroll
                        for (int r = 0; r <= c; ++r) z[r] = fmaf(sS[jj][LT(c, r)], dl, z[r]);
                    } else {
                        z[c] = dl;
                    }
                }
#pragma unroll
                for (int c = 0; c < D; ++c) {                                          // y = W z
#pragma unroll
                    for (int r = c; r < D; ++r) y[r] = fmaf(L[LT(r, c)], z[c], y[r]);
                }
#pragma unroll
                for (int a = 0; a < D; ++a) {                                          // u = W' y = M^-1 Delta
                    float s = 0.f;
#pragma unroll
                    for (int q = a; q < D; ++q) s = fmaf(L[LT(q, a)], y[q], s);
                    u[a] = s;
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
                    float mi = 0.f;
#pragma unroll
                    for (int q = a; q < D; ++q) mi = fmaf(L[LT(q, a)], L[LT(q, b)], mi);   // Minv(a,b) = sum_q W[q][a] W[q][b]
                    push(dp * (u[a] * u[b] - mi));                                     // GPz.m:174
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
#define PHI_CASE(DD)                                                                                                     \
    do {                                                                                                                 \
        if (diag)                                                                                                        \
            hipLaunchKernelGGL((k_psi32_phi<DD, true>), grid, dim3(256), 0, st, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, lnS, Phi, \
                               ld, jgroup);                                                                              \
        else                                                                                                             \
            hipLaunchKernelGGL((k_psi32_phi<DD, false>), grid, dim3(256), 0, st, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, lnS, Phi, \
                               ld, jgroup);                                                                              \
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
            hipLaunchKernelGGL((k_psi32_moments<DD, true>), dim3(nchunk, (m + 7) / 8), dim3(64), 0, st, Phi, T, ld, rowscal, w, \
                               v, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, rows_per_chunk, slab, nrec);                  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_psi32_moments<DD, false>), dim3(nchunk, (m + 7) / 8), dim3(64), 0, st, Phi, T, ld, rowscal, \
                               w, v, Xr, de, d, PsiT, ldp, n, m, P, Sig, Rc, rows_per_chunk, slab, nrec);               \
    } while (0)
    PSI32_CASES(MOM_CASE)
#undef MOM_CASE
    return 0;
}
