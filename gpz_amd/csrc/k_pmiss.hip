// Prediction for inputs with missing dimensions, diagonal kinds (GL/VL/GD/VD):
//   predictMissing       predictDiag.m:127-209   (no input noise)
//   predictNoisyMissing  predictDiag.m:211-297   (input noise Psi, n x d)
// for one group of rows that share a NaN pattern (predict.m:45-69 groups the rows; predictDiag.m:3 takes the
// pattern from the group's first row).  o = observed dimensions, u = missing ones.
//
// The reference's pair loop costs O(n m^3): for every basis pair (i,j) it forms N = No*Nu' (n x m) and
// EcCij = sum(N.*Pio,2).  That is No .* (Pio*Nu), and over all pairs a GEMM  Pio (n x m) * Nu (m x pairs) — it runs
// on the f64 MFMA T-GEMM kernel here, a chunk of mp pairs per launch; PHI = No .* (Pio*Nij') likewise.
#include "gpz_dev.h"
#include "gpz_kernels.h"

// No(i,j) = exp(-1/2 sum_o (x-p_j)^2/(sigma_j [+psi_i]) - 1/2 sum_o ln(sigma_j [+psi_i]))   predictDiag.m:142-148 / :226-233
// written to No[i*ld + j]; columns j >= m and rows i >= n are zero.
template <typename OBS>
__global__ void k_pm_no(const double *__restrict__ Xr, const double *__restrict__ Psir, int de, int n, long n_pad, int m,
                        int ld, int d, OBS obs, const double *__restrict__ P, const double *__restrict__ G,
                        double *__restrict__ No) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= ld) return;
    double v = 0.0;
    if (j < m && i < n) {
        double q = 0.0, ls = 0.0;
        for (int c = 0; c < d; ++c) {
            if (!obs_bit(obs, c)) continue;
            const double g = G[(size_t)j * de + c];
            double s = 1.0 / (g * g);                                  // Sigma = Gamma.^-2
            if (Psir) s += Psir[(size_t)i * de + c];
            const double dl = Xr[(size_t)i * de + c] - P[(size_t)j * de + c];
            q += dl * dl / s;
            ls += log(s);
        }
        v = exp(-0.5 * q - 0.5 * ls);
    }
    No[(size_t)i * ld + j] = v;
}

// Pio = (No .* priors) ./ rowsum   (predictDiag.m:147-152); one wave per row.
__global__ __launch_bounds__(256) void k_pm_pio(const double *__restrict__ No, int ld, int n, int m,
                                                 const double *__restrict__ priors, double *__restrict__ Pio) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j < m; j += 64) s += No[(size_t)i * ld + j] * priors[j];
    s = wave_sum(s);
    for (int j = lane; j < ld; j += 64) Pio[(size_t)i * ld + j] = (j < m) ? No[(size_t)i * ld + j] * priors[j] / s : 0.0;
}

// B[j*ld + i] = Nij(i,j) = exp(-1/2 sum_u (p_i-p_j)^2/(sigma_i+sigma_j) - 1/2 sum_u ln(sigma_i+sigma_j))   predictDiag.m:158
template <typename OBS>
__global__ void k_pm_nij(int m, int ld, int d, int de, OBS obs, const double *__restrict__ P,
                         const double *__restrict__ G, double *__restrict__ B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= ld) return;
    double v = 0.0;
    if (i < m && j < m) {
        double q = 0.0, ls = 0.0;
        for (int c = 0; c < d; ++c) {
            if (obs_bit(obs, c)) continue;
            const double gi = G[(size_t)i * de + c], gj = G[(size_t)j * de + c];
            const double s = 1.0 / (gi * gi) + 1.0 / (gj * gj);
            const double dl = P[(size_t)i * de + c] - P[(size_t)j * de + c];
            q += dl * dl / s;
            ls += log(s);
        }
        v = exp(-0.5 * q - 0.5 * ls);
    }
    B[(size_t)j * ld + i] = v;
}

// PHI(i,j) = exp(lnz_j) * No(i,j) * T1(i,j),  lnz_j = -1/2 sum_c ln(gamma_jc^2)   predictDiag.m:138,160-161
__global__ void k_pm_phi(const double *__restrict__ No, const double *__restrict__ T1, int ld, int n, long n_pad, int m,
                         int d, int de, const double *__restrict__ G, double *__restrict__ Phi) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= ld) return;
    double v = 0.0;
    if (j < m && i < n) {
        double lz = 0.0;
        for (int c = 0; c < d; ++c) { const double g = G[(size_t)j * de + c]; lz += log(g * g); }
        v = exp(-0.5 * lz) * No[(size_t)i * ld + j] * T1[(size_t)i * ld + j];
    }
    Phi[(size_t)i * ld + j] = v;
}

// pair index q = i(i+1)/2 + j, j <= i
__device__ __forceinline__ void pair_of(long q, int *pi, int *pj) {
    int i = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((long)(i + 1) * (i + 2) / 2 <= q) ++i;
    while ((long)i * (i + 1) / 2 > q) --i;
    *pi = i;
    *pj = (int)(q - (long)i * (i + 1) / 2);
}

// Tables of one chunk of pairs [q0, q0+ld):
//   B[l*ld + qq]  = Nu^{q}_l = exp(-1/2 sum_u (p_l - cij)^2/(sigma_l + Cij) - 1/2 sum_u ln(sigma_l + Cij))     :181-183 / :266-268
//   rec[qq]       = [cij (d) | Cij (d) | lnZ | c2*w_i*w_j (k) | c2*v_i*v_j (k) | c2*iSigma_w(i,j,:) (k)]
//     lnZ = lnz_i + lnz_j - 1/2 sum_c (p_i-p_j)^2/(sigma_i+sigma_j) - 1/2 sum_c ln(sigma_i+sigma_j)            :189 / :274
//           [- 1/2 sum_o ln Cij when there is no input noise: the x-independent part of No,                    :178]
//     c2  = 2 for j < i, 1 for j == i (2x inside the loop, minus 1x for the diagonal term after it,            :191-199)
// Columns past the last pair are zero (B) / c2 = 0 (rec).
// WIDE (d > GPZ_PM_MAXD_DIAG): the tile would not fit a CU's LDS; cij / Cij are formed where they are used (four gathers and a
// division per dimension and row of B instead of two LDS reads: the slow, correct route for inputs of any width).
template <typename OBS, bool WIDE>
__global__ __launch_bounds__(256) void k_pm_pairtab(long q0, long npairs, int m, int ld, int ldb, int d, int de, int k, OBS obs,
                                                    int has_psi, const double *__restrict__ P,
                                                    const double *__restrict__ G, const double *__restrict__ w,
                                                    const double *__restrict__ v, const double *__restrict__ iS,
                                                    double *__restrict__ B, double *__restrict__ rec, int nrec) {
    // 64 pairs per workgroup, lanes along the pairs: a wave writes 512 contiguous bytes of a row of B (one pair per workgroup
    // wrote a column of B, 8 bytes every ldb doubles).  cij / Cij of the 64 pairs sit in LDS as [dimension][pair].
    extern __shared__ double sh[];
    double *cij = sh, *Cij = sh + (size_t)d * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qq = blockIdx.x * 64 + lane;              // pair within the chunk (gridDim.x * 64 = chunk width = ldb)
    const long q = q0 + qq;
    const bool live = q < npairs;
    int i = 0, j = 0;
    if (live) pair_of(q, &i, &j);
    auto pair_c = [&](int c, double *Cv, double *cv) {
        const double gi = G[(size_t)i * de + c], gj = G[(size_t)j * de + c];
        const double isi = gi * gi, isj = gj * gj;
        const double C = 1.0 / (isi + isj);                                      // :173 / :257
        *Cv = live ? C : 1.0;
        *cv = live ? (P[(size_t)i * de + c] * isi + P[(size_t)j * de + c] * isj) * C : 0.0;   // :174 / :258
    };
    if (!WIDE) {
        for (int c = wave; c < d; c += 4) pair_c(c, &Cij[c * 64 + lane], &cij[c * 64 + lane]);
        __syncthreads();
    }
    // blockIdx.y splits the rows of B (64 pairs per workgroup alone would leave the chip to 64 workgroups per chunk)
    const int lper = (ld + gridDim.y - 1) / gridDim.y, l0 = blockIdx.y * lper, l1 = min(ld, l0 + lper);
    for (int l = l0 + wave; l < l1; l += 4) {
        double val = 0.0;
        if (l < m && live) {
            double qd = 0.0, ls = 0.0;
            for (int c = 0; c < d; ++c) {
                if (obs_bit(obs, c)) continue;
                const double gl = G[(size_t)l * de + c];
                double Cv, cv;
                if (WIDE) pair_c(c, &Cv, &cv);
                else { Cv = Cij[c * 64 + lane]; cv = cij[c * 64 + lane]; }
                const double s = 1.0 / (gl * gl) + Cv;
                const double dl = P[(size_t)l * de + c] - cv;
                qd += dl * dl / s;
                ls += log(s);
            }
            val = exp(-0.5 * qd - 0.5 * ls);
        }
        B[(size_t)l * ldb + qq] = val;
    }
    if (wave == 0 && blockIdx.y == 0) {
        double *r = rec + (size_t)qq * nrec;
        if (!live) {
            for (int e = 0; e < nrec; ++e) r[e] = (e >= d && e < 2 * d) ? 1.0 : 0.0;
            return;
        }
        double lz = 0.0, qd = 0.0, ls = 0.0, lo = 0.0;
        for (int c = 0; c < d; ++c) {
            const double gi = G[(size_t)i * de + c], gj = G[(size_t)j * de + c];
            lz += log(gi * gi) + log(gj * gj);
            const double s = 1.0 / (gi * gi) + 1.0 / (gj * gj);
            const double dl = P[(size_t)i * de + c] - P[(size_t)j * de + c];
            qd += dl * dl / s;
            ls += log(s);
            double Cv, cv;
            if (WIDE) pair_c(c, &Cv, &cv);
            else { Cv = Cij[c * 64 + lane]; cv = cij[c * 64 + lane]; }
            if (obs_bit(obs, c) && !has_psi) lo += log(Cv);
            r[c] = cv;
            r[d + c] = Cv;
        }
        r[2 * d] = -0.5 * lz - 0.5 * qd - 0.5 * ls - 0.5 * lo;
        const double c2 = (j < i) ? 2.0 : 1.0;
        for (int o = 0; o < k; ++o) {
            r[2 * d + 1 + o] = c2 * w[i + (size_t)m * o] * w[j + (size_t)m * o];
            r[2 * d + 1 + k + o] = v ? c2 * v[i + (size_t)m * o] * v[j + (size_t)m * o] : 0.0;
            r[2 * d + 1 + 2 * k + o] = c2 * iS[i + (size_t)m * j + (size_t)m * m * o];
        }
    }
}

// sums[3][k][n_pad] += over the pairs of the chunk:  Z = exp(lnZ + lnNo_q(x)) * T2(row, qq)
//   lnNo_q = -1/2 sum_o (x - cij)^2/(Cij [+ psi]) [- 1/2 sum_o ln(Cij + psi)]                                  :176-178 / :260-263
// One WAVE per row, lanes along the pairs of the chunk (T2 is read coalesced, the 3k sums are reduced over the wave at
// the end): a NaN-pattern group is often a few dozen rows, and with one THREAD per row the chunk's 512 pairs were a
// serial chain of 512 exp / divide iterations per launch (1 ms per launch whatever the group's size).
// gridDim.y = S splits of the chunk's pairs, each accumulating into its own slab sums[s][3k][n_pad] (summed in fixed order at
// the end): with a few dozen rows per group one wave per row left the chip empty.
// WIDE (d > GPZ_PM_MAXD): the row's x / psi are read where they are used (wave-uniform addresses) instead of from LDS.
template <typename OBS, bool WIDE>
__global__ __launch_bounds__(256) void k_pm_accum(const double *__restrict__ Xr, const double *__restrict__ Psir, int de,
                                                   int n, long n_pad, int ld, int d, int k, OBS obs, int npq,
                                                   const double *__restrict__ T2, const double *__restrict__ rec, int nrec,
                                                   double *__restrict__ sums_all) {
    double *sums = sums_all + (size_t)blockIdx.y * 3 * k * n_pad;
    const int per = (npq + gridDim.y - 1) / gridDim.y;
    const int qlo = blockIdx.y * per, qhi = min(npq, qlo + per);
    __shared__ double sx[4][WIDE ? 1 : GPZ_PM_MAXD], sps[4][WIDE ? 1 : GPZ_PM_MAXD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + wave;
    const bool live = i < n;
    if (!WIDE) {
        if (live)
            for (int c = lane; c < d; c += 64) {
                sx[wave][c] = Xr[(size_t)i * de + c];
                sps[wave][c] = Psir ? Psir[(size_t)i * de + c] : 0.0;
            }
        __syncthreads();
    }
    if (!live) return;
    // the 3k sums in blocks of 24 (k <= 8: one pass; more outputs walk the pairs again per block)
    for (int e0 = 0; e0 < 3 * k; e0 += 24) {
        double acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.0;
        for (int qq = qlo + lane; qq < qhi; qq += 64) {
            const double *r = rec + (size_t)qq * nrec;
            double qd = 0.0, ls = 0.0;
            for (int c = 0; c < d; ++c) {
                if (!obs_bit(obs, c)) continue;
                const double pv = WIDE ? (Psir ? Psir[(size_t)i * de + c] : 0.0) : sps[wave][c];
                const double xv = WIDE ? Xr[(size_t)i * de + c] : sx[wave][c];
                const double s = r[d + c] + pv;
                const double dl = xv - r[c];
                qd += dl * dl / s;
                if (Psir) ls += log(s);
            }
            const double Z = exp(r[2 * d] - 0.5 * qd - 0.5 * ls) * T2[(size_t)i * ld + qq];
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e0 + e < 3 * k) acc[e] = fma(Z, r[2 * d + 1 + e0 + e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            if (e0 + e >= 3 * k) break;
            double v = acc[e];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) sums[(size_t)(e0 + e) * n_pad + i] += v;
        }
    }
}

template <typename OBS>
static void pm_no_t(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int m, int ld, int d,
                    OBS obs, const double *P, const double *G, const double *priors, double *No, double *Pio) {
    hipLaunchKernelGGL(k_pm_no<OBS>, dim3((ld + 255) / 256, (unsigned)n_pad), dim3(256), 0, st, Xr, Psir, de, n, n_pad, m, ld, d, obs,
                       P, G, No);
    hipLaunchKernelGGL(k_pm_pio, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double *)No, ld, n, m, priors, Pio);
}
void launch_pm_no(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int m, int ld, int d,
                  ObsMask obs, const double *P, const double *G, const double *priors, double *No, double *Pio) {
    pm_no_t(st, Xr, Psir, de, n, n_pad, m, ld, d, obs, P, G, priors, No, Pio);
}
void launch_pm_no(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int m, int ld, int d,
                  ObsFlags obs, const double *P, const double *G, const double *priors, double *No, double *Pio) {
    pm_no_t(st, Xr, Psir, de, n, n_pad, m, ld, d, obs, P, G, priors, No, Pio);
}
void launch_pm_pio(hipStream_t st, const double *No, int ld, int n, int m, const double *priors, double *Pio) {
    hipLaunchKernelGGL(k_pm_pio, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, No, ld, n, m, priors, Pio);
}
void launch_pm_nij(hipStream_t st, int m, int ld, int d, int de, ObsMask obs, const double *P, const double *G, double *B) {
    hipLaunchKernelGGL(k_pm_nij<ObsMask>, dim3((ld + 255) / 256, ld), dim3(256), 0, st, m, ld, d, de, obs, P, G, B);
}
void launch_pm_nij(hipStream_t st, int m, int ld, int d, int de, ObsFlags obs, const double *P, const double *G, double *B) {
    hipLaunchKernelGGL(k_pm_nij<ObsFlags>, dim3((ld + 255) / 256, ld), dim3(256), 0, st, m, ld, d, de, obs, P, G, B);
}
void launch_pm_phi(hipStream_t st, const double *No, const double *T1, int ld, int n, long n_pad, int m, int d, int de,
                   const double *G, double *Phi) {
    hipLaunchKernelGGL(k_pm_phi, dim3((ld + 255) / 256, (unsigned)n_pad), dim3(256), 0, st, No, T1, ld, n, n_pad, m, d, de, G,
                       Phi);
}
// width = number of pairs of the chunk = row stride of B (ld rows: the K dimension of the following T-GEMM)
void launch_pm_pairtab(hipStream_t st, long q0, long npairs, int m, int ld, int width, int d, int de, int k, ObsMask obs,
                       int has_psi, const double *P, const double *G, const double *w, const double *v, const double *iS,
                       double *B, double *rec, int nrec) {
    const size_t lds = (size_t)2 * d * 64 * sizeof(double);   // cij / Cij of 64 pairs: d KB (d <= GPZ_PM_MAXD_DIAG keeps it inside a CU's 160 KB)
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)k_pm_pairtab<ObsMask, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_pm_pairtab<ObsMask, false>), dim3(width / 64, 16), dim3(256), lds, st, q0, npairs, m, ld, width, d, de, k, obs,
                       has_psi, P, G, w, v, iS, B, rec, nrec);
}
void launch_pm_pairtab(hipStream_t st, long q0, long npairs, int m, int ld, int width, int d, int de, int k, ObsFlags obs,
                       int has_psi, const double *P, const double *G, const double *w, const double *v, const double *iS,
                       double *B, double *rec, int nrec) {
    hipLaunchKernelGGL((k_pm_pairtab<ObsFlags, true>), dim3(width / 64, 16), dim3(256), 0, st, q0, npairs, m, ld, width, d, de, k, obs,
                       has_psi, P, G, w, v, iS, B, rec, nrec);
}
// Pair splits of the accumulation kernel.  A constant: the order in which a row's sums are formed must not depend on how many
// rows the call holds (gpz_mgpu_predict cuts a group into row blocks and promises the single-device bits).
int pm_accum_splits(int n) { (void)n; return 16; }
void launch_pm_accum(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int ld, int d, int k,
                     ObsMask obs, int npq, const double *T2, const double *rec, int nrec, double *sums, int nsplit) {
    hipLaunchKernelGGL((k_pm_accum<ObsMask, false>), dim3((unsigned)((n + 3) / 4), nsplit), dim3(256), 0, st, Xr, Psir, de, n, n_pad, ld,
                       d, k, obs, npq, T2, rec, nrec, sums);
}
void launch_pm_accum(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int ld, int d, int k,
                     ObsFlags obs, int npq, const double *T2, const double *rec, int nrec, double *sums, int nsplit) {
    hipLaunchKernelGGL((k_pm_accum<ObsFlags, true>), dim3((unsigned)((n + 3) / 4), nsplit), dim3(256), 0, st, Xr, Psir, de, n, n_pad, ld,
                       d, k, obs, npq, T2, rec, nrec, sums);
}
