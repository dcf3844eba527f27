// GC/VC with input noise in fp64 for 10 < d <= 32 (k_cpsi4w.hip: the same templates up to d = 48): FOUR (sample, basis) pairs per wave on
// v_mfma_f64_4x4x4_4b_f64.
//
//   getPHI.m:78-89   ln PHI_ij = -1/2 Delta' M^-1 Delta + 1/2 ln|Sigma_j| - 1/2 ln|M|,   M = Psi_i + Sigma_j
//   GPz.m:164-185    sum_i dPHI_ij * [1, M^-1 Delta, (M^-1 Delta)(M^-1 Delta)' - M^-1]
//
// Same elimination as k_cpsi.hip (symmetric sweep in panels of four pivots, the pivot block through its Cholesky factor), but the
// pair matrix is cut into 4 x 4 TILES, one f64 per lane and tile: the instruction multiplies four independent 4 x 4 x 4 blocks,
// lane = 16 hi + 4 b + lo with b the block (here: the pair), and (measured, tools/mfma_f64_4x4_probe.hip)
//       A operand: lane (hi = k, lo = i) supplies A[i][k]      B operand: lane (hi = k, lo = j) supplies B[k][j]
//       result:    lane (hi = i, lo = j) receives D[i][j]
// so a tile held in the RESULT layout is a B operand as it stands and, used as the A operand, its own transpose:
// mfma(X, Z, C) = C + X'Z on result-layout registers.  Every product of the sweep has that shape - nothing moves between lanes
// except the pivot block (16 doubles per pair through LDS, so that every lane can factorise it) and one transpose per
// pivot-column tile (the lower triangle only is stored):
//       W = inv(L), A11 = L L'                       per lane, 4 pairs per wave (k_cpsi.hip: 1 pair per wave for the same work)
//       Y_J' = W A_1J              = mfma(W', A_1J)          A_1J = tile (p, J), or the transposed tile (J, p) for J > p
//       B_IJ = A_IJ - Y_I Y_J'     = mfma(-Y_I', Y_J', A_IJ)
//       B_1J = W' Y_J'  (J < p)    = mfma(W, Y_J')       B_J1 = Y_J W  (J > p) = mfma(Y_J', W)       B_11 = -W'W = mfma(-W, W)
// Delta rides along as row 0 of an extra tile row (index ND): after the sweep that row is (M^-1 Delta)', its corner -Delta' M^-1 Delta,
// and u u' for the moment sums is one more mfma of two tiles of that row (rows 1..3 are zero).  16 cycles per instruction
// (PMC: the same 32 flop / clock / SIMD as the 16x16x4 form), and the per-panel scalar work (4 x 4 Cholesky, inverse) is shared by four pairs.
// d is padded to 4 ND with identity; missing dimensions are marginalised as in k_psi.hip (identity block in M, zero Delta).
#include "k_cpsi4_impl.h"

bool cpsi4_available(int d) {
    return !gpz_opts().cpsi4_off && d > 10 && d <= 32;   // (developer switch: the 16 x 16 tile kernels of k_cpsi.hip instead)
}

#define CPSI4_CASES(MACRO)          \
    switch ((d + 3) / 4) {          \
        case 3: MACRO(3); break;    \
        case 4: MACRO(4); break;    \
        case 5: MACRO(5); break;    \
        case 6: MACRO(6); break;    \
        case 7: MACRO(7); break;    \
        case 8: MACRO(8); break;    \
        default: return -1;         \
    }

int launch_cpsi4_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                     const double *lnS, double *Phi, int ld, const unsigned char *pat, bool shared) {
    if (!cpsi4_available(d)) return -1;
    if (r.n <= 0) return 0;
#define PHI_LAUNCH(ND, MS, SH)                                                                                                \
    hipLaunchKernelGGL((k_cpsi4_phi<ND, MS, SH>), dim3((r.n + 15) / 16), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, d, P, Sig, lnS, \
                       Phi, ld, (MS) ? r.gid : nullptr, (MS) ? pat : nullptr)
#define PHI_CASE(ND)                                                                                                          \
    do {                                                                                                                      \
        if (pat && shared) PHI_LAUNCH(ND, true, true);                                                                        \
        else if (pat) PHI_LAUNCH(ND, true, false);                                                                            \
        else if (shared) PHI_LAUNCH(ND, false, true);                                                                         \
        else PHI_LAUNCH(ND, false, false);                                                                                    \
    } while (0)
    CPSI4_CASES(PHI_CASE)
#undef PHI_CASE
#undef PHI_LAUNCH
    return 0;
}

int launch_cpsi4_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                         const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                         int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                         const int *chunktab, const double *minv) {
    if (!cpsi4_available(d)) return -1;
    if (nchunk <= 0) return 0;
#define MOM_LAUNCH(ND, MS, SH)                                                                                                \
    hipLaunchKernelGGL((k_cpsi4_moments<ND, MS, SH>), dim3(nchunk, (m + 15) / 16), dim3(256), 0, st, Phi, T, ld, rowscal, w, v,  \
                       r.Xr, de, r.Psi3, r.n, m, d, P, Sig, rows_per_chunk, slab, nrec, (MS) ? r.gid : nullptr,                \
                       (MS) ? pat : nullptr, chunktab, minv)
#define MOM_CASE(ND)                                                                                                          \
    do {                                                                                                                      \
        if (pat && minv) MOM_LAUNCH(ND, true, true);                                                                          \
        else if (pat) MOM_LAUNCH(ND, true, false);                                                                            \
        else if (minv) MOM_LAUNCH(ND, false, true);                                                                           \
        else MOM_LAUNCH(ND, false, false);                                                                                    \
    } while (0)
    CPSI4_CASES(MOM_CASE)
#undef MOM_CASE
#undef MOM_LAUNCH
    return 0;
}

// GC: -inv(Sigma + Psi_i) of every row as 4 x 4 tiles (ND x ND x 16 doubles per row), read back by k_cpsi4_moments<.., SHARED>
size_t cpsi4_minv_len(int d) {
    const int nd = (d + 3) / 4;
    return (size_t)nd * nd * 16;
}
// qA != nullptr (rows without missing dimensions): also the rows of the dense form of the PHI build (k_cpsi4_minv<.., QROW>), lda doubles apart
int launch_cpsi4_minv(hipStream_t st, const GenRows &r, int d, int de, const double *Sig, const double *lnS, const unsigned char *pat,
                      double *minv, double *qA, int lda, const double *ctr) {
    if (!cpsi4_available(d)) return -1;
    if (r.n <= 0) return 0;
#define MINV_LAUNCH(ND, MS, QR)                                                                                               \
    hipLaunchKernelGGL((k_cpsi4_minv<ND, MS, QR>), dim3((r.n + 15) / 16), dim3(256), 0, st, r.Psi3, r.n, d, Sig,               \
                       (MS) ? r.gid : nullptr, (MS) ? pat : nullptr, minv, r.Xr, de, lnS, qA, lda, ctr)
#define MINV_CASE(ND)                                                                                                         \
    do {                                                                                                                      \
        if (pat) MINV_LAUNCH(ND, true, false);                                                                                \
        else if (qA) MINV_LAUNCH(ND, false, true);                                                                            \
        else MINV_LAUNCH(ND, false, false);                                                                                   \
    } while (0)
    CPSI4_CASES(MINV_CASE)
#undef MINV_CASE
#undef MINV_LAUNCH
    return 0;
}

// The dense form expands (x - p)' M^-1 (x - p) into x' M^-1 x - 2 p' M^-1 x + p' M^-1 p, which cancels (|x| / |x - p|)^2 eps of the
// result when the inputs sit far from the origin.  Both sides therefore work on x - c and p - c, c = the mean basis centre.
__global__ void k_gcq_centre(int m, int d, int de, const double *__restrict__ P, double *__restrict__ ctr) {
    const int a = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < m; j += 64) s += P[(size_t)j * de + a];
    s = wave_sum(s);
    if (threadIdx.x == 0 && a < d) ctr[a] = s / m;
}
// The m-sized table of the dense form: B[e][j], e over [q_a q_b (a >= b, packed) ; -2 q_a ; 1] with q = p - c, zero rows up to kpad
__global__ void k_gcq_tab(int m, int d, int de, int kpad, int ldb, const double *__restrict__ P, const double *__restrict__ ctr,
                          double *__restrict__ B) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (j >= ldb || e >= kpad) return;
    const int K1 = d * (d + 1) / 2;
    double v = 0.0;
    if (j < m) {
        if (e < K1) {
            int a = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (a * (a + 1) / 2 > e) --a;
            while ((a + 1) * (a + 2) / 2 <= e) ++a;
            const int b = e - a * (a + 1) / 2;
            v = (P[(size_t)j * de + a] - ctr[a]) * (P[(size_t)j * de + b] - ctr[b]);
        } else if (e < K1 + d) v = -2.0 * (P[(size_t)j * de + (e - K1)] - ctr[e - K1]);
        else if (e == K1 + d) v = 1.0;
    }
    B[(size_t)e * ldb + j] = v;
}
// PHI = exp(-1/2 Q) on the n x m block, Q from the product above
__global__ void k_gcq_exp(const double *__restrict__ Q, int ld, int n, int m, double *__restrict__ Phi) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= m || i >= n) return;
    Phi[(size_t)i * ld + j] = exp(-0.5 * Q[(size_t)i * ld + j]);
}
int gcq_kpad(int d) { return (d * (d + 1) / 2 + d + 1 + 15) / 16 * 16; }
void launch_gcq_centre(hipStream_t st, int m, int d, int de, const double *P, double *ctr) {
    hipLaunchKernelGGL(k_gcq_centre, dim3(d), dim3(64), 0, st, m, d, de, P, ctr);
}
void launch_gcq_tab(hipStream_t st, int m, int d, int de, int ldb, const double *P, const double *ctr, double *B) {
    const int kpad = gcq_kpad(d);
    hipLaunchKernelGGL(k_gcq_tab, dim3((ldb + 255) / 256, kpad), dim3(256), 0, st, m, d, de, kpad, ldb, P, ctr, B);
}
void launch_gcq_exp(hipStream_t st, const double *Q, int ld, int n, int m, double *Phi) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gcq_exp, dim3((m + 255) / 256, n), dim3(256), 0, st, Q, ld, n, m, Phi);
}

int launch_cpsi4_predict_noisy(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                               const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                               long pairs_per_chunk, double *part, bool shared) {
    if (!cpsi4_available(d) || k > 8) return -1;
    if (n <= 0) return 0;
#define PN_CASE(ND)                                                                                                           \
    do {                                                                                                                      \
        if (k == 1 && shared)                                                                                                 \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 1, true>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
        else if (k == 1)                                                                                                      \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 1, false>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
        else if (shared)                                                                                                      \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 8, true>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 8, false>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
    } while (0)
    CPSI4_CASES(PN_CASE)
#undef PN_CASE
    return 0;
}
