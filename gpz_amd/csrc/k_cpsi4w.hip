// The kernels of k_cpsi4.hip for 32 < d <= 48 (ND = 9 .. 12 tiles per dimension): a translation unit of its own so that the unrolled
// sweeps compile beside the others.  One wave per SIMD, tiles and sums beyond 512 registers spill (moments: 0.5 - 2 KB of scratch per
// lane) - still three times faster than the one-pair-per-wave kernels of k_cpsi.hip, which keep the rows with missing values here.
#include "k_cpsi4_impl.h"

bool cpsi4w_available(int d) {
    return !gpz_opts().cpsi4_off && d > 32 && d <= 48;
}

#define CPSI4W_CASES(MACRO)         \
    switch ((d + 3) / 4) {          \
        case 9: MACRO(9); break;    \
        case 10: MACRO(10); break;  \
        case 11: MACRO(11); break;  \
        case 12: MACRO(12); break;  \
        default: return -1;         \
    }

int launch_cpsi4w_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                      const double *lnS, double *Phi, int ld) {
    if (!cpsi4w_available(d)) return -1;
    if (r.n <= 0) return 0;
#define PHI_CASE(ND)                                                                                                          \
    hipLaunchKernelGGL((k_cpsi4_phi<ND, false>), dim3((r.n + 15) / 16), dim3(256), 0, st, r.Xr, de, r.Psi3, r.n, m, d, P, Sig, lnS, \
                       Phi, ld, nullptr, nullptr)
    CPSI4W_CASES(PHI_CASE)
#undef PHI_CASE
    return 0;
}

int launch_cpsi4w_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                          const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                          int nchunk, int rows_per_chunk, double *slab, int nrec, const int *chunktab) {
    if (!cpsi4w_available(d)) return -1;
    if (nchunk <= 0) return 0;
#define MOM_CASE(ND)                                                                                                          \
    hipLaunchKernelGGL((k_cpsi4_moments<ND, false>), dim3(nchunk, (m + 15) / 16), dim3(256), 0, st, Phi, T, ld, rowscal, w, v,   \
                       r.Xr, de, r.Psi3, r.n, m, d, P, Sig, rows_per_chunk, slab, nrec, nullptr, nullptr, chunktab, nullptr)
    CPSI4W_CASES(MOM_CASE)
#undef MOM_CASE
    return 0;
}
