// predictNoisy for GC/VC at 32 < d <= 48 on the templates of k_cpsi4_impl.h (ND = 9 .. 12), single output: a translation unit of its
// own so that the unrolled sweeps compile beside the others (see k_cpsi4.hip / k_cpsi4w.hip).  More outputs take the general kernel.
#include "k_cpsi4_impl.h"

int launch_cpsi4w_predict_noisy(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                                const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                                long pairs_per_chunk, double *part, bool shared) {
    if (!cpsi4w_available(d) || k != 1) return -1;
    if (n <= 0) return 0;
#define PN_CASE(ND)                                                                                                           \
    do {                                                                                                                      \
        if (shared)                                                                                                           \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 1, true>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_cpsi4_predict_noisy<ND, 1, false>), dim3((n + 15) / 16, nchunk), dim3(256), 0, st, n, ldx, m, d, \
                               de, k, Xr, Psi3, tab, rec, w, v, iS, pairs_per_chunk, part);                                   \
    } while (0)
    switch ((d + 3) / 4) {
        case 9: PN_CASE(9); break;
        case 10: PN_CASE(10); break;
        case 11: PN_CASE(11); break;
        case 12: PN_CASE(12); break;
        default: return -1;
    }
#undef PN_CASE
    return 0;
}
