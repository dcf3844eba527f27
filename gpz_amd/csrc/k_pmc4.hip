// Prediction with missing values, GC/VC, 10 < d <= 32: the two record sums of k_pmiss_cov.hip (predictCov.m:178-207,190-201 / :276-318,
// :300-313) on the four-per-wave 4 x 4-tile sweeps of k_cpsi4_impl.h,
//     out(row, r) = exp(lnZ_r) * sum_l N(X_hat(row,l) - c_r ; C_r + Psi_hat_l(row)) * Pio(row,l)
// over a table of records r = [C_r (d x d) | c_r (d) | lnZ_r | weights (nw)] (the basis pairs with their 3k weights, or the basis
// functions themselves with no weights: PHI).  The four blocks of a wave take four components l.
//   k_pmc4_sum_n  input noise: Psi_hat depends on (row, component): one sweep without the inverse per (row, record, component)
//   k_pmc4_sum_s  no input noise: Psi_hat_l is CU_l whatever the row is: one sweep WITH the inverse per (record, component), then
//                 Delta' M^-1 Delta per row as ND(ND+1)/2 tile products (as the GC branch of k_cpsi4_predict_noisy)
// Layouts as the register-resident kernels (k_pmc_sum_s / k_pmc_sum_n): X_hat [row][d][m], Psi_hat packed lower [row][d(d+1)/2][m],
// CUT [d(d+1)/2][m].
#include "k_cpsi4_impl.h"

#define PLT4(r, c) ((r) * ((r) + 1) / 2 + (c))

template <int ND>
__global__ __launch_bounds__(64) void k_pmc4_sum_n(int nrows, int row0, int m, int d, int ld, long R, long rec_per_chunk,
                                                    const double *__restrict__ tab, int ntab, int nw,
                                                    const double *__restrict__ Pio, const double *__restrict__ XhT,
                                                    const double *__restrict__ PhT, double *__restrict__ Phi, long ldx,
                                                    double *__restrict__ part) {
    __shared__ double ex[64];
    const C4Lane L = c4_lane();
    const int rr = blockIdx.x, chunk = blockIdx.y, np = d * (d + 1) / 2;
    const long r0 = (long)chunk * rec_per_chunk, r1 = min(R, r0 + rec_per_chunk);
    double acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.0;
    for (long r = r0; r < r1; ++r) {
        const double *t = tab + (size_t)r * ntab;
        double ec = 0.0;
        for (int l0 = 0; l0 < m; l0 += 4) {
            const int l = l0 + L.b, lc = min(l, m - 1);
            double T[C4_NT(ND)];
#pragma unroll
            for (int I = 0; I < ND; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J) {
                    const int row = 4 * I + L.hi, col = 4 * J + L.lo;
                    const bool k = row < d && col < d;
                    const int rc = min(row, d - 1), cc = min(col, d - 1), hi_ = max(rc, cc), lo_ = min(rc, cc);
                    const double sv = t[rc * d + cc] + PhT[((size_t)rr * np + PLT4(hi_, lo_)) * m + lc];
                    T[c4_lt(I, J)] = k ? sv : ((row == col) ? 1.0 : 0.0);
                }
#pragma unroll
            for (int J = 0; J < ND; ++J) {
                const int col = 4 * J + L.lo, cc = min(col, d - 1);
                const double dv = XhT[((size_t)rr * d + cc) * m + lc] - t[d * d + cc];
                T[c4_lt(ND, J)] = (L.hi == 0 && col < d) ? dv : 0.0;
            }
            T[c4_lt(ND, ND)] = 0.0;
            double logdet;
            c4_sweep<ND, false>(T, ex, L, &logdet);
            const double quad = -__shfl(T[c4_lt(ND, ND)], 4 * L.b, 64);
            if (l < m) ec += exp(-0.5 * quad - 0.5 * logdet) * Pio[(size_t)rr * ld + l];
        }
        ec = __shfl(ec, 0, 64) + __shfl(ec, 4, 64) + __shfl(ec, 8, 64) + __shfl(ec, 12, 64);   // the four components of a wave
        const double Z = exp(t[d * d + d]) * ec;
        if (nw == 0 && L.lane == 0) Phi[(size_t)(row0 + rr) * ld + r] = Z;
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) acc[e] = fma(Z, t[d * d + d + 1 + e], acc[e]);
    }
    if (nw > 0 && L.lane == 0) {
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) part[((size_t)chunk * nw + e) * ldx + row0 + rr] = acc[e];
    }
}

template <int ND>
__global__ __launch_bounds__(64) void k_pmc4_sum_s(int nrows, int row0, int m, int d, int ld, long R, long rec_per_chunk,
                                                    const double *__restrict__ tab, int ntab, int nw,
                                                    const double *__restrict__ Pio, const double *__restrict__ XhT,
                                                    const double *__restrict__ CUT, double *__restrict__ Phi, long ldx,
                                                    double *__restrict__ part) {
    __shared__ double ex[64];
    __shared__ double ecs[64 * 4];
    const C4Lane L = c4_lane();
    const int chunk = blockIdx.x;
    const long r0 = (long)chunk * rec_per_chunk, r1 = min(R, r0 + rec_per_chunk);
    double acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.0;
    for (long r = r0; r < r1; ++r) {
        const double *t = tab + (size_t)r * ntab;
        for (int e = L.lane; e < 64 * 4; e += 64) ecs[e] = 0.0;
        for (int l0 = 0; l0 < m; l0 += 4) {
            const int l = l0 + L.b, lc = min(l, m - 1);
            double S[C4_NT(ND)];
#pragma unroll
            for (int I = 0; I < ND; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J) {
                    const int row = 4 * I + L.hi, col = 4 * J + L.lo;
                    const bool k = row < d && col < d;
                    const int rc = min(row, d - 1), cc = min(col, d - 1), hi_ = max(rc, cc), lo_ = min(rc, cc);
                    const double sv = t[rc * d + cc] + CUT[(size_t)PLT4(hi_, lo_) * m + lc];
                    S[c4_lt(I, J)] = k ? sv : ((row == col) ? 1.0 : 0.0);
                }
#pragma unroll
            for (int J = 0; J <= ND; ++J) S[c4_lt(ND, J)] = 0.0;
            double logdet;
            c4_sweep<ND, true>(S, ex, L, &logdet);                       // S = -M^-1
#pragma unroll
            for (int I = 0; I < ND; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J) S[c4_lt(I, J)] *= (I == J) ? -1.0 : -2.0;
            const double prd = exp(-0.5 * logdet);
            for (int rr = 0; rr < nrows; ++rr) {
                double qq = 0.0;
                double dc[ND];
#pragma unroll
                for (int I = 0; I < ND; ++I) {
                    const int rw = 4 * I + L.hi, rc = min(rw, d - 1);
                    const double dv = XhT[((size_t)rr * d + rc) * m + lc] - t[d * d + rc];
                    dc[I] = (L.lo == 0 && rw < d) ? dv : 0.0;            // Delta_I as column 0 of a tile
                }
#pragma unroll
                for (int J = 0; J < ND; ++J) {
                    const int col = 4 * J + L.lo, cc = min(col, d - 1);
                    const double dv = XhT[((size_t)rr * d + cc) * m + lc] - t[d * d + cc];
                    const double dr = (L.hi == 0 && col < d) ? dv : 0.0;  // Delta_J' as row 0
                    double h = 0.0;
#pragma unroll
                    for (int I = J; I < ND; ++I) h = MFMA4(dc[I], S[c4_lt(I, J)], h);
                    qq = fma(h, dr, qq);
                }
                qq += __shfl_xor(qq, 1, 64);
                qq += __shfl_xor(qq, 2, 64);
                if (L.hi == 0 && L.lo == 0 && l < m) ecs[rr * 4 + L.b] += exp(-0.5 * qq) * prd * Pio[(size_t)rr * ld + l];
            }
        }
        __syncthreads();
        if (L.lane < nrows) {                      // lanes re-dealt along the rows
            const double ec = (ecs[L.lane * 4] + ecs[L.lane * 4 + 1]) + (ecs[L.lane * 4 + 2] + ecs[L.lane * 4 + 3]);
            const double Z = exp(t[d * d + d]) * ec;
            if (nw == 0) Phi[(size_t)(row0 + L.lane) * ld + r] = Z;
#pragma unroll
            for (int e = 0; e < 24; ++e)
                if (e < nw) acc[e] = fma(Z, t[d * d + d + 1 + e], acc[e]);
        }
        __syncthreads();
    }
    if (nw > 0 && L.lane < nrows) {
#pragma unroll
        for (int e = 0; e < 24; ++e)
            if (e < nw) part[((size_t)chunk * nw + e) * ldx + row0 + L.lane] = acc[e];
    }
}

bool pmc4_available(int d) { return d > 10 && d <= 32; }

bool launch_pmc4_sum(hipStream_t st, int d, bool noisy, int nrows, int row0, int m, int ld, long R, int nchunk, const double *tab,
                     int ntab, int nw, const double *Pio, const double *XhT, const double *PsT, double *Phi, long ldx,
                     double *part) {
    if (!pmc4_available(d)) return false;
    const long rpc = (R + nchunk - 1) / nchunk;
    const int nch = (int)((R + rpc - 1) / rpc);
#define PMC4_CASE(ND)                                                                                                         \
    case ND:                                                                                                                  \
        if (noisy)                                                                                                            \
            hipLaunchKernelGGL((k_pmc4_sum_n<ND>), dim3(nrows, nch), dim3(64), 0, st, nrows, row0, m, d, ld, R, rpc, tab, ntab, nw, \
                               Pio, XhT, PsT, Phi, ldx, part);                                                                \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_pmc4_sum_s<ND>), dim3(nch), dim3(64), 0, st, nrows, row0, m, d, ld, R, rpc, tab, ntab, nw, Pio, \
                               XhT, PsT, Phi, ldx, part);                                                                     \
        return true;
    switch ((d + 3) / 4) {
        PMC4_CASE(3) PMC4_CASE(4) PMC4_CASE(5) PMC4_CASE(6) PMC4_CASE(7) PMC4_CASE(8)
        default: return false;
    }
#undef PMC4_CASE
}
