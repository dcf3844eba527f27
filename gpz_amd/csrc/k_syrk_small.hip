// S = PHI' * diag(w) * PHI  (GPz.m:63-65) for FEW basis functions: mp = 16 NB <= 256 columns.
//
// k_syrk (k_gemm.hip) works in 128 x 128 output tiles: at mp = 208 (BASELINE config 2) its three tiles cover 49 152 entries of which
// 21 736 are wanted, and every tile re-reads its columns of PHI.  Here the WHOLE upper triangle of 16 x 16 blocks - NB (NB + 1) / 2 of
// them, 91 at mp = 208: exactly the wanted products - lives in the accumulators of ONE workgroup of 8 waves, and the workgroup walks a
// contiguous range of rows (the K dimension), so PHI is read exactly once:
//   * the same fragment serves both operands: for the K step of rows i0 .. i0+3, lane l of block b holds f_b = PHI[i0 + (l >> 4)][16 b +
//     (l & 15)] - as A operand (m = l & 15, k = l >> 4) it is column block b of PHI', as B operand (k = l >> 4, n = l & 15) column block
//     b of PHI.  A wave owns whole block rows of the triangle: row R needs the fragments R .. NB-1 as B and w * f_R as A - one multiply
//     per owned row and K step (the weight rides on the A operand, as in k_syrk: the products are (w phi_i) phi_j);
//   * block rows are dealt so that the eight waves carry the same number of products: with NB > 8 the first 16 - NB waves take rows
//     0 .. 15-NB alone, wave w behind them rows w and its mirror (NB-1 .. ): 2 NB - 15 products each (NB = 13: 13 12 11 11 11 11 11 11;
//     NB = 16: 17 each).  The roles are compile-time (one instantiation of the loop per wave): every accumulator and fragment a fixed
//     register, every LDS address a register plus an immediate, no branch around an MFMA;
//   * rows arrive through LDS-DMA loads (global_load_lds_dwordx4: no registers, no vector ALU) in chunks of 32, double-buffered: the
//     requests of chunk c + 1 go out right after the barrier that opens chunk c and have its 8 K steps to land - ONE barrier per chunk.
//     LDS row stride = 16 (mod 32) doubles: the 16 lanes of a row cover all 32 banks once, rows k and k + 1 the two halves of the 64;
//   * every workgroup leaves its triangle as one record [tile][register][lane] (512-byte stores); k_syrk_small_reduce sums the records
//     in a fixed order and writes S and its mirror (diagonal blocks: the upper half mirrored, so S is exactly symmetric).
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define SS_GLDS(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g), (__attribute__((address_space(3))) void *)(l), 16, 0, 0)
#define SS_ROWS 32        // rows per chunk (8 K steps)

template <int NB>
struct SyrkSmallShape {
    static constexpr int LD = 16 * NB + ((NB & 1) ? 0 : 16);     // LDS row stride in doubles: 16 (mod 32)
    static constexpr int PAIRS = NB > 8 ? NB - 8 : 0;            // waves that carry two block rows
    static constexpr int SINGLES = 8 - PAIRS;
    static constexpr int tile0(int r) { return r * NB - r * (r - 1) / 2; }   // first tile of block row r (row-major over the upper triangle)
};

#ifdef GPZ_SYRK_SMALL_TRACE   // developer builds only (tools/syrk_small_bench.hip): s_memtime stamps per wave
__device__ unsigned long long *g_syrk_small_trace = nullptr;
#define SS_MARK(slot)                                                                                                              \
    do {                                                                                                                           \
        if (g_syrk_small_trace && lane == 0) g_syrk_small_trace[((size_t)blockIdx.x * 8 + W) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SS_MARK(slot) do { } while (0)
#endif

// The loop of wave W (compile-time): K steps [kb, ke) of 4 rows each.
template <int NB, int W>
__device__ __forceinline__ void syrk_small_wave(const double *__restrict__ Phi, int ld, const double *__restrict__ wgt, int n_rows, int kb,
                                                int ke, double *__restrict__ rec, double *smem) {
    using SH = SyrkSmallShape<NB>;
    constexpr int LD = SH::LD;
    constexpr int R1 = W < NB ? W : -1;                                                // (NB < 8: waves NB .. 7 only stage rows)
    constexpr int R2 = (NB > 8 && W >= SH::SINGLES) ? NB - 1 - (W - SH::SINGLES) : -1;
    constexpr int N1 = R1 >= 0 ? NB - R1 : 0, N2 = R2 >= 0 ? NB - R2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    double *sP = smem;                        // [2][32][LD]
    double *sW = smem + 2 * SS_ROWS * LD;     // [2][32]
    d4_t acc1[N1 > 0 ? N1 : 1], acc2[N2 > 0 ? N2 : 1];
#pragma unroll
    for (int b = 0; b < (N1 > 0 ? N1 : 1); ++b) acc1[b] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < (N2 > 0 ? N2 : 1); ++b) acc2[b] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int nk = ke - kb, nch = (nk + 7) >> 3;
    // Staging: the chunk's LDS image (32 rows x LD doubles) is LD / 4 wave-wide requests of 128 doubles; request q = W + 8 j is this wave's,
    // lane l moves the two doubles at flat index 128 q + 2 l = (row, col) - its offset inside the chunk's rows of PHI is the same for
    // every chunk (padding columns re-read the row's last pair).  EVERY request runs with all 64 lanes: a request inside a divergent
    // branch is not safe - the compiler merges differently-masked requests and picks the LDS base of one of them with a
    // v_readfirstlane (seen with a per-row "lanes below the row's end" form of this loop: rows overwritten by their neighbours').
    constexpr int NQ = LD / 4, QW = (NQ + 7) / 8;
    unsigned off[QW], rowq[QW], colq[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const unsigned e = 128u * (unsigned)(W + 8 * j) + 2u * (unsigned)lane;
        rowq[j] = e / (unsigned)LD;
        colq[j] = e % (unsigned)LD;
        if (colq[j] >= 16u * NB) colq[j] = 16u * NB - 2u;
        off[j] = rowq[j] * (unsigned)ld + colq[j];
    }
    auto stage = [&](int ch) {
        const long row0 = 4L * (kb + 8 * ch);
        const double *g0 = Phi + (size_t)row0 * ld;
        double *l0 = sP + (ch & 1) * SS_ROWS * LD;
        if (row0 + SS_ROWS <= (long)n_rows) {
#pragma unroll
            for (int j = 0; j < QW; ++j)
                if (W + 8 * j < NQ) SS_GLDS(g0 + off[j], l0 + 128 * (W + 8 * j));
        } else {   // the matrix ends inside the chunk (wave-uniform): rows past its end read the last row - finite, and never multiplied: the K steps stop at ke
            const unsigned lastr = (unsigned)((long)n_rows - 1 - row0);
#pragma unroll
            for (int j = 0; j < QW; ++j)
                if (W + 8 * j < NQ) SS_GLDS(g0 + ((rowq[j] < lastr ? rowq[j] : lastr) * (unsigned)ld + colq[j]), l0 + 128 * (W + 8 * j));
        }
    };
    const long last = (long)n_rows - 1;
    auto wload = [&](int ch) -> double {
        const long row = 4L * (kb + 8 * ch) + tid;
        return wgt[row < last ? row : last];
    };
    double wreg = 0.0;
    stage(0);
    if (W == 0 && tid < SS_ROWS) {
        sW[tid] = wload(0);
        if (nch > 1) wreg = wload(1);
    }
    const int fo = (lane >> 4) * LD + (lane & 15);       // this lane's place in a K step's four rows
    SS_MARK(0);
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA loads of chunk ch have landed (the compiler does not track them)
        __syncthreads();                                     // ... everyone's have, and everyone has left chunk ch - 1
        if (ch + 1 < nch) {
            stage(ch + 1);
            if (W == 0 && tid < SS_ROWS) {
                sW[((ch + 1) & 1) * SS_ROWS + tid] = wreg;
                if (ch + 2 < nch) wreg = wload(ch + 2);
            }
        }
        if (N1 > 0) {
            const int nkk = nk - 8 * ch < 8 ? nk - 8 * ch : 8;
            const double *pf = sP + (ch & 1) * SS_ROWS * LD + fo + 16 * (R1 > 0 ? R1 : 0);
            const double *pw = sW + (ch & 1) * SS_ROWS + (lane >> 4);
            auto frags = [&](double (&f)[N1 > 0 ? N1 : 1], double &w, int kk) {
                w = pw[4 * kk];
#pragma unroll
                for (int b = 0; b < N1; ++b) f[b] = pf[4 * kk * LD + 16 * b];
            };
            auto burst = [&](const double (&f)[N1 > 0 ? N1 : 1], double w) {
                const double a1 = w * f[0];
                const double a2 = N2 > 0 ? w * f[N2 > 0 ? R2 - R1 : 0] : 0.0;
#pragma unroll
                for (int b = 0; b < N1; ++b) acc1[b] = MFMA_F64(a1, f[b], acc1[b]);
#pragma unroll
                for (int b = 0; b < N2; ++b) acc2[b] = MFMA_F64(a2, f[R2 - R1 + b], acc2[b]);
            };
            // (measured and dropped, tools/syrk_small_bench.hip at c2's shape: fragments of step kk + 1 requested before the products of
            // step kk: 97 -> 106 us; the eight steps of a chunk unrolled: 101 us, and NB = 16 spills; s_setprio around the products: 103 us -
            // the two waves of a SIMD already alternate: one requests and waits while the other's products run)
            double f[N1 > 0 ? N1 : 1], w;
            for (int kk = 0; kk < nkk; ++kk) {
                frags(f, w, kk);
                burst(f, w);
            }
        }
    }
    SS_MARK(1);
    // the record: [tile][register r][lane] - entry (16 bi + (lane >> 4) + 4 r, 16 bj + (lane & 15))
    if (N1 > 0) {
        double *o = rec + (size_t)SH::tile0(R1 > 0 ? R1 : 0) * 256 + lane;
#pragma unroll
        for (int b = 0; b < N1; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[b * 256 + r * 64] = acc1[b][r];
    }
    if (N2 > 0) {
        double *o = rec + (size_t)SH::tile0(R2 > 0 ? R2 : 0) * 256 + lane;
#pragma unroll
        for (int b = 0; b < N2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[b * 256 + r * 64] = acc2[b][r];
    }
    SS_MARK(2);
}

template <int NB>
__global__ __launch_bounds__(512, 1) void k_syrk_small(const double *__restrict__ Phi, int ld, const double *__restrict__ wgt, int n_rows,
                                                        int ksteps_per_wg, double *__restrict__ slab) {
    extern __shared__ double ss_smem[];
    const int ktot = n_rows >> 2;
    const int kb = blockIdx.x * ksteps_per_wg;
    if (kb >= ktot) return;                                   // (whole workgroup: the launch is sized to the rows, see launch_syrk_small)
    const int ke = kb + ksteps_per_wg < ktot ? kb + ksteps_per_wg : ktot;
    double *rec = slab + (size_t)blockIdx.x * (NB * (NB + 1) / 2) * 256;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wv) {
    case 0: syrk_small_wave<NB, 0>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 1: syrk_small_wave<NB, 1>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 2: syrk_small_wave<NB, 2>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 3: syrk_small_wave<NB, 3>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 4: syrk_small_wave<NB, 4>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 5: syrk_small_wave<NB, 5>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    case 6: syrk_small_wave<NB, 6>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    default: syrk_small_wave<NB, 7>(Phi, ld, wgt, n_rows, kb, ke, rec, ss_smem); break;
    }
}

// S (mp x mp, row stride lds) = sum of the nrec records, in a fixed order: a block of 1024 threads takes 64 entries of one tile (one
// accumulator register of its 64 lanes) x 16 groups of records; group g sums records g, g + 16, ... (eight loads in flight), the groups
// are added in order.  Diagonal tiles: entries on and above the diagonal, mirrored.
__global__ __launch_bounds__(1024) void k_syrk_small_reduce(const double *__restrict__ slab, int nrec, int nb, double *__restrict__ S, int lds,
                                                            int accumulate) {
    __shared__ double part[16][64];
    const int t = blockIdx.x >> 2, r = blockIdx.x & 3, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t stride = (size_t)(nb * (nb + 1) / 2) * 256;
    const double *p = slab + (size_t)t * 256 + r * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    int q = g;
    for (; q + 7 * 16 < nrec; q += 8 * 16) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(q + 16 * u) * stride];
        s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3];
        s0 += v[4]; s1 += v[5]; s0 += v[6]; s1 += v[7];
    }
    for (; q < nrec; q += 16) s0 += p[(size_t)q * stride];
    part[g][lane] = s0 + s1;
    __syncthreads();
    if (g) return;
    double tot = part[0][lane];
#pragma unroll
    for (int u = 1; u < 16; ++u) tot += part[u][lane];
    int bi = 0, rem = t;                       // tile t -> (bi, bj), row-major over the upper triangle
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    const int i = 16 * bi + (lane >> 4) + 4 * r, j = 16 * bj + (lane & 15);
    if (i > j) return;
    S[(size_t)i * lds + j] = accumulate ? S[(size_t)i * lds + j] + tot : tot;
    if (i != j) S[(size_t)j * lds + i] = accumulate ? S[(size_t)j * lds + i] + tot : tot;
}

bool syrk_small_fits(int mp) { return mp >= 16 && mp <= 256 && (mp & 15) == 0; }
// Workgroups (= records) of a launch over n_rows rows (a multiple of 4): one per CU, at least one chunk of 32 rows each.
int syrk_small_plan(int n_rows, int *ksteps_per_wg) {
    const int ktot = n_rows >> 2, cu = gpz_cu_count();
    int kpw = (ktot + cu - 1) / cu;
    if (kpw < 8) kpw = 8;
    *ksteps_per_wg = kpw;
    const int nwg = (ktot + kpw - 1) / kpw;
    return nwg > 0 ? nwg : 1;
}
size_t syrk_small_slab_count(int n_rows, int mp) {
    int kpw;
    const int nb = mp / 16;
    return (size_t)syrk_small_plan(n_rows, &kpw) * (nb * (nb + 1) / 2) * 256;
}

template <int NB>
static void launch_syrk_small_nb(hipStream_t st, const double *Phi, int ld, const double *wgt, int n_rows, int nwg, int kpw, double *slab) {
    using SH = SyrkSmallShape<NB>;
    const size_t lds = (size_t)(2 * SS_ROWS * SH::LD + 2 * SS_ROWS) * sizeof(double);
    static bool attr = false;                 // (idempotent: a race between contexts sets the same value twice)
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)k_syrk_small<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(k_syrk_small<NB>, dim3(nwg), dim3(512), lds, st, Phi, ld, wgt, n_rows, kpw, slab);
}

// S = PHI' diag(wgt) PHI over n_rows rows (a multiple of 4; rows >= n carry weight 0), mp = 16 .. 256 columns; slab holds
// syrk_small_slab_count(n_rows, mp) doubles.  accumulate: S += (the next row tile of a streamed evaluation).
void launch_syrk_small(hipStream_t st, const double *Phi, int ld, const double *wgt, int n_rows, int mp, double *slab, double *S, int lds,
                       int accumulate) {
    int kpw;
    const int nwg = syrk_small_plan(n_rows, &kpw), nb = mp / 16;
    switch (nb) {
#define SS_CASE(NBV) case NBV: launch_syrk_small_nb<NBV>(st, Phi, ld, wgt, n_rows, nwg, kpw, slab); break;
        SS_CASE(1) SS_CASE(2) SS_CASE(3) SS_CASE(4) SS_CASE(5) SS_CASE(6) SS_CASE(7) SS_CASE(8)
        SS_CASE(9) SS_CASE(10) SS_CASE(11) SS_CASE(12) SS_CASE(13) SS_CASE(14) SS_CASE(15) SS_CASE(16)
#undef SS_CASE
    default: return;
    }
    hipLaunchKernelGGL(k_syrk_small_reduce, dim3(4 * (nb * (nb + 1) / 2)), dim3(1024), 0, st, slab, nwg, nb, S, lds, accumulate);
}
