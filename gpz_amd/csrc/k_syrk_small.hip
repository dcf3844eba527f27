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
//   * block rows are dealt to the SIMDs in snake order so that every SIMD issues (nearly) the same number of products per K step; a wave
//     carries at most two block rows.  The roles are compile-time (one instantiation of the loop per wave): every accumulator and
//     fragment a fixed register, every LDS address a register plus an immediate, no branch around an MFMA;
//   * rows (and their weights) arrive through LDS-DMA loads (global_load_lds_dwordx4: no registers, no vector ALU) in chunks of 32, in 2 - 4
//     buffers: the requests of chunk c + D go out right after the barrier that opens chunk c - ONE barrier per chunk.
//     LDS row stride = 16 (mod 32) doubles: the 16 lanes of a row cover all 32 banks once, rows k and k + 1 the two halves of the 64;
//   * every workgroup leaves its triangle as one record [tile][register][lane] (512-byte stores); k_syrk_small_reduce sums the records
//     in a fixed order and writes S and its mirror (diagonal blocks: the upper half mirrored, so S is exactly symmetric).
#include "gpz_dev.h"
#include "gpz_kernels.h"

#define SS_ROWS 32        // rows per chunk (8 K steps)
#define SS_GLDS(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g), (__attribute__((address_space(3))) void *)(l), 16, 0, 0)
template <int NB>
struct SyrkSmallShape {
    static constexpr int LD = 16 * NB + ((NB & 1) ? 0 : 16);     // LDS row stride in doubles: 16 (mod 32)
    static constexpr int BUF = SS_ROWS * LD + 128;                // doubles per chunk buffer: the rows of PHI, then the 32 weights (+ the unused lanes' copies)
    // chunk buffers: as many as the 160 KB hold, at most 4 - NB <= 9: 4, NB <= 13: 3, else 2.  A chunk's products are short when NB is
    // small (NB = 8: 3.5 us): ONE chunk of 36 KB per CU in flight bounds the launch at (256 CUs x 36 KB) / memory latency = 2.4 TB/s
    static constexpr int NBUF = (160 * 1024 / 8) / BUF >= 4 ? 4 : (160 * 1024 / 8) / BUF;
    static_assert(NBUF >= 2, "two chunk buffers must fit the LDS");
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
    // Block rows are dealt to the four SIMDs in snake order (row r of the triangle holds NB - r products): SIMD s carries rows s, 7 - s,
    // 8 + s and 15 - s as far as they exist - the same number of products on every SIMD at NB = 8 and 16, within one block row's length
    // of it elsewhere (NB = 13: 24 23 22 22) - split between its two waves s (rows s, 15 - s) and s + 4 (rows 7 - s, 8 + s).
    constexpr int RA = (W >> 2) ? 7 - (W & 3) : (W & 3), RB = (W >> 2) ? 8 + (W & 3) : 15 - (W & 3);
    constexpr int R1 = RA < NB ? RA : -1;                                              // (NB < 8: some waves only stage rows)
    constexpr int R2 = RB < NB ? RB : -1;
    constexpr int N1 = R1 >= 0 ? NB - R1 : 0, N2 = R2 >= 0 ? NB - R2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NBUF = SH::NBUF, D = NBUF - 1, BUF = SH::BUF;   // D chunks are requested ahead of the one being multiplied
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
    constexpr int CNT = (NQ - W + 7) / 8 + (W == 7 ? 1 : 0);      // requests of this wave per chunk (wave 7 also moves the weights)
    const long last2 = (long)n_rows - 2;
    auto stage = [&](int ch, int buf) {
        const long row0 = 4L * (kb + 8 * ch);
        const double *g0 = Phi + (size_t)row0 * ld;
        double *l0 = smem + buf * BUF;
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
        if (W == 7) {   // the chunk's 32 weights: lanes 0 .. 15 two each, the other lanes the same pairs again into the buffer's spare doubles
            const long r = row0 + 2 * (lane & 15);
            SS_GLDS(wgt + (r < last2 ? r : last2), l0 + SS_ROWS * LD);
        }
    };
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < nch) stage(c, c);
    const int fo = (lane >> 4) * LD + (lane & 15);       // this lane's place in a K step's four rows
    SS_MARK(0);
    int bc = 0, bs = D % NBUF;            // buffer of the chunk being multiplied / of the chunk requested next
    for (int ch = 0; ch < nch; ++ch) {
        // this wave's requests of chunk ch have landed (the compiler does not track LDS-DMA loads; they complete in order): all but the
        // D - 1 chunks requested after it - fewer are in flight at the end of the range, where the wait is for everything
        if (ch + D <= nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * CNT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                   // ... everyone's have, and everyone has left chunk ch - 1 (whose buffer the next request takes)
        if (ch + D < nch) stage(ch + D, bs);
        bs = bs + 1 == NBUF ? 0 : bs + 1;
        if (N1 > 0) {
            const int nkk = nk - 8 * ch < 8 ? nk - 8 * ch : 8;
            const double *pf = smem + bc * BUF + fo + 16 * (R1 > 0 ? R1 : 0);
            const double *pw = smem + bc * BUF + SS_ROWS * LD + (lane >> 4);
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
            // step kk - no change, 24 more registers; the eight steps of a chunk unrolled: 101 us, and NB = 16 spills; s_setprio around
            // the products: 103 us.  The loop runs at 0.85 - 0.89 of what its busiest SIMD's MFMA count allows at the clock the kernel gets)
            double f[N1 > 0 ? N1 : 1], w;
            for (int kk = 0; kk < nkk; ++kk) {
                frags(f, w, kk);
                burst(f, w);
            }
        }
        bc = bc + 1 == NBUF ? 0 : bc + 1;
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

// inv(SIGMA) = W' W for the lower-triangular W = inv(L) (mq x mq, zeros above the diagonal; GPz.m:67 through the Cholesky factor), mq <= 512:
// one workgroup per 16 x 16 tile of the upper triangle, its four waves take every fourth K step of rows 16 bj .. mq-1 (the rows above hold
// zeros in column block bj) straight from L2, the four partial tiles are added in a fixed order, the tile and its mirror are written
// (diagonal tiles: the upper half mirrored).  One launch of ~5 us where k_syrk<tri> + k_syrk_reduce were two of 8 + 7.
__global__ __launch_bounds__(256) void k_ltl_small(const double *__restrict__ W, int mq, double *__restrict__ S) {
    __shared__ double red[3][4][64];
    const int nb = mq >> 4, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bi = 0, rem = blockIdx.x;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    const size_t rstep = (size_t)4 * mq;
    const double *pa = W + (size_t)(16 * bj + (lane >> 4)) * mq + 16 * bi + (lane & 15);
    const double *pb = W + (size_t)(16 * bj + (lane >> 4)) * mq + 16 * bj + (lane & 15);
    const int nsteps = (mq - 16 * bj) >> 2;
    d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int s = wv; s < nsteps; s += 4) acc = MFMA_F64(pa[s * rstep], pb[s * rstep], acc);
    if (wv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wv) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        const int i = 16 * bi + (lane >> 4) + 4 * r, j = 16 * bj + (lane & 15);
        if (i <= j) {
            S[(size_t)i * mq + j] = v;
            if (i != j) S[(size_t)j * mq + i] = v;
        }
    }
}
bool ltl_small_fits(int mq) { return mq >= 16 && mq <= 512 && (mq & 15) == 0; }
void launch_ltl_small(hipStream_t st, const double *W, int mq, double *S) {
    const int nb = mq >> 4;
    hipLaunchKernelGGL(k_ltl_small, dim3(nb * (nb + 1) / 2), dim3(256), 0, st, W, mq, S);
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
    const size_t lds = (size_t)SH::NBUF * SH::BUF * sizeof(double);
    // per launch, not once per process: the attribute belongs to the CURRENT device's copy of the kernel (gpz_mgpu drives several devices
    // from one process); a host-side call on the recording / eager path only - a replayed graph does not come through here
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)k_syrk_small<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
