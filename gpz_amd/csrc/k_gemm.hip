// f64 MFMA contractions of the GPz objective/gradient path (gfx950, v_mfma_f64_16x16x4_f64).
//
//   k_syrk        S = PHI' * diag(w) * PHI   (GPz.m:63-65, upper 128x128 tiles, split over rows)
//   k_syrk_reduce sum the row-split slabs, mirror to the lower triangle
//   k_tgemm       T = PHI * B,  B = [inv(SIGMA) | w]   (GPz.m:69,72,77 in one product)
//   k_trtri_level one level of the recursive triangular inverse (64x64 tile GEMMs on the structurally non-zero K range)
//
// Tiling: a workgroup is 8 waves (2 x 4), each wave owns a 64x32 block of the 128x128 output tile as 4x2 MFMA tiles
// (8 accumulators x 4 f64 = 64 accumulator registers; 128 VGPRs per lane, four waves per SIMD with two workgroups per
// CU).  Operands are staged global -> registers -> LDS in 16-deep K slices, double buffered, one barrier per slice.
// The K loops keep the vector ALU for the MFMAs: a VALU instruction of any wave takes issue cycles the MFMA pipe of
// that SIMD does not get back (tools/mfma_f64_operands.hip), so addresses live in SGPR base pointers, loop-invariant
// offsets and immediates.
#include <type_traits>
#include "gpz_dev.h"
#include "gpz_kernels.h"

// Operand type of a contraction.  double: v_mfma_f64_16x16x4_f64, accumulators in fp64.  float (dtype = f32 with the fp32
// pair kernels active, BASELINE config 5): the operands are rounded to fp32 while they are staged into LDS (half the LDS
// traffic) and multiplied on v_mfma_f32_16x16x4_f32 (same 16x16x4 fragment layout, twice the rate); the fp32 accumulators
// are FLUSHED into fp64 accumulators every GPZ_F32_FLUSH K slices (16 * GPZ_F32_FLUSH products per element), so the
// accumulation error stays at the level of the operand rounding (tools/f32_operand_experiment.py: rounding PHI and
// [inv(SIGMA)|w] to fp32 moves f by 2.4e-10 and g by 3.6e-7 of max|g| on a 250 000-row shard of config 5).
#ifdef GPZ_GEMM_TRACE   // developer builds only (tools/gemm_trace.hip): per-wave timestamps of the K loop of k_tgemm
__device__ unsigned long long *g_gemm_trace = nullptr;
#define GPZ_TRACE_MARK(slot)                                                                                              \
    do {                                                                                                                  \
        if (g_gemm_trace && (threadIdx.x & 63) == 0)                                                                      \
            g_gemm_trace[((size_t)item * 8 + (threadIdx.x >> 6)) * 6 + (slot)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define GPZ_TRACE_MARK(slot) do { } while (0)
#endif
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
#ifndef GPZ_F32_FLUSH
#define GPZ_F32_FLUSH 8
#endif
template <typename OT> struct MfmaOf;
template <> struct MfmaOf<double> {
    typedef d4_t acc_t;
    typedef d2_t pair_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return MFMA_F64(a, b, c); }
};
template <> struct MfmaOf<float> {
    typedef f4_t acc_t;
    typedef f2_t pair_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <typename OT>
__device__ __forceinline__ typename MfmaOf<OT>::pair_t to_pair(const f2_t v) {   // an operand stored in fp32 (B of the fp32-operand T-GEMM)
    typename MfmaOf<OT>::pair_t r;
    r.x = (OT)v.x;
    r.y = (OT)v.y;
    return r;
}
template <typename BT> struct Pair2 { typedef d2_t t; };
template <> struct Pair2<float> { typedef f2_t t; };
template <typename OT>
__device__ __forceinline__ typename MfmaOf<OT>::pair_t to_pair(const d2_t v) {
    typename MfmaOf<OT>::pair_t r;
    r.x = (OT)v.x;
    r.y = (OT)v.y;
    return r;
}

// ---------------------------------------------------------------------------------------------
// S = PHI' W PHI
// ---------------------------------------------------------------------------------------------
// PHI: n_rows x ld row-major.  Output slab[split] is mp x mp row-major; only tiles (ti <= tj) are written.
// TRI: rows < tj*128 contribute nothing (PHI is lower triangular: used for inv(L)' * inv(L)).
// WC = wave columns of the 2 x WC wave grid: WC = 2 -> 4 waves of 64x64 each, WC = 4 -> 8 waves of 64x32 each
// (64 accumulator registers per wave, 4 waves per SIMD with two workgroups per CU).
template <bool WEIGHTED, int WC, bool DIAGT, typename OT>
__device__ __forceinline__ void syrk_body(const double *__restrict__ Phi, int ld, const double *__restrict__ wgt,
                                          int mp, int i0, int j0, int r_begin, int r_end,
                                          double *__restrict__ out, OT (*sA)[16][LDS_LD128],
                                          OT (*sB)[16][LDS_LD128], OT (*sW)[16]) {
    constexpr int NT = 128 * WC;          // threads
    constexpr int NI = 8 / WC;            // 16-column MFMA tiles per wave
    constexpr int Q = 1024 / NT;          // double2 per thread per operand slice (16 x 128 doubles)
    constexpr bool diag_tile = DIAGT;
    // The A operand is scaled by the row weight once, while it is staged (Q multiplies per thread and slice; f64 VALU multiplies
    // run on the MFMA pipe, a fragment scaled in the MFMA loop costs one per fragment and K step).  Diagonal tiles stage the one
    // operand they load twice: scaled into sA, as it is into sB.
    constexpr bool PRESCALE = WEIGHTED;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;

    // fp32 operands: K = rows here (thousands of POSITIVE products per element), where an fp32 running sum loses
    // eps32 * sqrt(K/3) - far more than the operand rounding, which averages out.  The fp32 accumulators are therefore
    // flushed into fp64 master sums every GPZ_F32_FLUSH slices (128 rows); the master costs 64 more registers, so this
    // variant runs two waves per SIMD instead of four.
    constexpr bool F32 = std::is_same<OT, float>::value;
    typedef typename MfmaOf<OT>::acc_t acc_t;
    typedef typename MfmaOf<OT>::pair_t pair_t;
    acc_t acc[4][NI];
    d4_t acc64[F32 ? 4 : 1][F32 ? NI : 1];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            acc[a][b] = (acc_t){0, 0, 0, 0};
            if (F32) acc64[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};
        }
    auto flush = [&]() {
        if constexpr (F32) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc64[a][b][r] += (double)acc[a][b][r];
                    acc[a][b] = (acc_t){0, 0, 0, 0};
                }
        }
    };

    // staging map: a wave reads one full 1 KiB tile row per q
    d2_t ra[Q], rb[Q];
    double rw = 0.0, rwq[PRESCALE ? Q : 1];
    // Edge tiles (mp not a multiple of 128): columns >= mp are read from the last valid column pair instead of being
    // zero-filled; they only feed accumulators whose rows / columns the guarded store below drops, so the K loop is
    // the same for every tile.  (c is the same for every q: NT is a multiple of 64.)
    const int ca = min(i0 + (tid & 63) * 2, mp - 2), cb = min(j0 + (tid & 63) * 2, mp - 2);
    // No address arithmetic on the vector ALU inside the K loop (see tgemm_body): wave-uniform base pointers advanced by
    // scalar adds, loop-invariant 32-bit byte offsets, LDS addresses as register + immediate (loop unrolled over the
    // two buffers).
    const unsigned voa = (unsigned)(((tid >> 6) * ld + ca) * 8), vob = (unsigned)(((tid >> 6) * ld + cb) * 8);
    const unsigned vow = (unsigned)((PRESCALE ? (tid >> 6) : (tid & 15)) * 8);
    const size_t qsp = (size_t)(NT / 64) * ld * 8, ksp = (size_t)16 * ld * 8;
    const char *pp = reinterpret_cast<const char *>(Phi + (size_t)r_begin * ld);
    const char *pw = reinterpret_cast<const char *>(WEIGHTED ? wgt + r_begin : nullptr);
    auto gload = [&]() {   // next 16-row slice
        unsigned oa = voa, ob = vob, ow = vow;
        asm volatile("" : "+v"(oa), "+v"(ob), "+v"(ow));   // keeps the offset extension next to the loads (SGPR-base mode)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            ra[q] = *reinterpret_cast<const d2_t *>(pp + q * qsp + oa);
            if (!diag_tile) rb[q] = *reinterpret_cast<const d2_t *>(pp + q * qsp + ob);
            if (PRESCALE) rwq[q] = *reinterpret_cast<const double *>(pw + q * (NT / 64) * 8 + ow);
        }
        if (WEIGHTED && !PRESCALE && tid < 16) rw = *reinterpret_cast<const double *>(pw + ow);
        pp += ksp;
        if (WEIGHTED) pw += 16 * 8;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            int idx = q * NT + tid;
            int row = idx >> 6, c = (idx & 63) * 2;
            if (diag_tile) *reinterpret_cast<pair_t *>(&sB[buf][row][c]) = to_pair<OT>(ra[q]);
            else *reinterpret_cast<pair_t *>(&sB[buf][row][c]) = to_pair<OT>(rb[q]);
            if (PRESCALE) { ra[q].x *= rwq[q]; ra[q].y *= rwq[q]; }
            *reinterpret_cast<pair_t *>(&sA[buf][row][c]) = to_pair<OT>(ra[q]);
        }
        if (WEIGHTED && !PRESCALE && tid < 16) sW[buf][tid] = (OT)rw;
    };

    // Software pipeline: the global loads of slice s+2 are issued in the middle of slice s, right after slice s+1
    // has been written to the other LDS buffer, so every load has a full slice of MFMA work to land and the
    // LDS writes sit between the two MFMA halves instead of in front of the barrier.
    const int nstage = (r_end > r_begin) ? (r_end - r_begin) / 16 : 0;
    if (nstage > 0) {
        gload();
        lstore(0);
        if (nstage > 1) gload();
    }
    __syncthreads();
    // Fragments one K step ahead of the MFMA burst, barrier in front of a slice's last burst (see tgemm_body).
    // Diagonal tiles: only the 36 16x16 products on and above the diagonal of the 8 x 8 grid are wanted (the 64x64-block deal of
    // rounds 1-3 issued 48: both diagonal blocks in full).  They are dealt so that every SIMD carries NINE per K step: wave s
    // (s = 0..3) takes five products of grid row s, wave s + 4 - its partner on SIMD s - the rest of row s and all of row 7 - s
    // (rows 0 / 7, 1 / 6, 2 / 5, 3 / 4 hold 8 + 1, 7 + 2, 6 + 3, 5 + 4 products).  Per product one A and one B fragment, each a
    // loop-invariant LDS address; the first product's A fragment serves the fifth (five-product waves stay in one row).
    // A diagonal workgroup costs 9/16 of an off-diagonal one per row (was 12/16): k_syrk splits the rows accordingly.
    static_assert(!DIAGT || NI == 2, "diagonal-tile wave roles are written for the 2 x 4 wave grid");
    constexpr int NA = 4, NB = DIAGT ? 5 : NI;   // fragment registers
    int dra[5] = {0, 0, 0, 0, 0}, dcb[5] = {0, 0, 0, 0, 0};
    bool has5 = false;
    if (DIAGT) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // grid row / column of product p of wave w, 4 bits each (product 4 of the four-product waves repeats product 3: not issued)
        const unsigned RA[8] = {0x00000u, 0x11111u, 0x22222u, 0x33333u, 0x77000u, 0x66611u, 0x55552u, 0x44444u};
        const unsigned CB[8] = {0x43210u, 0x54321u, 0x65432u, 0x76543u, 0x77765u, 0x77676u, 0x77657u, 0x77654u};
        has5 = wv < 4;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            dra[p] = (RA[wv] >> (4 * p)) & 15;
            dcb[p] = (CB[wv] >> (4 * p)) & 15;
        }
    }
    auto rdfrag = [&](int cur, int kk, OT (&a)[NA], OT (&b)[NB]) {
        const OT(*tA)[LDS_LD128] = sA[cur];
        const OT(*tB)[LDS_LD128] = sB[cur];
        const int krow = kk * 4 + (lane >> 4);
        if (DIAGT) {
#pragma unroll
            for (int p = 0; p < 4; ++p) a[p] = tA[krow][dra[p] * 16 + (lane & 15)];
#pragma unroll
            for (int p = 0; p < 5; ++p) b[p] = tB[krow][dcb[p] * 16 + (lane & 15)];
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = tA[krow][wr * 64 + mi * 16 + (lane & 15)];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = tB[krow][wc * (16 * NI) + ni * 16 + (lane & 15)];
        }
    };
    auto burst = [&](const OT (&a)[NA], const OT (&b)[NB]) {
        __builtin_amdgcn_s_setprio(1);   // see tgemm_body
        if (DIAGT) {
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p][0] = MfmaOf<OT>::run(a[p], b[p], acc[p][0]);
            if (has5) acc[0][1] = MfmaOf<OT>::run(a[0], b[DIAGT ? 4 : 0], acc[0][1]);
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = MfmaOf<OT>::run(a[mi], b[ni], acc[mi][ni]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    OT fa0[NA] = {}, fb0[NB] = {}, fa1[NA] = {}, fb1[NB] = {};
    if (nstage > 0) rdfrag(0, 0, fa0, fb0);
    auto stage = [&](auto curc, int s) {
        constexpr int cur = decltype(curc)::value;
        rdfrag(cur, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        rdfrag(cur, 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nstage) {
            lstore(cur ^ 1);
            if (s + 2 < nstage) gload();
        }
        rdfrag(cur, 3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (s + 1 < nstage) rdfrag(cur ^ 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int s = 0;
    for (; s + 1 < nstage; s += 2) {
        stage(std::integral_constant<int, 0>{}, s);
        stage(std::integral_constant<int, 1>{}, s + 1);
        if (F32 && ((s + 2) % GPZ_F32_FLUSH) == 0) flush();
    }
    if (s < nstage) stage(std::integral_constant<int, 0>{}, s);
    flush();
    auto res = [&](int mi, int ni, int r) -> double {
        if constexpr (F32) return acc64[mi][ni][r];
        else return acc[mi][ni][r];
    };
    // row of accumulator register r inside a 16x16 tile: the f64 instruction deals rows (lane >> 4) + 4r, the f32 one 4(lane >> 4) + r
    auto crow = [&](int r) -> int { return F32 ? 4 * (lane >> 4) + r : (lane >> 4) + 4 * r; };

    if (DIAGT) {
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            if (p == 4 && !has5) break;
            const int col = j0 + dcb[p] * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + dra[p] * 16 + crow(r);
                if (row < mp && col < mp) out[(size_t)row * mp + col] = p < 4 ? res(p, 0, r) : res(0, 1, r);
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = j0 + wc * (16 * NI) + ni * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + wr * 64 + mi * 16 + crow(r);
                    if (row < mp && col < mp) out[(size_t)row * mp + col] = res(mi, ni, r);
                }
            }
    }
}

template <bool WEIGHTED, bool TRI, int WC, typename OT>
__global__ __launch_bounds__(128 * WC, (std::is_same<OT, float>::value ? 2 : WC)) void k_syrk(
    const double *__restrict__ Phi, int ld, const double *__restrict__ wgt, int n_rows, int mp, int ntile, int nsplit,
    int rows_per_split, int nsplit_d, int rows_per_split_d, double *__restrict__ slab) {
    __shared__ OT sA[2][16][LDS_LD128];
    __shared__ OT sB[2][16][LDS_LD128];
    __shared__ OT sW[2][16];

    const int npairs = ntile * (ntile + 1) / 2;
    // XCD-aware remap (see k_tgemm): the tiles of one row split run on one XCD and walk the same 16-row slices of
    // PHI at the same time, so a slice is fetched once per split instead of once per tile.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    // Off-diagonal tiles take nsplit row ranges of rows_per_split rows; diagonal tiles cost 3/4 as much per row and
    // take nsplit_d longer ranges (rows_per_split_d).  Enumeration: group g = the noff off-diagonal tiles of row range
    // g followed by an even share of the ntile*nsplit_d diagonal workgroups, so every contiguous stretch of logical
    // blocks (= every XCD) carries the same mix.
    const int noff = npairs - ntile, ndiag = ntile * nsplit_d;
    auto gstart = [&](int g) { return g * noff + g * ndiag / nsplit; };   // all products < 2^31 (grid < 2^16 groups)
    int g = (int)((unsigned)lb * (unsigned)nsplit / (unsigned)(noff * nsplit + ndiag));
    g = g > nsplit - 1 ? nsplit - 1 : g;
    while (g > 0 && gstart(g) > lb) --g;
    while (g + 1 < nsplit && gstart(g + 1) <= lb) ++g;
    const int within = lb - gstart(g);
    int ti, tj, split, rps;
    if (within < noff) {
        split = g;
        rps = rows_per_split;
        int rem = within;   // row-major over the strict upper triangle
        ti = 0;
        while (rem >= ntile - 1 - ti) { rem -= ntile - 1 - ti; ++ti; }
        tj = ti + 1 + rem;
    } else {
        const int q = g * ndiag / nsplit + within - noff;
        split = q / ntile;
        rps = rows_per_split_d;
        ti = tj = q % ntile;
    }
    const bool diag_tile = (ti == tj);
    const int i0 = ti * 128, j0 = tj * 128;

    int r_begin = split * rps;
    int r_end = min(n_rows, r_begin + rps);
    if (TRI) r_begin = max(r_begin, j0 & ~15);
    double *out = slab + (size_t)split * mp * mp;
    if (diag_tile)
        syrk_body<WEIGHTED, WC, true, OT>(Phi, ld, wgt, mp, i0, j0, r_begin, r_end, out, sA, sB, sW);
    else
        syrk_body<WEIGHTED, WC, false, OT>(Phi, ld, wgt, mp, i0, j0, r_begin, r_end, out, sA, sB, sW);
}

// S[i][j] = S[j][i] = sum_s slab[s][min-tile-order(i,j)]
__global__ void k_syrk_reduce(const double *__restrict__ slab, int nsplit_off, int nsplit_d, int mp,
                              double *__restrict__ S, int lds, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= mp || i > j) return;                            // the upper triangle is summed (coalesced reads of the slabs) and written to
                                                             // both places: summing the mirror from its own side read every slab by columns
    const int ti = i >> 7, tj = j >> 7;
    const int nsplit = (ti == tj) ? nsplit_d : nsplit_off;
    const size_t stride = (size_t)mp * mp;
    const double *p = slab + (size_t)i * mp + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;           // four chains, eight loads in flight: the sum is latency bound
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(k + q) * stride];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; k < nsplit; ++k) s0 += p[(size_t)k * stride];
    const double t = (s0 + s1) + (s2 + s3);
    // accumulate: the next row tile of a streamed evaluation (the mirror is added from its own previous value: S stays symmetric)
    S[(size_t)i * lds + j] = accumulate ? S[(size_t)i * lds + j] + t : t;
    if (i != j) S[(size_t)j * lds + i] = accumulate ? S[(size_t)j * lds + i] + t : t;
}

// ---------------------------------------------------------------------------------------------
// T = PHI * B
// ---------------------------------------------------------------------------------------------
// PHI: n_pad x ld (row-major), B: mp x ldb (row-major), T: n_pad x ld.  n_pad % 128 == 0, mp % 16 == 0.
// Optional fused epilogue (nupart != nullptr):
//   nupart[(ct*WC + wc)*n_pad + row] = sum over this wave's columns (< m) of PHI[row][col]*T[row][col]   (GPz.m:69)
//   phiw[row] = T[row][mcol]  (= (PHI w)_row, GPz.m:77)
template <bool EDGE, int WC, typename OT, typename BT>
__device__ __forceinline__ void tgemm_body(const double *__restrict__ Phi, int ld, const BT *__restrict__ B,
                                           int ldb, double *__restrict__ T, int ldt, int mp, int i0, int j0,
                                           OT (*sA)[128][18], OT (*sB)[16][LDS_LD128],
                                           double *__restrict__ nupart, double *__restrict__ phiw, int m, int mcol,
                                           long n_pad, int ct, int kdim, int ncol16 = 8, int slot = 0, int nslot = 1, int item = 0) {
    constexpr int NT = 128 * WC;
    constexpr int NI = 8 / WC;
    constexpr int Q = 1024 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;

    // fp32 operands: K = m here (a few thousand products per element), where fp32 accumulation adds an error of the size of the
    // operand rounding itself (both ~ eps32 * sqrt(K) * rms|term|): no fp64 master sums, the result is converted at the end.
    typedef typename MfmaOf<OT>::acc_t acc_t;
    typedef typename MfmaOf<OT>::pair_t pair_t;
    acc_t acc[4][NI];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = (acc_t){0, 0, 0, 0};

    // A slice: 128 rows x 16 doubles, 8 consecutive lanes cover one 128-byte row segment.
    // Addressing is kept off the vector ALU (a VALU instruction takes issue cycles from the MFMA pipe of its SIMD:
    // tools/mfma_f64_operands.hip, 6 v_mov per 8 MFMAs cost 6 %): the global loads use a wave-uniform base pointer
    // (SGPR pair, advanced by scalar adds) plus one loop-invariant 32-bit byte offset per operand, and the K loop is
    // unrolled over the two LDS buffers so that every LDS address is a loop-invariant register plus an immediate.
    // Columns >= mp of the last column tile are read from the last valid pair (never stored, see the epilogue).
    typedef typename Pair2<BT>::t bpair_t;   // B as it is stored: fp64, or fp32 rounded once per evaluation (config 5: half the bytes of the
    d2_t ra[Q];                              // operand every 128-row panel streams again)
    bpair_t rb[Q];
    const unsigned voa = (unsigned)(((tid >> 3) * ld + (tid & 7) * 2) * 8);
    const unsigned vob = (unsigned)(((tid >> 6) * ldb + min(j0 + (tid & 63) * 2, mp - 2)) * sizeof(BT));
    const size_t qsa = (size_t)(NT / 8) * ld * 8, qsb = (size_t)(NT / 64) * ldb * sizeof(BT), ksb = (size_t)16 * ldb * sizeof(BT);
    const char *pa = reinterpret_cast<const char *>(Phi + (size_t)i0 * ld);
    const char *pb = reinterpret_cast<const char *>(B);
    auto gload = [&]() {   // next 16-deep slice
        // the empty asm keeps the 32 -> 64 bit extension of the offsets in this block, where instruction selection
        // folds it into the load's (SGPR base + VGPR offset) addressing mode; hoisted out of the loop it would come back
        // as a 64-bit vector add per load
        unsigned oa = voa, ob = vob;
        asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            ra[q] = *reinterpret_cast<const d2_t *>(pa + q * qsa + oa);
            rb[q] = *reinterpret_cast<const bpair_t *>(pb + q * qsb + ob);
        }
        pa += 16 * 8;
        pb += ksb;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            int idx = q * NT + tid;
            int arow = idx >> 3, ac = (idx & 7) * 2;
            *reinterpret_cast<pair_t *>(&sA[buf][arow][ac]) = to_pair<OT>(ra[q]);
            int row = idx >> 6, c = (idx & 63) * 2;
            *reinterpret_cast<pair_t *>(&sB[buf][row][c]) = to_pair<OT>(rb[q]);
        }
    };

    // The last column tile can be partial (m = 1000: 112 of its 128 columns exist).  Skipping the missing 16-column MFMA tiles in
    // the 2 x WC wave grid leaves the SIMD that hosts the waves of the last wave column with half the work and the others with all
    // of it - no time saved.  EDGE tiles are therefore dealt in ROW STRIPS: wave w takes rows 16 w .. 16 w + 15 and every valid
    // column tile (one A fragment, nvalid <= 8 B fragments per K step), so all four SIMDs run 2 nvalid MFMAs per K step instead of 16.
    int nvalid = NI;
    if (EDGE) {
        nvalid = (mp - j0 + 15) / 16;
        nvalid = nvalid < 0 ? 0 : (nvalid > ncol16 ? ncol16 : nvalid);   // ncol16 < 8: a column piece of a split tile (k_tgemm)
        nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    }
    static_assert(!EDGE || 4 * NI == 8, "the row-strip deal keeps its eight accumulators in acc[4][2]");
    constexpr int NFA = EDGE ? 1 : 4, NFB = EDGE ? 8 : NI;

    // Software pipeline as in syrk_body: loads of slice s+2 issued mid-slice s, LDS writes between the MFMA halves.
    const int nstage = kdim / 16;        // K (= mp for the evaluation's T = PHI*[inv(SIGMA)|w]; the output may be wider: mp columns)
    gload();
    lstore(0);
    if (nstage > 1) gload();
    __syncthreads();
    // Operand fragments are fetched from LDS one K step ahead of the MFMA burst that uses them, and the barrier of a
    // slice sits in front of its last burst, so the first fragment of the next slice is fetched under that burst.
    auto rdfrag = [&](int cur, int kk, OT (&a)[NFA], OT (&b)[NFB]) {
        const int kc = kk * 4 + (lane >> 4);
        if (EDGE) {
            a[0] = sA[cur][wave * 16 + (lane & 15)][kc];
#pragma unroll
            for (int p = 0; p < NFB; ++p) b[p] = sB[cur][kc][p * 16 + (lane & 15)];   // (columns past mp were staged from the last valid pair)
        } else {
#pragma unroll
            for (int mi = 0; mi < NFA; ++mi) a[mi] = sA[cur][wr * 64 + mi * 16 + (lane & 15)][kc];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = sB[cur][kc][wc * (16 * NI) + ni * 16 + (lane & 15)];
        }
    };
    // Raised wave priority over the MFMA burst only: the SIMD's issue arbiter then prefers a wave whose operands are
    // ready over the waves that are staging (global loads, LDS writes, address VALU), which otherwise take issue
    // slots in front of it.  k_tgemm 33.15 -> 31.95 ms at c4; raising it over the LDS operand reads as well gives
    // nothing, priority 3 the same as 1.  Fragment prefetch on top: 32.1 -> 31.6 ms.
    auto burst = [&](const OT (&a)[NFA], const OT (&b)[NFB]) {
        __builtin_amdgcn_s_setprio(1);
        if (EDGE) {
#pragma unroll
            for (int p = 0; p < NFB; ++p)
                if (p < nvalid) acc[p >> 1][p & 1] = MfmaOf<OT>::run(a[0], b[p], acc[p >> 1][p & 1]);
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int mi = 0; mi < NFA; ++mi) acc[mi][ni] = MfmaOf<OT>::run(a[mi], b[ni], acc[mi][ni]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    OT fa0[NFA], fb0[NFB], fa1[NFA], fb1[NFB];
    rdfrag(0, 0, fa0, fb0);
    GPZ_TRACE_MARK(1);
    auto stage = [&](auto curc, int s) {
        constexpr int cur = decltype(curc)::value;
        rdfrag(cur, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        rdfrag(cur, 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nstage) {
            lstore(cur ^ 1);
            if (s + 2 < nstage) gload();
        }
        rdfrag(cur, 3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (s + 1 < nstage) rdfrag(cur ^ 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        burst(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int s = 0;
    for (; s + 1 < nstage; s += 2) {
        stage(std::integral_constant<int, 0>{}, s);
        stage(std::integral_constant<int, 1>{}, s + 1);
    }
    if (s < nstage) stage(std::integral_constant<int, 0>{}, s);
    GPZ_TRACE_MARK(2);
    auto res = [&](int mi, int ni, int r) -> double { return (double)acc[mi][ni][r]; };
    // row of accumulator register r inside a 16x16 tile: the f64 instruction deals rows (lane >> 4) + 4r, the f32 one 4(lane >> 4) + r
    auto crow = [&](int r) -> int { return std::is_same<OT, float>::value ? 4 * (lane >> 4) + r : (lane >> 4) + 4 * r; };

    // Epilogue.  The PHI values of the fused nu sum are ALL requested before the first is used, ahead of the T stores: a load
    // that is waited for on its own costs a memory round trip behind every store issued before it (vmcnt counts both), and 32 of
    // those in sequence were 83 000 of a workgroup's 536 000 cycles (tools/gemm_trace.hip), with the compute unit's other
    // workgroup usually in the same phase.  No lane-dependent branches: out-of-range columns are dropped by selects, addresses are
    // a wave-uniform base plus one 32-bit lane offset.
    if (EDGE) {
        // row strip of this wave: rows i0 + 16 wave + crow(r), column tile p.  (One tile in eight: its K loop holds eight B fragments
        // twice beside the accumulators, and a batched form of this epilogue made the compiler spill INSIDE that loop - k_tgemm
        // 29 -> 63 ms.  The loads stay where they are used.)
#pragma unroll
        for (int p = 0; p < NFB; ++p) {
            const int col = j0 + p * 16 + (lane & 15);
            if (p < nvalid && col < mp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(size_t)(i0 + wave * 16 + crow(r)) * ldt + col] = res(p >> 1, p & 1, r);
            }
        }
        if (nupart) {
            // the consumers sum WC slots per column tile and row: this wave's sum over ALL its columns of the tile goes to slot `slot`
            // (0 for a whole edge tile, the piece index for a split tile); the first piece zeroes the slots no piece writes
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wave * 16 + crow(r);
                double pp = 0.0;
#pragma unroll
                for (int p = 0; p < NFB; ++p) {
                    const int col = j0 + p * 16 + (lane & 15);
                    if (p < nvalid && col < m) pp = fma(Phi[(size_t)row * ld + col], res(p >> 1, p & 1, r), pp);
                    if (p < nvalid && col == mcol) phiw[row] = res(p >> 1, p & 1, r);
                }
                pp += __shfl_xor(pp, 1, 64);
                pp += __shfl_xor(pp, 2, 64);
                pp += __shfl_xor(pp, 4, 64);
                pp += __shfl_xor(pp, 8, 64);
                if ((lane & 15) == 0) {
                    nupart[(size_t)(ct * WC + slot) * n_pad + row] = pp;
                    if (slot == 0)
#pragma unroll
                        for (int q = 1; q < WC; ++q)
                            if (q >= nslot) nupart[(size_t)(ct * WC + q) * n_pad + row] = 0.0;
                }
            }
        }
        return;
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wru = wv / WC, wcu = wv % WC;
    const char *pbase = reinterpret_cast<const char *>(Phi + (size_t)(i0 + wru * 64) * ld + j0 + wcu * (16 * NI));
    char *tbase = reinterpret_cast<char *>(T + (size_t)(i0 + wru * 64) * ldt + j0 + wcu * (16 * NI));
    double *slotp = nupart ? nupart + (size_t)(ct * WC + wc) * n_pad : nullptr;
    // two halves (rows mi = 0, 1 and 2, 3 of the wave's block): 16 loads in flight beside the 64 accumulator registers
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        double ph[2][4][NI];
        if (nupart) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        ph[mh][r][ni] = *reinterpret_cast<const double *>(pbase + ((size_t)((hf * 2 + mh) * 16 + crow(r)) * ld + ni * 16 + (lane & 15)) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<double *>(tbase + ((size_t)((hf * 2 + mh) * 16 + crow(r)) * ldt + ni * 16 + (lane & 15)) * 8) = res(hf * 2 + mh, ni, r);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (nupart) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = hf * 2 + mh;
                    const int row = i0 + wr * 64 + mi * 16 + crow(r);
                    double p = 0.0;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int col = j0 + wc * (16 * NI) + ni * 16 + (lane & 15);
                        const double t = fma(ph[mh][r][ni], res(mi, ni, r), p);
                        p = (col < m) ? t : p;
                        if (col == mcol) phiw[row] = res(mi, ni, r);
                    }
                    // sum over the 16 lanes that share this row (lane & 15 runs over columns)
                    p += __shfl_xor(p, 1, 64);
                    p += __shfl_xor(p, 2, 64);
                    p += __shfl_xor(p, 4, 64);
                    p += __shfl_xor(p, 8, 64);
                    if ((lane & 15) == 0) slotp[row] = p;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int WC, typename OT, typename BT>
__global__ __launch_bounds__(128 * WC, WC) void k_tgemm(const double *__restrict__ Phi, int ld,
                                                        const BT *__restrict__ B, int ldb,
                                                        double *__restrict__ T, int ldt, int mp, int nct,
                                                        double *__restrict__ nupart, double *__restrict__ phiw, int m,
                                                        int mcol, long n_pad, int kdim, int nfull, int npiece) {
    __shared__ OT sA[2][128][18];
    __shared__ OT sB[2][16][LDS_LD128];
    // One item per workgroup: the grid is the nfull whole tiles followed by the remaining tiles cut into npiece column pieces each
    // (launch_tgemm): when the tile count leaves the last round of resident workgroups mostly empty (c2: 1568 tiles on 512 slots = 3.06
    // rounds), the pieces of the last 0.06 round fill the chip for a fraction of a tile's time instead of 32 tiles holding it for a whole one.
    // (Round 5 measured two other schedules and dropped them - profiles/r05_dropped_kernel_experiments.tar.gz: PERSISTENT workgroups, two
    // per compute unit walking the item list, stay in phase, both in their store epilogue at once (33.8 ms against 29.3); FOUR 4-wave
    // workgroups per CU on 128 x 64 tiles overlap the epilogues but lose more inside the K loop, the LDS holding no four 16-deep
    // double-buffered stages (29.9 - 30.5 ms).)
    const int item = blockIdx.x;
#ifdef GPZ_GEMM_TRACE
    GPZ_TRACE_MARK(0);
    if (g_gemm_trace && (threadIdx.x & 63) == 0)
        g_gemm_trace[((size_t)item * 8 + (threadIdx.x >> 6)) * 6 + 4] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
    struct TraceEnd { int item; __device__ ~TraceEnd() { GPZ_TRACE_MARK(3); } } trace_end{item};
#endif
    if (item >= nfull) {
        const int q = item - nfull;
        const int tile = nfull + q / npiece, piece = q % npiece;
        const int rt = tile / nct, ct = tile % nct;
        tgemm_body<true, WC, OT, BT>(Phi, ld, B, ldb, T, ldt, mp, rt * 128, ct * 128 + piece * (128 / npiece), sA, sB, nupart, phiw, m, mcol,
                                 n_pad, ct, kdim, 8 / npiece, piece, npiece, item);
        return;
    }
    // XCD-aware remap (blocks are dispatched round-robin over the 8 XCDs): give every XCD a contiguous range of
    // logical tiles, so the 8 column tiles of a row panel run back to back on one XCD and the 1 MB PHI panel is
    // fetched from HBM once instead of once per XCD.  Bijective for any grid size; affects speed only.
    const int nwg = nfull, xcd = item & 7, q = nwg >> 3, r = nwg & 7;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (item >> 3);
    const int rt = lb / nct, ct = lb % nct;
    const int i0 = rt * 128, j0 = ct * 128;
    if (j0 + 128 <= mp)
        tgemm_body<false, WC, OT, BT>(Phi, ld, B, ldb, T, ldt, mp, i0, j0, sA, sB, nupart, phiw, m, mcol, n_pad, ct, kdim, 8, 0, 1, item);
    else
        tgemm_body<true, WC, OT, BT>(Phi, ld, B, ldb, T, ldt, mp, i0, j0, sA, sB, nupart, phiw, m, mcol, n_pad, ct, kdim, 8, 0, 1, item);
}

// ---------------------------------------------------------------------------------------------
// tile GEMM for the m x m factorisation (sizes <= a few thousand; L2 resident)
// ---------------------------------------------------------------------------------------------
// C[M x N] (row-major, stride ldc) = alpha * A[M x K] * B[K x N] over k in [kbeg, kend) only (both multiples of 16):
// the callers' operands are triangular, so half of every product is structurally zero.  A and B row-major (lda, ldb,
// both multiples of 4 with 32-byte aligned rows).  One workgroup computes the 64x64 tile (tm, tn).  The products are short and
// alone on their compute unit, so a 16-deep slice is bound by the latency of its operands (they come from another XCD's writes):
// slices are fetched into registers TWO ahead (two register sets, row-wise, 32 bytes per lane) and the LDS tiles are double
// buffered - one barrier per slice (one slice ahead and two barriers: 2 600 cycles per slice for 1 024 of MFMA work).
__device__ void gemm_tile_64(const double *__restrict__ A, long lda, const double *__restrict__ B, long ldb,
                             double *__restrict__ C, long ldc, int M, int N, int kbeg, int kend, double alpha, int tm,
                             int tn) {
    __shared__ double sA[2][16][LDS_LD64];
    __shared__ double sB[2][16][LDS_LD64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i0 = tm * 64, j0 = tn * 64;
    d4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};

    const int ar = tid >> 2, ak = (tid & 3) * 4;       // A slice: 64 rows x 16 k, four consecutive k per lane
    const int bk = tid >> 4, bc = (tid & 15) * 4;      // B slice: 16 k x 64 columns, four consecutive columns per lane
    const bool aok = i0 + ar < M, bok = j0 + bc < N;   // N is a multiple of 4
    const double *ap = A + (size_t)(i0 + ar) * lda + ak;
    const double *bp = B + (size_t)bk * ldb + j0 + bc;
    struct Regs { d2_t a[2], b[2]; } r0, r1;
    auto gload = [&](Regs &r, int k0) {
        const d2_t z = {0.0, 0.0};
        r.a[0] = aok ? *reinterpret_cast<const d2_t *>(ap + k0) : z;
        r.a[1] = aok ? *reinterpret_cast<const d2_t *>(ap + k0 + 2) : z;
        r.b[0] = bok ? *reinterpret_cast<const d2_t *>(bp + (size_t)k0 * ldb) : z;
        r.b[1] = bok ? *reinterpret_cast<const d2_t *>(bp + (size_t)k0 * ldb + 2) : z;
    };
    auto park = [&](const Regs &r, int buf) {
        sA[buf][ak + 0][ar] = r.a[0][0]; sA[buf][ak + 1][ar] = r.a[0][1]; sA[buf][ak + 2][ar] = r.a[1][0]; sA[buf][ak + 3][ar] = r.a[1][1];
        *reinterpret_cast<d2_t *>(&sB[buf][bk][bc]) = r.b[0];
        *reinterpret_cast<d2_t *>(&sB[buf][bk][bc + 2]) = r.b[1];
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kr = kk * 4 + (lane >> 4);
            double a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = sA[buf][kr][wr * 32 + mi * 16 + (lane & 15)];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = sB[buf][kr][wc * 32 + ni * 16 + (lane & 15)];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = MFMA_F64(a[mi], b[ni], acc[mi][ni]);
        }
    };
    if (kbeg < kend) {
        gload(r0, kbeg);
        if (kbeg + 16 < kend) gload(r1, kbeg + 16);
        park(r0, 0);
        __syncthreads();
    }
    // slice s sits in LDS buffer s & 1; slice s + 1 in the other register set; slice s + 2 is requested into the set slice s came from
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        if (k0 + 32 < kend) gload(r0, k0 + 32);
        multiply(0);
        if (k0 + 16 >= kend) break;
        park(r1, 1);
        __syncthreads();
        if (k0 + 48 < kend) gload(r1, k0 + 48);
        multiply(1);
        if (k0 + 32 < kend) park(r0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = j0 + wc * 32 + ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wr * 32 + mi * 16 + (lane >> 4) + 4 * r;
                if (row < M && col < N) C[row * ldc + col] = alpha * acc[mi][ni][r];
            }
        }
}

// One level of the recursive triangular inverse W = inv(L) (both lower triangular, mq x ld):
// for pair p with left block [a, a+gs) and right block [a+gs, a+2gs):
//   phase 0:  Tmp(right,left) = L(right,left) * W(left,left)
//   phase 1:  W(right,left)   = - W(right,right) * Tmp(right,left)
__global__ __launch_bounds__(256) void k_trtri_level(const double *__restrict__ L, double *__restrict__ W,
                                                      double *__restrict__ Tmp, int ld, int mq, int gs, int phase) {
    const int p = blockIdx.z;
    const int a = p * 2 * gs;
    const int rb = a + gs;
    if (rb >= mq) return;
    const int Mr = min(gs, mq - rb);
    const int tm = blockIdx.y, tn = blockIdx.x;
    if (tm * 64 >= Mr) return;
    if (phase == 0) {   // W(left,left) is lower triangular: rows k < 64 tn of it are zero in this tile's columns
        gemm_tile_64(L + (size_t)rb * ld + a, ld, W + (size_t)a * ld + a, ld, Tmp + (size_t)rb * ld + a, ld, Mr, gs,
                     tn * 64, gs, 1.0, tm, tn);
    } else {            // W(right,right) is lower triangular: columns k >= 64 (tm + 1) of this tile's rows are zero
        gemm_tile_64(W + (size_t)rb * ld + rb, ld, Tmp + (size_t)rb * ld + a, ld, W + (size_t)rb * ld + a, ld, Mr, gs,
                     0, min(Mr, tm * 64 + 64), -1.0, tm, tn);
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
#ifndef GPZ_GEMM_WC
#define GPZ_GEMM_WC 4   // 8 waves per workgroup (gpz_gemm_wave_cols() reports it to the host code)
#endif
int gpz_gemm_wave_cols() { return GPZ_GEMM_WC; }

void launch_syrk(hipStream_t st, const double *Phi, int ld, const double *wgt, int n_rows, int mp,
                 int nsplit, int rows_per_split, int nsplit_d, int rows_per_split_d, double *slab, bool tri,
                 bool f32_operands) {
    constexpr int WC = GPZ_GEMM_WC;
    const int ntile = (mp + 127) / 128;
    const int noff = ntile * (ntile - 1) / 2;
    dim3 grid(noff * nsplit + ntile * nsplit_d), block(128 * WC);
    if (tri)
        hipLaunchKernelGGL((k_syrk<false, true, WC, double>), grid, block, 0, st, Phi, ld, wgt, n_rows, mp, ntile, nsplit,
                           rows_per_split, nsplit_d, rows_per_split_d, slab);
    else if (wgt && f32_operands)
        hipLaunchKernelGGL((k_syrk<true, false, WC, float>), grid, block, 0, st, Phi, ld, wgt, n_rows, mp, ntile, nsplit,
                           rows_per_split, nsplit_d, rows_per_split_d, slab);
    else if (wgt)
        hipLaunchKernelGGL((k_syrk<true, false, WC, double>), grid, block, 0, st, Phi, ld, wgt, n_rows, mp, ntile, nsplit,
                           rows_per_split, nsplit_d, rows_per_split_d, slab);
    else
        hipLaunchKernelGGL((k_syrk<false, false, WC, double>), grid, block, 0, st, Phi, ld, wgt, n_rows, mp, ntile, nsplit,
                           rows_per_split, nsplit_d, rows_per_split_d, slab);
}

void launch_syrk_reduce(hipStream_t st, const double *slab, int nsplit, int nsplit_d, int mp, double *S, int lds, int accumulate) {
    dim3 block(256), grid((mp + 255) / 256, mp);
    hipLaunchKernelGGL(k_syrk_reduce, grid, block, 0, st, slab, nsplit, nsplit_d, mp, S, lds, accumulate);
}

// T (n_pad x mp, row stride ldt) = PHI (n_pad x kdim, row stride ld) * B (kdim x mp, row stride ldb).  kdim = 0: the square case
// of the evaluation (K = mp, ldt = ld).  kdim % 16 == 0.
// compute units of the current device (256 on MI355X); the tile kernels keep two workgroups resident on each
int gpz_cu_count() {
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        return cu;
    }();
    return n;
}

__global__ void k_round_f32(const double *__restrict__ src, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
void launch_round_f32(hipStream_t st, const double *src, float *dst, size_t n) {
    hipLaunchKernelGGL(k_round_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

void launch_tgemm(hipStream_t st, const double *Phi, int ld, const double *B, int ldb, double *T, int n_pad, int mp,
                  double *nupart, double *phiw, int m, int mcol, bool f32_operands, int kdim, int ldt, const float *B32) {
    constexpr int WC = GPZ_GEMM_WC;
    const int nct = (mp + 127) / 128;
    if (kdim <= 0) kdim = mp;
    if (ldt <= 0) ldt = ld;
    // tail balance (see k_tgemm): the tiles of the last, partly filled round of resident workgroups are cut into column pieces
    const int W = (n_pad / 128) * nct, C = 2 * gpz_cu_count();
    int tail = W % C, npiece = 1;
    const bool no_split = gpz_opts().tgemm_no_split;
    // quarter pieces only: a piece stages the same A and B slices as a whole tile, so a half costs ~0.9 of one (125k-row shard of c4,
    // tail 136: 3.83 ms either way) and a quarter ~0.6 (c2: 231 -> 211 us, c3: 862 -> 826 us)
    if (tail > 0 && !no_split && tail * 4 <= C) npiece = 4;
    if (npiece == 1) tail = 0;
    const int nfull = W - tail;
    dim3 grid(nfull + tail * npiece), block(128 * WC);
    if (f32_operands && B32)   // B already rounded to fp32 (launch_round_f32, once per evaluation): same bits, half the operand stream
        hipLaunchKernelGGL((k_tgemm<WC, float, float>), grid, block, 0, st, Phi, ld, B32, ldb, T, ldt, mp, nct, nupart, phiw, m, mcol,
                           (long)n_pad, kdim, nfull, npiece);
    else if (f32_operands)
        hipLaunchKernelGGL((k_tgemm<WC, float, double>), grid, block, 0, st, Phi, ld, B, ldb, T, ldt, mp, nct, nupart, phiw, m, mcol,
                           (long)n_pad, kdim, nfull, npiece);
    else
        hipLaunchKernelGGL((k_tgemm<WC, double, double>), grid, block, 0, st, Phi, ld, B, ldb, T, ldt, mp, nct, nupart, phiw, m, mcol,
                           (long)n_pad, kdim, nfull, npiece);
}

void launch_trtri_level(hipStream_t st, const double *L, double *W, double *Tmp, int ld, int mq, int gs) {
    const int npair = (mq + 2 * gs - 1) / (2 * gs);
    const int nt = (gs + 63) / 64;
    hipLaunchKernelGGL(k_trtri_level, dim3(nt, nt, npair), dim3(256), 0, st, L, W, Tmp, ld, mq, gs, 0);
    hipLaunchKernelGGL(k_trtri_level, dim3(nt, nt, npair), dim3(256), 0, st, L, W, Tmp, ld, mq, gs, 1);
}
