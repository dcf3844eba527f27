// The tail of an evaluation for FEW basis functions (m + k <= 256 columns) as ONE kernel (gfx950, v_mfma_f64_16x16x4_f64).
//
//   T = PHI [inv(SIGMA) | w]                                 GPz.m:69,72,77
//   nu_i = sum_j PHI_ij T_ij,  delta, beta, c, dbeta         GPz.m:69,77-79,93
//   dPHI_ij = (-omega beta_i T_ij - c_i w_j + dbeta_i v_j) PHI_ij            GPz.m:72,90,106,113
//   sum_i dPHI_ij [1, x_i, x_i x_i'],  PHI'c,  PHI'dbeta     GPz.m:89,104,142-159,189-194
//
// k_tgemm + k_row_scalars + k_moments_fused write T (n x m), read it back with PHI, and are bound by exactly that traffic once m is
// small: at n = 1e5, m = 200 (BASELINE config 2) they take 197 + 7 + 156 us for 106 us of MFMA work.  With mp = ceil16(m + k) <= 256 a
// workgroup holds WHOLE ROWS of T in its accumulators - 32 rows x mp columns over 4 waves - so everything after the product happens
// on registers and T never exists in memory (the context does not allocate it):
//   * wave w owns the 16-column blocks gb = 4 q + w' (q < NQ; w' = w, or 3 - w in the second workgroup of a compute unit, so that the
//     SIMD which hosts wave w of both carries 2 NQ - 1 blocks where the count is not a multiple of four: mp = 208 is 4 + 3 + 3 + 3)
//     for ALL 32 rows: two row strips x NQ blocks of accumulators;
//   * the block's rows of PHI are staged ONCE into LDS (32 x mp, one barrier) and serve as the A operand of every K step and as
//     PHI_ij of the epilogue; the B operand [inv(SIGMA) | w] needs no LDS and no barrier - a wave's columns are its own, so it
//     streams them from L2 with buffer loads four K steps ahead;
//   * nu: the waves' partial row sums over their own columns meet in LDS; one thread per row forms the row scalars and the
//     evaluation's scalar sums;
//   * dPHI_ij = -ob_i PHI_ij U_ij with U = T + delta_i w_j - (dbeta_i / ob_i) v_j: U is one more K step, -ob_i goes into the row
//     FEATURES, PHI_ij U_ij stays in the accumulators (see small_tail_run);
//   * the moment sums are MFMAs again: accumulator register r of a 16 x 16 block IS the A operand (j along M, the four rows
//     4r .. 4r + 3 along K) of a product with the block's features -ob_i [1 | x - mu | (x - mu)^2 or the packed products
//     (x - mu)(x - mu)'], staged in LDS as the B operand - 2 feature blocks of 16 cover d <= 15 (diagonal kinds) and d <= 6
//     (covariance kinds); the sums about the basis centres follow from these raw sums per basis function (k_small_finish; mu = the
//     column means of the training inputs, so |x - mu| is of the order of the data's spread and the expansion loses
//     spread^2 / length^2 ulps);
//   * workgroups are PERSISTENT (two per compute unit) and walk the 32-row blocks, so the moment sums stay in registers for the whole
//     launch and leave as ONE record per workgroup; rows are summed in a fixed order (no atomics: repeatable bit for bit).
// Measured at c2 (profiles/r06_*): 232 us (with its five small follow-up launches) against 377 us for the three kernels it replaces;
// a block takes 68 000 cycles of which 33 500 are its K loop (26 600 of MFMA issue) - the two workgroups of a compute unit overlap
// each other's staging and epilogue only partly; 3136 blocks on 512 workgroups are 7 rounds for 6.1 of work
// (tools/small_trace.hip prints the phase times: profiles/r06_small_tail_timeline.txt).
#include "gpz_dev.h"
#include "gpz_kernels.h"

#ifdef GPZ_SMALL_TRACE   // developer builds only (tools/small_trace.hip): s_memtime stamps of wave `wave` at the phase boundaries of every block
__device__ unsigned long long *g_small_trace = nullptr;
#define SM_MARK(slot)                                                                                                                  \
    do {                                                                                                                               \
        if (g_small_trace && lane == 0 && it < 8)                                                                                      \
            g_small_trace[(((size_t)blockIdx.x * 4 + wv) * 8 + it) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                       \
    } while (0)
#else
#define SM_MARK(slot) do { } while (0)
#endif
#define SM_GLDS(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g), (__attribute__((address_space(3))) void *)(l), 16, 0, 0)
#define SM_LDA 262    // row stride of the PHI block in LDS (doubles; mp <= 256, plus the four columns of the U step's A operand): 2 (mod 4), i.e. 4 (mod 8) banks - the 16 rows of an A-operand
                      // read (one k each) start 4 banks apart, 8 bytes each: conflict-free (260 = 8 banks apart: rows r and r + 8 collided,
                      // SQ_LDS_BANK_CONFLICT was half of the kernel's LDS cycles); the accumulator-layout read is 16 consecutive doubles per row
#define SM_LDE 48     // row stride of the feature tile (doubles): 32 features + padding, 16 (mod 32)
#ifndef SM_PD
#define SM_PD 4       // K steps of B (4 k each) in flight per wave: three steps = 24 MFMAs of this wave (48 with its partner on the SIMD) cover an L2 round trip
#endif

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: four rotate-and-add steps on the vector ALU
// (v_mov_b32_dpp row_ror) instead of four LDS crossbar round trips (ds_bpermute, what __shfl_xor compiles to).
template <int N>
__device__ __forceinline__ double row_ror(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0x120 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x120 + N, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row_sum16(double p) {
    p += row_ror<8>(p);
    p += row_ror<4>(p);
    p += row_ror<2>(p);
    p += row_ror<1>(p);
    return p;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// EIGHT values per lane summed over the 16 lanes of a DPP row with a transposing butterfly: 52 vector instructions instead of the 96 of
// eight row_sum16.  Four pairings that together generate all 16 lanes - row mirror (l <-> 15 - l), half-row mirror (l <-> l ^ 7), quad
// mirror (l <-> l ^ 3), neighbour (l <-> l ^ 1); at each of the first three a lane keeps the half of its values that its bit 3 / 2 / 1
// selects and hands the other half to its partner.  Afterwards lane l holds the row's total of value 4 b3 + 2 b2 + b1 (bits of l & 15).
__device__ __forceinline__ double row_sum16x8(const double (&v)[8], int lane) {
    double a[4], b[2];
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double keep = b3 ? v[k + 4] : v[k], send = b3 ? v[k] : v[k + 4];
        a[k] = keep + dpp_f64<0x140>(send);                      // row_mirror
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double keep = b2 ? a[k + 2] : a[k], send = b2 ? a[k] : a[k + 2];
        b[k] = keep + dpp_f64<0x141>(send);                      // row_half_mirror
    }
    const double keep = b1 ? b[1] : b[0], send = b1 ? b[0] : b[1];
    double c = keep + dpp_f64<0x1B>(send);                       // quad_perm [3, 2, 1, 0]
    c += dpp_f64<0xB1>(c);                                       // quad_perm [1, 0, 3, 2]
    return c;
}

// NQW = column blocks of THIS wave (wave-uniform; a launch mixes NQ and NQ - 1 where the block count is not a multiple of four), F2 = a
// second block of 16 features exists.  Everything the wave indexes its blocks with is a compile-time constant: no branch around an MFMA,
// every LDS address a register plus an immediate.
//
// The vector ALU is the scarce resource here, not the matrix pipe: while the other workgroup of the compute unit is inside its K loop,
// every vector instruction of this one waits behind an MFMA in flight (tools/small_trace.hip: a phase of n vector instructions takes
// ~64 n cycles then), so whatever can be expressed as one more MFMA step or as a scalar / memory / LDS instruction is:
//   * dPHI_ij = (-ob_i T_ij - c_i w_j + dbeta_i v_j) PHI_ij = -ob_i PHI_ij U_ij,  U = T + delta_i w_j - (dbeta_i / ob_i) v_j  (c = ob delta):
//     U is ONE MORE K STEP of the product - A = [delta_i, -dbeta_i / ob_i, 0, 0] parked behind the PHI block in LDS, B = [w; v; 0; 0] -
//     and the row factor -ob_i goes into the FEATURES (32 x 32 values per block instead of 32 x mp elements): one multiply per element;
//   * operands arrive through buffer loads (resource in SGPRs, loop-invariant VGPR offset, scalar row / K offset): no address arithmetic;
//   * the feature table of a thread (which two entries of the centred row [1 | x - mu] it multiplies) is fixed before the block loop.
// DP: the variant that writes dPHI and takes no features (no feature staging, no moment accumulators, no moment products).
template <int NQW, bool F2, bool DP>
__device__ __forceinline__ void small_tail_run(const SmallTailArgs &a, double *smem, int wce) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, mp = a.mp, ld = a.ld;
    double *sA = smem;                          // [32][SM_LDA]: the block's rows of PHI (A operand of the product, PHI_ij of the epilogue); columns mp .. mp+3: [delta, -dbeta/ob, 0, 0]
    double *sE = sA + 32 * SM_LDA;              // [32][SM_LDE]: row features times -omega beta
    double *sNu = sE + 32 * SM_LDE;             // [4][32]: partial nu of the four column groups
    double *sPw = sNu + 4 * 32;                 // [32]: (PHI w)_i
    double *sRs = sPw + 32;                     // [32][4]: c, dbeta of the block's rows

    d4_t Mq[NQW > 0 ? NQW : 1][2];        // moment sums of this wave's column blocks: [block][feature block]
    double r1[NQW > 0 ? NQW : 1], r2[NQW > 0 ? NQW : 1];   // PHI'c, PHI'dbeta partial sums over this lane's rows (column 16 gb + (lane & 15))
    double fbu[NQW > 0 ? NQW : 1];        // B fragment of the U step: row 0 = w, row 1 = v (times vscale), rows 2, 3 = 0
    int vo[NQW > 0 ? NQW : 1];
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        Mq[q][0] = (d4_t){0.0, 0.0, 0.0, 0.0};
        Mq[q][1] = (d4_t){0.0, 0.0, 0.0, 0.0};
        r1[q] = r2[q] = 0.0;
        const int col = (4 * q + wce) * 16 + (lane & 15), cc = col < m ? col : m - 1;   // (PHI_ij = 0 beyond m: any finite value does)
        fbu[q] = (lane >> 4) == 0 ? a.w[cc] : (lane >> 4) == 1 ? a.vscale * a.v[cc] : 0.0;
        // B operand straight from global memory (L2): the column blocks are this wave's alone, so LDS would add a copy and a barrier
        // and no reuse.  Lane l of K step kg reads B[4 kg + (l >> 4)][16 gb + (l & 15)]: four 128-byte runs, as a BUFFER load - resource
        // in SGPRs, one loop-invariant 32-bit byte offset per column block, the K step as the scalar offset: no vector ALU per load
        vo[q] = ((lane >> 4) * a.ldb + (4 * q + wce) * 16 + (lane & 15)) * 8;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // scalar sums (threads 0..31: one row of every block each)
    const int kstep = 4 * a.ldb * 8;                           // bytes per K step
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)a.B, 0, mp * a.ldb * 8, 0x00020000);
    auto bload = [&](int voff, int soff) -> double { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rB, voff, soff, 0)); };
    const int nk = ((m + 15) >> 4) << 2;                       // K steps: K = m rounded up to 16 (rows >= m of B are zero; SM_PD = 4 steps per turn of the ring)
    const int nblocks = (a.n_pad + 31) >> 5;
    // LDS addresses of this lane: A operand (row lane & 15, k lane >> 4) and accumulator layout (row lane >> 4, column lane & 15)
    const double *pa0 = sA + (lane & 15) * SM_LDA + (lane >> 4);
    const double *pc = sA + (lane >> 4) * SM_LDA + wce * 16 + (lane & 15);
    const double *pe = sE + (lane >> 4) * SM_LDE + (lane & 15);
    const double *prs = sRs + (lane >> 4) * 4;
    // column mcol = m + (output) of B is w: T[:, mcol] = PHI w (GPz.m:77) sits in block (m >> 4) = 4 qm + wce of ONE wave (the columns
    // m .. m+k-1 share a block: small_tail_fits), lanes with (lane & 15) == (mcol & 15)
    const int qm = (((m >> 4) - wce) & 3) == 0 ? ((m >> 4) - wce) >> 2 : -1;
    const bool pwlane = (lane & 15) == (a.mcol & 15);
    const double cmask = (lane & 15) < (m & 15) ? 1.0 : 0.0;   // block qm: its columns >= m hold y and padding (PHI_ij = 0 there)
    // staging: wave w takes rows 8 w .. 8 w + 7 of the block, a lane the double2 at columns 2 lane and 128 + 2 lane
    const int c2 = lane * 2;
    const bool in0 = c2 < mp, in1 = c2 + 128 < mp;
    // features: thread (row tid >> 3) multiplies entries pa, pb of its centred row [1 | x - mu | 0] for the four features 4 (tid & 7) + u
    // (entry 0 = the constant 1, entry d + 1 = 0 for the features past nf); the eight indices packed in two registers
    unsigned fpa = 0, fpb = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = (tid & 7) * 4 + u;
        int ia = 0, ib = 0;
        if (f >= a.nf) ia = ib = a.xs_ld - 1;                    // (the row's last entry is 0)
        else if (a.missing) {                                    // row = [1 | x' mk (d) | mk (d) | 0]: features [mk | x' mk | (x' mk)^2]
            if (f < a.d) ia = 1 + a.d + f;
            else if (f < 2 * a.d) ia = 1 + f - a.d;
            else ia = ib = 1 + f - 2 * a.d;
        } else if (f >= 1) {
            if (f <= a.d) ia = f;
            else if (a.kind == GPZ_KIND_DIAG) ia = ib = f - a.d;
            else {
                int e2 = f - 1 - a.d, aa = 0;                    // packed upper triangle, row aa: a.d - aa entries
                while (e2 >= a.d - aa) { e2 -= a.d - aa; ++aa; }
                ia = aa + 1; ib = aa + e2 + 1;
            }
        }
        fpa |= (unsigned)ia << (8 * u);
        fpb |= (unsigned)ib << (8 * u);
    }
    const int xs_ld = a.xs_ld;
    if (tid < 32) { sA[tid * SM_LDA + mp + 2] = 0.0; sA[tid * SM_LDA + mp + 3] = 0.0; }   // rows 2, 3 of the U step's A operand
    int it = 0;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x, ++it) {
        const int i0 = blk * 32;
        SM_MARK(0);
        // An opaque zero in every address of this iteration: the optimiser otherwise hoists each of the (unrolled) epilogue's loop-invariant
        // addresses and constants out of the block loop - some 150 registers held for the whole kernel, spilled around the MFMA loop.
        unsigned oz = 0;
        asm volatile("" : "+v"(oz));
        // ---- stage the block's rows of PHI (32 x mp) with LDS-DMA loads (global_load_lds_dwordx4: lane l of a wave puts its 16 bytes at
        // LDS base + 16 l, i.e. a wave moves 128 consecutive doubles of one row): no registers, no vector ALU, all sixteen requests of
        // a wave in flight at once.  (Through registers the sixteen double2 per lane were 64 VGPRs on top of the moment accumulators:
        // the four-block waves spilled here, 4 600 cycles per block - profiles/r06_small_tail_timeline.txt.)
        {
            const double *g0 = a.Phi + (size_t)(i0 + wv * 8) * ld + c2;
            double *l0 = sA + (wv * 8) * SM_LDA;
            if (in0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) SM_GLDS(g0 + (size_t)r * ld, l0 + r * SM_LDA);
            }
            if (in1) {
#pragma unroll
                for (int r = 0; r < 8; ++r) SM_GLDS(g0 + (size_t)r * ld + 128, l0 + r * SM_LDA + 128);
            }
            // features (times -omega beta of the row: the row factor of dPHI lives here, see above)
            if constexpr (!DP) {
                const int r = tid >> 3, i = i0 + r;
                const double *xs = a.Xs + (size_t)i * xs_ld;
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xa[u] = xs[(fpa >> (8 * u)) & 255]; xb[u] = xs[(fpb >> (8 * u)) & 255]; }
                const double nob = -a.wbeta[i];                    // (rows >= n: omega beta = 0)
#pragma unroll
                for (int u = 0; u < 4; ++u) sE[r * SM_LDE + (tid & 7) * 4 + u] = nob * (xa[u] * xb[u]);
            }
        }
        // ---- T block = PHI(i0 .. i0+31, :) * B: no barrier inside, every wave runs its own columns
        d4_t acc[2][NQW > 0 ? NQW : 1];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < NQW; ++q) acc[t][q] = (d4_t){0.0, 0.0, 0.0, 0.0};
        double fb[SM_PD][NQW > 0 ? NQW : 1];
#pragma unroll
        for (int p = 0; p < SM_PD; ++p)
#pragma unroll
            for (int q = 0; q < NQW; ++q) fb[p][q] = bload(vo[q], p * kstep);
        SM_MARK(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA loads of this wave have landed (the compiler does not track them for the barrier)
        __syncthreads();
        SM_MARK(2);
        double fa0 = (pa0 + oz)[0], fa1 = (pa0 + oz)[16 * SM_LDA];
        int so = SM_PD * kstep;                                // byte offset of the next K step to request
        for (int kg = 0; kg < nk; kg += SM_PD) {
#pragma unroll
            for (int p = 0; p < SM_PD; ++p) {
                // A fragments of the NEXT step are requested before this step's burst (the row is wider than K: the read past the last step is harmless)
                const double na0 = (pa0 + oz)[4 * (kg + p + 1)], na1 = (pa0 + oz)[16 * SM_LDA + 4 * (kg + p + 1)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    acc[0][q] = MFMA_F64(fa0, fb[p][q], acc[0][q]);
                    acc[1][q] = MFMA_F64(fa1, fb[p][q], acc[1][q]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQW; ++q) fb[p][q] = bload(vo[q], so);   // SM_PD steps ahead; past the last row of B the buffer load returns 0 (unused)
#ifndef SM_EXPERIMENT_B_L1   // (tools/small_trace.hip -DSM_EXPERIMENT_B_L1: every K step re-reads the same rows of B - L1 hits - to price the L2 traffic)
                so += kstep;
#endif
                fa0 = na0;
                fa1 = na1;
            }
        }
        SM_MARK(3);
        // the row-scalar inputs of this block's rows (threads 0..31), requested now: they are needed a phase and a barrier later
        double ry = 0.0, rlb = 0.0, rom = 1.0, rom1 = 1.0, rob = 0.0;
        if (tid < 32 && i0 + tid < a.n) {
            const int i = i0 + tid;
            ry = a.y[i]; rlb = a.lnbeta[i]; rob = a.wbeta[i];
            if (a.omega) { rom = a.omega[i]; rom1 = a.omega1[i]; }
        }
        // ---- nu partials and PHI w (PHI_ij from the LDS block, accumulator layout: row (lane >> 4) + 4 r, column lane & 15)
        {
            double pv[8];                                        // value 4 t + r: this lane's columns of row 16 t + (lane >> 4) + 4 r
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double p = 0.0;
#pragma unroll
                    for (int q = 0; q < NQW; ++q) {
                        double ph = (pc + oz)[(t * 16 + 4 * r) * SM_LDA + 64 * q];
                        if (q == qm) ph *= cmask;                  // (wave-uniform condition)
                        p = fma(ph, acc[t][q][r], p);
                    }
                    pv[4 * t + r] = p;
                }
            const double tot = row_sum16x8(pv, lane);            // lane l: value k = (l >> 1) & 7 summed over the 16 columns-lanes of its row group
            const int k = (lane >> 1) & 7;
            if ((lane & 1) == 0) sNu[wce * 32 + (k >> 2) * 16 + (lane >> 4) + 4 * (k & 3)] = tot;
        }
#pragma unroll
        for (int q = 0; q < NQW; ++q)
            if (q == qm) {                                       // (wave-uniform)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (pwlane) sPw[t * 16 + (lane >> 4) + 4 * r] = acc[t][q][r];
            }
        SM_MARK(4);
        __syncthreads();
        // ---- row scalars (GPz.m:43,48,77-79,93) and the scalar sums of GPz.m:81,94,236-237: one thread per row
        if (tid < 32) {
            const int i = i0 + tid;
            double cc = 0.0, db = 0.0, delta = 0.0, g = 0.0;
            if (i < a.n) {
                const double nu = ((sNu[tid] + sNu[32 + tid]) + sNu[64 + tid]) + sNu[96 + tid];
                const double pw = sPw[tid];
                delta = pw - ry;
                const double lb = rlb, om = rom, ob = rob;               // ob = omega beta, GPz.m:48
                // dbeta = 0.5 (-beta) (1/beta - (delta^2 + nu)) omega  (GPz.m:93)  =  0.5 (omega beta (delta^2 + nu) - omega): no exp, no divide
                db = 0.5 * (ob * fma(delta, delta, nu) - om);
                cc = ob * delta;
                g = ob > 0.0 ? db * gpz_rcp(ob) : 0.0;
                s0 = fma(cc, delta, s0);
                s1 = fma(rom1, delta * delta, s1);                       // omega(training): the first column  GPz.m:236
                s2 += -0.5 * (cc * delta + om * lb);                     // omega (-0.5 beta delta^2 + 0.5 ln beta), ln beta = -lnBeta_i   GPz.m:237
                s3 += db;
                a.phiw[i] = pw;
            }
            sRs[tid * 4 + 0] = cc; sRs[tid * 4 + 1] = db; sRs[tid * 4 + 2] = -rob;   // (-omega beta: the row factor of dPHI where it is written out)
            sA[tid * SM_LDA + mp] = delta; sA[tid * SM_LDA + mp + 1] = -g;   // A operand of the U step
        }
        __syncthreads();
        SM_MARK(5);
        // ---- U = T + delta w' - (dbeta / ob) v': one more K step
        {
            const double ua0 = (pa0 + oz)[mp], ua1 = (pa0 + oz)[16 * SM_LDA + mp];
#pragma unroll
            for (int q = 0; q < NQW; ++q) {
                acc[0][q] = MFMA_F64(ua0, fbu[q], acc[0][q]);
                acc[1][q] = MFMA_F64(ua1, fbu[q], acc[1][q]);
            }
        }
        // ---- PHI_ij U_ij in the accumulators (the row factor -ob_i is in the features), PHI'c / PHI'dbeta, then the moment products
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double cc = (prs + oz)[(t * 16 + 4 * r) * 4 + 0], db = (prs + oz)[(t * 16 + 4 * r) * 4 + 1];
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    double ph = (pc + oz)[(t * 16 + 4 * r) * SM_LDA + 64 * q];
                    if (q == qm) ph *= cmask;
                    acc[t][q][r] *= ph;
                    r1[q] = fma(ph, cc, r1[q]);
                    r2[q] = fma(ph, db, r2[q]);
                }
            }
        if constexpr (DP) {   // dPHI_ij = -omega beta_i PHI_ij U_ij for the moment kernels that cannot take features
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double nob = (prs + oz)[(t * 16 + 4 * r) * 4 + 2];
                    double *dr = a.dphi + (size_t)(i0 + t * 16 + (lane >> 4) + 4 * r) * a.ldd + wce * 16 + (lane & 15);
#pragma unroll
                    for (int q = 0; q < NQW; ++q) dr[64 * q] = nob * acc[t][q][r];
                }
        }
        if constexpr (!DP) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double e0 = (pe + oz)[(t * 16 + 4 * r) * SM_LDE], e1 = F2 ? (pe + oz)[(t * 16 + 4 * r) * SM_LDE + 16] : 0.0;
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    Mq[q][0] = MFMA_F64(acc[t][q][r], e0, Mq[q][0]);
                    if (F2) Mq[q][1] = MFMA_F64(acc[t][q][r], e1, Mq[q][1]);
                }
            }
        }
        SM_MARK(6);
        __syncthreads();   // sA, sE, sNu, sRs are rewritten by the next block
        SM_MARK(7);
    }
    // ---- leave: one record per workgroup: [j][nf raw sums | PHI'c | PHI'dbeta]
    const int nrec = a.nf + 2;
    double *rec = a.slab + (size_t)blockIdx.x * m * nrec;
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        // Mq[q][fb][r]: row (lane >> 4) + 4 r of the block = basis j, column lane & 15 of the feature block
        if constexpr (!DP) {
#pragma unroll
        for (int fb2 = 0; fb2 < (F2 ? 2 : 1); ++fb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = (4 * q + wce) * 16 + (lane >> 4) + 4 * r, f = fb2 * 16 + (lane & 15);
                if (j < m && f < a.nf) rec[(size_t)j * nrec + f] = Mq[q][fb2][r];
            }
        }
        double t1 = r1[q], t2 = r2[q];                    // this lane's rows -> all rows of the wave: lanes l, l + 16, l + 32, l + 48
        t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
        t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
        const int j = (4 * q + wce) * 16 + lane;
        if (lane < 16 && j < m) { rec[(size_t)j * nrec + a.nf] = t1; rec[(size_t)j * nrec + a.nf + 1] = t2; }
    }
    // scalar sums of the workgroup (the first 32 lanes of wave 0 hold them)
    if (wv == 0) {
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
        if (lane == 0) {
            double *pw = a.partial + (size_t)blockIdx.x * GPZ_NS;
            pw[0] = s0; pw[1] = s1; pw[2] = s2; pw[3] = s3;
            for (int q = 4; q < GPZ_NS; ++q) pw[q] = 0.0;
        }
    }
}

template <int NQ, bool F2, bool DP>
__global__ __launch_bounds__(256, 2) void k_small_tail(SmallTailArgs a) {
    extern __shared__ double smem[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // column blocks of this wave: gb = 4 q + wce.  The two workgroups of a compute unit deal them in opposite order, so that the SIMD
    // which hosts wave w of both carries 2 NQ - 1 blocks where the block count is not a multiple of four (mp = 208: 4 + 3 + 3 + 3)
    const int wce = ((blockIdx.x / a.ncu) & 1) ? 3 - wv : wv;
    const int nblk = a.mp >> 4;
    const int nqw = (nblk - wce + 3) >> 2;                 // blocks gb = wce, wce + 4, ... < nblk
    // The second workgroup of a compute unit starts half a block late: the two then ALTERNATE between the phases that use memory and
    // LDS (staging, epilogue) and the K loop instead of running them in step (measured with tools/small_trace.hip: in step, every
    // workgroup of the chip stages its 53 KB at the same moment - 27 MB in one burst, 13 000 cycles - and both K loops share the SIMDs)
    if ((blockIdx.x / a.ncu) & 1)
        for (int q = 0; q < a.stagger; ++q) __builtin_amdgcn_s_sleep(127);
    if (nqw == NQ) small_tail_run<NQ, F2, DP>(a, smem, wce);
    else small_tail_run<NQ - 1, F2, DP>(a, smem, wce);         // (NQ = ceil(nblk / 4): every wave has NQ or NQ - 1)
}

// ONE launch behind k_small_tail (it replaced two record sums, a copy, the conversion and the split: five ~5 us launches of an evaluation of
// 600): block j < m sums the nwg records of basis function j - nf + 2 values, up to 32 record lanes per value (lane q adds records q,
// q + L, ...: four loads in flight), the lanes combined in a fixed order - converts the raw sums and writes the moments mom[j][nm] and the
// two column sums cols[.][j]; block m sums the workgroups' scalar partials into scal[0..3] (GPz.m:81-82,236-237).
__global__ __launch_bounds__(1024) void k_small_finish(const double *__restrict__ slab, const double *__restrict__ partial, int nwg, int m, int d,
                                                       int kind, int nf, int missing, const double *__restrict__ P,
                                                       const double *__restrict__ xmu, int nm, int mp, double *__restrict__ mom,
                                                       double *__restrict__ cols, double *__restrict__ scal, int accumulate, int cols_only) {
    __shared__ double part[32][36];
    __shared__ double R[36];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == m) {
        const int v = tid & 3, sl = tid >> 2;                 // 4 values x 256 record lanes
        double s = 0.0;
        for (int r = sl; r < nwg; r += 256) s += partial[(size_t)r * GPZ_NS + v];
        double *sp = &part[0][0];                             // (1152 doubles: 1024 are used here)
        sp[sl * 4 + v] = s;
        __syncthreads();
        if (tid < 4) {
            double t = 0.0;
            for (int q = 0; q < 256; ++q) t += sp[q * 4 + tid];
            scal[tid] = t;
        }
        return;
    }
    const int j = blockIdx.x, nv = nf + 2;
    const int L = 1024 / nv < 32 ? 1024 / nv : 32;
    const int f = tid % nv, sl = tid / nv;
    if (sl < L) {
        const double *p = slab + (size_t)j * nv + f;
        const size_t stride = (size_t)m * nv;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = sl;
        for (; r + 3 * L < nwg; r += 4 * L) {
            s0 += p[(size_t)r * stride];
            s1 += p[(size_t)(r + L) * stride];
            s2 += p[(size_t)(r + 2 * L) * stride];
            s3 += p[(size_t)(r + 3 * L) * stride];
        }
        for (; r < nwg; r += L) s0 += p[(size_t)r * stride];
        part[sl][f] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < nv) {
        double t = 0.0;
        for (int q = 0; q < L; ++q) t += part[q][tid];
        R[tid] = t;
    }
    __syncthreads();
    // raw sums about mu -> the moment records of k_moments_fused: [M1 (d) | S (d, or the packed d(d+1)/2)] and PHI'c, PHI'dbeta:
    //   sum dp (x - p)             = R1 - q R0,                                   q = p - mu
    //   sum dp (x - p)_a (x - p)_b = R2_ab - q_a R1_b - q_b R1_a + q_a q_b R0
    // (with missing values, diagonal kinds: per dimension R0_c = sum dp mk_c, R1_c = sum dp mk_c x'_c, R2_c = sum dp mk_c x'_c^2)
    const int q = tid;
    if (q >= nm + 2) return;
    if (q >= nm) {
        cols[(size_t)(q - nm) * mp + j] = R[nf + (q - nm)];
        return;
    }
    if (cols_only) return;
    double val;
    if (q < d || kind == GPZ_KIND_DIAG) {
        const int c = q < d ? q : q - d;
        const double qc = P[(size_t)j * d + c] - xmu[c];
        const double R0 = missing ? R[c] : R[0], R1 = missing ? R[d + c] : R[1 + c];
        if (q < d) val = R1 - qc * R0;
        else val = fma(qc, fma(qc, R0, -2.0 * R1), missing ? R[2 * d + c] : R[1 + d + c]);
    } else {
        int e = q - d, aa = 0;
        while (e >= d - aa) { e -= d - aa; ++aa; }                       // packed pair index -> (aa, bb), aa <= bb
        const int bb = aa + e;
        const double qa = P[(size_t)j * d + aa] - xmu[aa], qb = P[(size_t)j * d + bb] - xmu[bb];
        val = R[1 + q] - qa * R[1 + bb] - qb * R[1 + aa] + qa * qb * R[0];
    }
    mom[(size_t)j * nm + q] = accumulate ? mom[(size_t)j * nm + q] + val : val;   // dPHI is a sum over the outputs (GPz.m:113)
}

// features per basis function: diagonal kinds 1 + 2d ([1 | x' | x'^2], x' = x - mu); with missing values 3d ([mk | x' mk | (x' mk)^2] per
// dimension, mk = 1 observed / 0 missing: every sum carries the mask of ITS dimension, getPHI.m:64-69, GPz.m:189-194); covariance kinds
// 1 + d + d(d+1)/2 (missing values there take the per-pattern path)
int small_tail_features(int kind, int d, bool missing) {
    return kind == GPZ_KIND_DIAG ? (missing ? 3 * d : 1 + 2 * d) : 1 + d + d * (d + 1) / 2;
}
// m .. m+k-1 (the columns of PHI that hold y, of B that hold w) in ONE 16-column block: the kernel masks that block's columns >= m
bool small_tail_fits(int kind, int d, int m, int k, int mp, bool missing) {
    return mp <= 256 && (mp & 15) == 0 && small_tail_features(kind, d, missing) <= 32 && !(missing && kind != GPZ_KIND_DIAG) &&
           ((m + k - 1) >> 4) == (m >> 4);
}
int small_tail_nwg() { return 2 * gpz_cu_count(); }   // persistent workgroups: two per compute unit

void launch_small_tail(hipStream_t st, const SmallTailArgs &a0, int nwg) {
    SmallTailArgs a = a0;
    a.ncu = gpz_cu_count();
    if (a.stagger <= 0) a.stagger = 1;   // (tools/r06_small_stagger.sh: 1 .. 8 within 2 % since the staging went to LDS-DMA loads; 1 is the fastest by a hair)
    const int nq = ((a.mp >> 4) + 3) / 4;
    const size_t lds = ((size_t)32 * SM_LDA + 32 * SM_LDE + 4 * 32 + 32 + 32 * 4) * sizeof(double);
    dim3 g(nwg), b(256);
#define SMALL_CASE(NQ_)                                                                                   \
    do {                                                                                                  \
        if (a.dphi) hipLaunchKernelGGL((k_small_tail<NQ_, false, true>), g, b, lds, st, a);               \
        else if (a.nf > 16) hipLaunchKernelGGL((k_small_tail<NQ_, true, false>), g, b, lds, st, a);       \
        else hipLaunchKernelGGL((k_small_tail<NQ_, false, false>), g, b, lds, st, a);                     \
    } while (0)
    if (nq <= 1) SMALL_CASE(1);
    else if (nq == 2) SMALL_CASE(2);
    else if (nq == 3) SMALL_CASE(3);
    else SMALL_CASE(4);
#undef SMALL_CASE
}
void launch_small_finish(hipStream_t st, const double *slab, const double *partial, int nwg, int m, int d, int kind, int nf, int missing,
                         const double *P, const double *xmu, int nm, int mp, double *mom, double *cols, double *scal, int accumulate,
                         int cols_only) {
    hipLaunchKernelGGL(k_small_finish, dim3(m + 1), dim3(1024), 0, st, slab, partial, nwg, m, d, kind, nf, missing, P, xmu, nm, mp, mom, cols,
                       scal, accumulate, cols_only);
}
