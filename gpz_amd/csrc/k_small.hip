// The tail of an evaluation for FEW basis functions (m + k <= 256 columns) as ONE kernel (gfx950, v_mfma_f64_16x16x4_f64).
//
//   T = PHI [inv(SIGMA) | w]                                 GPz.m:69,72,77
//   nu_i = sum_j PHI_ij T_ij,  delta, beta, c, dbeta         GPz.m:69,77-79,93
//   dPHI_ij = (-omega beta_i T_ij - c_i w_j + dbeta_i v_j) PHI_ij            GPz.m:72,90,106,113
//   sum_i dPHI_ij [1, x_i, x_i x_i'],  PHI'c,  PHI'dbeta     GPz.m:89,104,142-159,189-194
//
// k_tgemm + k_row_scalars + k_moments_fused write T (n x m), read it back with PHI, and are bound by exactly that traffic once m is
// small: at n = 1e5, m = 200 (BASELINE config 2) they take 197 + 7 + 156 us for 106 us of MFMA work.  With mp = ceil16(m + k) <= 256 a
// workgroup holds WHOLE ROWS of T in its accumulators - 64 rows x mp columns over 8 waves - so everything after the product happens
// on registers and T never exists in memory:
//   * waves (wr, wc), wr = 0..1, wc = 0..3: rows 32 wr .. 32 wr + 31 of the block, 16-column blocks gb = 4 q + wc' (q < NQ) of the
//     columns, wc' = wc for wr = 0 and 3 - wc for wr = 1, so that the two waves of a SIMD carry 2 NQ - 1 blocks where the count is
//     not a multiple of four (mp = 208: 13 blocks = 4 + 3 + 3 + 3);
//   * K loop as in k_tgemm (16-deep slices of PHI and B staged global -> registers -> LDS, double buffered, one barrier per slice);
//   * nu: every wave's partial row sums over its own columns meet in LDS; one thread per row forms the row scalars and the
//     evaluation's scalar sums; the accumulators are overwritten with dPHI;
//   * the moment sums are MFMAs again: accumulator register r of a 16 x 16 block of dPHI IS the A operand (j along M, the four rows
//     4r .. 4r + 3 along K) of a product with the block's row FEATURES [1 | x - mu | (x - mu)^2 or the packed products (x - mu)(x - mu)'],
//     staged in LDS as the B operand - 2 feature blocks of 16 cover d <= 15 (diagonal kinds) and d <= 6 (covariance kinds);
//     the sums about the basis centres follow from these raw sums per basis function (k_small_convert; mu = the column means
//     of the training inputs, so that |x - mu| is of the order of the data's spread and the expansion loses spread^2 / length^2 ulps);
//   * workgroups are PERSISTENT (one per compute unit) and walk the 64-row blocks, so the moment sums stay in registers for the
//     whole launch and leave as ONE record per workgroup half; rows are summed in a fixed order (no atomics: repeatable bit for bit).
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
#define SM_LDA 258    // row stride of the PHI block in LDS (doubles; mp <= 256): 2 (mod 4), i.e. 4 (mod 8) banks - the 16 rows of an A-operand
                      // read (one k each) start 4 banks apart, 8 bytes each: conflict-free (260 = 8 banks apart: rows r and r + 8 collided,
                      // SQ_LDS_BANK_CONFLICT was half of the kernel's LDS cycles); the accumulator-layout read is 16 consecutive doubles per row
#define SM_LDE 48     // row stride of the feature tile (doubles): 32 features + padding, 16 (mod 32)
#define SM_PD 4       // K steps of B (4 k each) in flight per wave: three steps = 24 MFMAs of this wave (48 with its partner on the SIMD) cover an L2 round trip

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

// NQW = column blocks of THIS wave (wave-uniform; a launch mixes NQ and NQ - 1 where the block count is not a multiple of four), F2 = a
// second block of 16 features exists.  Everything the wave indexes its blocks with is a compile-time constant: no branch around an MFMA,
// every LDS address a register plus an immediate.
template <int NQW, bool F2>
__device__ __forceinline__ void small_tail_run(const SmallTailArgs &a, double *smem, int wce) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, mp = a.mp, ld = a.ld;
    double *sA = smem;                          // [32][SM_LDA]: the block's rows of PHI (A operand of the product, PHI_ij of the epilogue)
    double *sE = sA + 32 * SM_LDA;              // [32][SM_LDE]: row features
    double *sNu = sE + 32 * SM_LDE;             // [4][32]: partial nu of the four column groups
    double *sPw = sNu + 4 * 32;                 // [32]: (PHI w)_i
    double *sRs = sPw + 32;                     // [32][4]: omega*beta, c, dbeta of the block's rows

    d4_t Mq[NQW > 0 ? NQW : 1][2];        // moment sums of this wave's column blocks: [block][feature block]
    double r1[NQW > 0 ? NQW : 1], r2[NQW > 0 ? NQW : 1];   // PHI'c, PHI'dbeta partial sums over this lane's rows (column 16 gb + (lane & 15))
    int vo[NQW > 0 ? NQW : 1];
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        Mq[q][0] = (d4_t){0.0, 0.0, 0.0, 0.0};
        Mq[q][1] = (d4_t){0.0, 0.0, 0.0, 0.0};
        r1[q] = r2[q] = 0.0;
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
    // column m of B is w: T[:, m] = PHI w (GPz.m:77) sits in block (m >> 4) = 4 qm + wce of ONE wave, lanes with (lane & 15) == (m & 15)
    const int qm = (((m >> 4) - wce) & 3) == 0 ? ((m >> 4) - wce) >> 2 : -1;
    const bool pwlane = (lane & 15) == (m & 15);
    int it = 0;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x, ++it) {
        const int i0 = blk * 32;
        SM_MARK(0);
        // An opaque zero in every address of this iteration: the optimiser otherwise hoists each of the (unrolled) epilogue's loop-invariant
        // addresses and constants out of the block loop - some 150 registers held for the whole kernel, spilled around the MFMA loop.
        unsigned oz = 0;
        asm volatile("" : "+v"(oz));
        // the row-scalar inputs of this block's rows (threads 0..31), requested now: they are needed two barriers later
        double ry = 0.0, rlb = 0.0, rom = 1.0, rob = 0.0;
        if (tid < 32 && i0 + tid < a.n) {
            const int i = i0 + tid;
            ry = a.y[i]; rlb = a.lnbeta[i]; rob = a.wbeta[i];
            if (a.omega) rom = a.omega[i];
        }
        // ---- stage the block's rows of PHI (32 x mp, coalesced double2 reads) and their features
        {   // wave w: rows 8 w .. 8 w + 7, a lane two double2 per row (mp <= 256).  ALL loads are issued before the first is used (one memory
            // round trip per block, not sixteen: the accumulators are not live here, so the 64 registers exist)
            const int c2 = lane * 2;
            const bool in0 = c2 < mp, in1 = c2 + 128 < mp;
            const double *src = a.Phi + (size_t)(i0 + wv * 8) * ld + (in0 ? c2 : 0);
            const double *src1 = a.Phi + (size_t)(i0 + wv * 8) * ld + (in1 ? c2 + 128 : 0);
            double *dst = sA + (wv * 8) * SM_LDA + c2;
            d2_t v0[8], v1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v0[r] = *reinterpret_cast<const d2_t *>(src + (size_t)r * ld);
                v1[r] = *reinterpret_cast<const d2_t *>(src1 + (size_t)r * ld);
            }
            __builtin_amdgcn_sched_barrier(0);
            // columns >= m (y, padding) become zero in LDS: they meet zero rows of B in the product, and the epilogue's PHI_ij, j < m, needs no mask
            const double k0 = c2 < m ? 1.0 : 0.0, k1 = c2 + 1 < m ? 1.0 : 0.0, k2 = c2 + 128 < m ? 1.0 : 0.0, k3 = c2 + 129 < m ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                d2_t u = v0[r], t = v1[r];
                u.x *= k0; u.y *= k1; t.x *= k2; t.y *= k3;
                if (in0) *reinterpret_cast<d2_t *>(dst + r * SM_LDA) = u;
                if (in1) *reinterpret_cast<d2_t *>(dst + r * SM_LDA + 128) = t;
            }
        }
        {   // thread: row tid >> 3, features 4 (tid & 7) .. + 3.  The four values' inputs are requested together (one memory round trip, not four)
            const int r = tid >> 3, f0 = (tid & 7) * 4, i = i0 + r;
            const bool rowin = i < a.n;
            const double *xi = a.Xr + (size_t)(rowin ? i : 0) * a.d;
            int ia[4], ib[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u;
                ia[u] = ib[u] = -1;                                      // -1: the constant 1 (f = 0) or an unused feature
                if (f >= 1 && f < a.nf) {
                    if (f <= a.d) ia[u] = f - 1;
                    else if (a.kind == GPZ_KIND_DIAG) ia[u] = ib[u] = f - 1 - a.d;
                    else {
                        int e2 = f - 1 - a.d, aa = 0;                    // packed upper triangle, row aa: a.d - aa entries
                        while (e2 >= a.d - aa) { e2 -= a.d - aa; ++aa; }
                        ia[u] = aa; ib[u] = aa + e2;
                    }
                }
            }
            double xa[4], xb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = ia[u] >= 0 ? xi[ia[u]] - a.xmu[ia[u]] : 1.0;
                xb[u] = ib[u] >= 0 ? xi[ib[u]] - a.xmu[ib[u]] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u;
                sE[r * SM_LDE + f] = (rowin && f < a.nf) ? xa[u] * xb[u] : 0.0;
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
#ifdef SM_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    acc[0][q] = MFMA_F64(fa0, fb[p][q], acc[0][q]);
                    acc[1][q] = MFMA_F64(fa1, fb[p][q], acc[1][q]);
                }
#ifdef SM_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
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
        // ---- nu partials and PHI w (PHI_ij from the LDS block, accumulator layout: row (lane >> 4) + 4 r, column lane & 15)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double p = 0.0;
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    const double ph = (pc + oz)[(t * 16 + 4 * r) * SM_LDA + 64 * q];
                    p = fma(ph, acc[t][q][r], p);
                }
                p = row_sum16(p);
                if ((lane & 15) == 0) sNu[wce * 32 + t * 16 + (lane >> 4) + 4 * r] = p;
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
            double ob = 0.0, cc = 0.0, db = 0.0;
            if (i < a.n) {
                const double nu = ((sNu[tid] + sNu[32 + tid]) + sNu[64 + tid]) + sNu[96 + tid];
                const double pw = sPw[tid];
                const double delta = pw - ry;
                const double lb = rlb, om = rom;
                ob = rob;                                                // omega beta, GPz.m:48
                // dbeta = 0.5 (-beta) (1/beta - (delta^2 + nu)) omega  (GPz.m:93)  =  0.5 (omega beta (delta^2 + nu) - omega): no exp, no divide
                db = 0.5 * (ob * fma(delta, delta, nu) - om);
                cc = ob * delta;
                s0 = fma(cc, delta, s0);
                s1 = fma(om, delta * delta, s1);
                s2 += -0.5 * (cc * delta + om * lb);                     // omega (-0.5 beta delta^2 + 0.5 ln beta), ln beta = -lnBeta_i   GPz.m:237
                s3 += db;
                a.phiw[i] = pw;
            }
            sRs[tid * 4 + 0] = ob; sRs[tid * 4 + 1] = cc; sRs[tid * 4 + 2] = db;
        }
        __syncthreads();
        SM_MARK(5);
        // ---- dPHI in the accumulators, PHI'c / PHI'dbeta, then the moment products
        double wj[NQW > 0 ? NQW : 1], vj[NQW > 0 ? NQW : 1];
#pragma unroll
        for (int q = 0; q < NQW; ++q) {
            const int col = (4 * q + wce) * 16 + (lane & 15), cc = col < m ? col : m - 1;   // (PHI_ij = 0 beyond m: any finite value does)
            wj[q] = (a.w + oz)[cc];
            vj[q] = a.vscale * (a.v + oz)[cc];               // (homoscedastic: v points at w, vscale = 0)
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ob = (prs + oz)[(t * 16 + 4 * r) * 4 + 0], cc = (prs + oz)[(t * 16 + 4 * r) * 4 + 1], db = (prs + oz)[(t * 16 + 4 * r) * 4 + 2];
#pragma unroll
                for (int q = 0; q < NQW; ++q) {
                    const double ph = (pc + oz)[(t * 16 + 4 * r) * SM_LDA + 64 * q];
                    acc[t][q][r] = (-ob * acc[t][q][r] - cc * wj[q] + db * vj[q]) * ph;
                    r1[q] = fma(ph, cc, r1[q]);
                    r2[q] = fma(ph, db, r2[q]);
                }
            }
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
#pragma unroll
        for (int fb2 = 0; fb2 < (F2 ? 2 : 1); ++fb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = (4 * q + wce) * 16 + (lane >> 4) + 4 * r, f = fb2 * 16 + (lane & 15);
                if (j < m && f < a.nf) rec[(size_t)j * nrec + f] = Mq[q][fb2][r];
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

template <int NQ, bool F2>
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
    if (nqw == NQ) small_tail_run<NQ, F2>(a, smem, wce);
    else small_tail_run<NQ - 1, F2>(a, smem, wce);         // (NQ = ceil(nblk / 4): every wave has NQ or NQ - 1)
}

// raw sums about mu -> the moment records of k_moments_fused: [M1 (d) | S (d, or the packed d(d+1)/2) | PHI'c, PHI'dbeta] per basis function.
//   sum dp (x - p)       = R1 - q R0,                        q = p - mu
//   sum dp (x - p)_a (x - p)_b = R2_ab - q_a R1_b - q_b R1_a + q_a q_b R0
__global__ void k_small_convert(const double *__restrict__ raw, int m, int d, int kind, int nf, const double *__restrict__ P,
                                const double *__restrict__ xmu, double *__restrict__ frec, int nm) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const double *R = raw + (size_t)j * (nf + 2);
    double *o = frec + (size_t)j * (nm + 2);
    const double R0 = R[0];
    for (int c = 0; c < d; ++c) o[c] = R[1 + c] - (P[(size_t)j * d + c] - xmu[c]) * R0;
    if (kind == GPZ_KIND_DIAG) {
        for (int c = 0; c < d; ++c) {
            const double q = P[(size_t)j * d + c] - xmu[c];
            o[d + c] = fma(q, fma(q, R0, -2.0 * R[1 + c]), R[1 + d + c]);
        }
    } else {
        int e = 0;
        for (int aa = 0; aa < d; ++aa)
            for (int bb = aa; bb < d; ++bb, ++e) {
                const double qa = P[(size_t)j * d + aa] - xmu[aa], qb = P[(size_t)j * d + bb] - xmu[bb];
                o[d + e] = R[1 + d + e] - qa * R[1 + bb] - qb * R[1 + aa] + qa * qb * R0;
            }
    }
    o[nm] = R[nf];
    o[nm + 1] = R[nf + 1];
}

int small_tail_features(int kind, int d) { return kind == GPZ_KIND_DIAG ? 1 + 2 * d : 1 + d + d * (d + 1) / 2; }
bool small_tail_fits(int kind, int d, int mp) { return mp <= 256 && (mp & 15) == 0 && small_tail_features(kind, d) <= 32; }
int small_tail_nwg() { return 2 * gpz_cu_count(); }   // persistent workgroups: two per compute unit

void launch_small_tail(hipStream_t st, const SmallTailArgs &a0, int nwg) {
    SmallTailArgs a = a0;
    a.ncu = gpz_cu_count();
    if (a.stagger <= 0) a.stagger = 5;
    const int nq = ((a.mp >> 4) + 3) / 4;
    const size_t lds = ((size_t)32 * SM_LDA + 32 * SM_LDE + 4 * 32 + 32 + 32 * 4) * sizeof(double);
    dim3 g(nwg), b(256);
#define SMALL_CASE(NQ_)                                                                                   \
    do {                                                                                                  \
        if (a.nf > 16) hipLaunchKernelGGL((k_small_tail<NQ_, true>), g, b, lds, st, a);                   \
        else hipLaunchKernelGGL((k_small_tail<NQ_, false>), g, b, lds, st, a);                            \
    } while (0)
    if (nq <= 1) SMALL_CASE(1);
    else if (nq == 2) SMALL_CASE(2);
    else if (nq == 3) SMALL_CASE(3);
    else SMALL_CASE(4);
#undef SMALL_CASE
}
void launch_small_convert(hipStream_t st, const double *raw, int m, int d, int kind, int nf, const double *P, const double *xmu,
                          double *frec, int nm) {
    hipLaunchKernelGGL(k_small_convert, dim3((m + 63) / 64), dim3(64), 0, st, raw, m, d, kind, nf, P, xmu, frec, nm);
}
