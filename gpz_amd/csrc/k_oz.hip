// T = PHI * [inv(SIGMA) | w] on the int8 matrix pipe (GPz.m:69,72,77) - the one route past the fp64 MFMA ceiling of gfx950.
//
// fp64 operands are cut into S = 7 balanced base-256 digit planes (Ozaki scheme I): x = scale * sum_p d_p 2^(-6 - 8p), d_p in
// [-128, 127].  The product is the sum over the 28 plane pairs (p, q) with p + q <= 6 of EXACT int32 GEMMs on
// v_mfma_i32_32x32x32_i8, one accumulator per level l = p + q (|sum| < 2^27 at K = 1024), recombined in fp64 by Horner in 2^-8:
//     T_ij = cs_j 2^-12 sum_l 2^(-8 l) C_l[i][j],       C_l = sum_{p + q = l} A_p B_q.
// PHI lies in [0, 1] (exp of a non-positive exponent; without input noise) and takes ONE fixed scale 2^-54 - no pass over a row for
// its maximum, and the planes could be written by the PHI build itself; B takes a power of two per column.  What is dropped
// (levels 7 ...) is 2^-52 of the scale per term: tools/ozaki_numerics.py at c4's shape, cond(SIGMA) = 5e8: max |dT| / max |T| =
// 1.3e-12 against 2.9e-13 of the fp64 product itself; tools/ozaki_probe.hip: 2.6e-15 on random data.
//
// Layouts.  A planes: [row panel of 128][K step of 32 columns][plane p][4 KB]: the 4 KB are the LDS image of a 128 x 32 byte tile,
// 16-byte chunk (row r, half h) at r * 2 + (h ^ bit 3 of r) - conflict-free for ds_read_b128 in the MFMA operand pattern.  B planes:
// [column panel of 64][K step][plane q][2 KB], chunk (column c, half h) likewise (k contiguous per column).  A workgroup stages one K
// step (7 x 4 KB + 7 x 2 KB = 42 KB, contiguous in both plane arrays) per LDS-DMA piece of 1 KiB (global_load_lds_dwordx4), three
// stages deep; 8 waves (2 per SIMD) on a 128 x 64 tile, each a 32 x 32 block with all seven levels (112 accumulator registers).
//
// STATUS (round 6): developer build only, not the default and not going to be - DESIGN.md section 8 has the arithmetic of the go / no-go.
// Known limitation of this measured-and-archived route: the slicing assumes FINITE operands with PHI in [0, 1]; a NaN /
// Inf in PHI or B (an evaluation whose Cholesky failed) becomes finite digits here, where the fp64 route propagates the NaN.
#include "gpz_dev.h"
#include "gpz_kernels.h"

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
#define OZ_S 7
#define OZ_ABLK 4096
#define OZ_BBLK 2048
#define OZ_STAGE (OZ_S * (OZ_ABLK + OZ_BBLK))
#define OZ_NSTAGE 3
#define OZ_GLDS(g, l, aux) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g), (__attribute__((address_space(3))) void *)(l), 16, 0, aux)

__host__ __device__ inline int oz_chunk(int r, int h) { return r * 2 + (h ^ ((r >> 3) & 1)); }
static inline int oz_ksteps(int mp) { return (mp + 31) / 32; }
size_t oz_a_bytes(long n_pad, int mp) { return (size_t)(n_pad / 128) * oz_ksteps(mp) * OZ_S * OZ_ABLK; }
size_t oz_b_bytes(int mp) { return (size_t)((mp + 63) / 64) * oz_ksteps(mp) * OZ_S * OZ_BBLK; }

// balanced base-256 digits of N (|N| <= 2^54), most significant first
__device__ __forceinline__ void oz_digits(long long N, signed char (&d)[OZ_S]) {
#pragma unroll
    for (int s = OZ_S - 1; s >= 0; --s) {
        const int dd = (int)(((N + 128) & 255) - 128);
        d[s] = (signed char)dd;
        N = (N - dd) >> 8;
    }
}

// A planes from PHI (n_pad x ld row-major, columns < m; everything else - the y columns, the padding up to a multiple of 32 - is
// zero: the matching rows of B are zero as well).  One thread per 16-byte chunk = 16 consecutive columns of one row.
__global__ __launch_bounds__(256) void k_oz_slice_a(const double *__restrict__ Phi, int ld, int m, int ksteps, char *__restrict__ Apl) {
    const int rp = blockIdx.y, ks = blockIdx.x, c = threadIdx.x;
    const int r = c >> 1, h = (c & 1) ^ ((r >> 3) & 1);
    const double *src = Phi + (size_t)(rp * 128 + r) * ld + ks * 32 + h * 16;
    signed char out[OZ_S][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int col = ks * 32 + h * 16 + e;
        const double x = col < m ? src[e] : 0.0;
        signed char d[OZ_S];
        oz_digits(__double2ll_rn(x * 0x1p54), d);
#pragma unroll
        for (int p = 0; p < OZ_S; ++p) out[p][e] = d[p];
    }
    char *dst = Apl + ((size_t)(rp * ksteps + ks) * OZ_S) * OZ_ABLK + c * 16;
#pragma unroll
    for (int p = 0; p < OZ_S; ++p) *reinterpret_cast<i4_t *>(dst + (size_t)p * OZ_ABLK) = *reinterpret_cast<const i4_t *>(out[p]);
}

// per column j of B: cs[j] = the power of two with |B_kj| / cs[j] <= 1/2 for every k (cs = 1 for an all-zero column)
__global__ __launch_bounds__(256) void k_oz_colscale(const double *__restrict__ B, int ldb, int krows, int mp, double *__restrict__ cs) {
    // 256 columns per workgroup; the 4 waves split the rows (coalesced row segments), LDS combines them
    __shared__ double smx[4][64];
    const int jb = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double mxw = 0.0;
    if (jb < mp)
        for (int k = w; k < krows; k += 4) mxw = fmax(mxw, fabs(B[(size_t)k * ldb + jb]));
    smx[w][threadIdx.x & 63] = mxw;
    __syncthreads();
    if (w != 0 || jb >= mp) return;
    const int j = jb;
    const double mx = fmax(fmax(smx[0][threadIdx.x], smx[1][threadIdx.x]), fmax(smx[2][threadIdx.x], smx[3][threadIdx.x]));
    int e = 0;
    if (mx > 0.0) (void)frexp(mx, &e);      // mx = f 2^e, f in [1/2, 1)
    cs[j] = (mx > 0.0 && mx < 1.0e300) ? ldexp(1.0, e + 1) : 1.0;
}
// B planes: one thread per chunk = 16 consecutive k of one column
__global__ __launch_bounds__(128) void k_oz_slice_b(const double *__restrict__ B, int ldb, int krows, int mp, int ksteps,
                                                     const double *__restrict__ cs, char *__restrict__ Bpl) {
    const int cp = blockIdx.y, ks = blockIdx.x, c = threadIdx.x;
    const int cc = c >> 1, h = (c & 1) ^ ((cc >> 3) & 1);
    const int j = cp * 64 + cc;
    const double inv = j < mp ? 0x1p54 / cs[j] : 0.0;
    signed char out[OZ_S][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = ks * 32 + h * 16 + e;
        const double x = (j < mp && k < krows) ? B[(size_t)k * ldb + j] : 0.0;
        signed char d[OZ_S];
        oz_digits(__double2ll_rn(x * inv), d);
#pragma unroll
        for (int q = 0; q < OZ_S; ++q) out[q][e] = d[q];
    }
    char *dst = Bpl + ((size_t)(cp * ksteps + ks) * OZ_S) * OZ_BBLK + c * 16;
#pragma unroll
    for (int q = 0; q < OZ_S; ++q) *reinterpret_cast<i4_t *>(dst + (size_t)q * OZ_BBLK) = *reinterpret_cast<const i4_t *>(out[q]);
}

// One 128 x 64 tile of T per workgroup, 8 waves (2 per SIMD), each a 32 x 32 block with all seven levels.  Every wave issues its share
// of the 42 LDS-DMA pieces of a K step (two dedicated loader waves were measured: 30.8 ms against 28.0 - the loop then waits for the
// slower of them at every step).  Epilogue as k_tgemm's - T out, nupart[(32-column group) * n_pad + row] = sum over the group's
// columns < m of PHI * T (GPz.m:69), phiw[row] = T[row][mcol] (GPz.m:77) - but through the LDS (free by then): the accumulators' layout
// (a lane = one column) gives 8-byte accesses, and the epilogue is bound by the ISSUE of memory instructions; transposed, a lane moves
// 16 bytes of a row.
#ifndef OZ_NLW
#define OZ_NLW 0   // dedicated loader waves (0: the compute waves issue the DMA themselves)
#endif
#ifndef OZ_AUX_B
#define OZ_AUX_B 0   // cache policy of the B pieces (re-read by every row panel); the A pieces are streamed once: nt
#endif
template <int NLW>
__global__ __launch_bounds__(512 + 64 * NLW) void k_oz_tgemm(const char *__restrict__ Apl, const char *__restrict__ Bpl, const double *__restrict__ cs,
                                                     const double *__restrict__ Phi, int ld, double *__restrict__ T, int ldt, int mp,
                                                     int ksteps, int ncp, double *__restrict__ nupart, double *__restrict__ phiw, int m, int mcol,
                                                     long n_pad, int nslots) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware order: every XCD a contiguous range of tiles, the ncp column panels of a row panel back to back on one XCD
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int rp = lb / ncp, cp = lb % ncp;
    const char *ga = Apl + (size_t)rp * ksteps * (OZ_S * OZ_ABLK);
    const char *gb = Bpl + (size_t)cp * ksteps * (OZ_S * OZ_BBLK);
    if (NLW > 0 && wave >= 8) {
        // loader wave lw of NLW: pieces lw, lw + NLW, ... of the 42 of a K step; meets the compute waves at ONE barrier per step, behind
        // which stage ks has landed and the buffer of stage ks - 1 is free for stage ks + 2
        const int lw = wave - 8;
        constexpr int NP = NLW > 0 ? (42 + NLW - 1) / NLW : 1;
        const int mine = (42 - lw + NLW - 1) / (NLW > 0 ? NLW : 1);      // pieces of this wave per stage
        auto ldma = [&](int ks) {
            const char *sa = ga + (size_t)ks * (OZ_S * OZ_ABLK), *sb = gb + (size_t)ks * (OZ_S * OZ_BBLK);
            char *l0 = lds + (ks % OZ_NSTAGE) * OZ_STAGE;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int pc = lw + NLW * i;
                if (pc < 28) OZ_GLDS(sa + pc * 1024 + lane * 16, l0 + pc * 1024, 2);
                else if (pc < 42) OZ_GLDS(sb + (pc - 28) * 1024 + lane * 16, l0 + pc * 1024, OZ_AUX_B);
            }
        };
        ldma(0);
        if (ksteps > 1) ldma(1);
        for (int ks = 0; ks < ksteps; ++ks) {
            if (ks + 1 < ksteps) {   // all but the newest stage's pieces have landed
                if (mine == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP - 1) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ks + 2 < ksteps) ldma(ks + 2);
        }
        __builtin_amdgcn_s_barrier();   // (the compute waves' barrier in front of the epilogue)
        __builtin_amdgcn_s_barrier();   // (... and the one behind their LDS stores)
        return;
    }

    i16_t acc[OZ_S];
#pragma unroll
    for (int l = 0; l < OZ_S; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[l][r] = 0;

    // DMA pieces of 1 KiB: 28 of A, 14 of B; wave w issues pieces w, w + 8, ... - five each, waves 0 and 1 a sixth
    auto dma = [&](int ks, int st) {
        const char *sa = ga + (size_t)ks * (OZ_S * OZ_ABLK), *sb = gb + (size_t)ks * (OZ_S * OZ_BBLK);
        char *l0 = lds + st * OZ_STAGE;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pc = wave + 8 * i;
            if (pc < 28) OZ_GLDS(sa + pc * 1024 + lane * 16, l0 + pc * 1024, 2);
            else OZ_GLDS(sb + (pc - 28) * 1024 + lane * 16, l0 + pc * 1024, OZ_AUX_B);
        }
        if (wave < 2) OZ_GLDS(sb + (wave + 12) * 1024 + lane * 16, l0 + (wave + 40) * 1024, OZ_AUX_B);
    };
    auto wait_next = [&](bool more) {   // all but the newest stage's pieces of this wave have landed
        if (more) {
            if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int ar = 32 * wr + (lane & 31), bj = 32 * wc + (lane & 31), hh = lane >> 5;
    const int aoff = oz_chunk(ar, hh) * 16, boff = OZ_S * OZ_ABLK + oz_chunk(bj, hh) * 16;

    if (NLW == 0) {
        dma(0, 0);
        if (ksteps > 1) dma(1, 1);
    }
    for (int ks = 0; ks < ksteps; ++ks) {
        if (NLW == 0) wait_next(ks + 1 < ksteps);
        __builtin_amdgcn_s_barrier();   // (not __syncthreads(): its fence would wait for the DMA of the NEXT stage as well)
        if (NLW == 0 && ks + 2 < ksteps) dma(ks + 2, (ks + 2) % OZ_NSTAGE);
        const char *l0 = lds + (ks % OZ_NSTAGE) * OZ_STAGE;
        i4_t a[OZ_S], b[OZ_S];
#pragma unroll
        for (int p = 0; p < OZ_S; ++p) a[p] = *reinterpret_cast<const i4_t *>(l0 + aoff + p * OZ_ABLK);
#pragma unroll
        for (int q = 0; q < OZ_S; ++q) b[q] = *reinterpret_cast<const i4_t *>(l0 + boff + q * OZ_BBLK);
#pragma unroll
        for (int p = 0; p < OZ_S; ++p)
#pragma unroll
            for (int q = 0; q < OZ_S - p; ++q) acc[p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[q], acc[p + q], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // every wave has read its last fragments: the ring is free for the output tile

    // accumulator register r of a 32 x 32 tile: row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32  ->  sT[128][64 + 2] in LDS
    double (*sT)[66] = reinterpret_cast<double (*)[66]>(lds);
    {
        const int col = cp * 64 + 32 * wc + (lane & 31);
        const double csj = (col < mp ? cs[col] : 0.0) * 0x1p-12;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double t = 0.0;
#pragma unroll
            for (int l = OZ_S - 1; l >= 1; --l) t = (t + (double)acc[l][r]) * 0x1p-8;   // Horner in 2^-8, smallest level first
            sT[32 * wr + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)][32 * wc + (lane & 31)] = (t + (double)acc[0][r]) * csj;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // 512 threads: lane pair of columns c2 = 2 (tid % 32), rows tid / 32 + 16 i
    const int c2 = 2 * (tid & 31), rb = tid >> 5;
    const int col0 = cp * 64 + c2;
    const bool ok0 = col0 < mp, ok1 = col0 + 1 < mp;      // (mp is even: both or neither)
    const int colc = ok0 ? col0 : mp - 2;
    d2_t ph[8];
    if (nupart) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ph[i] = *reinterpret_cast<const d2_t *>(Phi + ((size_t)rp * 128 + rb + 16 * i) * ld + colc);
    }
    d2_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i].x = sT[rb + 16 * i][c2];
        v[i].y = sT[rb + 16 * i][c2 + 1];
    }
    if (ok0 && ok1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<d2_t *>(T + ((size_t)rp * 128 + rb + 16 * i) * ldt + col0) = v[i];
    }
    if (nupart) {
        const int grp = cp * 2 + (c2 >> 5);                  // 32-column group of this lane (16 lanes each)
        double *slotp = nupart + (size_t)grp * n_pad;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t row = (size_t)rp * 128 + rb + 16 * i;
            double p = (ok0 && col0 < m) ? ph[i].x * v[i].x : 0.0;
            p = (ok1 && col0 + 1 < m) ? fma(ph[i].y, v[i].y, p) : p;
            if (col0 == mcol) phiw[row] = v[i].x;
            if (col0 + 1 == mcol) phiw[row] = v[i].y;
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 8, 64);
            if ((tid & 15) == 0) {
                slotp[row] = p;
                // the consumers sum nslots 32-column groups per row (k_tgemm's count, a multiple of four): the ones past the last column
                // panel are zeroed here
                if (cp == ncp - 1 && 2 * ncp + (c2 >> 5) < nslots) nupart[(size_t)(2 * ncp + (c2 >> 5)) * n_pad + row] = 0.0;
            }
        }
    }
}

void launch_oz_slice_a(hipStream_t st, const double *Phi, int ld, long n_pad, int m, int mp, char *Apl) {
    const int ks = oz_ksteps(mp);
    hipLaunchKernelGGL(k_oz_slice_a, dim3(ks, (unsigned)(n_pad / 128)), dim3(256), 0, st, Phi, ld, m, ks, Apl);
}
void launch_oz_slice_b(hipStream_t st, const double *B, int ldb, int krows, int mp, double *cs, char *Bpl) {
    const int ks = oz_ksteps(mp);
    hipLaunchKernelGGL(k_oz_colscale, dim3((mp + 63) / 64), dim3(256), 0, st, B, ldb, krows, mp, cs);
    hipLaunchKernelGGL(k_oz_slice_b, dim3(ks, (mp + 63) / 64), dim3(128), 0, st, B, ldb, krows, mp, ks, (const double *)cs, Bpl);
}
// the kernel's 126 KB of dynamic LDS have to be allowed once per device (called at context creation: not a stream operation)
int oz_prepare_device() {
    return hipFuncSetAttribute((const void *)k_oz_tgemm<OZ_NLW>, hipFuncAttributeMaxDynamicSharedMemorySize, OZ_NSTAGE * OZ_STAGE) == hipSuccess ? 0 : -1;
}
void launch_oz_tgemm(hipStream_t st, const char *Apl, const char *Bpl, const double *cs, const double *Phi, int ld, double *T, int ldt,
                     long n_pad, int mp, double *nupart, double *phiw, int m, int mcol) {
    const int ncp = (mp + 63) / 64, nslots = gpz_gemm_wave_cols() * ((mp + 127) / 128);
    hipLaunchKernelGGL(k_oz_tgemm<OZ_NLW>, dim3((unsigned)((n_pad / 128) * ncp)), dim3(512 + 64 * OZ_NLW), OZ_NSTAGE * OZ_STAGE, st, Apl, Bpl, cs, Phi, ld, T, ldt, mp,
                       oz_ksteps(mp), ncp, nupart, phiw, m, mcol, n_pad, nslots);
}
