// Shared device-side definitions for the gfx950 kernels of libgpz_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one wave.
//   A operand: lane l supplies A[i = l&15][k = l>>4]
//   B operand: lane l supplies B[k = l>>4][j = l&15]
//   C/D:       lane l, reg r holds C[row = (l>>4) + 4r][col = l&15]     (f64-specific map)
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define GPZ_KIND_DIAG 0   // GL, VL, GD, VD: Gamma expanded to m x d   (getPHI.m:28-35)
#define GPZ_KIND_COV  1   // GC, VC:         Gamma expanded to d x d x m (getPHI.m:36-39)

#define GPZ_LOG2   0.69314718055994530942
#define GPZ_LOG2PI 1.83787706640934548356

// padded leading dimension of K-major LDS tiles read with the MFMA operand pattern
// (16 consecutive doubles x 4 k-rows per ds_read_b64): stride == 16 (mod 32) doubles is conflict-free.
#define LDS_LD128 144
#define LDS_LD64   80

// 1/u for u > 0 without the library divide (~25 instructions, correctly rounded): v_rcp_f64 seed + two Newton steps, <= 1 ulp.
__device__ __forceinline__ double gpz_rcp(double u) {
    double y = __builtin_amdgcn_rcp(u);
    double e = fma(-u, y, 1.0);
    y = fma(y, e, y);
    e = fma(-u, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// The same with ONE Newton step: the v_rcp_f64 seed is good to ~2^-27, one step squares that - a few ulps, not correctly rounded.  For the
// per-(row, basis function, dimension) factors 1 / (1 + psi gamma^2) of the input-noise kernels, whose results are summed over d dimensions
// and n rows and gated at 1e-8: two multiply-adds less of the ~14 a triple costs.
__device__ __forceinline__ double gpz_rcp1(double u) {
    double y = __builtin_amdgcn_rcp(u);
    const double e = fma(-u, y, 1.0);
    return fma(y, e, y);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the block (256 threads = 4 waves); result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double *sh4) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh4[w] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}
