// gamd_f16x3.h — fp32-grade GEMMs on the fp16 matrix pipe by operand splitting.
//
// gfx950's fp32 matrix rate (v_mfma_f32_32x32x2_f32) is 1/16 of its fp16 rate (v_mfma_f32_32x32x16_f16).  Every
// fp32 operand is split into two fp16 numbers, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 of fp32's 24
// significand bits; fp16 subnormals are honoured by the conversion and by the MFMA, verified by
// probes/mfma_f16x3_probe.hip), and a product is evaluated as
//       W x  ~=  W_hi x_hi + (W_hi x_lo + W_lo x_hi)            (3 MFMAs, fp32 accumulate; W_lo x_lo ~ 2^-22 dropped)
// at 3/16 of the fp32 matrix time.  Measured error of a 128-term dot product: 3.6e-7 relative, the same as the
// fp32 MFMA (3.5e-7); forces of the reference goldens differ from the fp32 path by < 2e-6 relative (tolerance 1e-5).
//
// Range: operands pass through fp16, |x| < 65504 (LayerNorm outputs, RBFs, unit vectors and SiLU activations are far
// below that); not checked on the device.
//
// Layout: the chain layout of gamd_common.h carries over exactly as for bf16 (gamd_bf16.h): a lane's 16 features of
// tile t feed two K=16 steps u = 0,1 of 8 packed values; weights are packed on the host as two fp16 fragment
// images (hi | lo, 32 KiB each = 64 KiB per 128x128 matrix, the size of the fp32 image):
//     Wp[part][((tp*4 + t)*2 + u)*64 + lane][j] = part(W[32tp + n][feat(t, 8u + j, half)]),  j = 0..7
#pragma once
#include "gamd_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gamd_f16x2 __attribute__((ext_vector_type(2)));
typedef float gamd_f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned gamd_u32x4_t __attribute__((ext_vector_type(4)));

// 8 consecutive registers of a chain-layout block -> (hi, lo) operand pair of one K step
__device__ __forceinline__ void gamd_split8(const f32x16& v, int u, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const gamd_f32x2_t x = {v[8 * u + j], v[8 * u + j + 1]};
        const gamd_f16x2 h = __builtin_convertvector(x, gamd_f16x2);
        const gamd_f32x2_t r = x - __builtin_convertvector(h, gamd_f32x2_t);
        const gamd_f16x2 l = __builtin_convertvector(r, gamd_f16x2);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// one K step (t, u) of a 128x128 GEMM for all four output tiles: 12 MFMAs on four independent accumulators.
// W: LDS image [hi | lo]; xh/xl: the activation operands of this step.
template <bool F2>
__device__ __forceinline__ void gamd_f16x3_step(const f16x8* W, int lane, int t, int u, f16x8 xh, f16x8 xl,
                                                f32x16 (&acc)[4]) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const f16x8 wh = W[((tp * 4 + t) * 2 + u) * 64 + lane];
        const f16x8 wl = W[2048 + ((tp * 4 + t) * 2 + u) * 64 + lane];
        if (F2) {
            acc[tp] = mfma_f16(xl, wh, acc[tp]);
            acc[tp] = mfma_f16(xh, wl, acc[tp]);
            acc[tp] = mfma_f16(xh, wh, acc[tp]);
        } else {
            acc[tp] = mfma_f16(wh, xl, acc[tp]);
            acc[tp] = mfma_f16(wl, xh, acc[tp]);
            acc[tp] = mfma_f16(wh, xh, acc[tp]);
        }
    }
}

// acc (+)= W X^T (F1, chain layout out) or X W^T (F2, row layout out); X is an fp32 chain-layout block that is split
// on the fly (8 registers of operands live at a time)
template <bool F2>
__device__ __forceinline__ void gemm128_f16x3(const f16x8* W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 xh, xl;
            gamd_split8(X[t], u, xh, xl);
            gamd_f16x3_step<F2>(W, lane, t, u, xh, xl, acc);
        }
}
