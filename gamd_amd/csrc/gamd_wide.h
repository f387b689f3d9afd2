// gamd_wide.h — shared by the generic-width node kernels (wide.hip: k_node_wide, wide_d.hip: k_node_wide_d).
#pragma once
#include "gamd_common.h"

// A wave's quarter (32 output features) of one packed 128 x 128 weight block, fetched from L2 in one batch of 16 float4 per lane
// one block-GEMM ahead of its use, and the 64 MFMAs that consume it (32-atom tile, chain layout).
struct WQ { f32x4 w[16]; };

__device__ __forceinline__ void wq_load(const float* __restrict__ Wp, int quarter, int lane, WQ& o) {
    const f32x4* W = reinterpret_cast<const f32x4*>(Wp) + (size_t)quarter * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) o.w[i] = W[i * 64];
}
__device__ __forceinline__ void wq_gemm(const WQ& wq, const f32x16 (&X)[4], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma32(wq.w[t * 4 + q][j], X[t][q * 4 + j], acc);
}
