// gamd_bf16.h — bf16 variant of the chain layout (BASELINE config 5: bf16 edge-MLP on MFMA, fp32 accumulate).
//
// Accumulators stay fp32 in the chain layout of gamd_common.h.  v_mfma_f32_32x32x16_bf16 consumes 8 bf16
// per lane per operand (K = 16 per instruction), so the 16 features a lane owns in tile t,
//     feat(t, r, half) = 32t + (r&3) + 8(r>>2) + 4half,   r = 0..15,
// are fed as two K-steps u = 0,1 of 8 packed values (r = 8u..8u+7).  The weight fragment for
// (output tile tp, input tile t, step u) is, for lane (n = lane&31, half):
//     Wp[((tp*4 + t)*2 + u)*64 + lane][j] = bf16( W[32tp + n][feat(t, 8u + j, half)] ),  j = 0..7
// (32 KiB per 128x128 matrix: all four matrices of a conv layer fit in LDS at once, so the bf16 kernel
// needs no weight streaming and no barriers in its main loop).  Verified on hardware by
// probes/mfma_bf16_layout_probe.hip.
#pragma once
#include "gamd_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gamd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gamd_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gamd_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gamd_u32x2 __attribute__((ext_vector_type(2)));

#define GAMD_WFRAG_BF16_BYTES (GAMD_H * GAMD_H * 2)      // 32 KiB

// two floats -> packed bf16 pair, round to nearest even (one v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned gamd_pk_bf16(float lo, float hi) {
    const gamd_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gamd_bf16x2));
}

// chain-layout fp32 block -> packed MFMA operands P[t][u]
__device__ __forceinline__ void pack_chain_bf16(const f32x16 (&X)[4], bf16x8 (&P)[4][2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            gamd_u32x4 w;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = gamd_pk_bf16(X[t][8 * u + 2 * k], X[t][8 * u + 2 * k + 1]);
            P[t][u] = __builtin_bit_cast(bf16x8, w);
        }
}

// acc (+)= W * P^T (F1, chain layout out) or P * W^T (F2, row layout out); 32 MFMAs, 4 independent
// accumulators rotate inside every K-step
// (HALFREAD: timing ablation of the profiling build — both K steps u use the u = 0 fragment, i.e. half the LDS reads)
template <bool F2, bool HALFREAD = false, typename WPtr>
__device__ __forceinline__ void gemm128_bf16(WPtr W, int lane, const bf16x8 (&P)[4][2], f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const bf16x8 w = W[((tp * 4 + t) * 2 + (HALFREAD ? 0 : u)) * 64 + lane];
                acc[tp] = F2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[t][u], w, acc[tp], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, P[t][u], acc[tp], 0, 0, 0);
            }
}

// The same GEMM with the weight fragments read D MFMAs ahead through an explicit ring of D x 4 registers.  hipcc's own schedule
// of gemm128_bf16 is read -> s_waitcnt -> MFMA with one or two fragments in flight, so every 32-cycle MFMA sits out most of an
// LDS round trip: 97 cycles per MFMA measured in k_conv_edge_bf16 (tools/bf16_variants.py: the kernel without its MFMAs took
// 17 us, with them 49 us, for 10 us of matrix time).  Scheduling barriers in front of and behind the GEMM keep other memory
// instructions out of it, sched_group_barrier pins the pattern (one LDS read, one MFMA) inside; the waits are hipcc's own
// counted lgkmcnt (LDS reads return in order), so they stay correct whatever else is in flight.
template <bool F2, int D>
__device__ __forceinline__ void gemm128_bf16_pf(const bf16x8* W, int lane, const bf16x8 (&P)[4][2], f32x16 (&acc)[4]) {
    // MFMA i = (t, u, tp) with tp fastest: four independent accumulators rotate; its fragment sits at ((tp*4 + t)*2 + u) * 64
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) w[i] = W[((((i & 3) * 4 + (i >> 3)) * 2) + ((i >> 2) & 1)) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int tp = i & 3, t = i >> 3, u = (i >> 2) & 1;
        const bf16x8 cur = w[i % D];
        acc[tp] = F2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[t][u], cur, acc[tp], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, P[t][u], acc[tp], 0, 0, 0);
        if (i + D < 32) {
            const int j = i + D;
            w[i % D] = W[((((j & 3) * 4 + (j >> 3)) * 2) + ((j >> 2) & 1)) * 64 + lane];
        }
    }
    // D reads up front, then (MFMA, read) pairs, then the last D MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int i = 0; i < 32 - D; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, D, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- operand-set form (wide_lp.hip; the structure of conv_edge_f16x3.hip on bf16 operands) -----------------------------------
struct OpSetB { gamd_u32x4 w[4][2]; };           // bf16 operand set of a 32 x 128 block: w[t][u] = the 8 values of K step (t, u)
struct SiluKB { gamd_f32x2 nl2e, one; };

// SiLU of two accumulator elements + rounding to bf16, written as dword (r0 & 7) / 2 of K step (t, r0 >> 3)
__device__ __forceinline__ void silu_pack_pair(OpSetB& P, int t, int r0, float x0, float x1, const SiluKB& k) {
    const gamd_f32x2 x = {x0, x1};
    const gamd_f32x2 a = x * k.nl2e;
    const gamd_f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const gamd_f32x2 d = e + k.one;
    const gamd_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const gamd_f32x2 y = x * r;
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(y, gamd_bf16x2));
    const int u = r0 >> 3, dw = (r0 & 7) >> 1;
    if (dw == 0) P.w[t][u] = gamd_u32x4{b, 0u, 0u, 0u};          // a NEW register quad (see silu_split_pair, gamd_f16x3.h)
    else P.w[t][u][dw] = b;
}

// 128x128 bf16 GEMM, output block by output block: acc[tp] initialised by init(tp) right in front of its 8 MFMAs, the post-op of
// block tp - 1 between the MFMAs of block tp, step(i) in front of K step i = 0..31 (gemm128_f16x3_lazy, gamd_f16x3.h)
template <bool F2, typename Init, typename Post, typename Step>
__device__ __forceinline__ void gemm128_bf16_lazy(const bf16x8* W, int lane, const OpSetB& P, f32x16 (&acc)[4], Init init, Post post, Step step) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        __builtin_amdgcn_sched_barrier(0);
        init(tp);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                step((tp * 4 + t) * 2 + u);
                const bf16x8 w = W[((tp * 4 + t) * 2 + u) * 64 + lane];
                const bf16x8 x = __builtin_bit_cast(bf16x8, P.w[t][u]);
                acc[tp] = F2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, acc[tp], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc[tp], 0, 0, 0);
                if (tp > 0) post(tp - 1, 2 * (t * 2 + u));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) post(3, 2 * k);
    __builtin_amdgcn_sched_barrier(0);
}
