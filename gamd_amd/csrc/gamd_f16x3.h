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
// Small operands: for |x| < 0.125 the lo half falls into fp16's subnormal range (absolute quantum 6e-8), so an operand of size 0.04
// (a typical weight) is represented to 7e-7 relative rather than 2^-22 = 2.4e-7; measured over random architectures the forces
// carry 3-4 x the rounding error of the fp32 kernels (tests/matrix_sweep.py).  Scaling the weights by a power
// of two on the host would recover about a factor 2 at the price of an unscale in every post-op of six kernels; not done.
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

// ---- pieces shared by the conv-layer edge kernels on this pipe (conv_edge_f16x3.hip, wide_lp.hip) ----------------------
// An operand set: the (hi, lo) fp16 images of a 32 x 128 activation block in MFMA operand order, 64 registers
// (the size of the fp32 block it replaces): w[t][u][part] = 4 dwords = 8 halves of K step (t, u).
struct OpSet { gamd_u32x4_t w[4][2][2]; };

// one 1 KiB piece (k = 0 .. 64/NW - 1 for this wave) of the same copy, to be issued between MFMAs: with one wave per
// SIMD the ~100 cycles each LDS-DMA instruction takes to issue are otherwise dead time of the matrix pipe
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "m0" on the clobber lists: see gamd_common.h
template <int NW, int KB = 64>                         // KB: size of the image in KiB (64: [hi | lo] fp16 or fp32; 32: bf16)
__device__ __forceinline__ void stage_chunk(const float* __restrict__ gw, float* ldsbuf, int wave, unsigned lane16, int k) {
    // (inline assembly: a compiler-tracked global_load_lds turns the next wait of any kind into vmcnt(0) lgkmcnt(0), see
    // gamd_stage_weight_raw in gamd_common.h; the landing is guaranteed by the counted vmcnt of phase_barrier.)  A wave's
    // 64 / NW KiB are contiguous and addressed by the instruction's immediate offset (which advances the global and the LDS
    // side alike): base pair + M0 are rebuilt per call from one opaque scalar instead of living in 16 x 3 loop-invariant,
    // spilled SGPRs per matrix.
    int woff = wave * (KB / NW) * 1024 + (k >> 2) * 4096;
    asm volatile("" : "+s"(woff));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf + (unsigned)woff;
    const char* g0 = reinterpret_cast<const char*>(gw) + woff;
    switch (k & 3) {
        case 0: asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(g0), "s"(lds0) : "memory", "m0"); break;
        case 1: asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(g0), "s"(lds0) : "memory", "m0"); break;
        case 2: asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane16), "s"(g0), "s"(lds0) : "memory", "m0"); break;
        default: asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(lane16), "s"(g0), "s"(lds0) : "memory", "m0"); break;
    }
}

#pragma clang diagnostic pop

// End of a phase: every wave has its own weight DMA (issued at the phase start, before the N most
// recent VMEM loads) landed, then the workgroup meets.  The N prefetch loads stay in flight.
// vmcnt retires in order, so "at most N outstanding" proves the older DMA is done only if at least
// N loads really were issued after it: callers pass 0 on paths that skip the prefetch.
template <int N>
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
}

// pre-split e fragments written by edge_encode_f16x3.hip: [tile][t][u][hi|lo][lane][8 halves], 16 KiB per tile; scalar tile
// base + 32-bit lane offset (the tile index is wave-uniform): no 64-bit per-lane pointer to keep
__device__ __forceinline__ void load_e_tile_s(const float* __restrict__ e_frag, int tile, unsigned lane16, OpSet& P) {
    const char* base = reinterpret_cast<const char*>(e_frag) + (size_t)__builtin_amdgcn_readfirstlane(tile) * 16384;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                P.w[t][u][p] = *reinterpret_cast<const gamd_u32x4_t*>(base + (lane16 + (unsigned)(((t * 2 + u) * 2 + p) * 1024)));
}
struct SiluK2 { gamd_f32x2_t nl2e, one; };

__device__ __forceinline__ void silu_split_pair(OpSet& P, int t, int r0, float x0, float x1, const SiluK2& k) {
    const gamd_f32x2_t x = {x0, x1};
    const gamd_f32x2_t a = x * k.nl2e;
    const gamd_f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const gamd_f32x2_t d = e + k.one;
    const gamd_f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const gamd_f32x2_t y = x * r;
    const gamd_f16x2 h = __builtin_convertvector(y, gamd_f16x2);
    const gamd_f32x2_t rem = y - __builtin_convertvector(h, gamd_f32x2_t);
    const gamd_f16x2 l = __builtin_convertvector(rem, gamd_f16x2);
    const int u = r0 >> 3, dw = (r0 & 7) >> 1;
    const unsigned hb = __builtin_bit_cast(unsigned, h), lb = __builtin_bit_cast(unsigned, l);
    // the first pair of a K step starts a NEW register quad: inserting into the old one would keep the set's previous
    // contents (the operand of two phases ago) alive next to the accumulators that are being turned into it
    if (dw == 0) { P.w[t][u][0] = gamd_u32x4_t{hb, 0u, 0u, 0u}; P.w[t][u][1] = gamd_u32x4_t{lb, 0u, 0u, 0u}; }
    else { P.w[t][u][0][dw] = hb; P.w[t][u][1][dw] = lb; }
}

// 128x128 split-fp16 GEMM, output block by output block; acc[tp] = init(tp) right in front of its K loop; the post-op of
// block tp - 1 rides between the K steps of block tp; step(i) in front of K step i = 0..31.
template <bool F2, typename Init, typename Post, typename Step>
__device__ __forceinline__ void gemm128_f16x3_lazy(const f16x8* W, int lane, const OpSet& P, f32x16 (&acc)[4], Init init, Post post,
                                                   Step step) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        __builtin_amdgcn_sched_barrier(0);       // output blocks stay in program order: interleaving them keeps all four alive
        init(tp);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                step((tp * 4 + t) * 2 + u);
                const f16x8 wh = W[((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 wl = W[2048 + ((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 xh = __builtin_bit_cast(f16x8, P.w[t][u][0]), xl = __builtin_bit_cast(f16x8, P.w[t][u][1]);
                if (F2) {
                    acc[tp] = mfma_f16(xl, wh, acc[tp]);
                    acc[tp] = mfma_f16(xh, wl, acc[tp]);
                    acc[tp] = mfma_f16(xh, wh, acc[tp]);
                } else {
                    acc[tp] = mfma_f16(wh, xl, acc[tp]);
                    acc[tp] = mfma_f16(wl, xh, acc[tp]);
                    acc[tp] = mfma_f16(wh, xh, acc[tp]);
                }
                if (tp > 0) post(tp - 1, 2 * (t * 2 + u));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) post(3, 2 * k);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ f32x16 bias_block(const float* vb, int t, int half) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&vb[32 * t + 8 * q + 4 * half]);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[q * 4 + j] = v[j];
    }
    return o;
}

