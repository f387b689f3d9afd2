// gamd_common.h — shared device helpers and data-layout contract of the gfx950 kernels.
//
// Layout contract:
//
// * Work unit of every MFMA kernel is a "wave tile": 32 rows (edges or atoms) owned by one
//   64-lane wavefront.  lane = (slot = lane & 31, half = lane >> 5).
// * A 128-wide activation row-block lives in registers as X[4] (f32x16 each), the "chain layout":
//       lane (slot, half), X[t][r]  <->  row `slot`, feature  32*t + (r&3) + 8*(r>>2) + 4*half
//   This is exactly the C/D fragment layout of v_mfma_f32_32x32x2_f32 when the MFMA computes
//   Y^T = W * X^T (rows = output features, cols = slots), AND exactly the A/B operand layout the
//   next MFMA wants for its K index (lanes 0-31 supply k_lo, lanes 32-63 supply k_hi = k_lo + 4).
//   So a GEMM chain  X -> act(W1 X) -> act(W2 .) -> ...  never leaves the register file.
// * The LDS/global-resident operand (the weights) is pre-packed on the host into fragment order
//       Wp[((tp*4 + t)*4 + q)*64 + lane]  (float4)  =  W[32*tp + slot][32*t + 8*q + 4*half + 0..3]
//   so that one conflict-free ds_read_b128 (lane-linear, 1 KiB per wave) feeds four MFMAs.
//   The same packed image serves both orientations:
//     F1 (A = W fragment, B = X reg):  D[row = out feature][col = slot]   -> chain layout again
//     F2 (A = X reg, B = W fragment):  D[row = slot][col = out feature]   -> "row layout":
//        lane (nu = lane&31, half), acc[tp][r] <-> out feature 32*tp + nu, slot (r&3)+8*(r>>2)+4*half
//   F2 is used for the last GEMM of a conv layer so that the per-destination segment sum becomes
//   an in-lane sum over r (no cross-lane traffic, no atomics).
// * Slot <-> CSR edge mapping inside a 32-edge tile: slot rho holds CSR edge
//       base + PI(rho),  PI(rho) = 16*((rho>>2)&1) + (rho&3) + 4*(rho>>3)
//   so that in the F2 row layout, (half, r) enumerates CSR edges base + 16*half + r in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GAMD_H 128            // feature width of every hidden tensor (all shipped configs)
#define GAMD_TILE 32          // rows per wave tile
#define GAMD_CHUNK 16         // edges per (tile, half) = granularity of partial-sum pieces
#define GAMD_WFRAG_FLOATS (GAMD_H * GAMD_H)   // one packed 128x128 weight = 64 KiB

__device__ __forceinline__ int gamd_pi(int rho) {
    return 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One 128x128 GEMM step of the chain.  W: packed fragment image (LDS or global), X: input in chain
// layout, acc: accumulators (pre-initialised with bias / zeros by the caller).
//   F2 == false:  acc (chain layout)  += W * X^T
//   F2 == true :  acc (row layout)    += X * W^T
template <bool F2, typename WPtr>
__device__ __forceinline__ void gemm128_tile(WPtr W, int lane, const f32x16 (&X)[4], f32x16& acc, int tp) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = W[((tp * 4 + t) * 4 + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = X[t][q * 4 + j];
                acc = F2 ? mfma32(x, w[j], acc) : mfma32(w[j], x, acc);
            }
        }
    }
}

template <bool F2, typename WPtr>
__device__ __forceinline__ void gemm128(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) gemm128_tile<F2>(W, lane, X, acc[tp], tp);
}

// ---- activations ---------------------------------------------------------------------------------
// Hardware-transcendental forms used inside the MFMA kernels.  On gfx950 fp32 MFMA issues at the fp32
// VALU rate and VALU work is NOT hidden behind it (measured: a software-pipelined SiLU costs the same
// 8 % as a trailing one), so activations are written for the fewest VALU cycles that keep fp32 accuracy.
__device__ __forceinline__ float gamd_silu_hw(float x) {    // v_exp_f32 + v_rcp_f32, ~1 ulp each
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GELU(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|), with the normal tail as ONE exponential of a polynomial:
//     Phi(-a) = 2^Q(a),  Q = degree-6 least-squares fit of log2 Phi(-a) on [0, 6] weighted by a Phi(-a) (the factor the
// error is multiplied by in the result).  |fit error| <= 1.1e-7 in |x| Phi(-|x|); evaluated in fp32 the result is within
// 0.95 (half-ulp + 1.2e-7) of the exact erf-GELU (nn.GELU(), nn_module.py:41-42) on [-8, 8] — closer than the Abramowitz &
// Stegun 7.1.26 erfc form used in round 1 (1.8) — with 11 VALU instructions (one v_exp_f32) instead of 17 (v_exp_f32 +
// v_rcp_f32): measured 465 -> 432 us on k_edge_encode at C2 (tools/enc_variants.py; a packed v_pk_fma_f32 Horner chain
// was slower than the scalar one).  Beyond |x| = 6 the tail is below 1e-9: the argument is
// clamped there (one v_min with the |.| source modifier), which also keeps the fit's positive leading coefficient harmless.
#define GAMD_GELU_Q0 -9.999880791e-01f
#define GAMD_GELU_Q1 -1.151242852e+00f
#define GAMD_GELU_Q2 -4.586574435e-01f
#define GAMD_GELU_Q3 -5.355345458e-02f
#define GAMD_GELU_Q4 8.167289197e-03f
#define GAMD_GELU_Q5 -7.945232792e-04f
#define GAMD_GELU_Q6 3.589583139e-05f
__device__ __forceinline__ float gamd_gelu_hw(float x) {
    // v_med3_f32 instead of fminf / fmaxf: those canonicalise their operands first (one extra v_max each)
    const float a = __builtin_amdgcn_fmed3f(fabsf(x), 0.0f, 6.0f);
    float q = fmaf(GAMD_GELU_Q6, a, GAMD_GELU_Q5);
    q = fmaf(q, a, GAMD_GELU_Q4);
    q = fmaf(q, a, GAMD_GELU_Q3);
    q = fmaf(q, a, GAMD_GELU_Q2);
    q = fmaf(q, a, GAMD_GELU_Q1);
    q = fmaf(q, a, GAMD_GELU_Q0);
    // max(x, 0) as x - clamp(x, -big, 0): exact for finite x, and unlike v_max / v_med3 (which return the non-NaN operand)
    // it lets a NaN input through, so non-finite positions or weights still surface in the forces (STICKY_NONFINITE)
    const float relu = x - __builtin_amdgcn_fmed3f(x, -3.0e38f, 0.0f);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), relu);
}
// The seven coefficients of gamd_gelu_hw's exponent polynomial as register PAIRS (c, c): v_pk_fma_f32 cannot take a literal,
// and left to itself hipcc keeps the Horner chain scalar (6 x v_fmaak_f32 per element).
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
struct GeluCoef { gelu_f2 q[7]; };
__device__ __forceinline__ GeluCoef gelu_coef() {
    GeluCoef k;
    const float c[7] = {GAMD_GELU_Q0, GAMD_GELU_Q1, GAMD_GELU_Q2, GAMD_GELU_Q3, GAMD_GELU_Q4, GAMD_GELU_Q5, GAMD_GELU_Q6};
#pragma unroll
    for (int i = 0; i < 7; ++i) { k.q[i] = gelu_f2{c[i], c[i]}; asm volatile("" : "+v"(k.q[i])); }
    return k;
}
__device__ __forceinline__ gelu_f2 pk_fma(gelu_f2 a, gelu_f2 b, gelu_f2 c) {
    gelu_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// GELU of two elements: per element exactly the operations of gamd_gelu_hw (same IEEE fused multiply-adds in the same
// order, so the bits are the same), the six Horner steps as packed instructions: 77 lane cycles per pair instead of 96.
__device__ __forceinline__ gelu_f2 gelu_pair(gelu_f2 x, const GeluCoef& k) {
    const gelu_f2 a = {__builtin_amdgcn_fmed3f(fabsf(x[0]), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(x[1]), 0.0f, 6.0f)};
    gelu_f2 q = pk_fma(k.q[6], a, k.q[5]);
    q = pk_fma(q, a, k.q[4]);
    q = pk_fma(q, a, k.q[3]);
    q = pk_fma(q, a, k.q[2]);
    q = pk_fma(q, a, k.q[1]);
    q = pk_fma(q, a, k.q[0]);
    const gelu_f2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const gelu_f2 relu = {x[0] - __builtin_amdgcn_fmed3f(x[0], -3.0e38f, 0.0f), x[1] - __builtin_amdgcn_fmed3f(x[1], -3.0e38f, 0.0f)};
    // The last step stays a compiler-visible instruction per element: its result is the next GEMM's MFMA operand, and gfx940+
    // needs two wait states between a VALU write and an MFMA read of the same register (and one behind a transcendental) —
    // hipcc's hazard recogniser inserts them for its own instructions but cannot see the operands of inline assembly.  The
    // packed Horner steps above only ever feed v_exp_f32.
    gelu_f2 r = {__builtin_fmaf(-a[0], e[0], relu[0]), __builtin_fmaf(-a[1], e[1], relu[1])};
    return r;
}

// RBF expansion exp(-gamma (d - mu_k)^2) of the standardised length on a UNIFORM grid of centres mu_k = c0 + k delta
// (nn_module.py:237-240: linspace(0, 1, 40), gamma = 40).  A lane holds every other centre (k = 2 j + half); along such a
// chain, with u = d - mu and step s = 2 delta,
//     g(u - s) = g(u) rho(u),   rho(u) = 2^(A u + B),   rho(u - s) = rho(u) C,
//     A = -2 gexp s,  B = gexp s^2,  C = 2^(2 gexp s^2)     (gexp = -gamma log2 e)
// so four chains of five centres cost 8 v_exp_f32 + 28 multiplies instead of 20 v_exp_f32 + 60 other instructions.  A chain is
// restarted every five centres to keep the accumulated rounding below 5e-7 relative.  Underflow of a chain's first value
// loses only values below 1e-27.  The host checks that the state_dict's centres are uniform (gamd_finalize_weights) and
// falls back to the table otherwise.
struct RbfGrid { int uniform; float c0, delta, A, B, C; };
__device__ __forceinline__ void gamd_rbf_chains(float d, int half, float gexp, const RbfGrid& g, float (&F)[24]) {
#pragma clang fp contract(off)      // the same bits in every kernel this is inlined into (see edge_features, edge_encode.hip)
    const float s = 2.0f * g.delta;
    const float dh = d - (g.c0 + (half ? g.delta : 0.0f));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float u0 = dh - (float)(5 * c) * s;
        float v = __builtin_amdgcn_exp2f(gexp * (u0 * u0));
        float rho = __builtin_amdgcn_exp2f(fmaf(g.A, u0, g.B));
        F[2 + 5 * c] = v;
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            v *= rho;
            F[2 + 5 * c + j] = v;
            if (j < 4) rho *= g.C;
        }
    }
}

// torch.remainder for floats (result takes the sign of the divisor)
__device__ __forceinline__ float gamd_remainder(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

// minimum image of nn_module.py:617-621, remainder(d + L/2, L) - L/2, for |d| <= L (differences of wrapped
// coordinates): fmod is exact, so this branch form is bit-identical to torch.remainder on that range
__device__ __forceinline__ float gamd_min_image_wrapped(float d, float L, float halfL) {
    float t = d + halfL;
    if (t < 0.0f) t += L;
    else if (t >= L) t -= L;
    return t - halfL;
}

// The edge embeddings are a STREAM: every conv-layer launch reads each e tile exactly once (256 - 512 B per edge, 15 - 30 KB per
// atom) next to gathers of node-table rows (hn, S: 1 KB per atom) that are re-used by the ~60 edges of each neighbour.  Loaded
// with the non-temporal hint (global_load ... nt) the stream does not push the rows out of the XCD's 4 MiB L2: measured where
// the tables no longer fit (10^5 - 10^6 atoms, profiles/r05_gather_hbm.md); at the BASELINE sizes everything is L2-resident
// either way.  -DGAMD_E_TEMPORAL builds the plain loads for the A/B.
__device__ __forceinline__ f32x4 gamd_load_stream(const f32x4* p) {
#ifdef GAMD_E_TEMPORAL
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}

// Message of one edge folded into the running sum of its partial-sum piece (nn_module.py:142, u_mul_e -> sum): a single fused
// multiply-add, i.e. hn[src] * e_emb is not rounded before it is added (the reference multiplies and adds separately; the
// two differ by at most half an ulp of the product).  Every fp32 conv-layer edge kernel uses this form, so they stay
// bit-identical to one another.
__device__ __forceinline__ float gamd_msg_acc(float hn, float e_emb, float prev) { return __builtin_fmaf(hn, e_emb, prev); }

// fp16 node tables of the bf16 edge MLP (NodeArgs::tab16): position, in fp16 elements, of feature f inside an S / D row.  A
// lane (slot, half) of the chain layout owns the 64 features with bit 2 == half; they are stored as eight 16-byte groups
// c = 2 t + k (t = f >> 5, k = bit 4 of f) holding X[t][8 k .. 8 k + 7], the two halves of a group side by side: one
// global_load_dwordx4 per group touches 32 bytes per row (half a fp32 row's worth of cache lines and instructions).
__host__ __device__ __forceinline__ int gamd_tab16_pos(int f) {
    return 16 * (2 * (f >> 5) + ((f >> 4) & 1)) + 8 * ((f >> 2) & 1) + 4 * ((f >> 3) & 1) + (f & 3);
}

// bias fragment in chain layout: 16 float4 (one per (t,q)), from a plain [128] vector
template <typename BPtr>
__device__ __forceinline__ void load_bias_chain(BPtr b, int half, f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&b[32 * t + 8 * q + 4 * half]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][q * 4 + j] = v[j];
        }
}

// row `row_ptr` (plain row-major [128] floats) -> chain layout registers (16 x 16-byte loads)
__device__ __forceinline__ void load_row_chain(const float* __restrict__ row, int half, f32x16 (&X)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * t + 8 * q + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
        }
}

__device__ __forceinline__ void store_row_chain(float* __restrict__ row, int half, const f32x16 (&X)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = X[t][q * 4 + j];
            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q + 4 * half) = v;
        }
}

// sum of a per-lane value over the two halves of a slot (lane l <-> l ^ 32)
__device__ __forceinline__ float gamd_xhalf_sum(float v) {
    return v + __shfl_xor(v, 32, 64);
}

// LayerNorm over the 128 features of each slot, in place on chain-layout registers.
// torch.nn.LayerNorm: biased variance, eps inside the sqrt.
template <typename GPtr>
__device__ __forceinline__ void layernorm_chain(f32x16 (&X)[4], GPtr gamma, GPtr beta, int half, float eps) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += X[t][r];
    const float mean = gamd_xhalf_sum(s) * (1.0f / 128.0f);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = X[t][r] - mean;
            v += d * d;
        }
    const float var = gamd_xhalf_sum(v) * (1.0f / 128.0f);
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(&gamma[32 * t + 8 * q + 4 * half]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&beta[32 * t + 8 * q + 4 * half]);
#pragma unroll
            for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = (X[t][q * 4 + j] - mean) * rstd * g[j] + b[j];
        }
}

// ---- shared building blocks of several kernels -------------------------------------------------------------------
// L2 -> LDS copy of one packed 64 KiB weight matrix by the NW waves of a workgroup (64 / NW x 1 KiB per wave) with
// global_load_lds.  lane16 = lane * 16 is made opaque so the 64-bit addresses are rebuilt (1 VALU each) instead of
// being hoisted out of the tile loop and spilled (a spilled pointer = scratch reload + s_waitcnt vmcnt(0) in front of
// every copy).
template <int NW = 8>
__device__ __forceinline__ void gamd_stage_weight(const float* __restrict__ gw, float* ldsbuf, int wave, unsigned lane16) {
    asm volatile("" : "+v"(lane16));
#pragma unroll
    for (int k = 0; k < 64 / NW; ++k) {
        const int chunk = k * NW + wave;      // 64 chunks of 1 KiB, lane-linear image == packed global image
        const char* base = reinterpret_cast<const char*>(gw) + chunk * 1024;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + lane16),
            (__attribute__((address_space(3))) void*)(ldsbuf + chunk * 256), 16, 0, 0);
    }
}

// M0 (the LDS address of an LDS-DMA) is written inside the inline assembly below and named in its clobber list, so that
// hipcc never assumes a value it put into M0 itself (for a tracked global_load_lds or a relative-indexed access) survives
// the statement; clang warns that M0 is a reserved register on this target — that is exactly why it is listed.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"

// The same copy issued from inline assembly, i.e. INVISIBLE to hipcc's s_waitcnt insertion.  For a tracked
// global_load_lds (FLAT encoding, touches both global memory and LDS) the compiler keeps a "pending flat" state that
// turns the next wait of ANY kind — e.g. the lgkmcnt wait of the first weight ds_read of the phase — into
// `s_waitcnt vmcnt(0) lgkmcnt(0)`: every wave would sit out the full L2 -> LDS round trip of the copy it has just issued,
// for the NEXT phase, before its first MFMA.  Callers own the ordering: the data may be read only after an explicit counted
// `s_waitcnt vmcnt(N)` (N = VMEM ops the wave issued after this copy) and a workgroup barrier.  Compiler-generated vmcnt
// waits for other loads stay correct (they do not count these eight ops, so they can only wait longer, never shorter).
template <int NW = 8>
__device__ __forceinline__ void gamd_stage_weight_raw(const float* __restrict__ gw, float* ldsbuf, int wave, unsigned lane16) {
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;
#pragma unroll
    for (int k = 0; k < 64 / NW; ++k) {
        const int chunk = k * NW + wave;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     ::"v"(lane16), "s"(reinterpret_cast<const char*>(gw) + chunk * 1024), "s"(lds0 + chunk * 1024u)
                     : "memory", "m0");
    }
}

// The same copy with each wave's 64 / NW KiB CONTIGUOUS (chunks wave * 64 / NW ... ) and addressed by the instruction's
// immediate offset, which advances the global and the LDS side alike: one scalar base pair + one M0 value per four
// KiB-sized copies instead of one of each per copy.  The per-copy form above keeps 8 x 3 loop-invariant SGPRs per matrix
// alive, which hipcc spills to VGPR lanes and restores with two v_readlane per copy (~160 per tile in k_conv_edge).
template <int NW = 8>
__device__ __forceinline__ void gamd_stage_weight_raw_contig(const float* __restrict__ gw, float* ldsbuf, int wave, unsigned lane16) {
    static_assert((64 / NW) % 4 == 0, "whole groups of four KiB-sized copies per wave");
    int woff = wave * (64 / NW) * 1024;
    asm volatile("" : "+s"(woff));                      // rebuilt per call (2 scalar adds), never kept across the tile loop
    const char* g0 = reinterpret_cast<const char*>(gw) + woff;
    const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf + (unsigned)woff;
#pragma unroll
    for (int h = 0; h < 64 / NW / 4; ++h)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:3072"
                     ::"v"(lane16), "s"(g0 + h * 4096), "s"(l0 + h * 4096u)
                     : "memory", "m0");
}

#pragma clang diagnostic pop

// Latency-oriented split of a 32-row tile over the 4 waves of a 256-thread workgroup (node.hip, conv_edge_small.hip,
// wide.hip's node kernel): wave `quarter` computes output features [32 quarter, 32 quarter + 32) of every GEMM from its
// 16 KiB weight quarter, fetched from L2 in ONE batch of 16 float4 per lane (a GEMM then costs one L2 round trip, not
// sixteen), and the 128-wide activation rows are re-assembled through a padded LDS exchange buffer between GEMMs.
struct WQuarter { f32x4 w[16]; };

__device__ __forceinline__ void load_wquarter(const float* __restrict__ Wp, int quarter, int lane, WQuarter& o) {
    const f32x4* W = reinterpret_cast<const f32x4*>(Wp) + (size_t)quarter * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) o.w[i] = W[i * 64];
}

// acc (this wave's 32 output features x 32 rows, one C tile) += W[quarter] * X^T   (F2: X * W[quarter]^T, row layout)
template <bool F2 = false>
__device__ __forceinline__ void gemm_quarter(const WQuarter& wq, const f32x16 (&X)[4], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = F2 ? mfma32(X[t][q * 4 + j], wq.w[t * 4 + q][j], acc) : mfma32(wq.w[t * 4 + q][j], X[t][q * 4 + j], acc);
}

// 16 floats of a plain row-major [128] row that belong to (quarter, half) in chain order
__device__ __forceinline__ f32x16 load_slice(const float* __restrict__ row, int quarter, int half) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(row + 32 * quarter + 8 * q + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[q * 4 + j] = x[j];
    }
    return v;
}

__device__ __forceinline__ void store_slice(float* __restrict__ row, int quarter, int half, const f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = v[q * 4 + j];
        *reinterpret_cast<f32x4*>(row + 32 * quarter + 8 * q + 4 * half) = x;
    }
}

// every wave contributes its quarter of each of NB 128-blocks of the rows; afterwards every wave holds the full rows
// in chain layout.  XLDW: padded row stride of the exchange buffer in floats (128 NB + 4).
template <int NB, int XLDW>
__device__ __forceinline__ void exchange_blocks(float* xbuf, int quarter, int slot, int half, const f32x16 (&mine)[NB],
                                                f32x16 (&X)[NB][4]) {
    __syncthreads();                                        // previous readers are done
#pragma unroll
    for (int b = 0; b < NB; ++b) store_slice(xbuf + slot * XLDW + 128 * b, quarter, half, mine[b]);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) load_row_chain(xbuf + slot * XLDW + 128 * b, half, X[b]);
}

constexpr int GAMD_XLD = 132;                  // exchange-buffer row stride of the 128-wide kernels

__device__ __forceinline__ void exchange(float* xbuf, int quarter, int slot, int half, const f32x16& mine, f32x16 (&X)[4]) {
    __syncthreads();
    store_slice(xbuf + slot * GAMD_XLD, quarter, half, mine);
    __syncthreads();
    load_row_chain(xbuf + slot * GAMD_XLD, half, X);
}

// The same for rows whose mean over the 128 features is zero BY CONSTRUCTION: the edge encoder's last Linear is packed
// with its output rows centred (W - mean_rows(W), b - mean(b): gamd_finalize_weights), which is LayerNorm's mean
// subtraction done once on the host instead of per edge (64 adds + 64 subtracts + a shuffle per 32 x 128 block).
// inv_width = 1 / (true width): rows narrower than 128 are zero-padded (their outputs are exact zeros and add nothing to the sum)
template <typename GPtr>
__device__ __forceinline__ void layernorm_chain_centered(f32x16 (&X)[4], GPtr gamma, GPtr beta, int half, float eps,
                                                         float inv_width = 1.0f / 128.0f) {
    // sum of squares per 32-feature block, then a fixed tree over the blocks: the order the small-system encoder
    // (one block per wave, edge_encode.hip) reproduces, so both give the same bits
    float vt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        vt[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[t] = fmaf(X[t][r], X[t][r], vt[t]);
    }
    const float var = gamd_xhalf_sum((vt[0] + vt[1]) + (vt[2] + vt[3])) * inv_width;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(&gamma[32 * t + 8 * q + 4 * half]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&beta[32 * t + 8 * q + 4 * half]);
#pragma unroll
            for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = fmaf(X[t][q * 4 + j] * rstd, g[j], b[j]);
        }
}

// XCD-aware persistent work split: workgroup b is observed to run on XCD b % 8 (speed only, never
// correctness).  Give each XCD one contiguous eighth of the tile range so that its private L2 sees a
// compact slice of the node tables.  Returns the first tile and the stride via out params; caller
// iterates  for (tile = first; tile < end; tile += step).
__device__ __forceinline__ void gamd_xcd_range(int n_tiles, int block, int n_blocks, int& first, int& end, int& step) {
    const int nx = 8;
    if (n_blocks % nx != 0 || n_blocks < nx) { first = block; end = n_tiles; step = n_blocks; return; }
    const int x = block % nx, w = block / nx, per = n_blocks / nx;
    const int q = n_tiles / nx, rem = n_tiles % nx;
    const int lo = x * q + (x < rem ? x : rem);
    const int cnt = q + (x < rem ? 1 : 0);
    first = lo + w; end = lo + cnt; step = per;
}

#define GAMD_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
