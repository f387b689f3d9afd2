// conv_edge_bf16.hip — bf16-MFMA variant of the conv-layer edge kernel (BASELINE config 5).
//
// Same math and data flow as conv_edge.hip (nn_module.py:135-142), but the four 128x128 GEMMs run on
// v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (RNE), fp32 accumulate, everything else (bias, S/D
// add, SiLU, message, segment sum) in fp32.  At 1/16 of the fp32 matrix time the kernel is no longer
// MFMA-bound: all four bf16 weight matrices (4 x 32 KiB) stay resident in LDS, so there is no weight
// streaming and no barrier in the main loop; waves run free and hide each other's gather latency.
//
// Round 6: the node tables of this mode are fp16 (NodeArgs::tab16, written by k_node: hn in natural order, S and D in the group
// order of gamd_tab16_pos and pre-multiplied by log2 e), and the pre-activations of the three SiLUs arrive multiplied by log2 e
// (W1, b1, b3 scaled at packing time, W4 by ln 2 to take it out again): y' = x' / (1 + 2^-x') needs no multiply in front of the
// exponential.  Half the gather instructions and cache lines, 96 vector registers less between GEMM 1 and GEMM 2 (which is what
// lets the next tile's e stream be fetched a whole tile ahead), S + D and the message as one v_fma_mix_f32 per element.
// Bound: the vector ALU (3 x 128 transcendental pairs per tile and wave) next to 8 KiB of e + 3 x 8 KiB of S/D/hn rows per tile.
#include "gamd_bf16.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

// fp16 tables: the 64 features of row `row_off / 256` that lane (slot, half) owns = eight 16-byte groups, 32 bytes apart
__device__ __forceinline__ void load_row_tab16(const float* __restrict__ base, unsigned row_off, gamd_u32x4 (&X)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
        X[c] = *reinterpret_cast<const gamd_u32x4*>(reinterpret_cast<const char*>(base) + (size_t)row_off + 32 * c);
}
// d = (float)a.h[HI_A] * 1.0 + (float)c.h[HI_C]  /  (float)a.h[HI_A] * b + c : one v_fma_mix_f32 each (fp16 sources converted
// inside the instruction, fp32 arithmetic, one rounding)
// (hi_a / hi_c / hi: which half of the dword; constants after unrolling)
__device__ __forceinline__ float add_h_h(unsigned a, unsigned c, bool HI_A, bool HI_C) {
    float d;
    if (HI_A && HI_C) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    else if (HI_A)    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    else if (HI_C)    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    else              asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_h_f_f(unsigned a, float b, float c, bool HI_A) {
    float d;
    if (HI_A) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

#ifndef BF16_NW
#define BF16_NW 8
#endif
#ifndef BF16_RING
#define BF16_RING 8
#endif
#ifndef BF16_FRING
#define BF16_FRING 4         // weight-fragment ring depth of the fused GEMM + SiLU form (7 vector instructions between two MFMAs)
#endif
#ifndef BF16_STAGGER
#define BF16_STAGGER 0
#endif
#ifndef BF16_FETCH_AFTER_GEMM4
#define BF16_FETCH_AFTER_GEMM4 1
#endif
#ifndef BF16_PREFETCH_E
#define BF16_PREFETCH_E 1
#endif
constexpr int CONVB_LDS_BYTES = 4 * GAMD_WFRAG_BF16_BYTES + 3 * 128 * 4;

// ABL (profiling build only, wrong results by construction): timing ablations selected with GAMD_BF16_VARIANT
//   1 SiLU -> x / 2   2 every gather from the zero row   4 no piece stores   8 no LDS weight fill   16 no MFMAs
//   32 the S / D (chain-layout) gathers alone from the zero row
//   128 GEMMs without their LDS weight reads (one fragment quad, read once)   256 GEMMs without MFMAs (operands and weight
//   reads stay live: an empty asm statement per MFMA consumes them and "writes" the accumulator)
//   1024 s_setprio 1 around every GEMM (its MFMAs win the issue arbitration against the partner wave's vector instructions)
//   2048 WITHOUT the s_setprio 1 around every SiLU block   4096 static s_setprio 1 for waves 4-7   8192 priority 1 everywhere
//   but in the GEMMs (results unchanged by all of them)
template <int ABL, bool F2>
__device__ __forceinline__ void gemm_abl(const bf16x8* W, int lane, const bf16x8 (&P)[4][2], f32x16 (&acc)[4]) {
    if (ABL & 16) return;
    if (!(ABL & (128 | 256))) {
        if (ABL & 1024) __builtin_amdgcn_s_setprio(1);
        if (ABL & 8192) __builtin_amdgcn_s_setprio(0);            // 8192: priority 1 everywhere BUT in the GEMMs
        gemm128_bf16_pf<F2, BF16_RING>(W, lane, P, acc);
        if (ABL & 1024) __builtin_amdgcn_s_setprio(0);
        if (ABL & 8192) __builtin_amdgcn_s_setprio(1);
        return;
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 w0 = W[lane];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int tp = i & 3, t = i >> 3, u = (i >> 2) & 1;
        const bf16x8 cur = (ABL & 128) ? w0 : W[((tp * 4 + t) * 2 + u) * 64 + lane];
        if (ABL & 256) asm volatile("" : "+v"(acc[tp]) : "v"(cur), "v"(P[t][u]));
        else acc[tp] = F2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[t][u], cur, acc[tp], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, P[t][u], acc[tp], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int ABL>
__device__ __forceinline__ float silu_abl(float x) { return (ABL & 1) ? 0.5f * x : gamd_silu_hw(x); }

// (Round 6: the block runs at s_setprio 1.  The two waves of a SIMD share its vector issue; a wave inside a SiLU block — 128
// transcendental pairs, the densest vector stretch of a tile — that keeps losing issue slots to its partner's scattered
// vector instructions finishes later AND delays the partner's next block.  Measured same-box: 42.3 -> 39.8 us per launch at
// C5, 113.1 -> 108.6 at 10 000 LJ atoms; priority around the GEMMs, for the younger half, or everywhere but the GEMMs: +-0.)
// SiLU of a 32 x 128 block + rounding to the bf16 operands of the next GEMM, two elements at a time.  The block arrives
// multiplied by log2 e (x' = x log2 e: the scale sits in the weights and tables that feed it), so
//     y' = x' * rcp(1 + exp2(-x')) = log2 e * SiLU(x)
// costs two transcendentals (the negation is a source modifier), one packed add, one packed multiply and the packed conversion
// per pair; the next GEMM's weights carry what takes the log2 e out again (W2 and W3 unchanged: their outputs are wanted
// times log2 e; W4 times ln 2).  The constant lives in a register pair because packed instructions take no literals.
struct SiluK { gamd_f32x2 one; };
__device__ __forceinline__ SiluK silu_consts() {
    SiluK k{{1.0f, 1.0f}};
    asm volatile("" : "+v"(k.one));
    return k;
}
template <int ABL>
__device__ __forceinline__ void silu_pack_bf16(const f32x16 (&X)[4], bf16x8 (&P)[4][2], const SiluK& k) {
    if (!(ABL & 2048)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            gamd_u32x4 w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gamd_f32x2 x = {X[t][8 * u + 2 * q], X[t][8 * u + 2 * q + 1]};
                gamd_f32x2 y;
                if (ABL & 1) {
                    y = x * gamd_f32x2{0.5f, 0.5f};
                } else {
                    const gamd_f32x2 e = {__builtin_amdgcn_exp2f(-x[0]), __builtin_amdgcn_exp2f(-x[1])};
                    const gamd_f32x2 d = e + k.one;
                    const gamd_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    y = x * r;
                }
                w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, gamd_bf16x2));
            }
            P[t][u] = __builtin_bit_cast(bf16x8, w);
        }
    if (!(ABL & 2048)) __builtin_amdgcn_s_setprio(0);
}

// SiLU of one pair of accumulator elements -> one dword of bf16 operands (the element-wise body of silu_pack_bf16)
__device__ __forceinline__ unsigned silu_pair_bf16(float x0, float x1, const SiluK& k) {
    const gamd_f32x2 x = {x0, x1};
    const gamd_f32x2 e = {__builtin_amdgcn_exp2f(-x[0]), __builtin_amdgcn_exp2f(-x[1])};
    const gamd_f32x2 d = e + k.one;
    const gamd_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const gamd_f32x2 y = x * r;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(y, gamd_bf16x2));
}

// ABL 32768 (variant): the SiLU of the PREVIOUS GEMM's output fused into this GEMM, inside the wave.  The GEMM walks K in four
// stages of 8 MFMAs (K block t = the previous GEMM's output block t); only the first quarter of the SiLU block has to be done
// before the first MFMA, quarter t + 1 runs between the MFMAs of stage t (one pair of elements = 7 vector instructions per MFMA of
// 32 matrix cycles).  Same operations on the same values in the same per-accumulator order: bit-identical results.
template <bool F2, int D, int PRIO>
__device__ __forceinline__ void gemm128_bf16_fused(const bf16x8* W, int lane, const f32x16 (&X)[4], bf16x8 (&P)[4][2], f32x16 (&acc)[4],
                                                   const SiluK& k) {
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    {
        gamd_u32x4 w0, w1;
#pragma unroll
        for (int q = 0; q < 4; ++q) { w0[q] = silu_pair_bf16(X[0][2 * q], X[0][2 * q + 1], k); w1[q] = silu_pair_bf16(X[0][8 + 2 * q], X[0][8 + 2 * q + 1], k); }
        P[0][0] = __builtin_bit_cast(bf16x8, w0); P[0][1] = __builtin_bit_cast(bf16x8, w1);
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) w[i] = W[((((i & 3) * 4 + (i >> 3)) * 2) + ((i >> 2) & 1)) * 64 + lane];
    gamd_u32x4 nw[2];
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
        if (t < 3) {                                       // pair (i & 7) of quarter t + 1
            const int pr = i & 7, uu = pr >> 2, q = pr & 3;
            nw[uu][q] = silu_pair_bf16(X[t + 1][8 * uu + 2 * q], X[t + 1][8 * uu + 2 * q + 1], k);
            if (pr == 7) { P[t + 1][0] = __builtin_bit_cast(bf16x8, nw[0]); P[t + 1][1] = __builtin_bit_cast(bf16x8, nw[1]); }
        }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + D < 32) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < 24) { __builtin_amdgcn_sched_group_barrier(0x400, 4, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
    }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
}

template <int ABL>
__global__ void __launch_bounds__(64 * BF16_NW, BF16_NW == 8 ? 2 : 1) k_conv_edge_bf16(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const bf16x8* W1 = reinterpret_cast<const bf16x8*>(ldsb);
    const bf16x8* W2 = W1 + 2048;
    // W3 / W4 start at byte 65 536 / 98 304 of the LDS image: beyond the 16-bit offset field of ds_read_b128.  With constant
    // addresses hipcc materialises one address register per 1 KiB fragment; an opaque base keeps it to one register per matrix
    // and immediate offsets (the same finding as in wide.hip's k_conv_edge_wide).
    unsigned w3_off = 2 * GAMD_WFRAG_BF16_BYTES, w4_off = 3 * GAMD_WFRAG_BF16_BYTES;
    asm volatile("" : "+s"(w3_off), "+s"(w4_off));
    const bf16x8* W3 = reinterpret_cast<const bf16x8*>(ldsb + w3_off);
    const bf16x8* W4 = reinterpret_cast<const bf16x8*>(ldsb + w4_off);
    float* vb1 = reinterpret_cast<float*>(ldsb + 4 * GAMD_WFRAG_BF16_BYTES);
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_wg_tiles = (n_tiles + BF16_NW - 1) / BF16_NW;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    const bf16x8* efrag = reinterpret_cast<const bf16x8*>(a.e_frag);

    // Software pipeline over this wave's tiles: the indices of tile i + 1 are fetched at the top of tile i and its e fragments
    // (8 KiB streamed from HBM) after tile i's phase 1, into the registers that phase's accumulators leave free — a tile no
    // longer starts with two dependent memory round trips (indices -> rows / e) in front of its first MFMA.  Round 3's kernel
    // did, and spent more than half of its time waiting there (47 us per launch against ~16 us of VALU work).
    auto tile_of = [&](int wt) { const int t = wt * BF16_NW + wave; return (wt < end && t < n_tiles) ? t : n_tiles; };
    auto fetch_idx = [&](int tile, int& src, int& dst) {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = tile < n_tiles && x < E;
        // padding slots of the last tile gather the all-zero row n of hn / S / D: their messages are exact zeros
        const int xc = valid ? x : 0;
        const int s0 = GAMD_CHK_RANGE(a.sticky, a.col[xc], 0, a.zero_row, GAMD_CHK_CONV_SRC), d0 = GAMD_CHK_RANGE(a.sticky, a.erow[xc], 0, a.zero_row, GAMD_CHK_CONV_DST);
        src = valid ? s0 : a.zero_row; dst = valid ? d0 : a.zero_row;
        if (ABL & 2) { src = a.zero_row; dst = a.zero_row; }
    };
    const unsigned lane16 = 16u * (unsigned)lane, half16 = 16u * (unsigned)half;
    auto fetch_e = [&](int tile, bf16x8 (&P)[4][2]) {
        // (wave-uniform tile base in scalar registers) + (lane * 16) + (immediate)
        const char* tb = reinterpret_cast<const char*>(efrag) + (size_t)((tile < n_tiles && !(ABL & 512)) ? tile : 0) * 8192;   // ABL 512: every tile reads e tile 0 (no HBM stream)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                P[t][u] = __builtin_bit_cast(bf16x8, gamd_load_stream(reinterpret_cast<const f32x4*>(tb + (size_t)lane16 + (t * 2 + u) * 1024)));
    };
    if ((ABL & 4096) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (ABL & 8192) __builtin_amdgcn_s_setprio(1);
#if BF16_STAGGER > 0
    // de-phase the two waves of a SIMD (waves w and w + 4): their MFMA phases and their VALU phases then interleave instead
    // of colliding on the matrix pipe / the vector ALU
    if (wave >= 4) __builtin_amdgcn_s_sleep(BF16_STAGGER);
#endif
    // profiling build, ABL bit 64: s_memtime between the segments of a tile, summed per wave -> a.tdbg[block][wave][16]
    constexpr bool TIME = (ABL & 64) != 0;
    long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
#define BT(I) do { if (TIME) { __builtin_amdgcn_sched_barrier(0); const long long now__ = (long long)__builtin_readcyclecounter(); \
                               tacc[I] += now__ - tprev; tprev = now__; __builtin_amdgcn_sched_barrier(0); } } while (0)
    const SiluK sk = silu_consts();
    int tile = tile_of(first);
    int src, dst;
    bf16x8 P[4][2], Pn[4][2];
    // Prologue: the first tile's indices and e fragments (HBM) are requested FIRST, then the four weight matrices go L2 -> LDS as
    // 128 one-KiB LDS-DMA copies (16 per wave, all in flight at once, no register round trip), then ONE wait and ONE barrier.
    // (Rounds 3-5 copied through registers in four dependent round trips of 4 loads + 4 ds_write_b128 per thread and only then
    // asked for the first indices: three more serial round trips before the first MFMA of every workgroup.)
    fetch_idx(tile, src, dst);
    fetch_e(tile, P);
    if (!(ABL & 8)) {
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
#pragma unroll
        for (int k = 0; k < 128 / BF16_NW; ++k) {
            const int chunk = k * BF16_NW + wave;               // 128 chunks of 1 KiB: matrix chunk >> 5, KiB chunk & 31 of it
            const float* gw = (chunk >> 5) == 0 ? a.w1p : (chunk >> 5) == 1 ? a.w2p : (chunk >> 5) == 2 ? a.w3p : a.w4p;
            const char* base = reinterpret_cast<const char*>(gw) + (chunk & 31) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + l16),
                                             (__attribute__((address_space(3))) void*)(ldsb + chunk * 1024), 16, 0, 0);
        }
    }
    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    __syncthreads();                                             // (its release fence waits for the DMA copies: vmcnt(0))

    for (int wt = first; wt < end; wt += step) {
        if (tile >= n_tiles) break;                  // this wave's tiles are exhausted (tile numbers grow with wt)
        asm volatile("" ::: "memory");               // keep loop-invariant LDS reads (bias, weights) inside the loop
        const int tile_n = tile_of(wt + step);
        int src_n, dst_n;
        fetch_idx(tile_n, src_n, dst_n);

        if (TIME) tprev = (long long)__builtin_readcyclecounter();
        f32x16 RB[4], RC[4];
        gamd_u32x4 S16[8], D16[8];                     // fp16 S[src] / D[dst] rows: the 64 features this lane owns, 8 groups of 8
        gamd_u32x2 H16[4][4];                          // fp16 hn[src] of the 16 edges of this half: features 4 slot .. 4 slot + 3
        // phase 1: T1 = SiLU(W1 e + b1).  The S[src] / D[dst] rows of phase 2 are gathered behind the GEMM (their round trip
        // rides under the SiLU block)
        load_bias_chain(vb1, half, RC);
        BT(0);                                         // bias init (+ wait for e)
        gemm_abl<ABL, false>(W1, lane, P, RC);
        BT(1);                                         // GEMM 1
        // (ABL 32: only the S / D gathers from the zero row, the hn gather as it is: what the chain-layout gather alone costs)
        load_row_tab16(a.S, ((unsigned)((ABL & 32) ? a.zero_row : src) << 8) + half16, S16);
        load_row_tab16(a.D, ((unsigned)((ABL & 32) ? a.zero_row : dst) << 8) + half16, D16);
        BT(2);                                         // S / D gather issue
        constexpr bool FUSE = (ABL & 32768) != 0;    // SiLU blocks fused into the GEMM that consumes them (gemm128_bf16_fused)
        constexpr int FPRIO = (ABL & 131072) ? 2 : (ABL & 2048) ? 0 : 1;      // 131072: the whole fused GEMM at priority 1
        if (!FUSE) silu_pack_bf16<ABL>(RC, P, sk);
        BT(3);                                         // SiLU 1 + pack
        if (BF16_PREFETCH_E) fetch_e(tile_n, Pn);     // next tile's e: three phases to land
        // phase 2: T3 = SiLU(W2 T1 + S[src] + D[dst]); group c = 2 t + k holds X[t][8 k .. 8 k + 7], two fp16 per dword
        if (ABL & 16384) __builtin_amdgcn_s_setprio(1);           // ABL 16384: the S + D block at priority too
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                RB[c >> 1][8 * (c & 1) + 2 * i] = add_h_h(S16[c][i], D16[c][i], false, false);
                RB[c >> 1][8 * (c & 1) + 2 * i + 1] = add_h_h(S16[c][i], D16[c][i], true, true);
            }
        if (ABL & 16384) { asm volatile("" : "+v"(RB[0]), "+v"(RB[3])); __builtin_amdgcn_s_setprio(0); }
        BT(4);                                         // S + D (waits for both gathers)
        if (FUSE) gemm128_bf16_fused<false, BF16_FRING, FPRIO>(W2, lane, RC, P, RB, sk); else gemm_abl<ABL, false>(W2, lane, P, RB);
        BT(5);                                         // GEMM 2
        if (!FUSE) silu_pack_bf16<ABL>(RB, P, sk);
        BT(6);                                         // SiLU 2 + pack
        // hn[src] rows for phase 4 (row layout: lane = feature, reg = edge)
        const int x0 = tile * GAMD_TILE + 16 * half;
        int nvalid = E - x0;
        nvalid = nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid);
        // (W4's output rows are packed permuted, gamd_finalize_weights: lane = features 4 slot .. 4 slot + 3, one 8-byte load
        // per edge, H16[r >> 2][r & 3]; one bpermute index register + immediate lane offsets, scalar base + 32-bit offset
        // addressing: conv_edge.hip's gather_hn2)
        {
            const unsigned soff = (unsigned)src << 8, idx0 = 16u * (unsigned)half, slot8 = 8u * (unsigned)slot;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                unsigned o0, o1, o2, o3;
                asm volatile("ds_bpermute_b32 %0, %4, %5 offset:%6\n\tds_bpermute_b32 %1, %4, %5 offset:%7\n\t"
                             "ds_bpermute_b32 %2, %4, %5 offset:%8\n\tds_bpermute_b32 %3, %4, %5 offset:%9\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                             : "v"(idx0), "v"(soff), "n"(4 * (0 + 8 * r4)), "n"(4 * (1 + 8 * r4)), "n"(4 * (2 + 8 * r4)), "n"(4 * (3 + 8 * r4)));
                const unsigned o[4] = {o0, o1, o2, o3};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    H16[r4][k] = *(const gamd_u32x2*)((const char*)a.hn + (o[k] + slot8));
            }
        }
        BT(7);                                         // hn gather issue (bpermutes + 16 loads)
        // phase 3: T4 = SiLU(W3 T3 + b3)
        load_bias_chain(vb3, half, RC);
        if (FUSE) gemm128_bf16_fused<false, BF16_FRING, FPRIO>(W3, lane, RB, P, RC, sk); else gemm_abl<ABL, false>(W3, lane, P, RC);
        BT(8);                                         // GEMM 3
        if (!FUSE) silu_pack_bf16<ABL>(RC, P, sk);
        BT(9);                                         // SiLU 3 + pack
        // phase 4: e_emb = T4 W4^T + b4 (F2), message, segment sum (fp32)
        const unsigned mask = a.chunk_mask[tile * 2 + half];
        int p = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float b = vb4[32 * tp + slot];
#pragma unroll
            for (int r = 0; r < 16; ++r) RB[tp][r] = b;
        }
        BT(10);                                        // chunk metadata loads + b4 init
        if (FUSE) gemm128_bf16_fused<true, BF16_FRING, FPRIO>(W4, lane, RC, P, RB, sk); else gemm_abl<ABL, true>(W4, lane, P, RB);
        BT(11);                                        // GEMM 4
        if (!BF16_PREFETCH_E && BF16_FETCH_AFTER_GEMM4) fetch_e(tile_n, P);      // P is free: the next tile's e rides under the message / store block
        const unsigned keep_bits = ~(mask << 1);
        if (ABL & 65536) __builtin_amdgcn_s_setprio(1);           // ABL 65536: message + segment sum at priority too
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                RB[tp][r] = fma_h_f_f(H16[r >> 2][r & 3][tp >> 1], RB[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RB[tp][r - 1] : 0.f, (tp & 1) != 0);
        if (ABL & 65536) { asm volatile("" : "+v"(RB[0]), "+v"(RB[3])); __builtin_amdgcn_s_setprio(0); }
        BT(12);                                        // message + segment sum (waits for hn)
        unsigned ends = mask;
        if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends |= 1u << (nvalid - 1);
        // the prefetched registers are waited for HERE, in front of the piece stores (they landed two GEMMs ago): with loads and
        // stores both outstanding hipcc can only wait with vmcnt(0), i.e. a wait placed at the top of the next tile would sit
        // out the write latency of this tile's pieces
        if (BF16_PREFETCH_E) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) asm volatile("" ::"v"(Pn[t][u]));
        }
        asm volatile("" ::"v"(src_n), "v"(dst_n));
        // One store per finished piece.  `ends` is uniform within each half-wave, so the closing edge of the next piece is a
        // SCALAR per half (v_readlane) and the running sum is picked with a register-indexed move (RB[tp][r] with a wave-uniform
        // r: s_set_gpr_idx / v_movrels) instead of a 15-deep select chain per output block: ~15 instead of ~65 VALU per piece
        unsigned e0 = __builtin_amdgcn_readlane(ends, 0), e1 = __builtin_amdgcn_readlane(ends, 32);
        while (e0 | e1) {
            const int r0 = e0 ? __builtin_ctz(e0) : 0, r1 = e1 ? __builtin_ctz(e1) : 0;
            f32x4 pv;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const float v0 = RB[tp][r0], v1 = RB[tp][r1];
                pv[tp] = half ? v1 : v0;
            }
            if (half ? (e1 != 0) : (e0 != 0)) {
                if (!(ABL & 4) || pv[0] == 123.456f) *(f32x4*)(a.partial + (size_t)p * GAMD_H + 4 * slot) = pv;
                ++p;
            }
            e0 &= e0 - 1; e1 &= e1 - 1;
        }
        BT(13);                                        // piece stores
        if (TIME) tacc[15] += 1;
        tile = tile_n; src = src_n; dst = dst_n;
        if (BF16_PREFETCH_E) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) P[t][u] = Pn[t][u];
        } else if (!BF16_FETCH_AFTER_GEMM4) {
            fetch_e(tile, P);
        }
    }
#ifdef GAMD_PROFILING
    if (TIME && a.tdbg && lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 16; ++i) a.tdbg[((size_t)blockIdx.x * 8 + wave) * 16 + i] = tacc[i];
#endif
#undef BT
}

}  // namespace

template <int ABL>
static int launch_bf16_abl(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)CONVB_LDS_BYTES, k_conv_edge_bf16<ABL>)) return e;
    hipLaunchKernelGGL(k_conv_edge_bf16<ABL>, dim3(n_blocks), dim3(64 * BF16_NW), CONVB_LDS_BYTES, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_conv_edge_bf16(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
#ifdef GAMD_PROFILING
    static int v = -1;
    if (v < 0) { const char* e = getenv("GAMD_BF16_VARIANT"); v = e ? atoi(e) : 0; }
    switch (v) {
        case 1: return launch_bf16_abl<1>(a, n_blocks, st);
        case 2: return launch_bf16_abl<2>(a, n_blocks, st);
        case 4: return launch_bf16_abl<4>(a, n_blocks, st);
        case 8: return launch_bf16_abl<8>(a, n_blocks, st);
        case 16: return launch_bf16_abl<16>(a, n_blocks, st);
        case 3: return launch_bf16_abl<3>(a, n_blocks, st);
        case 7: return launch_bf16_abl<7>(a, n_blocks, st);
        case 23: return launch_bf16_abl<23>(a, n_blocks, st);
        case 31: return launch_bf16_abl<31>(a, n_blocks, st);
        case 32: return launch_bf16_abl<32>(a, n_blocks, st);
        case 64: return launch_bf16_abl<64>(a, n_blocks, st);
        case 128: return launch_bf16_abl<128>(a, n_blocks, st);
        case 256: return launch_bf16_abl<256>(a, n_blocks, st);
        case 257: return launch_bf16_abl<257>(a, n_blocks, st);
        case 263: return launch_bf16_abl<263>(a, n_blocks, st);
        case 9: return launch_bf16_abl<9>(a, n_blocks, st);
        case 512: return launch_bf16_abl<512>(a, n_blocks, st);
        case 391: return launch_bf16_abl<391>(a, n_blocks, st);
        case 775: return launch_bf16_abl<775>(a, n_blocks, st);
        case 903: return launch_bf16_abl<903>(a, n_blocks, st);
        case 384: return launch_bf16_abl<384>(a, n_blocks, st);
        case 1024: return launch_bf16_abl<1024>(a, n_blocks, st);
        case 2048: return launch_bf16_abl<2048>(a, n_blocks, st);
        case 4096: return launch_bf16_abl<4096>(a, n_blocks, st);
        case 5120: return launch_bf16_abl<5120>(a, n_blocks, st);
        case 8192: return launch_bf16_abl<8192>(a, n_blocks, st);
        case 10240: return launch_bf16_abl<10240>(a, n_blocks, st);
        case 16384: return launch_bf16_abl<16384>(a, n_blocks, st);
        case 65536: return launch_bf16_abl<65536>(a, n_blocks, st);
        case 81920: return launch_bf16_abl<81920>(a, n_blocks, st);
        case 32768: return launch_bf16_abl<32768>(a, n_blocks, st);      // SiLU fused into the consuming GEMM, first quarter at priority
        case 34816: return launch_bf16_abl<34816>(a, n_blocks, st);      // ... no priority anywhere
        case 163840: return launch_bf16_abl<163840>(a, n_blocks, st);    // ... the whole fused GEMM at priority
        default: break;
    }
#endif
    return launch_bf16_abl<0>(a, n_blocks, st);
}
