// conv_edge_bf16.hip — bf16-MFMA variant of the conv-layer edge kernel (BASELINE config 5).
//
// Same math and data flow as conv_edge.hip (nn_module.py:135-142), but the four 128x128 GEMMs run on
// v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (RNE), fp32 accumulate, everything else (bias, S/D
// add, SiLU, message, segment sum) in fp32.  At 1/16 of the fp32 matrix time the kernel is no longer
// MFMA-bound: all four bf16 weight matrices (4 x 32 KiB) stay resident in LDS, so there is no weight
// streaming and no barrier in the main loop; waves run free and hide each other's gather latency.
// Bound: L2 / HBM gather traffic (8 KiB of e + 3 x 16 KiB of S/D/hn rows per 32-edge tile).
#include "gamd_bf16.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

// row `row` of a [.][128] fp32 table -> chain layout, addressed as (scalar base) + (32-bit per-lane byte offset) + (immediate):
// one offset register per gather instead of a 64-bit per-lane pointer (the round-3 kernel spilled those)
__device__ __forceinline__ void load_row_chain_off(const float* __restrict__ base, unsigned row_off, f32x16 (&X)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (size_t)row_off + (32 * t + 8 * q) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
        }
}

#ifndef BF16_NW
#define BF16_NW 8
#endif
#ifndef BF16_RING
#define BF16_RING 8
#endif
#ifndef BF16_STAGGER
#define BF16_STAGGER 0
#endif
#ifndef BF16_FETCH_AFTER_GEMM4
#define BF16_FETCH_AFTER_GEMM4 1
#endif
#ifndef BF16_PREFETCH_E
#define BF16_PREFETCH_E 0
#endif
constexpr int CONVB_LDS_BYTES = 4 * GAMD_WFRAG_BF16_BYTES + 3 * 128 * 4;

// ABL (profiling build only, wrong results by construction): timing ablations selected with GAMD_BF16_VARIANT
//   1 SiLU -> x / 2   2 every gather from the zero row   4 no piece stores   8 no LDS weight fill   16 no MFMAs
//   32 the S / D (chain-layout) gathers alone from the zero row
template <int ABL>
__device__ __forceinline__ float silu_abl(float x) { return (ABL & 1) ? 0.5f * x : gamd_silu_hw(x); }

// SiLU of a 32 x 128 block + rounding to the bf16 operands of the next GEMM, two elements at a time: the three simple
// operations of x * rcp(1 + exp2(-x log2 e)) as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 on register pairs; the
// constants live in registers because packed instructions take no literals), the two transcendentals per element as they
// are.  Same IEEE operations per element as gamd_silu_hw, so the bits do not change.
struct SiluK { gamd_f32x2 nl2e, one; };
__device__ __forceinline__ SiluK silu_consts() {
    SiluK k{{-1.4426950408889634f, -1.4426950408889634f}, {1.0f, 1.0f}};
    asm volatile("" : "+v"(k.nl2e), "+v"(k.one));
    return k;
}
template <int ABL>
__device__ __forceinline__ void silu_pack_bf16(const f32x16 (&X)[4], bf16x8 (&P)[4][2], const SiluK& k) {
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
                    const gamd_f32x2 a = x * k.nl2e;
                    const gamd_f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    const gamd_f32x2 d = e + k.one;
                    const gamd_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    y = x * r;
                }
                w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, gamd_bf16x2));
            }
            P[t][u] = __builtin_bit_cast(bf16x8, w);
        }
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
    {
        f32x4* dst = reinterpret_cast<f32x4*>(ldsb);
        const f32x4* s1 = reinterpret_cast<const f32x4*>(a.w1p);
        const f32x4* s2 = reinterpret_cast<const f32x4*>(a.w2p);
        const f32x4* s3 = reinterpret_cast<const f32x4*>(a.w3p);
        const f32x4* s4 = reinterpret_cast<const f32x4*>(a.w4p);
        if (!(ABL & 8))
            for (int i = tid; i < 2048; i += 64 * BF16_NW) {
                dst[i] = s1[i]; dst[2048 + i] = s2[i]; dst[4096 + i] = s3[i]; dst[6144 + i] = s4[i];
            }
        if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    }
    __syncthreads();

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
        const int s0 = a.col[xc], d0 = a.erow[xc];
        src = valid ? s0 : a.zero_row; dst = valid ? d0 : a.zero_row;
        if (ABL & 2) { src = a.zero_row; dst = a.zero_row; }
    };
    const unsigned lane16 = 16u * (unsigned)lane, half16 = 16u * (unsigned)half;
    auto fetch_e = [&](int tile, bf16x8 (&P)[4][2]) {
        // (wave-uniform tile base in scalar registers) + (lane * 16) + (immediate)
        const char* tb = reinterpret_cast<const char*>(efrag) + (size_t)(tile < n_tiles ? tile : 0) * 8192;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                P[t][u] = __builtin_bit_cast(bf16x8, gamd_load_stream(reinterpret_cast<const f32x4*>(tb + (size_t)lane16 + (t * 2 + u) * 1024)));
    };
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
    fetch_idx(tile, src, dst);
    fetch_e(tile, P);

    for (int wt = first; wt < end; wt += step) {
        if (tile >= n_tiles) break;                  // this wave's tiles are exhausted (tile numbers grow with wt)
        asm volatile("" ::: "memory");               // keep loop-invariant LDS reads (bias, weights) inside the loop
        const int tile_n = tile_of(wt + step);
        int src_n, dst_n;
        fetch_idx(tile_n, src_n, dst_n);

        if (TIME) tprev = (long long)__builtin_readcyclecounter();
        f32x16 RA[4], RB[4], RC[4];
        // phase 1: T1 = SiLU(W1 e + b1).  The S[src] / D[dst] rows of phase 2 are gathered behind the GEMM (their round trip
        // rides under the SiLU block): issued at the top of the tile they hold 128 registers through the GEMM, which leaves no
        // room for the weight-fragment ring
        load_bias_chain(vb1, half, RC);
        BT(0);                                         // bias init (+ wait for e)
        if (!(ABL & 16)) gemm128_bf16_pf<false, BF16_RING>(W1, lane, P, RC);
        BT(1);                                         // GEMM 1
        // (ABL 32: only the S / D gathers from the zero row, the hn gather as it is: what the chain-layout gather alone costs)
        load_row_chain_off(a.S, ((unsigned)((ABL & 32) ? a.zero_row : src) << 9) + half16, RA);
        load_row_chain_off(a.D, ((unsigned)((ABL & 32) ? a.zero_row : dst) << 9) + half16, RB);
        BT(2);                                         // S / D gather issue
        silu_pack_bf16<ABL>(RC, P, sk);
        BT(3);                                         // SiLU 1 + pack
        if (BF16_PREFETCH_E) fetch_e(tile_n, Pn);     // next tile's e: three phases to land
        // phase 2: T3 = SiLU(W2 T1 + S[src] + D[dst])
#pragma unroll
        for (int t = 0; t < 4; ++t) RB[t] += RA[t];
        BT(4);                                         // S + D (waits for both gathers)
        if (!(ABL & 16)) gemm128_bf16_pf<false, BF16_RING>(W2, lane, P, RB);
        BT(5);                                         // GEMM 2
        silu_pack_bf16<ABL>(RB, P, sk);
        BT(6);                                         // SiLU 2 + pack
        // hn[src] rows for phase 4 (row layout: lane = feature, reg = edge): RA is free now
        const int x0 = tile * GAMD_TILE + 16 * half;
        int nvalid = E - x0;
        nvalid = nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid);
        // (W4's output rows are packed permuted, gamd_finalize_weights: lane = features 4 slot .. 4 slot + 3, one 16-byte load
        // per edge, landing in RA[r >> 2][4 (r & 3) + tp]; one bpermute index register + immediate lane offsets, scalar base +
        // 32-bit offset addressing: conv_edge.hip's gather_hn2)
        {
            const unsigned soff = (unsigned)src << 9, idx0 = 16u * (unsigned)half, slot16 = 16u * (unsigned)slot;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                unsigned o0, o1, o2, o3;
                asm volatile("ds_bpermute_b32 %0, %4, %5 offset:%6\n\tds_bpermute_b32 %1, %4, %5 offset:%7\n\t"
                             "ds_bpermute_b32 %2, %4, %5 offset:%8\n\tds_bpermute_b32 %3, %4, %5 offset:%9\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                             : "v"(idx0), "v"(soff), "n"(4 * (0 + 8 * r4)), "n"(4 * (1 + 8 * r4)), "n"(4 * (2 + 8 * r4)), "n"(4 * (3 + 8 * r4)));
                const unsigned o[4] = {o0, o1, o2, o3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 hv = *(const f32x4*)((const char*)a.hn + (o[k] + slot16));
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) RA[r4][k * 4 + tp] = hv[tp];
                }
            }
        }
        BT(7);                                         // hn gather issue (bpermutes + 16 loads)
        // phase 3: T4 = SiLU(W3 T3 + b3)
        load_bias_chain(vb3, half, RC);
        if (!(ABL & 16)) gemm128_bf16_pf<false, BF16_RING>(W3, lane, P, RC);
        BT(8);                                         // GEMM 3
        silu_pack_bf16<ABL>(RC, P, sk);
        BT(9);                                         // SiLU 3 + pack
        // phase 4: e_emb = T4 W4^T + b4 (F2), message, segment sum (fp32)
        const unsigned mask = a.chunk_mask[tile * 2 + half];
        int p = a.chunk_piece[tile * 2 + half];
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float b = vb4[32 * tp + slot];
#pragma unroll
            for (int r = 0; r < 16; ++r) RB[tp][r] = b;
        }
        BT(10);                                        // chunk metadata loads + b4 init
        if (!(ABL & 16)) gemm128_bf16_pf<true, BF16_RING>(W4, lane, P, RB);
        BT(11);                                        // GEMM 4
        if (!BF16_PREFETCH_E && BF16_FETCH_AFTER_GEMM4) fetch_e(tile_n, P);      // P is free: the next tile's e rides under the message / store block
        const unsigned keep_bits = ~(mask << 1);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                RB[tp][r] = gamd_msg_acc(RA[r >> 2][(r & 3) * 4 + tp], RB[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RB[tp][r - 1] : 0.f);
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
        default: break;
    }
#endif
    return launch_bf16_abl<0>(a, n_blocks, st);
}
