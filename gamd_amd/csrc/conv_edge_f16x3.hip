// conv_edge_f16x3.hip — the conv-layer edge kernel (see conv_edge.hip for the data flow) with its four 128x128 GEMMs on the
// fp16 matrix pipe via operand splitting (gamd_f16x3.h): fp32-grade results at 3/16 of the fp32 matrix time.  Persistent
// 512-thread workgroups (two waves per SIMD, 256 registers each), one 32-edge tile per wave, 2-slot 64 KiB weight ring (the
// [hi | lo] fp16 image of a matrix is as large as its fp32 image) refilled by LDS DMA one phase ahead, one barrier per GEMM
// phase, partial-sum pieces as in the fp32 kernel.  A GEMM's input is an operand set (hi, lo fp16 images, 64 registers like
// the fp32 block it stands for); the post-op of a phase (SiLU, split) turns the accumulators into the next phase's operand
// set.  e arrives pre-split from the encoder.
//
// Round 4 rewrite.  Rounds 1-3 ran ONE wave per SIMD with six 64-register sets alive (two operand sets, S, D, accumulators,
// hn) in the AGPR half of the file: 800 of ~2 700 VALU instructions per tile were v_accvgpr moves, and nothing hid a wave's
// own VALU / memory-issue time from the matrix pipe (36.5 k cycles per tile for 12.3 k of MFMAs; 0.33-0.36 ms per launch at
// C2).  Now a GEMM's accumulators are initialised lazily (output block tp right in front of its 24 MFMAs) and become the next
// GEMM's operands block by block, so a phase holds its input set (64), the accumulator / output blocks (<= 80) and one
// gathered set: S[src] / D[dst] arrive as 16-byte quads of output block tp while it accumulates, hn[src] at the phase 3 / 4
// boundary, the next tile's e behind GEMM 4.  SiLU and the hi / lo split run on register pairs (v_pk_mul_f32 / v_pk_add_f32);
// the message is the fp32 kernels' fused multiply-add on zero-row padding.  253 VGPRs, no AGPRs, no scratch, ~1 800 VALU
// per tile; 0.26 ms per launch (0.36 for the old kernel on the same box).  What the register allocator needed to get there is
// noted where it matters: register sets scoped to the active branch, fresh operand quads, opaque LDS / global bases.
#include "gamd_f16x3.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

constexpr int CONV_LDS_FLOATS = 2 * GAMD_WFRAG_FLOATS + 3 * 128;

template <bool TIME>
__global__ void __launch_bounds__(512, 2) k_conv_edge_f16x3(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + GAMD_WFRAG_FLOATS;
    // the bias vectors sit behind the two 64 KiB slots: with a constant address every bias read gets its own address register
    // (byte offsets >= 131 072 do not fit the 16-bit offset field of ds_read) -- one opaque base + immediates instead
    unsigned boff = (unsigned)(2 * GAMD_WFRAG_FLOATS * sizeof(float));
    asm volatile("" : "+s"(boff));
    float* vb1 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + boff);
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_units = (n_tiles + NW - 1) / NW;              // work unit = 8 tiles, one per wave
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;
    const int n_iter = (end - first + step - 1) / step;
    auto tile_of = [&](int it) {
        const int u = first + it * step;
        return (it < n_iter && u * NW + wave < n_tiles) ? u * NW + wave : n_tiles;
    };

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    gamd_stage_weight_raw_contig<NW>(a.w1p, buf0, wave, lane16);

    OpSet PA;                         // e of the current tile (the only register set that crosses the tile loop)
    SiluK2 sk;
    sk.nl2e = gamd_f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
    sk.one = gamd_f32x2_t{1.0f, 1.0f};
    asm volatile("" : "+v"(sk.nl2e), "+v"(sk.one));          // register pairs for the packed instructions (no literals there)

    int tile = tile_of(0);
    bool active = tile < n_tiles;
    // padding slots of the last tile gather the all-zero row n of hn / S / D: their messages are exact zeros without a mask
    int src = a.zero_row, dst = a.zero_row;
    {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        if (active && x < E) { src = GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst = GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST); }
        if (active) load_e_tile_s(a.e_frag, tile, lane16, PA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the W1 copy is not tracked by hipcc
    __syncthreads();
    long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
#define FT(I) do { if (TIME) { __builtin_amdgcn_sched_barrier(0); const long long now__ = (long long)__builtin_readcyclecounter(); \
                               tacc[I] += now__ - tprev; tprev = now__; __builtin_amdgcn_sched_barrier(0); } } while (0)

    for (int it = 0; it < n_iter; ++it) {
        const int tile_n = tile_of(it + 1);
        const bool active_n = tile_n < n_tiles;
        int src_n = a.zero_row, dst_n = a.zero_row;
        if (TIME) tprev = (long long)__builtin_readcyclecounter();
        if (!active) {
            // a wave without a tile in this unit (the tail of the grid) still copies its share of the four matrices and meets
            // the barriers; its register sets are not touched (everything but PA lives inside the other branch: a set written
            // under a condition would be carried around the loop through this path and count against the 256 registers)
            gamd_stage_weight_raw_contig<NW>(a.w2p, buf1, wave, lane16);
            phase_barrier<0>();
            gamd_stage_weight_raw_contig<NW>(a.w3p, buf0, wave, lane16);
            phase_barrier<0>();
            gamd_stage_weight_raw_contig<NW>(a.w4p, buf1, wave, lane16);
            phase_barrier<0>();
            gamd_stage_weight_raw_contig<NW>(a.w1p, buf0, wave, lane16);
            phase_barrier<0>();
        } else {
            const int x0 = tile * GAMD_TILE + 16 * half;
            int nvalid = E - x0;
            nvalid = nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid);
            OpSet PB, PC;                     // T1 / T4; T3
            f32x16 ACC[4], RC[4];             // accumulators of phases 1, 3 / 2, 4 (piece sums)
            f32x4 HN[16];
            // gathers: scalar base + 32-bit lane offset + immediate (no 64-bit per-lane pointers)
            const unsigned soff = ((unsigned)src << 9) + 16u * (unsigned)half, doff = ((unsigned)dst << 9) + 16u * (unsigned)half;
            const unsigned mask = a.chunk_mask[tile * 2 + half];
            const int p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE);
            // ===== phase 1: T1 = SiLU(W1 e + b1) =====
            gemm128_f16x3_lazy<false>((const f16x8*)buf0, lane, PA, ACC, [&](int tp) { ACC[tp] = bias_block(vb1, tp, half); },
                [&](int tp, int r0) { silu_split_pair(PB, tp, r0, ACC[tp][r0], ACC[tp][r0 + 1], sk); },
                [&](int i) { if (i < 8) stage_chunk<NW>(a.w2p, buf1, wave, lane16, i); });
            FT(0);
            phase_barrier<0>();
            FT(1);
            // ===== phase 2: T3 = SiLU((W2 T1 + D[dst]) + S[src]): the rows arrive as 16-byte quads of output block tp while
            // block tp accumulates and are read by its post-op, which rides in the next block's K loop =====
            {
                f32x4 SQ[4][4], DQ[4][4];     // each quad is written once (a value, not a buffer): live from its load to its use
                gemm128_f16x3_lazy<false>((const f16x8*)buf1, lane, PB, RC,
                    [&](int tp) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) RC[tp][r] = 0.f;
                    },
                    [&](int tp, int r0) {
                        const int q = r0 >> 2, j = r0 & 3;
                        silu_split_pair(PC, tp, r0, (RC[tp][r0] + DQ[tp][q][j]) + SQ[tp][q][j],
                                        (RC[tp][r0 + 1] + DQ[tp][q][j + 1]) + SQ[tp][q][j + 1], sk);
                    },
                    [&](int i) {
                        const int tp = i >> 3, k = i & 7;
                        if (i < 8) stage_chunk<NW>(a.w3p, buf0, wave, lane16, i);
                        // second half of the K loop (the quads of block tp - 1 are being used up meanwhile: fewer of both alive);
                        // the last block's post-op follows its K loop directly, so its rows are fetched in the first half
                        const int q = tp == 3 ? k : k - 4;
                        if (q >= 0 && q < 4) {
                            SQ[tp][q] = *(const f32x4*)((const char*)a.S + (soff + (unsigned)(128 * tp + 32 * q)));
                            DQ[tp][q] = *(const f32x4*)((const char*)a.D + (doff + (unsigned)(128 * tp + 32 * q)));
                        }
                    });
            }
            FT(2);
            phase_barrier<0>();
            FT(3);
            // ===== phase 3: T4 = SiLU(W3 T3 + b3); hn[src] rows for phase 4 =====
            if (active_n) {
                const int xn = tile_n * GAMD_TILE + gamd_pi(slot);
                if (xn < E) { src_n = GAMD_CHK_RANGE(a.sticky, a.col[xn], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst_n = GAMD_CHK_RANGE(a.sticky, a.erow[xn], 0, a.zero_row, GAMD_CHK_CONV_DST); }
            }
            // hn[src] rows of edges 4 r4 .. 4 r4 + 3 of this half (row layout: lane = feature, register = edge; rows are stored
            // permuted, node.hip hn_perm: one 16-byte load per edge): the row offset of edge r lives in lane rho(r, half) -- one
            // bpermute index register + immediate lane offsets.  Edges 0-7 are fetched at the end of phase 3, edges 8-15 (first
            // used in the second K loop of phase 4) at the start of phase 4: 32 registers less across phase 3.
            auto gather_hn = [&](int r4) {
                const unsigned rowoff = (unsigned)src << 9, idx0 = 16u * (unsigned)half, slot16 = 16u * (unsigned)slot;
                unsigned o0, o1, o2, o3;
                switch (r4) {
#define HN_BPERM(R4) asm volatile("ds_bpermute_b32 %0, %4, %5 offset:%6\n\tds_bpermute_b32 %1, %4, %5 offset:%7\n\t" \
                                  "ds_bpermute_b32 %2, %4, %5 offset:%8\n\tds_bpermute_b32 %3, %4, %5 offset:%9\n\ts_waitcnt lgkmcnt(0)" \
                                  : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) \
                                  : "v"(idx0), "v"(rowoff), "n"(4 * (0 + 8 * R4)), "n"(4 * (1 + 8 * R4)), "n"(4 * (2 + 8 * R4)), "n"(4 * (3 + 8 * R4)))
                    case 0: HN_BPERM(0); break;
                    case 1: HN_BPERM(1); break;
                    case 2: HN_BPERM(2); break;
                    default: HN_BPERM(3); break;
#undef HN_BPERM
                }
                HN[4 * r4 + 0] = *(const f32x4*)((const char*)a.hn + (o0 + slot16));
                HN[4 * r4 + 1] = *(const f32x4*)((const char*)a.hn + (o1 + slot16));
                HN[4 * r4 + 2] = *(const f32x4*)((const char*)a.hn + (o2 + slot16));
                HN[4 * r4 + 3] = *(const f32x4*)((const char*)a.hn + (o3 + slot16));
            };
            gemm128_f16x3_lazy<false>((const f16x8*)buf0, lane, PC, ACC, [&](int tp) { ACC[tp] = bias_block(vb3, tp, half); },
                [&](int tp, int r0) { silu_split_pair(PB, tp, r0, ACC[tp][r0], ACC[tp][r0 + 1], sk); },
                [&](int i) {
                    if (i < 8) stage_chunk<NW>(a.w4p, buf1, wave, lane16, i);
                    else if (i == 24) gather_hn(0);
                    else if (i == 28) gather_hn(1);
                });
            FT(4);
            phase_barrier<8>();                                       // the hn gathers stay in flight
            FT(5);
            // ===== phase 4: e_emb = T4 W4^T + b4 (F2), message, segment sum =====
            const unsigned keep_bits = ~(mask << 1);
            gemm128_f16x3_lazy<true>((const f16x8*)buf1, lane, PB, RC,
                [&](int tp) {
                    const float b = vb4[32 * tp + slot];
#pragma unroll
                    for (int r = 0; r < 16; ++r) RC[tp][r] = b;
                },
                [&](int tp, int r0) {
                    // message folded into the running sum of its piece as the fp32 kernels do it (gamd_msg_acc: one fused
                    // multiply-add; RC[tp][r] restarts after every edge that closes a destination segment)
#pragma unroll
                    for (int r = r0; r < r0 + 2; ++r)
                        RC[tp][r] = gamd_msg_acc(HN[r][tp], RC[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RC[tp][r - 1] : 0.f);
                },
                [&](int i) {
                    if (i < 8) stage_chunk<NW>(a.w1p, buf0, wave, lane16, i);
                    if (i == 1) gather_hn(2);
                    else if (i == 3) gather_hn(3);
                });
            unsigned pend_ends = mask;
            if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) pend_ends |= 1u << (nvalid - 1);
            // the next tile's e, behind GEMM 4 (the operand sets are dead, the piece sums and the barrier cover the latency)
            // (unconditional: "PA keeps its old value when there is no next tile" would keep the CURRENT tile's e alive through all
            //  four phases; the clamped index re-reads a tile that exists on the last iteration)
            load_e_tile_s(a.e_frag, active_n ? tile_n : tile, lane16, PA);
            FT(6);
            phase_barrier<16>();                                      // the next e tile stays in flight
            FT(7);
            int p = p0;
            int slot_o = slot;
            asm volatile("" : "+v"(slot_o));      // the store address is built here: hoisted out of the tile loop it is a 64-bit
                                                  // per-lane pointer that does not fit next to the register sets (2 spilled registers)
            while (__any(pend_ends != 0)) {
                if (pend_ends != 0) {
                    const int r = __builtin_ctz(pend_ends);
                    pend_ends &= pend_ends - 1;
                    float* prow = a.partial + ((size_t)p * GAMD_H + slot_o);
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        float v = RC[tp][0];
#pragma unroll
                        for (int k = 1; k < 16; ++k) v = (r == k) ? RC[tp][k] : v;
                        prow[32 * tp] = v;
                    }
                    ++p;
                }
            }
            FT(8);
            if (TIME) tacc[15] += 1;
        }
        tile = tile_n; active = active_n; src = src_n; dst = dst_n;
    }
#ifdef GAMD_PROFILING
    if (TIME && a.tdbg && lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 16; ++i) a.tdbg[((size_t)blockIdx.x * 8 + wave) * 16 + i] = tacc[i];
#endif
#undef FT
}

}  // namespace

template <bool TIME>
static int launch_f16x3(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * CONV_LDS_FLOATS;
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_conv_edge_f16x3<TIME>)) return e;
    hipLaunchKernelGGL(k_conv_edge_f16x3<TIME>, dim3(n_blocks), dim3(512), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_conv_edge_f16x3(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
#ifdef GAMD_PROFILING
    static const bool timed = getenv("GAMD_F16X3_TIME") != nullptr;
    if (timed) return launch_f16x3<true>(a, n_blocks, st);
#endif
    return launch_f16x3<false>(a, n_blocks, st);
}
