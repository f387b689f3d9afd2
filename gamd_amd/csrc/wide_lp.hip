// wide_lp.hip — the reduced-precision conv-layer edge kernels for the generic widths of wide.hip: Eh = 128 EHT, H = 128 HT
// (EHT, HT in {1, 2}), hidden_dim = 128 -- the DFT-water configuration (256 / 128 / 256 x 5, water/test_script/
// test_nosehoover_hb.py:69-81), the trainers' default widths (LJ/train_network_lj.py:394-396), anything zero-padded.
//   k_conv_edge_f16x3_wide: split-fp16, fp32-grade (W x ~= W_hi x_hi + (W_hi x_lo + W_lo x_hi), gamd_f16x3.h); the fp32 goldens (1e-5)
//   k_conv_edge_bf16_wide:  operands rounded to bf16, fp32 accumulate (BASELINE config 5's arithmetic; tolerance restated, 1e-2)
//
// One body for both (the Mode structs below carry what differs: operand set, MFMA step, post-op, image and fragment sizes).
// Structure of conv_edge_f16x3.hip -- 512-thread persistent workgroups, two waves per SIMD, one 32-edge tile per wave, lazily
// initialised accumulators that become the next GEMM's operands, gathers as quads -- with the block decomposition of wide.hip:
// a tile runs NP = EHT + 2 + HT GEMM phases over the weight blocks W1[:, kb] | W2 | W3 | W4[ob, :] (contiguous images from
// a.w1p), streamed through the 2-slot ring (64 KiB stride) one phase ahead.  Phase 1 accumulates over the EHT blocks of e
// (written in operand form by k_edge_encode_wide, e_format 2 / 1); phase 4 runs once per 128-wide output block: hn[src] block,
// message, piece sums, one store per finished piece and block.
#include "gamd_bf16.h"
#include "gamd_f16x3.h"
#include "gamd_internal.h"

namespace {

constexpr int WCONV_LDS_FLOATS = 2 * GAMD_WFRAG_FLOATS + 2 * 128 + 256;

struct ModeF16X3 {
    using Set = OpSet;
    using Frag = f16x8;
    using Consts = SiluK2;
    static constexpr int IMG_KB = 64;            // one weight block: [hi | lo] fp16 images
    static constexpr int E_BYTES = 16384;        // one (tile, block) of e: [t][u][hi | lo][lane][8 halves]
    static constexpr int E_LOADS = 16;           // 16-byte loads per lane of it
    static __device__ __forceinline__ Consts consts() {
        Consts k{{-1.4426950408889634f, -1.4426950408889634f}, {1.0f, 1.0f}};
        asm volatile("" : "+v"(k.nl2e), "+v"(k.one));
        return k;
    }
    static __device__ __forceinline__ void put_e(Set& P, int k, gamd_u32x4_t v) { P.w[k >> 2][(k >> 1) & 1][k & 1] = v; }
    static __device__ __forceinline__ void post(Set& P, int t, int r0, float x0, float x1, const Consts& k) { silu_split_pair(P, t, r0, x0, x1, k); }
    template <bool F2, typename Init, typename Post, typename Step>
    static __device__ __forceinline__ void gemm(const Frag* W, int lane, const Set& P, f32x16 (&acc)[4], Init init, Post post_, Step step) {
        gemm128_f16x3_lazy<F2>(W, lane, P, acc, init, post_, step);
    }
};
struct ModeBF16 {
    using Set = OpSetB;
    using Frag = bf16x8;
    using Consts = SiluKB;
    static constexpr int IMG_KB = 32;            // one weight block: a bf16 image (half of a ring slot)
    static constexpr int E_BYTES = 8192;         // one (tile, block) of e: [t][u][lane][8 values]
    static constexpr int E_LOADS = 8;
    static __device__ __forceinline__ Consts consts() {
        Consts k{{-1.4426950408889634f, -1.4426950408889634f}, {1.0f, 1.0f}};
        asm volatile("" : "+v"(k.nl2e), "+v"(k.one));
        return k;
    }
    static __device__ __forceinline__ void put_e(Set& P, int k, gamd_u32x4_t v) { P.w[k >> 1][k & 1] = __builtin_bit_cast(gamd_u32x4, v); }
    static __device__ __forceinline__ void post(Set& P, int t, int r0, float x0, float x1, const Consts& k) { silu_pack_pair(P, t, r0, x0, x1, k); }
    template <bool F2, typename Init, typename Post, typename Step>
    static __device__ __forceinline__ void gemm(const Frag* W, int lane, const Set& P, f32x16 (&acc)[4], Init init, Post post_, Step step) {
        gemm128_bf16_lazy<F2>(W, lane, P, acc, init, post_, step);
    }
};

// one (tile, block) of e in operand form: scalar base per 4 KiB group (the immediate offset of a global load ends at 4 095) + the
// lane offset every wave holds anyway -- left to itself hipcc keeps a 64-bit per-lane offset pair per group alive across the loop
template <typename M>
__device__ __forceinline__ void load_e_block(const float* __restrict__ e_frag, int idx, unsigned lane16, typename M::Set& P) {
    const char* base = reinterpret_cast<const char*>(e_frag) + (size_t)__builtin_amdgcn_readfirstlane(idx) * M::E_BYTES;
#pragma unroll
    for (int grp = 0; grp < M::E_LOADS / 4; ++grp) {
        const char* bk = base + 4096 * grp;
        asm volatile("" : "+s"(bk));
#pragma unroll
        for (int k = 0; k < 4; ++k) M::put_e(P, 4 * grp + k, *reinterpret_cast<const gamd_u32x4_t*>(bk + (lane16 + (unsigned)(1024 * k))));
    }
}
// this wave's share of a weight image, all at once (prologue / waves without a tile)
template <typename M, int NW>
__device__ __forceinline__ void stage_block(const float* gw, float* ldsbuf, int wave, unsigned lane16) {
#pragma unroll
    for (int k = 0; k < M::IMG_KB / NW; ++k) stage_chunk<NW, M::IMG_KB>(gw, ldsbuf, wave, lane16, k);
}

template <typename M, int EHT, int HT>
__device__ __forceinline__ void conv_lp_wide(const ConvEdgeArgs& a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;
    constexpr int NW = 8, NP = EHT + 2 + HT, H = 128 * HT, CHUNKS = M::IMG_KB / NW;
    using Frag = typename M::Frag;
    using Set = typename M::Set;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // bias vectors behind the two 64 KiB slots, addressed from one opaque base (see conv_edge_f16x3.hip)
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
    const int n_units = (n_tiles + NW - 1) / NW;
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;
    const int n_iter = (end - first + step - 1) / step;
    auto tile_of = [&](int it) {
        const int u = first + it * step;
        return (it < n_iter && u * NW + wave < n_tiles) ? u * NW + wave : n_tiles;
    };

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; }
    if (tid < H) vb4[tid] = a.b4[tid];
    stage_block<M, NW>(a.w1p, lds, wave, lane16);

    // running phase counter: block blk = g % NP sits in slot g & 1; the slot offset stays a run-time scalar (one base register +
    // immediate offsets for all 64 fragment reads of a phase, wide.hip)
    unsigned g = 0;
    int blk = 0;
    auto cur_w = [&]() -> const Frag* {
        unsigned off = (g & 1u) * (unsigned)(GAMD_WFRAG_FLOATS * sizeof(float));
        asm volatile("" : "+s"(off));
        return (const Frag*)((const char*)lds + off);
    };
    auto next_block = [&]() { return a.w1p + (size_t)((blk + 1 == NP) ? 0 : blk + 1) * (M::IMG_KB * 256); };      // floats per image
    auto next_slot = [&]() { return lds + ((g + 1) & 1u) * GAMD_WFRAG_FLOATS; };
    auto advance = [&]() { ++g; blk = (blk + 1 == NP) ? 0 : blk + 1; };

    Set PA;                           // e block 0 of the current tile (the only register set that crosses the tile loop)
    const typename M::Consts sk = M::consts();           // SiLU constants as register pairs (packed instructions take no literals)

    int tile = tile_of(0);
    bool active = tile < n_tiles;
    int src = a.zero_row, dst = a.zero_row;       // padding slots gather the all-zero row n of hn / S / D
    {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        if (active && x < E) { src = GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst = GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST); }
        if (active) load_e_block<M>(a.e_frag, tile * EHT, lane16, PA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first block's copy is not tracked by hipcc
    __syncthreads();

    for (int it = 0; it < n_iter; ++it) {
        const int tile_n = tile_of(it + 1);
        const bool active_n = tile_n < n_tiles;
        int src_n = a.zero_row, dst_n = a.zero_row;
        if (!active) {
            // no tile in this unit: copy this wave's share of every block and meet the barriers (register sets untouched)
#pragma unroll
            for (int ph = 0; ph < NP; ++ph) {
                stage_block<M, NW>(next_block(), next_slot(), wave, lane16);
                phase_barrier<0>();
                advance();
            }
        } else {
            const int x0 = tile * GAMD_TILE + 16 * half;
            int nvalid = E - x0;
            nvalid = nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid);
            Set PB, PC;                     // T1 / T4; e block 1 / T3
            f32x16 ACC[4], RC[4];
            const unsigned soff = ((unsigned)src << 9) + 16u * (unsigned)half, doff = ((unsigned)dst << 9) + 16u * (unsigned)half;
            const unsigned mask = a.chunk_mask[tile * 2 + half];
            const int p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE);
            // ===== phase 1: T1 = SiLU(W1 e + b1), K = Eh: one GEMM per 128-wide block of e =====
            {
                const Frag* W = cur_w();
                const float* nb = next_block();
                float* ns = next_slot();
                if (EHT == 1) {
                    M::template gemm<false>(W, lane, PA, ACC, [&](int tp) { ACC[tp] = bias_block(vb1, tp, half); },
                        [&](int tp, int r0) { M::post(PB, tp, r0, ACC[tp][r0], ACC[tp][r0 + 1], sk); },
                        [&](int i) { if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i); });
                } else {
                    // the second block of e is fetched during the second half of this GEMM (one 16-byte load per K step) and
                    // stays in flight across the barrier
                    const char* eb = reinterpret_cast<const char*>(a.e_frag) + (size_t)__builtin_amdgcn_readfirstlane(tile * EHT + 1) * M::E_BYTES;
                    M::template gemm<false>(W, lane, PA, ACC, [&](int tp) { ACC[tp] = bias_block(vb1, tp, half); },
                        [&](int, int) {},
                        [&](int i) {
                            if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i);
                            if (i >= 16 && i < 16 + M::E_LOADS) {
                                const int k = i - 16;
                                const char* bk = eb + 4096 * (k >> 2);
                                asm volatile("" : "+s"(bk));
                                M::put_e(PC, k, *reinterpret_cast<const gamd_u32x4_t*>(bk + (lane16 + (unsigned)(1024 * (k & 3)))));
                            }
                        });
                }
                if (EHT == 2) phase_barrier<M::E_LOADS>(); else phase_barrier<0>();      // the e loads stay in flight
                advance();
            }
            if (EHT == 2) {
                const Frag* W = cur_w();
                const float* nb = next_block();
                float* ns = next_slot();
                M::template gemm<false>(W, lane, PC, ACC, [&](int) {},
                    [&](int tp, int r0) { M::post(PB, tp, r0, ACC[tp][r0], ACC[tp][r0 + 1], sk); },
                    [&](int i) { if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i); });
                phase_barrier<0>();
                advance();
            }
            // ===== phase 2: T3 = SiLU((W2 T1 + D[dst]) + S[src]) =====
            {
                const Frag* W = cur_w();
                const float* nb = next_block();
                float* ns = next_slot();
                f32x4 SQ[4][4], DQ[4][4];
                M::template gemm<false>(W, lane, PB, RC,
                    [&](int tp) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) RC[tp][r] = 0.f;
                    },
                    [&](int tp, int r0) {
                        const int q = r0 >> 2, j = r0 & 3;
                        M::post(PC, tp, r0, (RC[tp][r0] + DQ[tp][q][j]) + SQ[tp][q][j],
                                        (RC[tp][r0 + 1] + DQ[tp][q][j + 1]) + SQ[tp][q][j + 1], sk);
                    },
                    [&](int i) {
                        const int tp = i >> 3, k = i & 7;
                        if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i);
                        const int q = tp == 3 ? k : k - 4;
                        if (q >= 0 && q < 4) {
                            SQ[tp][q] = *(const f32x4*)((const char*)a.S + (soff + (unsigned)(128 * tp + 32 * q)));
                            DQ[tp][q] = *(const f32x4*)((const char*)a.D + (doff + (unsigned)(128 * tp + 32 * q)));
                        }
                    });
                phase_barrier<0>();
                advance();
            }
            // ===== phase 3: T4 = SiLU(W3 T3 + b3) =====
            if (active_n) {
                const int xn = tile_n * GAMD_TILE + gamd_pi(slot);
                if (xn < E) { src_n = GAMD_CHK_RANGE(a.sticky, a.col[xn], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst_n = GAMD_CHK_RANGE(a.sticky, a.erow[xn], 0, a.zero_row, GAMD_CHK_CONV_DST); }
            }
            {
                const Frag* W = cur_w();
                const float* nb = next_block();
                float* ns = next_slot();
                M::template gemm<false>(W, lane, PC, ACC, [&](int tp) { ACC[tp] = bias_block(vb3, tp, half); },
                    [&](int tp, int r0) { M::post(PB, tp, r0, ACC[tp][r0], ACC[tp][r0 + 1], sk); },
                    [&](int i) { if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i); });
                phase_barrier<0>();
                advance();
            }
            // ===== phase 4, once per 128-wide output block: e_emb = T4 W4[ob]^T + b4 (F2), message with hn[src], segment sum =====
            const unsigned keep_bits = ~(mask << 1);
            unsigned ends = mask;
            if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends |= 1u << (nvalid - 1);
#pragma unroll
            for (int ob = 0; ob < HT; ++ob) {
                const Frag* W = cur_w();
                const float* nb = next_block();
                float* ns = next_slot();
                // hn[src] block ob in the row layout (lane = feature 128 ob + 32 tp + slot, register = edge): natural [n][H] rows,
                // 4 coalesced dword loads per edge; the row offset of edge r lives in lane rho(r, half) (one bpermute index
                // register + immediate lane offsets); edges 4 r4 .. 4 r4 + 3 in front of K step r4 (first used 8 steps later)
                f32x4 HN[16];
                auto gather_hn = [&](int r4) {
                    const unsigned rowoff = (unsigned)src * (unsigned)(H * 4), idx0 = 16u * (unsigned)half, slot4 = 4u * (unsigned)slot;
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
                    const unsigned o[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp)
                            HN[4 * r4 + k][tp] = *(const float*)((const char*)a.hn + (o[k] + slot4 + (unsigned)(512 * ob + 128 * tp)));
                };
                M::template gemm<true>(W, lane, PB, RC,
                    [&](int tp) {
                        const float b = vb4[128 * ob + 32 * tp + slot];
#pragma unroll
                        for (int r = 0; r < 16; ++r) RC[tp][r] = b;
                    },
                    [&](int tp, int r0) {
#pragma unroll
                        for (int r = r0; r < r0 + 2; ++r)
                            RC[tp][r] = gamd_msg_acc(HN[r][tp], RC[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RC[tp][r - 1] : 0.f);
                    },
                    [&](int i) {
                        if (i < CHUNKS) stage_chunk<NW, M::IMG_KB>(nb, ns, wave, lane16, i);
                        if (i < 4) gather_hn(i);
                    });
                // the next tile's first e block behind the last GEMM (clamped index on the last iteration: see conv_edge_f16x3.hip)
                if (ob == HT - 1) {
                    load_e_block<M>(a.e_frag, (active_n ? tile_n : tile) * EHT, lane16, PA);
                    phase_barrier<M::E_LOADS>();
                } else {
                    phase_barrier<0>();
                }
                advance();
                // one store per finished piece and output block
                int p = p0;
                unsigned pe = ends;
                while (__any(pe != 0)) {
                    if (pe != 0) {
                        const int r = __builtin_ctz(pe);
                        pe &= pe - 1;
                        float* prow = a.partial + ((size_t)p * H + 128 * ob + slot);
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
            }
        }
        tile = tile_n; active = active_n; src = src_n; dst = dst_n;
    }
}

template <int EHT, int HT>
__global__ void __launch_bounds__(512, 2) k_conv_edge_f16x3_wide(ConvEdgeArgs a) { conv_lp_wide<ModeF16X3, EHT, HT>(a); }
template <int EHT, int HT>
__global__ void __launch_bounds__(512, 2) k_conv_edge_bf16_wide(ConvEdgeArgs a) { conv_lp_wide<ModeBF16, EHT, HT>(a); }

template <bool BF, int EHT, int HT>
int conv_launch(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * WCONV_LDS_FLOATS;
    static PerDeviceOnce once;
    const void* fn = BF ? (const void*)k_conv_edge_bf16_wide<EHT, HT> : (const void*)k_conv_edge_f16x3_wide<EHT, HT>;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, fn)) return e;
    if (BF) hipLaunchKernelGGL((k_conv_edge_bf16_wide<EHT, HT>), dim3(n_blocks), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((k_conv_edge_f16x3_wide<EHT, HT>), dim3(n_blocks), dim3(512), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
template <bool BF>
int conv_dispatch(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) {
    if (eht == 1 && ht == 1) return conv_launch<BF, 1, 1>(a, n_blocks, st);
    if (eht == 1 && ht == 2) return conv_launch<BF, 1, 2>(a, n_blocks, st);
    if (eht == 2 && ht == 1) return conv_launch<BF, 2, 1>(a, n_blocks, st);
    if (eht == 2 && ht == 2) return conv_launch<BF, 2, 2>(a, n_blocks, st);
    return -22;
}

}  // namespace

int launch_conv_edge_f16x3_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) { return conv_dispatch<false>(a, eht, ht, n_blocks, st); }
int launch_conv_edge_bf16_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) { return conv_dispatch<true>(a, eht, ht, n_blocks, st); }
