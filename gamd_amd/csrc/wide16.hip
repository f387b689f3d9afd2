// wide16.hip — the conv layer's edge side (nn_module.py:135-142) for Eh = 128 EHT, H = 128 HT on 16-EDGE work units.
//
// k_conv_edge_wide (wide.hip) gives every wave a 32-edge tile on v_mfma_f32_32x32x2_f32; a SIMD therefore works in quanta of a whole
// tile (6 phases x 256 MFMAs x 64 cycles at the DFT-water widths), and a launch with 1 < tiles / SIMD < 1.5 — the 774-atom
// DFT-water configuration: 1 470 tiles on 1 024 SIMDs — takes two quanta where 1.44 would do (0.54 of the fp32 matrix peak).
// Here a wave owns one 16-edge chunk (GAMD_CHUNK: the unit of the partial-sum pieces already) on v_mfma_f32_16x16x4_f32: half the
// quantum, 2.87 chunks per SIMD -> three half-quanta.  One wave per SIMD (256-thread workgroups: an iteration of a workgroup is 4
// chunks, so the launch is quantised per SIMD, not per pair of waves), same 2-slot LDS ring of 64 KiB weight blocks, one
// barrier per phase.
//
// Bit-identical to k_conv_edge_wide.  Both MFMA shapes are chains of IEEE fused multiply-adds over K in ascending lane-group
// order (probes/mfma_korder_probe.hip: 2 x 32x32x2 == 16x16x4 == fmaf chain on 2 M elements), so it is enough to feed K in the
// order the 32-edge kernel accumulates it: position p = ((t 4 + q) 4 + j) 2 + half <-> input feature 32 t + 8 q + 4 half + j.
// MFMA m (0..31) of an output block covers positions 4 m .. 4 m + 3, lane group g = lane >> 4 supplies position 4 m + g:
//     kfeat(m, g) = 32 (m >> 3) + 8 ((m >> 1) & 3) + 4 (g & 1) + (g >> 1) + 2 (m & 1).
// "chain16x" layout of a 16 x 128 activation block: lane (a = lane & 15: edge, g), X[m] <-> feature kfeat(m, g) (32 registers).
// The weights are packed (pack16x, gamd_api.hip) with output row 16 ob + 4 g + r of a chained matrix = feature kfeat(4 ob + r, g):
// the C/D registers of one GEMM (acc[ob][r], lane (a, g)) ARE the B operands X[4 ob + r] of the next, as in the 32-edge kernels.
// Last GEMM in the F2 orientation (A = activations, B = weights): lane (n = lane & 15, g), acc[ob][r] <-> edge 4 g + r, output
// feature 64 (ob >> 2) + 4 n + (ob & 3) of the 128-block (rows packed so that a lane owns four consecutive features: hn rows are
// gathered and the pieces stored 16 bytes at a time).  The per-destination running sum of the messages is sequential over the 16
// edges of the chunk exactly as in the 32-edge kernel (one fused multiply-add per edge, in edge order): four lane groups hold
// four edges each, so it runs in four rounds, the carry crossing to the next group through ds_bpermute.
#include "gamd_common.h"
#include "gamd_internal.h"

namespace {

__device__ __forceinline__ f32x4 mfma16x(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void wide16_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// acc[ob] (+)= block GEMM over K = 128: W = packed image [ob 8][m4 8][lane 64] float4 (component c <-> MFMA m = 4 m4 + c).
// Output blocks in pairs: two independent accumulator chains (40-cycle dependent latency, 32-cycle issue) keep the pipe full.
// One wave per SIMD has no partner to hide the LDS round trip of its weight fragments, and hipcc's own schedule is read ->
// s_waitcnt -> 8 MFMAs (measured: phases at half the matrix rate): the fragments of step s + D are requested while step s
// computes (a step = one (o2, m4): 2 x ds_read_b128 feeding 8 MFMAs = 256 matrix cycles), pinned with sched_group_barrier as in
// gemm128_bf16_pf (gamd_bf16.h); the waits are hipcc's counted lgkmcnt.
template <bool F2, int D = 3>
__device__ __forceinline__ void gemm16x(const f32x4* W, int lane, const f32x4 (&X)[8], f32x4 (&acc)[8]) {
    __builtin_amdgcn_sched_barrier(0);
    f32x4 w0[D], w1[D];
#pragma unroll
    for (int s = 0; s < D; ++s) {
        w0[s] = W[((2 * (s >> 3)) * 8 + (s & 7)) * 64 + lane];
        w1[s] = W[((2 * (s >> 3) + 1) * 8 + (s & 7)) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int o2 = s >> 3, m4 = s & 7;
        const f32x4 a0 = w0[s % D], a1 = w1[s % D];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x = X[m4][c];                  // X[m], m = 4 m4 + c
            acc[2 * o2] = F2 ? mfma16x(x, a0[c], acc[2 * o2]) : mfma16x(a0[c], x, acc[2 * o2]);
            acc[2 * o2 + 1] = F2 ? mfma16x(x, a1[c], acc[2 * o2 + 1]) : mfma16x(a1[c], x, acc[2 * o2 + 1]);
        }
        if (s + D < 32) {
            const int n = s + D;
            w0[s % D] = W[((2 * (n >> 3)) * 8 + (n & 7)) * 64 + lane];
            w1[s % D] = W[((2 * (n >> 3) + 1) * 8 + (n & 7)) * 64 + lane];
        }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * D, 0);
#pragma unroll
    for (int s = 0; s < 32 - D; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * D, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// a [128] vector in plain feature order -> chain16x registers: X[m] = v[kfeat(m, g)].  For t = m >> 3, q = (m >> 1) & 3 the pair
// m = 8 t + 2 q + {0, 1} is components (g >> 1), (g >> 1) + 2 of the float4 at feature 32 t + 8 q + 4 (g & 1).
template <typename Ptr>
__device__ __forceinline__ void load_row16x(Ptr row, int g, f32x4 (&X)[8]) {
    const int base = 4 * (g & 1), j0 = g >> 1;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&row[32 * t + 8 * q + base]);
            const float lo = j0 ? v[1] : v[0], hi = j0 ? v[3] : v[2];
            X[2 * t + (q >> 1)][2 * (q & 1)] = lo;         // m = 8 t + 2 q  ->  X[m >> 2][m & 3]
            X[2 * t + (q >> 1)][2 * (q & 1) + 1] = hi;     // m = 8 t + 2 q + 1
        }
}

template <int EHT, int HT>
__global__ void __launch_bounds__(256, 1) k_conv_edge_wide16(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    constexpr int NP = EHT + 2 + HT;
    constexpr int H = 128 * HT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* vb1 = lds + 2 * GAMD_WFRAG_FLOATS;
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, la = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_chunks = 2 * n_tiles;
    const int n_wg_iters = (n_chunks + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_wg_iters, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; }
    if (tid < H) vb4[tid] = a.b4[tid];
    gamd_stage_weight_raw_contig<4>(a.w16p, lds, wave, lane16);
    wide16_barrier();

    unsigned ph = 0;                   // running phase counter: block ph % NP sits in slot ph & 1
    int blk = 0;
    // start the DMA of the next block into the other slot, hand back this phase's slot (k_conv_edge_wide's scheme: the copy is
    // issued from inline assembly and waited for by the vmcnt(0) of the barrier that ends the phase)
    auto begin_phase = [&]() -> const f32x4* {
        const int nb = (blk + 1 == NP) ? 0 : blk + 1;
        gamd_stage_weight_raw_contig<4>(a.w16p + (size_t)nb * GAMD_WFRAG_FLOATS, lds + ((ph + 1) & 1u) * GAMD_WFRAG_FLOATS, wave, lane16);
        unsigned off = (ph & 1u) * (unsigned)(GAMD_WFRAG_FLOATS * sizeof(float));
        asm volatile("" : "+s"(off));
        return (const f32x4*)((const char*)lds + off);
    };
    auto end_phase = [&]() { wide16_barrier(); ++ph; blk = (blk + 1 == NP) ? 0 : blk + 1; };

    // Everything a phase needs from global memory is requested one phase EARLIER, in front of that phase's weight DMA, and
    // waited for by the vmcnt(0) of the barrier in between: the DMA copies are invisible to hipcc's wait counting (they are
    // issued from inline assembly), so any wait the compiler places behind them for a younger load sits out the 64 KiB copy
    // as well — with one wave per SIMD and 8 192-cycle phases that was a third of the kernel.
    struct ChunkIdx { int chunk, src, dst, nvalid, p0; unsigned mask; int active; };
    auto load_idx = [&](int wi) -> ChunkIdx {
        ChunkIdx c;
        c.chunk = wi * 4 + wave;
        c.active = (wi < end && c.chunk < n_chunks) ? 1 : 0;
        const int x = c.chunk * GAMD_CHUNK + la;            // lane's CSR edge (F1 phases: lane = edge)
        const bool valid = c.active && x < E;
        c.src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC) : 0;
        c.dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST) : 0;
        int nv = E - c.chunk * GAMD_CHUNK;
        c.nvalid = !c.active ? 0 : (nv >= 16 ? 16 : (nv <= 0 ? 0 : nv));
        c.mask = 0; c.p0 = 0;
        if (c.active) { c.mask = a.chunk_mask[c.chunk]; c.p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[c.chunk], 0, a.piece_cap - 17, GAMD_CHK_PIECE); }
        return c;
    };
    // e fragments of K block kb: float4 (t, q) of lane (rho, half = g & 1) holds features 32 t + 8 q + 4 (g & 1) + 0..3 of the
    // lane's edge; rho = slot of that edge inside the 32-edge fragment image the encoder writes (gamd_pi^-1 of 16 hc + la)
    auto load_e = [&](const ChunkIdx& c, int kb, f32x4 (&X)[8]) {
        const int tile = c.active ? c.chunk >> 1 : 0, hc = c.chunk & 1;
        const int rho = 8 * (la >> 2) + 4 * hc + (la & 3);
        const f32x4* ef = (const f32x4*)a.e_frag + ((size_t)tile * EHT + kb) * 16 * 64 + rho + 32 * (g & 1);
        const int j0 = g >> 1;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = ef[(t * 4 + q) * 64];
                X[2 * t + (q >> 1)][2 * (q & 1)] = j0 ? v[1] : v[0];
                X[2 * t + (q >> 1)][2 * (q & 1) + 1] = j0 ? v[3] : v[2];
            }
    };
    // hn[src] of this lane group's four edges: two float4 per edge (features 64 G + 4 n .. + 3, G = 0, 1) of block OB
    auto load_hn = [&](const ChunkIdx& c, int OB, f32x4 (&HN)[4][2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sr = __shfl(c.src, 4 * g + r, 64);                   // src of edge 4 g + r sits in lanes (a = 4 g + r, *)
            const float* hrow = a.hn + (size_t)sr * H + 128 * OB + 4 * la;
            HN[r][0] = *reinterpret_cast<const f32x4*>(hrow);
            HN[r][1] = *reinterpret_cast<const f32x4*>(hrow + 64);
        }
    };

    ChunkIdx cur = load_idx(first);
    f32x4 E0[8];                                            // e, K block 0, of the chunk about to start
    load_e(cur, 0, E0);

    for (int wi = first; wi < end; wi += step) {
        asm volatile("" ::: "memory");
        const bool active = cur.active != 0;
        const int nvalid = cur.nvalid, p0 = cur.p0;
        const unsigned mask = cur.mask;

        f32x4 T[8], U[8], X[8], DQ[8], E1[8];
        f32x4 HN[2][4][2];
        // ---- T = SiLU(W1 e + b1), K = Eh --------------------------------------------------------
        load_row16x(vb1, g, T);
#pragma unroll
        for (int kb = 0; kb < EHT; ++kb) {
            // ahead: the second K block of e; S[src] (straight into the next accumulators) and D[dst]
            if (kb == 0 && EHT > 1) load_e(cur, 1, E1);
            if (kb == EHT - 1) { load_row16x(a.S + (size_t)cur.src * 128, g, U); load_row16x(a.D + (size_t)cur.dst * 128, g, DQ); }
            const f32x4* W = begin_phase();
            if (active) { if (kb == 0) gemm16x<false>(W, lane, E0, T); else gemm16x<false>(W, lane, E1, T); }
            end_phase();
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[o][r] = gamd_silu_hw(T[o][r]);
        // ---- U = SiLU(W2 T + S[src] + D[dst]) ---------------------------------------------------
        {
            const f32x4* W = begin_phase();
            if (active) {
#pragma unroll
                for (int o = 0; o < 8; ++o) U[o] += DQ[o];
                gemm16x<false>(W, lane, T, U);
#pragma unroll
                for (int o = 0; o < 8; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) U[o][r] = gamd_silu_hw(U[o][r]);
            }
            end_phase();
        }
        // ---- T = SiLU(W3 U + b3) ----------------------------------------------------------------
        {
            load_hn(cur, 0, HN[0]);                                        // ahead: hn rows of the first output block
            const f32x4* W = begin_phase();
            if (active) {
                load_row16x(vb3, g, T);
                gemm16x<false>(W, lane, U, T);
#pragma unroll
                for (int o = 0; o < 8; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[o][r] = gamd_silu_hw(T[o][r]);
            }
            end_phase();
        }
        // ---- e_emb block OB = T W4[OB]^T + b4 (F2), message with hn[src], segment sum ------------
        // lane (n = la, g): acc[ob][r] <-> edge 4 g + r of the chunk, feature 128 OB + 64 (ob >> 2) + 4 n + (ob & 3)
        ChunkIdx nxt = cur;
#pragma unroll
        for (int OB = 0; OB < HT; ++OB) {
            // ahead: the next block's hn rows, or (last block) the next chunk's indices and its first e block
            if (OB + 1 < HT) load_hn(cur, OB + 1, HN[(OB + 1) & 1]);
            else { nxt = load_idx(wi + step); load_e(nxt, 0, E0); }
            const f32x4* W = begin_phase();
            if (active) {
                const f32x4 (&HNb)[4][2] = HN[OB & 1];
#pragma unroll
                for (int ob = 0; ob < 8; ++ob) {
                    const float b = vb4[128 * OB + 64 * (ob >> 2) + 4 * la + (ob & 3)];
                    X[ob] = f32x4{b, b, b, b};
                }
                gemm16x<true>(W, lane, T, X);
                // message + per-destination running sum, sequential over the chunk's 16 edges (edge e = 4 g + r): round gg finishes
                // lane group gg; the sum at its last edge crosses to group gg + 1 (lane + 16) for the next round
                const unsigned keep_bits = ~(mask << 1);                   // bit e set: edge e continues the segment of edge e - 1
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    float cin[8];
#pragma unroll
                    for (int ob = 0; ob < 8; ++ob) {
                        const float last = X[ob][3];                       // (of lane - 16: the running sum at edge 4 gg - 1)
                        cin[ob] = gg == 0 ? 0.f : __shfl(last, lane - 16, 64);
                    }
                    if (g == gg) {
#pragma unroll
                        for (int ob = 0; ob < 8; ++ob) {
                            float run = cin[ob];
                            float o4[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int e = 4 * gg + r;
                                const float hnv = (e < nvalid) ? HNb[r][ob >> 2][ob & 3] : 0.f;
                                run = gamd_msg_acc(hnv, X[ob][r], (e > 0 && ((keep_bits >> e) & 1u)) ? run : 0.f);
                                o4[r] = run;
                            }
                            X[ob] = f32x4{o4[0], o4[1], o4[2], o4[3]};
                        }
                    }
                }
                // one 16-byte store per (finished piece, 64-feature group): the lane group that holds the closing edge writes it
                unsigned ends = mask;
                if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends |= 1u << (nvalid - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 4 * g + r;
                    if ((ends >> e) & 1u) {
                        const int p = p0 + __popc(ends & ((1u << e) - 1u));
                        float* prow = a.partial + (size_t)p * H + 128 * OB + 4 * la;
                        *reinterpret_cast<f32x4*>(prow) = f32x4{X[0][r], X[1][r], X[2][r], X[3][r]};
                        *reinterpret_cast<f32x4*>(prow + 64) = f32x4{X[4][r], X[5][r], X[6][r], X[7][r]};
                    }
                }
            }
            end_phase();
        }
        cur = nxt;
    }
}

template <int EHT, int HT>
int conv16_launch(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    constexpr int LDS_BYTES = (2 * GAMD_WFRAG_FLOATS + 256 + 128 * HT) * (int)sizeof(float);
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, LDS_BYTES, k_conv_edge_wide16<EHT, HT>)) return e;
    hipLaunchKernelGGL((k_conv_edge_wide16<EHT, HT>), dim3(n_blocks), dim3(256), LDS_BYTES, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_conv_edge_wide16(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) {
    if (eht == 1 && ht == 1) return conv16_launch<1, 1>(a, n_blocks, st);
    if (eht == 1 && ht == 2) return conv16_launch<1, 2>(a, n_blocks, st);
    if (eht == 2 && ht == 1) return conv16_launch<2, 1>(a, n_blocks, st);
    if (eht == 2 && ht == 2) return conv16_launch<2, 2>(a, n_blocks, st);
    return -22;
}
