// conv_edge_f16x3.hip — the conv-layer edge kernel (see conv_edge.hip for the data flow and the schedule) with its four
// 128x128 GEMMs on the fp16 matrix pipe via operand splitting (gamd_f16x3.h): fp32-grade results at 3/16 of the fp32
// matrix time.  Same persistent workgroups, 2-slot 64 KiB weight ring (the [hi | lo] fp16 image of a matrix is as
// large as its fp32 image), one barrier per GEMM phase and partial-sum pieces as the fp32 kernel.  A GEMM's input is
// an operand set (hi, lo fp16 images, 64 registers like the fp32 block it stands for); the post-op of a phase
// (SiLU, split) writes the next phase's operand set directly.  e arrives pre-split from the encoder.
// With the matrix time cut 5x the kernel is no longer MFMA-bound: probes/f16x3_chain_bench.hip puts the GEMM chain
// itself at ~7 000 cycles per 4-tile round (LDS operand feed + SiLU/split VALU), memory-instruction issue comes next.
#include "gamd_f16x3.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

constexpr int CONV_LDS_FLOATS = 2 * GAMD_WFRAG_FLOATS + 3 * 128;

// An operand set: the (hi, lo) fp16 images of a 32 x 128 activation block in MFMA operand order, 64 registers
// (the size of the fp32 block it replaces): w[t][u][part] = 4 dwords = 8 halves of K step (t, u).
struct OpSet { gamd_u32x4_t w[4][2][2]; };

// fp32 pair -> dword d of (hi, lo)
__device__ __forceinline__ void put_pair(OpSet& P, int t, int r0, float x0, float x1) {
    const gamd_f32x2_t x = {x0, x1};
    const gamd_f16x2 h = __builtin_convertvector(x, gamd_f16x2);
    const gamd_f32x2_t rem = x - __builtin_convertvector(h, gamd_f32x2_t);
    const gamd_f16x2 l = __builtin_convertvector(rem, gamd_f16x2);
    const int u = r0 >> 3, d = (r0 & 7) >> 1;
    P.w[t][u][0][d] = __builtin_bit_cast(unsigned, h);
    P.w[t][u][1][d] = __builtin_bit_cast(unsigned, l);
}

// 128x128 split-fp16 GEMM, output tile by output tile (24 back-to-back MFMAs per accumulator).  The element-wise
// post-op of the PREVIOUS output tile is issued between the K steps of the current one (post(tp, r0) handles elements
// r0, r0+1 of acc[tp]); step(i) is called in front of K step i = 0..31 for work that should ride in the MFMA shadow.
template <bool F2, typename Post, typename Step>
__device__ __forceinline__ void gemm128_f16x3_post(const f16x8* W, int lane, const OpSet& P, f32x16 (&acc)[4], Post post,
                                                   Step step) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                step((tp * 4 + t) * 2 + u);              // K step 0..31: hook for work that rides in the MFMA shadow
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
#pragma unroll
    for (int k = 0; k < 8; ++k) post(3, 2 * k);
}

// one 1 KiB piece (k = 0 .. 64/NW - 1 for this wave) of the same copy, to be issued between MFMAs: with one wave per
// SIMD the ~100 cycles each LDS-DMA instruction takes to issue are otherwise dead time of the matrix pipe
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "m0" on the clobber lists: see gamd_common.h
template <int NW>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ gw, float* ldsbuf, int wave, unsigned lane16, int k) {
    // (inline assembly: a compiler-tracked global_load_lds turns the next wait of any kind into vmcnt(0) lgkmcnt(0), see
    // gamd_stage_weight_raw in gamd_common.h; the landing is guaranteed by the counted vmcnt of phase_barrier.)  A wave's
    // 64 / NW KiB are contiguous and addressed by the instruction's immediate offset (which advances the global and the LDS
    // side alike): base pair + M0 are rebuilt per call from one opaque scalar instead of living in 16 x 3 loop-invariant,
    // spilled SGPRs per matrix.
    int woff = wave * (64 / NW) * 1024 + (k >> 2) * 4096;
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

// piece i (0..15) of load_row_chain / load_e_tile, so that a gather can be spread over the K steps of a GEMM
__device__ __forceinline__ void load_row_piece(const float* __restrict__ row, int half, f32x16 (&X)[4], int i) {
    const int t = i >> 2, q = i & 3;
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * t + 8 * q + 4 * half);
#pragma unroll
    for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
}

// End of a phase: every wave has its own weight DMA (issued at the phase start, before the N most
// recent VMEM loads) landed, then the workgroup meets.  The N prefetch loads stay in flight.
// vmcnt retires in order, so "at most N outstanding" proves the older DMA is done only if at least
// N loads really were issued after it: callers pass 0 on paths that skip the prefetch.
template <int N>
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
}

// pre-split e fragments written by edge_encode_f16x3.hip: [tile][t][u][hi|lo][lane][8 halves], 16 KiB per tile
__device__ __forceinline__ void load_e_tile(const float* __restrict__ e_frag, int tile, int lane, OpSet& P) {
    const gamd_u32x4_t* ef = reinterpret_cast<const gamd_u32x4_t*>(e_frag) + (size_t)tile * 16 * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 2; ++p) P.w[t][u][p] = ef[((t * 2 + u) * 2 + p) * 64 + lane];
}
__device__ __forceinline__ void load_e_piece(const float* __restrict__ e_frag, int tile, int lane, OpSet& P, int i) {
    const gamd_u32x4_t* ef = reinterpret_cast<const gamd_u32x4_t*>(e_frag) + (size_t)tile * 16 * 64;
    P.w[i >> 2][(i >> 1) & 1][i & 1] = ef[i * 64 + lane];
}

// One wave per SIMD (256-thread workgroups, the whole 512-entry register file per wave): the operand sets, the
// accumulators and all gathered rows of a tile stay in registers without spilling, and every gather is issued a full
// GEMM ahead of its use.  (Two waves per SIMD at 256 registers each spill ~100 registers and are slower.)
// TIME (profiling build, GAMD_F16X3_TIME=1): s_memtime between the segments of a tile, summed per wave -> a.tdbg[block][wave][16]
template <bool TIME>
__global__ void __launch_bounds__(256, 1) k_conv_edge_f16x3(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + GAMD_WFRAG_FLOATS;
    float* vb1 = buf1 + GAMD_WFRAG_FLOATS;
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    // work unit = 4 tiles, one per wave / SIMD; units are dealt to the workgroups XCD by XCD (gamd_xcd_range)
    const int n_units = (n_tiles + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;
    const int n_iter = (end - first + step - 1) / step;
    auto tile_of = [&](int it) {              // this wave's tile in iteration `it`, or n_tiles (inactive)
        const int u = first + it * step;
        return (it < n_iter && u * 4 + wave < n_tiles) ? u * 4 + wave : n_tiles;
    };

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    gamd_stage_weight_raw_contig<NW>(a.w1p, buf0, wave, lane16);

    // 64-register sets: two operand sets PA / PB alternate as GEMM input / output (the post-op of a phase writes its
    // activation directly as the split operands of the next phase); RA = S[src], RC = D[dst] -> accumulators of
    // phases 2 and 4 (piece sums), ACC = accumulators of phases 1 and 3, HN = hn[src] rows.
    OpSet PA, PB;
    f32x16 RA[4], RC[4], ACC[4];
    f32x4 HN[16];                     // HN[r][tp] = hn[src of edge r][32 tp + slot]

    int tile = tile_of(0);
    bool active = tile < n_tiles;
    int src = 0, dst = 0;
    {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        if (active && x < E) { src = a.col[x]; dst = a.erow[x]; }
        if (active) load_e_tile(a.e_frag, tile, lane, PA);
    }
    // the W1 copy above is issued from inline assembly: hipcc does not count it, so the wait for it is explicit (without it a
    // wave could read chunks of buf0 that another wave's copy has not filled yet on a cold first tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned pend_ends = 0;           // piece stores of the tile just finished (issued after its last barrier)
    int pend_p = 0;
    long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
#define FT(I) do { if (TIME) { __builtin_amdgcn_sched_barrier(0); const long long now__ = (long long)__builtin_readcyclecounter(); \
                               tacc[I] += now__ - tprev; tprev = now__; __builtin_amdgcn_sched_barrier(0); } } while (0)

    for (int it = 0; it < n_iter; ++it) {
        const int x0 = tile * GAMD_TILE + 16 * half;            // this half's 16 CSR edges: x0 + r
        int nvalid = E - x0;
        nvalid = !active ? 0 : (nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid));
        const int tile_n = tile_of(it + 1);
        const bool active_n = tile_n < n_tiles;
        int src_n = 0, dst_n = 0;

        if (TIME) tprev = (long long)__builtin_readcyclecounter();
        // ===== phase 1: T1 = SiLU(W1 e + b1) =====
        unsigned mask = 0;
        int p0 = 0;
        if (active) {
            // All memory instructions of the iteration ride between the MFMAs of the GEMMs (step hooks): with one wave
            // per SIMD every VMEM issue stall is otherwise dead time of the matrix pipe.  S[src], D[dst] (phase 2) here.
            const float* srow = a.S + (size_t)src * GAMD_H;
            const float* drow = a.D + (size_t)dst * GAMD_H;
            mask = a.chunk_mask[tile * 2 + half];
            p0 = a.chunk_piece[tile * 2 + half];
            load_bias_chain(vb1, half, ACC);
            gemm128_f16x3_post<false>((const f16x8*)buf0, lane, PA, ACC, [&](int tp, int r0) {
                put_pair(PB, tp, r0, gamd_silu_hw(ACC[tp][r0]), gamd_silu_hw(ACC[tp][r0 + 1]));
            }, [&](int i) {
                if (i < 16) stage_chunk<NW>(a.w2p, buf1, wave, lane16, i);
                else { load_row_piece(srow, half, RA, i - 16); load_row_piece(drow, half, RC, i - 16); }
            });
        } else {
            gamd_stage_weight_raw_contig<NW>(a.w2p, buf1, wave, lane16);
        }
        FT(0);                                                        // phase 1 (GEMM + SiLU/split, DMA + gather issue inside)
        if (active) phase_barrier<32>(); else phase_barrier<0>();     // S/D gathers (issued after the DMA) stay in flight
        FT(1);                                                        // barrier 1
        // ===== phase 2: T3 = SiLU(W2 T1 + S[src] + D[dst]) =====
        if (!active) gamd_stage_weight_raw_contig<NW>(a.w3p, buf0, wave, lane16);
        if (active) {
#pragma unroll
            for (int t = 0; t < 4; ++t) RC[t] += RA[t];
            gemm128_f16x3_post<false>((const f16x8*)buf1, lane, PB, RC, [&](int tp, int r0) {
                put_pair(PA, tp, r0, gamd_silu_hw(RC[tp][r0]), gamd_silu_hw(RC[tp][r0 + 1]));
            }, [&](int i) {
                if (i < 16) { stage_chunk<NW>(a.w3p, buf0, wave, lane16, i); return; }
                // hn[src] rows for phase 4 (row layout: lane = feature, reg = edge); the source index of edge (half, r)
                // lives in lane rho(r, half) of `src`.  hn rows are stored permuted (node.hip, hn_perm): features slot,
                // 32 + slot, 64 + slot, 96 + slot are adjacent, so one 16-byte load per edge.
                const int r = i - 16;
                const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int sr = __shfl(src, rho, 64);
                HN[r] = *reinterpret_cast<const f32x4*>(a.hn + (size_t)sr * GAMD_H + 4 * slot);
            });
        }
        FT(2);                                                        // phase 2
        if (active) phase_barrier<16>(); else phase_barrier<0>();     // hn gathers stay in flight
        FT(3);                                                        // barrier 2
        // ===== phase 3: T4 = SiLU(W3 T3 + b3) =====
        if (!active) gamd_stage_weight_raw_contig<NW>(a.w4p, buf1, wave, lane16);
        if (active_n) {
            const int xn = tile_n * GAMD_TILE + gamd_pi(slot);
            if (xn < E) { src_n = a.col[xn]; dst_n = a.erow[xn]; }
        }
        if (active) {
            load_bias_chain(vb3, half, ACC);
            gemm128_f16x3_post<false>((const f16x8*)buf0, lane, PA, ACC, [&](int tp, int r0) {
                put_pair(PB, tp, r0, gamd_silu_hw(ACC[tp][r0]), gamd_silu_hw(ACC[tp][r0 + 1]));
            }, [&](int i) { if (i < 16) stage_chunk<NW>(a.w4p, buf1, wave, lane16, i); });
        }
        FT(4);                                                        // idx loads + phase 3
        phase_barrier<0>();
        FT(5);                                                        // barrier 3
        // ===== phase 4: e_emb = T4 W4^T + b4 (F2: 16 edges x 4 features per lane), message, segment sum =====
        if (!active) gamd_stage_weight_raw_contig<NW>(a.w1p, buf0, wave, lane16);       // next tile's W1 (harmless on the last iteration)
        if (active) {
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const float b = vb4[32 * tp + slot];
#pragma unroll
                for (int r = 0; r < 16; ++r) RC[tp][r] = b;
            }
            // message + segment sum (nn_module.py:142 u_mul_e -> sum), branch-free: RC[tp][r] becomes the running sum
            // of the messages of the current piece (reset after every edge that closes a destination segment).  This kernel
            // is EXEMPT from gamd_msg_acc (gamd_common.h): it keeps the reference's multiply-then-add (two roundings) and a
            // per-element nvalid mask instead of the zero-row padding; the f32 / bf16 / wide / small kernels fuse the two.
            // Both forms are within the 1e-5 goldens; only the fp32 kernels are required to be bit-identical to one another.
            const unsigned keep_bits = ~(mask << 1);          // bit r set: edge r continues edge r-1's piece
            gemm128_f16x3_post<true>((const f16x8*)buf1, lane, PB, RC, [&](int tp, int r0) {
#pragma unroll
                for (int r = r0; r < r0 + 2; ++r) {
                    const float prod = (r < nvalid) ? HN[r][tp] * RC[tp][r] : 0.f;
                    if (r == 0) RC[tp][0] = prod;
                    else RC[tp][r] = (((keep_bits >> r) & 1u) ? RC[tp][r - 1] : 0.f) + prod;
                }
            }, [&](int i) {
                if (i < 16) stage_chunk<NW>(a.w1p, buf0, wave, lane16, i);
                else if (active_n) load_e_piece(a.e_frag, tile_n, lane, PA, i - 16);    // PA is free since phase 3
            });
            pend_ends = mask;
            if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) pend_ends |= 1u << (nvalid - 1);
            pend_p = p0;
        }
        FT(6);                                                        // phase 4 (GEMM + message + segment sum)
        if (active && active_n) phase_barrier<16>(); else phase_barrier<0>();   // next e tile stays in flight
        FT(7);                                                        // barrier 4
        // one store per finished piece: closing edges (mask bits) and, if the chunk's last valid edge does not close a
        // segment, that edge too (the run continues in the next chunk as its own piece)
        while (__any(pend_ends != 0)) {
            if (pend_ends != 0) {
                const int r = __builtin_ctz(pend_ends);
                pend_ends &= pend_ends - 1;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float v = RC[tp][0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) v = (r == k) ? RC[tp][k] : v;
                    a.partial[(size_t)pend_p * GAMD_H + 32 * tp + slot] = v;
                }
                ++pend_p;
            }
        }
        FT(8);                                                        // piece stores
        if (TIME && active) tacc[15] += 1;
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
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)k_conv_edge_f16x3<TIME>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return (int)e1;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_conv_edge_f16x3<TIME>, dim3(n_blocks), dim3(256), lds, st, a);
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
