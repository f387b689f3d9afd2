// conv_edge.hip — the edge side of one SmoothConvLayerNew, fully fused:
//
//   e (fragment order) -> edge_affine[Lin,SiLU,Lin] (+ S[src] + D[dst]) -> theta_edge[SiLU,Lin,SiLU,Lin]
//   -> multiply by hn[src] -> per-destination segment sum -> partial-sum pieces
//
// Replaces (reference, code/nn_module.py:135-142): six E-row Linears, two row gathers, two SiLU passes
// and DGL's u_mul_e -> sum SpMM, i.e. ~9 materialised [E,128] temporaries per layer.  src_affine /
// dst_affine are hoisted to node rows (S, D tables, computed by node.hip) — algebraically identical.
//
// Structure: persistent 512-thread workgroups (8 waves, 2 per SIMD), one per CU.  Each wave owns a
// 32-edge tile and carries its activations through four 128x128 fp32 MFMA GEMMs in registers
// (gamd_common.h).  The four weight matrices of the layer (4 x 64 KiB) are streamed L2 -> LDS with
// global_load_lds into a two-slot ring, one matrix ahead of the GEMM that consumes it; one
// workgroup barrier per GEMM phase.
//
// What the schedule is built around (profiles/r01_conv_edge_cycles.md, r02_ / r03_conv_edge_experiments.md):
// the two waves of a SIMD serialise their MFMA streams — whoever gets the matrix pipe first after a barrier keeps it
// until its GEMM is done, then the other one runs.  A phase therefore costs 2 GEMMs + whatever sits between the barrier
// and the first MFMA.  So:
// (1) everything a GEMM needs besides its LDS weights is loaded a phase ahead, into whichever of the three 64-register
// sets is free — by waves 0-3 BEFORE the barrier (the loads stay in flight across it: raw s_barrier + counted vmcnt),
// by waves 4-7 right AFTER it (see "Gather schedule" below);
// (2) nothing but the accumulator initialisation stands between a barrier and the first MFMA: the weight copy for the
// next phase, the piece stores of the previous tile and the D[dst] gather are issued 64 MFMAs INTO the GEMM (mid()
// hook).  In front of the GEMM they queued behind the other waves' gathers on the CU's address path, and the 8 x 3
// loop-invariant SGPRs of the per-copy addresses were restored from VGPR lanes (two v_readlane per copy) on that
// critical stretch; the copy now takes one base pair + immediate offsets (gamd_stage_weight_raw_contig);
// (3) element-wise post-ops (SiLU, message, segment sum) of output tile tp-1 are issued between the MFMA groups of tile
// tp.  fp32 MFMA and VALU share the SIMD's lanes, so this hides latency, not work: the kernel without any post-op runs
// at 0.87-0.89 of the matrix peak, with them at 0.81 (timing ablations in r03_conv_edge_experiments.md).
//
// The last GEMM runs in the F2 orientation so each lane ends up with 16 edges x 4 features: the
// multiply by hn[src] and the segment sum are in-lane, and every maximal run of edges (same
// destination, same 16-edge chunk) is written once as a "piece".  Pieces are summed per atom, in
// order, by the node kernel -> no atomics, bit-reproducible.
//
// Roofline: MFMA-bound.  8*128*128 = 131072 FLOP per edge per launch against ~1.6 KB of traffic.
#include "gamd_common.h"
#include "gamd_internal.h"

#include <cstdlib>

namespace {

constexpr int CONV_LDS_FLOATS = 2 * GAMD_WFRAG_FLOATS + 3 * 128;

// SiLU of an input that already carries the factor log2 e (CV_SILU_PRE)
__device__ __forceinline__ float silu_pre(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x)); }

// SiLU of a whole 16-register output block on PAIRS of elements: the multiply by -log2(e), the "+ 1" and the final product
// as packed instructions with the constants in registers (hipcc keeps them scalar because v_pk_* cannot take a literal);
// per element the same IEEE operations as gamd_silu_hw, so the bits do not change.  add: S[src] block of phase 2, or null.
// Used by the CV_BUNCH variant only (measured slower than the interleaved scalar form, r03_conv_edge_experiments.md §3).
__device__ __forceinline__ void silu_block16(f32x16& v, const f32x16* add, float c_neg_log2e, float c_one) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 c = {c_neg_log2e, c_neg_log2e}, one = {c_one, c_one};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f2 x = {v[2 * k], v[2 * k + 1]};
        if (add) { const f2 s2 = {(*add)[2 * k], (*add)[2 * k + 1]}; x = x + s2; }
        f2 y = x * c;
        y[0] = __builtin_amdgcn_exp2f(y[0]); y[1] = __builtin_amdgcn_exp2f(y[1]);
        y = y + one;
        y[0] = __builtin_amdgcn_rcpf(y[0]); y[1] = __builtin_amdgcn_rcpf(y[1]);
        x = x * y;
        v[2 * k] = x[0]; v[2 * k + 1] = x[1];
    }
}

// 128x128 GEMM of the chain with a software-pipelined element-wise post-op: while output tile tp is being accumulated (64
// MFMAs in 16 groups of 4), post(tp-1, g) finishes element g of the previous, already complete, output tile.  Only tile 3's
// post-op trails the last MFMA.
// mid() runs once, between output tiles 0 and 1 (64 MFMAs into the GEMM): memory instructions that are due "some time
// during this phase" (the weight copy for the next phase, piece stores, C-in gathers) are issued there instead of in front
// of the first MFMA, where they would queue behind the other waves' gathers on the CU's address path and keep the matrix
// pipe idle after every barrier (CV_INGEMM).
// BUNCH (variant): the 16 post-op elements of output tile tp - 1 run as ONE fenced block behind the first MFMA group of tile
// tp (post(tp - 1, -1)) instead of one element per MFMA group.
// PRIO (variants CV_PRIO_TAIL / CV_PRIO_BUNCH, round 6: conv_edge_bf16.hip's SiLU blocks gained 10 % from it): bit 0 = the
// trailing post-ops of output tile 3 (a pure vector stretch behind the last MFMA) at s_setprio 1, bit 1 = the fenced blocks of
// BUNCH at s_setprio 1.
template <bool F2, bool BUNCH = false, int PRIO = 0, typename WPtr, typename Post, typename Mid>
__device__ __forceinline__ void gemm128_post(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4], Post post, Mid mid) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        if (tp == 1) mid();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = W[((tp * 4 + t) * 4 + q) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = X[t][q * 4 + j];
                    acc[tp] = F2 ? mfma32(x, w[j], acc[tp]) : mfma32(w[j], x, acc[tp]);
                }
                if (!BUNCH) { if (tp > 0) post(tp - 1, t * 4 + q); }
                else if (tp > 0 && t == 0 && q == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (PRIO & 2) __builtin_amdgcn_s_setprio(1);
                    post(tp - 1, -1);                                   // g = -1: the whole block
                    if (PRIO & 2) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (PRIO) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); }
    if (BUNCH) { __builtin_amdgcn_sched_barrier(0); post(3, -1); }
    else {
#pragma unroll
        for (int g = 0; g < 16; ++g) post(3, g);
    }
    if (PRIO) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
}

// The same GEMM with an explicit ring of D fragment buffers (4 registers each): the read of fragment g + D is issued as soon
// as the four MFMAs of fragment g have issued, so every read has D - 1 full groups (256 cycles each) to land (hipcc's own
// schedule re-uses one or two buffers: read, s_waitcnt lgkmcnt(0), 4 MFMAs).  Measured neutral in the full kernel (CV_PF3,
// profiles/r03_conv_edge_experiments.md): the LDS latency is already covered by the other wave of the SIMD.  The reads are
// inline assembly and the waits are tied to the registers they guard, so no MFMA can be scheduled above its wait.
template <bool F2, int D, typename Post, typename Mid>
__device__ __forceinline__ void gemm128_post_pf(const float* Wlds, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4], Post post, Mid mid) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)Wlds + (unsigned)lane * 16u;
    f32x4 w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[d]) : "v"(addr), "n"(d * 1024));
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        if (tp == 1) mid();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = (tp * 4 + t) * 4 + q;
                f32x4& cur = w[g % D];
                // reads return in order: at most min(D - 1, 63 - g) younger ones may still be in flight
                if (63 - g >= D - 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(D - 1));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = X[t][q * 4 + j];
                    acc[tp] = F2 ? mfma32(x, cur[j], acc[tp]) : mfma32(cur[j], x, acc[tp]);
                }
                if (g + D < 64) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cur) : "v"(addr), "n"((g + D) * 1024));
                if (tp > 0) post(tp - 1, t * 4 + q);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

// End of a phase: every wave has its own weight DMA (issued at the phase start, before the N most
// recent VMEM loads) landed, then the workgroup meets.  The N prefetch loads stay in flight.
// vmcnt retires in order, so "at most N outstanding" proves the older DMA is done only if at least
// N loads really were issued after it: callers pass 0 on paths that skip the prefetch.
template <int N, bool SYNC = true>
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    if (SYNC) __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void load_e_tile(const float* __restrict__ e_frag, int tile, int lane, f32x16 (&X)[4]) {
    const f32x4* ef = (const f32x4*)e_frag + (size_t)tile * 16 * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = gamd_load_stream(&ef[(t * 4 + q) * 64 + lane]);
#pragma unroll
            for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
        }
}

// Kernel variants (template bit mask).  CV_TIME is the s_memtime instrumentation; the others are scheduling choices
// that do not change any result bit.  The release library instantiates CONV_PRODUCTION only; libgamd_hip_prof.so
// (-DGAMD_PROFILING) also builds the other combinations and selects one with GAMD_CONV_VARIANT for A/B timing.
// (Round-2 variants that were measured and dropped — DMA by waves 4-7 only, s_setprio schemes, gather / store ablations —
// are recorded in profiles/r02_conv_edge_experiments.md.)
enum {
    CV_TIME = 1,         // per-segment cycle counters -> a.tdbg
    CV_SYM_GATHER = 2,   // every wave issues its gathers BEFORE the phase barrier (round-1 schedule; see below)
    CV_TRACKED_DMA = 4,  // the weight copy through __builtin_amdgcn_global_load_lds (compiler-tracked: see gamd_stage_weight_raw)
    CV_INGEMM = 8,       // weight copy / piece stores / D gather issued 64 MFMAs into the GEMM instead of at the phase boundary
    CV_PRIO_TAIL = 16,   // the trailing post-ops of a GEMM (output tile 3's, behind the last MFMA) at s_setprio 1
    CV_PRIO_BUNCH = 64,  // with CV_BUNCH: the fenced post-op blocks at s_setprio 1
    CV_NOBARRIER = 32,   // TIMING ONLY (results invalid): the phase barriers are skipped, waves run free
    CV_ROW0 = 128,       // TIMING ONLY: every S / D / hn gather reads row 0 (cache-hot)
    CV_BUNCH = 256,      // post-ops of an output tile as one fenced block (gemm128_post<.., BUNCH>)
    CV_CONTIG_DMA = 512, // weight copy: 8 contiguous KiB per wave, immediate offsets (gamd_stage_weight_raw_contig)
    CV_HN2 = 1024,       // hn gather with one bpermute index register and scalar-base addressing (gather_hn2)
    CV_ZROW = 2048,      // padding slots of the last tile gather the all-zero row n instead of being masked per element
    CV_NOPOST = 8192,    // TIMING ONLY: no element-wise post-ops at all (SiLU, S add, message / segment sum skipped)
    CV_NOGATHER = 16384, // TIMING ONLY: no S / D / hn gathers and no e prefetch (registers keep stale values)
    CV_PF3 = 32768,      // weight fragments through an explicit ring of three buffers (gemm128_post_pf<3>)
    CV_NODMA = 65536,    // TIMING ONLY: no weight copies after the prologue
    CV_NOBIAS = 131072,  // TIMING ONLY: accumulators not initialised
    CV_SILU_PRE = 262144, // TIMING ONLY (the host does not scale the weights): SiLU without the multiply by log2 e in front of its
                         // exponential, as conv_edge_bf16.hip has it with log2 e folded into W1 / b1 / S / D / b3
};
#ifndef CONV_PRODUCTION
#define CONV_PRODUCTION (CV_INGEMM | CV_CONTIG_DMA | CV_HN2 | CV_ZROW)
#endif

// Gather schedule.  Of the two waves of a SIMD the one with the lower id is served first after a barrier (measured:
// waves 0-3 wait ~18 000 ticks at the barrier behind their own GEMM, waves 4-7 wait as long in front of theirs).  A gather
// (32 distinct rows per instruction) or a streaming prefetch occupies the CU's address path for thousands of cycles, and
// whatever a wave issues next queues behind it.  So:
//   * waves 0-3 issue the gathers for the next phase right after their GEMM, BEFORE the barrier: the address path is idle
//     then (waves 4-7 are in their GEMM) and the loads are long done when the barrier opens;
//   * waves 4-7 reach the barrier last; gathers issued there would sit in the queue in front of the next phase.  They issue
//     them AFTER the barrier, at the start of the phase, where they have a whole GEMM of waiting in front of them anyway;
//   * the weight copy of the phase after next is issued by every wave 64 MFMAs into its GEMM (CV_INGEMM), behind the gathers.
template <int V>
__global__ void __launch_bounds__(512, 2) k_conv_edge(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    constexpr bool TIME = (V & CV_TIME) != 0;
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
    // Work unit = 4 tiles, one per SIMD.  Waves 0-3 and 4-7 of the workgroup take successive units of its list, so
    // the work is balanced to half an iteration (an iteration with only waves 0-3 active takes about half the
    // time: the two waves of a SIMD serialise their MFMA streams anyway).
    const int n_units = (n_tiles + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;
    const int n_iter = ((end - first + step - 1) / step + 1) / 2;
    const int wsub = wave & 3, whalf = wave >> 2;
    const bool early = (V & CV_SYM_GATHER) ? true : whalf == 0;       // gathers before (true) / after (false) the barrier
    constexpr bool INGEMM = (V & CV_INGEMM) != 0;
    float c_nl2e = -1.4426950408889634f, c_one = 1.0f;          // SiLU constants in registers (silu_block16)
    asm volatile("" : "+v"(c_nl2e), "+v"(c_one));
    // vmcnt budget of a late wave at a boundary: with the copy issued inside the GEMM nothing younger than it is in flight
    constexpr int LATE_N = INGEMM ? 0 : 16;
    auto tile_of = [&](int it) {              // this wave's tile in iteration `it`, or n_tiles (inactive)
        const int u = first + (2 * it + whalf) * step;
        return (it < n_iter && u < end) ? u * 4 + wsub : n_tiles;
    };
    // L2 -> LDS copy of the next phase's weight matrix
    bool dma_on = true;
    auto stage = [&](const float* gw, float* buf) {
        if ((V & CV_NODMA) && !dma_on) return;
        if (V & CV_TRACKED_DMA) gamd_stage_weight<8>(gw, buf, wave, lane16);
        else if (V & CV_CONTIG_DMA) gamd_stage_weight_raw_contig<8>(gw, buf, wave, lane16);
        else gamd_stage_weight_raw<8>(gw, buf, wave, lane16);
    };

    long long tacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tacc[i] = 0;
    long long tprev = 0;
#define TMARK(i) do { if (TIME) { const long long tn__ = (long long)__builtin_readcyclecounter(); tacc[i] += tn__ - tprev; tprev = tn__; } } while (0)

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    stage(a.w1p, buf0);
    if (!INGEMM) stage(a.w2p, buf1);             // phase 1's copy (inside the loop it is issued at the previous tile's boundary 4)

    // three 64-register sets rotate through the roles {GEMM input, GEMM output, prefetched gather}
    f32x16 RA[4], RB[4], RC[4];
    auto init_b4 = [&]() {
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float b = vb4[32 * tp + slot];
#pragma unroll
            for (int r = 0; r < 16; ++r) RC[tp][r] = b;
        }
    };

    // per-lane edge of the current tile (slot order) and prefetch for the first tile
    int tile = tile_of(0);
    bool active = tile < n_tiles;
    int src = 0, dst = 0;
    {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        if (active && x < E) { src = GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst = GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST); }
        else if (V & CV_ZROW) { src = a.zero_row; dst = a.zero_row; }
        if (V & CV_ROW0) { src = 0; dst = 0; }
        if (active) {
            load_e_tile(a.e_frag, tile, lane, RA);
            if (!INGEMM) load_row_chain(a.D + (size_t)dst * GAMD_H, half, RC);
        }
    }
    asm volatile("" ::"v"(src), "v"(dst));      // compiler-visible wait for the index loads (see phase 4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (TIME) tprev = (long long)__builtin_readcyclecounter();
    unsigned pend_ends = 0;           // piece stores of the tile just finished (issued after its last barrier)
    int pend_p = 0;

    // S[src] rows, phase 2's post-op (chain layout, -> RA)
    auto gather_S = [&]() { load_row_chain(a.S + (size_t)src * GAMD_H, half, RA); };
    // hn[src] rows for phase 4 (-> RA; row layout: lane = features 4 slot .. 4 slot + 3 -- W4's output rows are packed in
    // that order, gamd_finalize_weights -- reg = edge): one 16-byte load per edge; the source index of edge (half, r)
    // lives in lane rho(r, half) of `src`
    auto gather_hn = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int s = __shfl(src, rho, 64);
            const f32x4 hv = *(const f32x4*)(a.hn + (size_t)s * GAMD_H + 4 * slot);
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) RA[r >> 2][(r & 3) * 4 + tp] = hv[tp];      // load lands in place
        }
    };

    // The same gather with ONE index register: ds_bpermute's immediate offset selects the lane (hipcc materialises the 16
    // lane indices of __shfl as 16 loop-invariant VGPRs), the byte offset src * 512 is what travels, and the load takes the
    // scalar base + 32-bit offset form (one v_add per edge instead of sign extension + 64-bit shift + 64-bit add).
    auto gather_hn2 = [&]() {
        const unsigned soff = (unsigned)src << 9;
        const unsigned idx0 = 16u * (unsigned)half;                  // bpermute address = 4 * lane: lanes 4 half + ...
        const unsigned slot16 = 16u * (unsigned)slot;
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
    };

    // Phase boundary: barrier (every wave's share of the weight copy issued one boundary earlier has landed: at most N
    // younger VMEM operations may still be in flight) + the copy for the phase after the next barrier.  Waves that gather
    // early pass it AFTER their gathers, the others BEFORE: the branch is around the barrier, not around the loads, so
    // both groups run the same gather code into the same registers.
#define BOUNDARY(COND, N, STAGE_STMT)                               \
    do {                                                            \
        if (COND) phase_barrier<N, !(V & CV_NOBARRIER)>(); else phase_barrier<0, !(V & CV_NOBARRIER)>();      \
        if (!INGEMM) { STAGE_STMT; }                                \
    } while (0)
    // one store per finished piece of the previous tile: closing edges (mask bits) and, if the chunk's last valid edge does
    // not close a segment, that edge too (the run continues in the next chunk as its own piece)
    auto piece_stores = [&]() {
        while (__any(pend_ends != 0)) {
            if (pend_ends != 0) {
                const int r = __builtin_ctz(pend_ends);
                pend_ends &= pend_ends - 1;
                f32x4 pv;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float v = RC[tp][0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) v = (r == k) ? RC[tp][k] : v;
                    pv[tp] = v;
                }
                *(f32x4*)(a.partial + (size_t)pend_p * GAMD_H + 4 * slot) = pv;
                ++pend_p;
            }
        }
    };
    auto nomid = []() {};
    (void)nomid;
    // INGEMM: what a phase owes the memory system, issued 64 MFMAs into its GEMM (inactive waves still owe their share
    // of the weight copy: the `else` branches below)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define GEMM(F2, BUF, IN, OUT, ...)                                                              \
    do {                                                                                         \
        if (V & CV_PF3) gemm128_post_pf<F2, 3>(BUF, lane, IN, OUT, __VA_ARGS__);                 \
        else gemm128_post<F2, (V & CV_BUNCH) != 0, ((V & CV_PRIO_TAIL) ? 1 : 0) | ((V & CV_PRIO_BUNCH) ? 2 : 0)>((const f32x4*)BUF, lane, IN, OUT, __VA_ARGS__); \
    } while (0)

    if (V & CV_NODMA) dma_on = false;
    for (int it = 0; it < n_iter; ++it) {
        const int x0 = tile * GAMD_TILE + 16 * half;            // this half's 16 CSR edges: x0 + r
        int nvalid = E - x0;
        nvalid = !active ? 0 : (nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid));
        // next tile of this wave (indices prefetched during phase 3)
        const int tile_n = tile_of(it + 1);
        const bool active_n = tile_n < n_tiles;
        int src_n = (V & CV_ZROW) ? a.zero_row : 0, dst_n = src_n;      // padding slots: the all-zero row (CV_ZROW)

        // ===== phase 1: RB = SiLU(W1 e + b1)        in RA = e (prefetched), RC = D[dst] (prefetched) =====
        // (W2 -> buf1 was issued at the previous boundary)
        if (active) {
            if (!(V & CV_NOBIAS)) load_bias_chain(vb1, half, RB);
            TMARK(0);
            GEMM(false, buf0, RA, RB,
                                [&](int tp, int g) {
                                    if (V & CV_NOPOST) return;
                                    if (g < 0) silu_block16(RB[tp], nullptr, c_nl2e, c_one); else RB[tp][g] = (V & CV_SILU_PRE) ? silu_pre(RB[tp][g]) : gamd_silu_hw(RB[tp][g]);
                                },
                                [&]() {
                                    if (INGEMM) {       // previous tile's pieces out of RC, then D[dst] (C-in of phase 2) into it, W2 -> buf1
                                        FENCE();
                                        piece_stores();
                                        if (!(V & CV_NOGATHER)) load_row_chain(a.D + (size_t)dst * GAMD_H, half, RC);
                                        stage(a.w2p, buf1);
                                        FENCE();
                                    }
                                });
            TMARK(1);
        } else if (INGEMM) {
            piece_stores();
            stage(a.w2p, buf1);
        }
        // boundary 1: S[src] -> RA for phase 2's post-op; W3 -> buf0.  Younger than the copy of W2: e, D (late) / D, S (early)
        if (!early) BOUNDARY(active, LATE_N, stage(a.w3p, buf0));
        TMARK(2);
        if (active && !(V & CV_NOGATHER)) gather_S();
        TMARK(3);
        if (early) BOUNDARY(active, 16, stage(a.w3p, buf0));
        // ===== phase 2: RC = SiLU(W2 T1 + D[dst] + S[src])        in RB, S in RA =====
        if (active) {
            GEMM(false, buf1, RB, RC,
                                [&](int tp, int g) {
                                    if (V & CV_NOPOST) return;
                                    if (g < 0) silu_block16(RC[tp], &RA[tp], c_nl2e, c_one); else RC[tp][g] = (V & CV_SILU_PRE) ? silu_pre(RC[tp][g] + RA[tp][g]) : gamd_silu_hw(RC[tp][g] + RA[tp][g]);
                                },
                                [&]() { if (INGEMM) { FENCE(); stage(a.w3p, buf0); FENCE(); } });
            TMARK(4);
        } else if (INGEMM) {
            stage(a.w3p, buf0);
        }
        // boundary 2: hn[src] -> RA for phase 4; W4 -> buf1.  Younger than the copy of W3: S (late) / hn (early)
        if (!early) BOUNDARY(active, LATE_N, stage(a.w4p, buf1));
        TMARK(5);
        if (active && !(V & CV_NOGATHER)) { if (V & CV_HN2) gather_hn2(); else gather_hn(); }
        TMARK(6);
        if (early) BOUNDARY(active, 16, stage(a.w4p, buf1));
        // ===== phase 3: RB = SiLU(W3 T3 + b3)        in RC =====
        unsigned mask = 0;
        int p0 = 0;
        // small index loads for phase 4 / the next tile go first: done long before the barrier needs vmcnt(0)
        if (active) {
            mask = a.chunk_mask[tile * 2 + half];
            p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE);
        }
        if (active_n) {
            const int xn = tile_n * GAMD_TILE + gamd_pi(slot);
            if (xn < E) { src_n = GAMD_CHK_RANGE(a.sticky, a.col[xn], 0, a.zero_row, GAMD_CHK_CONV_SRC); dst_n = GAMD_CHK_RANGE(a.sticky, a.erow[xn], 0, a.zero_row, GAMD_CHK_CONV_DST); }
            if (V & CV_ROW0) { src_n = 0; dst_n = 0; }
        }
        if (active) {
            if (!(V & CV_NOBIAS)) load_bias_chain(vb3, half, RB);
            TMARK(7);
            GEMM(false, buf0, RC, RB,
                                [&](int tp, int g) {
                                    if (V & CV_NOPOST) return;
                                    if (g < 0) silu_block16(RB[tp], nullptr, c_nl2e, c_one); else RB[tp][g] = (V & CV_SILU_PRE) ? silu_pre(RB[tp][g]) : gamd_silu_hw(RB[tp][g]);
                                },
                                [&]() { if (INGEMM) { FENCE(); stage(a.w4p, buf1); FENCE(); } });
            TMARK(8);
        } else if (INGEMM) {
            stage(a.w4p, buf1);
        }
        // boundary 3 (nothing to gather): everything has landed behind it
        phase_barrier<0, !(V & CV_NOBARRIER)>();
        // The index loads of phase 3 are complete, but hipcc cannot see a wait written in assembly: it would keep them on
        // its scoreboard and later flush vmcnt(0) — in front of the piece-store loop, at the next tile's first use of src,
        // and before it re-initialises src_n.  Naming the registers here makes it emit its wait now, where it costs nothing.
        asm volatile("" ::"v"(mask), "v"(p0), "v"(src_n), "v"(dst_n));
        if (!INGEMM) stage(a.w1p, buf0);       // next tile's W1 (harmless on the last iteration: drained at boundary 4)
        TMARK(9);
        // ===== phase 4: RC = T4 W4^T + b4 (F2: 16 edges x 4 features per lane), message, segment sum =====
        if (active) {
            if (!(V & CV_NOBIAS)) init_b4();
            // e_emb for this lane's 16 edges x 4 features, then message + segment sum (nn_module.py:142
            // u_mul_e -> sum).  In-stream part is branch-free: RC[tp][r] becomes the running sum of the
            // messages of the current piece (reset after every edge that closes a destination segment).
            const unsigned keep_bits = ~(mask << 1);          // bit r set: edge r continues edge r-1's piece
            GEMM(true, buf1, RB, RC, [&](int tp, int rr) {
              if (V & CV_NOPOST) return;
              _Pragma("unroll") for (int r = (rr < 0 ? 0 : rr); r < (rr < 0 ? 16 : rr + 1); ++r) {
                // padding edges (r >= nvalid, last tile only): masked here, or hn[zero_row] = 0 makes the product an exact zero
                const float hnv = (V & CV_ZROW) ? RA[r >> 2][(r & 3) * 4 + tp] : ((r < nvalid) ? RA[r >> 2][(r & 3) * 4 + tp] : 0.f);
                RC[tp][r] = gamd_msg_acc(hnv, RC[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RC[tp][r - 1] : 0.f);
              }
            }, [&]() { if (INGEMM) { FENCE(); if (it + 1 < n_iter) stage(a.w1p, buf0); FENCE(); } });
            // piece stores are deferred past the boundary (vmcnt counts stores too: issued here they would sit in front of
            // the prefetch loads and a counted wait would wait for their write latency)
            pend_ends = mask;
            if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) pend_ends |= 1u << (nvalid - 1);
            pend_p = p0;
            TMARK(10);
        } else if (INGEMM) {
            if (it + 1 < n_iter) stage(a.w1p, buf0);
        }
        // boundary 4: the next tile's e -> RA; its W2 -> buf1 (not behind the last tile: a copy must not outlive the
        // workgroup).  Younger than the copy of W1: nothing (late) / e (early)
        if (!early) BOUNDARY(false, 0, if (it + 1 < n_iter) stage(a.w2p, buf1));      // (late waves: nothing younger than the copy)
        TMARK(11);
        if (active_n && !(V & CV_NOGATHER)) load_e_tile(a.e_frag, tile_n, lane, RA);
        TMARK(12);
        if (early) BOUNDARY(active_n, 16, if (it + 1 < n_iter) stage(a.w2p, buf1));
        TMARK(13);
        if (!INGEMM) {
            piece_stores();
            // D[dst] of the next tile (C-in of its phase 2) -> RC, now free; lands during phase 1
            if (active_n) load_row_chain(a.D + (size_t)dst_n * GAMD_H, half, RC);
        }
        TMARK(14);
        tile = tile_n; active = active_n; src = src_n; dst = dst_n;
    }
#undef BOUNDARY
#undef FENCE
#undef GEMM
    if (INGEMM) piece_stores();            // the last tile's pieces
    if (TIME && a.tdbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a.tdbg[((size_t)blockIdx.x * 8 + wave) * 16 + i] = tacc[i];
    }
#undef TMARK
}

template <int V>
int launch_variant(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * CONV_LDS_FLOATS;
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_conv_edge<V>)) return e;
    hipLaunchKernelGGL(k_conv_edge<V>, dim3(n_blocks), dim3(512), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_conv_edge(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
#ifdef GAMD_PROFILING
    static int v = -1;
    if (v < 0) { const char* s = getenv("GAMD_CONV_VARIANT"); v = s ? atoi(s) : CONV_PRODUCTION; }
    switch (v) {
#define CASE(X) case X: return launch_variant<X>(a, n_blocks, st)
        // (CONV_PRODUCTION = 3592; | 1 = cycle marks)
        CASE(0); CASE(1); CASE(2); CASE(4); CASE(6); CASE(8); CASE(520); CASE(1544); CASE(3592); CASE(3593); CASE(3624); CASE(3720);
        CASE(3848); CASE(3608); CASE(3864); CASE(3912); CASE(3928); CASE(11784); CASE(19976); CASE(28168); CASE(36360); CASE(60936); CASE(126472); CASE(192008); CASE(257544); CASE(265736);
#undef CASE
        default: break;
    }
#endif
    return launch_variant<CONV_PRODUCTION>(a, n_blocks, st);
}
