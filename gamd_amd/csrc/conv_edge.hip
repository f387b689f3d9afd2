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
// workgroup barrier per GEMM phase.  The last GEMM runs in the F2 orientation so each lane ends up
// with 16 edges x 1 feature: the multiply by hn[src] and the segment sum are then in-lane, and every
// maximal run of edges (same destination, same 16-edge chunk) is written once as a "piece".  Pieces
// are summed per atom, in order, by the node kernel -> no atomics, bit-reproducible.
//
// Roofline: MFMA-bound.  8*128*128 = 131072 FLOP per edge per launch against ~1.6 KB of traffic.
#include "gamd_common.h"
#include "gamd_internal.h"

namespace {

constexpr int CONV_LDS_FLOATS = 2 * GAMD_WFRAG_FLOATS + 3 * 128;

__device__ __forceinline__ void stage_weight(const float* __restrict__ gw, float* ldsbuf, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int chunk = k * 8 + wave;      // 64 chunks of 1 KiB, lane-linear image == packed global image
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(gw + chunk * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(ldsbuf + chunk * 256), 16, 0, 0);
    }
}

__global__ void __launch_bounds__(512, 2) k_conv_edge(ConvEdgeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + GAMD_WFRAG_FLOATS;
    float* vb1 = buf1 + GAMD_WFRAG_FLOATS;
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = lane & 31, half = lane >> 5;
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    const int n_wg_tiles = (n_tiles + 7) / 8;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    stage_weight(a.w1p, buf0, wave, lane);
    __syncthreads();

    for (int wt = first; wt < end; wt += step) {
        const int tile = wt * 8 + wave;
        const bool active = tile < n_tiles;
        const long long base = (long long)tile * GAMD_TILE;
        const long long x = base + gamd_pi(slot);
        const bool valid = active && x < E;
        const int src = valid ? a.col[x] : 0;
        const int dst = valid ? a.erow[x] : 0;

        f32x16 X[4], acc[4];
        // ================= phase 1: T1 = SiLU(W1 e + b1) =================
        stage_weight(a.w2p, buf1, wave, lane);
        if (active) {
            const f32x4* ef = (const f32x4*)a.e_frag + (size_t)tile * 16 * 64;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = ef[(t * 4 + q) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
                }
            load_bias_chain(vb1, half, acc);
            gemm128<false>((const f32x4*)buf0, lane, X, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[t][r] = gamd_silu(acc[t][r]);
            // C-in of GEMM 2 = D[dst]; the loads fly across the barrier
            load_row_chain(a.D + (size_t)dst * GAMD_H, half, acc);
        }
        __syncthreads();
        // ================= phase 2: T3 = SiLU(W2 T1 + S[src] + D[dst]) =================
        stage_weight(a.w3p, buf0, wave, lane);
        if (active) {
            f32x16 SD[4];
            load_row_chain(a.S + (size_t)src * GAMD_H, half, SD);     // issued now, consumed after the GEMM
            gemm128<false>((const f32x4*)buf1, lane, X, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[t][r] = gamd_silu(acc[t][r] + SD[t][r]);
        }
        __syncthreads();
        // ================= phase 3: T4 = SiLU(W3 T3 + b3) =================
        stage_weight(a.w4p, buf1, wave, lane);
        if (active) {
            load_bias_chain(vb3, half, acc);
            gemm128<false>((const f32x4*)buf0, lane, X, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[t][r] = gamd_silu(acc[t][r]);
        }
        __syncthreads();
        // ================= phase 4: e_emb = T4 W4^T + b4 (F2), message, segment sum =================
        stage_weight(a.w1p, buf0, wave, lane);       // next tile's W1 (harmless on the last iteration)
        if (active) {
            const int chunk = tile * 2 + half;
            const long long x0 = base + 16 * half;   // this half's 16 CSR edges: x0 + r
            long long nv = E - x0;
            const int nvalid = nv >= 16 ? 16 : (nv <= 0 ? 0 : (int)nv);
            int srcs[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4 c4 = *reinterpret_cast<const int4*>(a.col + x0 + 4 * g);
                srcs[4 * g + 0] = c4.x; srcs[4 * g + 1] = c4.y; srcs[4 * g + 2] = c4.z; srcs[4 * g + 3] = c4.w;
            }
            f32x16 hv[4];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s = r < nvalid ? srcs[r] : 0;
                const float* hrow = a.hn + (size_t)s * GAMD_H + slot;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) hv[tp][r] = hrow[32 * tp];
            }
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const float b = vb4[32 * tp + slot];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tp][r] = b;
            }
            gemm128<true>((const f32x4*)buf1, lane, X, acc);
            const unsigned mask = a.chunk_mask[chunk];
            const int p0 = a.chunk_piece[chunk];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                float* out = a.partial + 32 * tp + slot;
                int p = p0;
                float sum = 0.f;
                bool open = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r < nvalid) {
                        sum = __fadd_rn(sum, __fmul_rn(hv[tp][r], acc[tp][r]));   // nn_module.py:142 u_mul_e, sum
                        open = true;
                        if ((mask >> r) & 1u) { out[(size_t)p * GAMD_H] = sum; sum = 0.f; ++p; open = false; }
                    }
                }
                if (open) out[(size_t)p * GAMD_H] = sum;       // run continues in the next chunk: own piece
            }
        }
        __syncthreads();
    }
}

}  // namespace

int launch_conv_edge(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * CONV_LDS_FLOATS;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)k_conv_edge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return (int)e1;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_conv_edge, dim3(n_blocks), dim3(512), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
