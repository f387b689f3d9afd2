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

namespace {

constexpr int CONVB_LDS_BYTES = 4 * GAMD_WFRAG_BF16_BYTES + 3 * 128 * 4;

__global__ void __launch_bounds__(512, 2) k_conv_edge_bf16(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const bf16x8* W1 = reinterpret_cast<const bf16x8*>(ldsb);
    const bf16x8* W2 = W1 + 2048;
    const bf16x8* W3 = W2 + 2048;
    const bf16x8* W4 = W3 + 2048;
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
        for (int i = tid; i < 2048; i += 512) {
            dst[i] = s1[i]; dst[2048 + i] = s2[i]; dst[4096 + i] = s3[i]; dst[6144 + i] = s4[i];
        }
        if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; vb4[tid] = a.b4[tid]; }
    }
    __syncthreads();

    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_wg_tiles = (n_tiles + 7) / 8;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    const bf16x8* efrag = reinterpret_cast<const bf16x8*>(a.e_frag);

    for (int wt = first; wt < end; wt += step) {
        const int tile = wt * 8 + wave;
        if (tile >= n_tiles) continue;
        asm volatile("" ::: "memory");               // keep loop-invariant LDS reads (bias, weights) inside the loop
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = x < E;
        // padding slots of the last tile gather the all-zero row n of hn / S / D: their messages are exact zeros
        const int src = valid ? a.col[x] : a.zero_row;
        const int dst = valid ? a.erow[x] : a.zero_row;

        bf16x8 P[4][2];
        f32x16 RA[4], RB[4], RC[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) P[t][u] = efrag[((size_t)tile * 8 + t * 2 + u) * 64 + lane];
        load_row_chain(a.S + (size_t)src * GAMD_H, half, RA);
        load_row_chain(a.D + (size_t)dst * GAMD_H, half, RB);

        // phase 1: T1 = SiLU(W1 e + b1)
        load_bias_chain(vb1, half, RC);
        gemm128_bf16<false>(W1, lane, P, RC);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) RC[t][r] = gamd_silu_hw(RC[t][r]);
        pack_chain_bf16(RC, P);
        // phase 2: T3 = SiLU(W2 T1 + S[src] + D[dst])
#pragma unroll
        for (int t = 0; t < 4; ++t) RB[t] += RA[t];
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
        gemm128_bf16<false>(W2, lane, P, RB);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) RB[t][r] = gamd_silu_hw(RB[t][r]);
        pack_chain_bf16(RB, P);
        // phase 3: T4 = SiLU(W3 T3 + b3)
        load_bias_chain(vb3, half, RC);
        gemm128_bf16<false>(W3, lane, P, RC);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) RC[t][r] = gamd_silu_hw(RC[t][r]);
        pack_chain_bf16(RC, P);
        // phase 4: e_emb = T4 W4^T + b4 (F2), message, segment sum (fp32)
        const unsigned mask = a.chunk_mask[tile * 2 + half];
        int p = a.chunk_piece[tile * 2 + half];
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float b = vb4[32 * tp + slot];
#pragma unroll
            for (int r = 0; r < 16; ++r) RB[tp][r] = b;
        }
        gemm128_bf16<true>(W4, lane, P, RB);
        const unsigned keep_bits = ~(mask << 1);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                RB[tp][r] = gamd_msg_acc(RA[r >> 2][(r & 3) * 4 + tp], RB[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? RB[tp][r - 1] : 0.f);
        unsigned ends = mask;
        if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends |= 1u << (nvalid - 1);
        while (__any(ends != 0)) {
            if (ends != 0) {
                const int r = __builtin_ctz(ends);
                ends &= ends - 1;
                f32x4 pv;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float v = RB[tp][0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) v = (r == k) ? RB[tp][k] : v;
                    pv[tp] = v;
                }
                *(f32x4*)(a.partial + (size_t)p * GAMD_H + 4 * slot) = pv;
                ++p;
            }
        }
    }
}

}  // namespace

int launch_conv_edge_bf16(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)k_conv_edge_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            CONVB_LDS_BYTES);
        if (e1 != hipSuccess) return (int)e1;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_conv_edge_bf16, dim3(n_blocks), dim3(512), CONVB_LDS_BYTES, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
