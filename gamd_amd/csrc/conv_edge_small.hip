// conv_edge_small.hip — the conv-layer edge kernel for SMALL edge counts (the reference's own drivers run 258-atom LJ and
// 774-atom water boxes: 200-800 tiles of 32 edges, a fraction of the 1024 SIMDs of the chip).
//
// conv_edge.hip gives every tile to one wave, which then runs the four 128x128 GEMMs of the layer back to back on one
// SIMD: ~4 x 16 400 matrix cycles of pure latency when there is less than one tile per SIMD.  Here one 32-edge tile is
// shared by the four waves of a 256-thread workgroup, each computing one 32-feature output block of every GEMM (64
// MFMAs instead of 256, its 16 KiB weight quarter read straight from L2 one GEMM ahead), with the 128-wide activation
// re-assembled through a 16.5 KiB LDS exchange buffer between the GEMMs — the scheme of node.hip applied to edges.
// Same data, same order of floating-point operations per output element as conv_edge.hip (bias / D first, K in the same
// order, then S, SiLU; same piece sums): the two kernels are bit-identical, so switching between them by size is
// invisible in the results.
#include "gamd_common.h"
#include "gamd_internal.h"

namespace {

constexpr int XLD = GAMD_XLD;

// EHT, HT: edge-embedding / node width in 128-blocks (wide.hip); the weight blocks W1[:, kb] | W2 | W3 | W4[ob, :] are
// contiguous from a.w1p (gamd_finalize_weights lays them out that way for every width).  WIDE selects the operation
// order of wide.hip's phase 2 ((S + D) first, then the GEMM) instead of conv_edge.hip's (D, GEMM, + S).
template <int EHT, int HT, bool WIDE>
__global__ void __launch_bounds__(256) k_conv_edge_small(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    constexpr int H = 128 * HT;
    __shared__ __attribute__((aligned(16))) float xbuf[32 * XLD];
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int quarter = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;

    // feature of (output block ob, this wave's row `slot`) in phase 4: the 128-wide fp32 W4 is packed with its output rows
    // permuted (packed row 32 q + s = feature 4 s + q, what conv_edge.hip's 16-byte hn loads / piece stores want)
    auto feat4 = [&](int ob) { return WIDE ? 128 * ob + 32 * quarter + slot : 4 * slot + quarter; };
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = x < E;
        const int src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC) : 0;
        const int dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST) : 0;
        f32x16 X[4], XE[EHT][4], acc;
        WQuarter wa, wb;
        const float* wblk = a.w1p;                       // block k at wblk + k * GAMD_WFRAG_FLOATS
        load_wquarter(wblk, quarter, lane, wa);
#pragma unroll
        for (int kb = 0; kb < EHT; ++kb) {               // e tile in fragment order = chain layout registers
            const f32x4* ef = (const f32x4*)a.e_frag + ((size_t)tile * EHT + kb) * 16 * 64;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = ef[(t * 4 + q) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) XE[kb][t][q * 4 + j] = v[j];
                }
        }
        // rows this wave needs later: its quarter of S[src], D[dst]; hn[src] of its 16 edges x its 32 features
        const f32x16 s_q = load_slice(a.S + (size_t)src * 128, quarter, half);
        const f32x16 d_q = load_slice(a.D + (size_t)dst * 128, quarter, half);
        f32x16 hn_q[HT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int s = __shfl(src, rho, 64);
#pragma unroll
            for (int ob = 0; ob < HT; ++ob) hn_q[ob][r] = a.hn[(size_t)s * H + feat4(ob)];
        }
        const unsigned mask = a.chunk_mask[tile * 2 + half];
        const int p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE);
        const int x0 = tile * GAMD_TILE + 16 * half;
        int nvalid = E - x0;
        nvalid = nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid);

        // phase 1: T1 = SiLU(W1 e + b1), K = Eh
        acc = load_slice(a.b1, quarter, half);
#pragma unroll
        for (int kb = 0; kb < EHT; ++kb) {
            WQuarter& cur = (kb & 1) ? wb : wa;
            WQuarter& nxt = (kb & 1) ? wa : wb;
            load_wquarter(wblk + (size_t)(kb + 1) * GAMD_WFRAG_FLOATS, quarter, lane, nxt);     // next K block, or W2
            gemm_quarter<false>(cur, XE[kb], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_silu_hw(acc[r]);
        exchange(xbuf, quarter, slot, half, acc, X);
        // block index parity decides which buffer holds what from here on
        constexpr int B2 = EHT;                          // W2, then W3 = B2 + 1, W4[ob] = B2 + 2 + ob
        WQuarter& w2 = (B2 & 1) ? wb : wa;
        WQuarter& w3 = (B2 & 1) ? wa : wb;
        // phase 2: conv_edge.hip: T3 = SiLU((D[dst] + W2 T1) + S[src]);  wide.hip: T3 = SiLU((S + D) + W2 T1)
        if (WIDE) acc = s_q + d_q; else acc = d_q;
        load_wquarter(wblk + (size_t)(B2 + 1) * GAMD_WFRAG_FLOATS, quarter, lane, w3);
        gemm_quarter<false>(w2, X, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_silu_hw(WIDE ? acc[r] : acc[r] + s_q[r]);
        exchange(xbuf, quarter, slot, half, acc, X);
        // phase 3: T4 = SiLU(W3 T3 + b3)
        acc = load_slice(a.b3, quarter, half);
        load_wquarter(wblk + (size_t)(B2 + 2) * GAMD_WFRAG_FLOATS, quarter, lane, w2);
        gemm_quarter<false>(w3, X, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_silu_hw(acc[r]);
        exchange(xbuf, quarter, slot, half, acc, X);
        // phase 4 (F2: lane = feature feat4(ob), register = edge): e_emb, message, segment sum
        const unsigned keep_bits = ~(mask << 1);
        unsigned ends0 = mask;
        if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends0 |= 1u << (nvalid - 1);
#pragma unroll
        for (int ob = 0; ob < HT; ++ob) {
            WQuarter& cur = (ob & 1) ? w3 : w2;
            WQuarter& nxt = (ob & 1) ? w2 : w3;
            if (ob + 1 < HT) load_wquarter(wblk + (size_t)(B2 + 3 + ob) * GAMD_WFRAG_FLOATS, quarter, lane, nxt);
            const float b = a.b4[128 * ob + 32 * quarter + slot];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b;
            gemm_quarter<true>(cur, X, acc);
            if (WIDE && a.emb_out) {                         // update_edge_emb: e_emb rows for launch_edge_update (wide.hip)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.emb_out[(size_t)(x0 + r) * H + feat4(ob)] = acc[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = gamd_msg_acc((r < nvalid) ? hn_q[ob][r] : 0.f, acc[r], (r > 0 && ((keep_bits >> r) & 1u)) ? acc[r - 1] : 0.f);
            }
            unsigned ends = ends0;
            int p = p0;
            while (__any(ends != 0)) {
                if (ends != 0) {
                    const int r = __builtin_ctz(ends);
                    ends &= ends - 1;
                    float v = acc[0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) v = (r == k) ? acc[k] : v;
                    a.partial[(size_t)p * H + feat4(ob)] = v;
                    ++p;
                }
            }
        }
    }
}

}  // namespace

int launch_conv_edge_small(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    hipLaunchKernelGGL((k_conv_edge_small<1, 1, false>), dim3(n_blocks), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_conv_edge_small_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) {
    if (eht == 1 && ht == 1) hipLaunchKernelGGL((k_conv_edge_small<1, 1, true>), dim3(n_blocks), dim3(256), 0, st, a);
    else if (eht == 1 && ht == 2) hipLaunchKernelGGL((k_conv_edge_small<1, 2, true>), dim3(n_blocks), dim3(256), 0, st, a);
    else if (eht == 2 && ht == 1) hipLaunchKernelGGL((k_conv_edge_small<2, 1, true>), dim3(n_blocks), dim3(256), 0, st, a);
    else if (eht == 2 && ht == 2) hipLaunchKernelGGL((k_conv_edge_small<2, 2, true>), dim3(n_blocks), dim3(256), 0, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}
