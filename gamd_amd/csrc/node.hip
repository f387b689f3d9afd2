// node.hip — the node side of the conv stack, one kernel per layer boundary.
//
// mode 0:  h0 = node_emb.repeat(N) | node_encoder(feat)        (nn_module.py:681 | :554)
//          then pre(0)
// mode 1:  post(l-1) then pre(l)
// mode 2:  post(L-1) then graph_decoder + denormalise           (nn_module.py:684, train_network_lj.py:128-131)
//
//   pre(l):  hn = LayerNorm_l(h)                                (nn_module.py:202)
//            S  = src_affine(hn) + b_src + b_dst + b_edge_affine.2   (hoisted from E rows, :136)
//            D  = dst_affine(hn)                                (hoisted, :137)
//            P  = phi_dst(hn) + b_phi_dst + b_phi_edge          (:147)
//   post(l): agg = sum of this atom's partial-sum pieces, in order   (:142)
//            h' = phi(P + phi_edge(agg)) + h                    (:147, :202 residual)
//
// N is small (258 ... 10^4 rows): the kernel is bound by the latency of five chained 128x128 GEMMs, not by throughput.
// Work unit: a tile of 16 atoms per 256-thread workgroup, on v_mfma_f32_16x16x4_f32.  Wave w computes output features
// [32 w, 32 w + 32) of every GEMM (two 16-feature row blocks = two independent accumulators, 64 MFMAs of 32 cycles =
// 2 048 matrix cycles per GEMM: half of what a 32-atom tile on the 32x32x2 form needs, because a dependent chain there
// cannot go below 64 MFMAs x 64 cycles however few atoms the tile holds) from its 16 KiB weight quarter, fetched from L2
// in one batch one GEMM ahead; the 128-wide rows are re-assembled through an 8 KiB LDS exchange buffer between GEMMs.
// Twice as many, half as long workgroups also spread better: 10 000 atoms = 625 tiles over 256 CUs (<= 3 per CU, 1.5
// old-tile times) instead of 313 (2 per CU on 57 CUs, 2 old-tile times).
//
// "chain16" register layout of a 16-atom x 128-feature block: lane (a = lane & 15 atom, g = lane >> 4),
//     XB[blk][r]  <->  atom a, feature 16 blk + 4 g + r          (blk 0..7, r 0..3: 8 float4 per lane)
// which is at once the C/D layout of Y^T = W X^T on 16x16x4 (D row = 4 g + r of the 16-feature block, column = atom) and
// the B operand of the next GEMM with the K index taken in the order (blk, r, g); the weights are packed to match
// (pack16 in gamd_api.hip):  Wp[((ob * 8 + blk) * 64 + lane)][r] = W[16 ob + (lane & 15)][16 blk + 4 (lane >> 4) + r].
#include "gamd_common.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

typedef _Float16 nf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 nf16x2 __attribute__((ext_vector_type(2)));
typedef float nf32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 16;                         // atoms per tile
// Row stride of the exchange buffers in floats.  A ds_write_b128 / ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, ..., MI355X_MICROARCH.md section LDS); lane (a, g) touches dword a * XLD + 4 g (+ const): with 132
// (round 2 - 4) lanes with equal a + g met in a bank (2.1e5 conflicts per launch, rocprofv3 SQ_LDS_BANK_CONFLICT), with 136
// every group covers 16 different 4-bank slots.
constexpr int XLD = 136;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// A wave's 16 KiB weight quarter is consumed as two K halves of 8 float4 per lane, h.w[o * 4 + b] = block (ob = 2 w + o,
// blk = 4 half + b), and fetched from L2 one HALF ahead of the MFMAs that use it (the first half of a GEMM during the
// exchange in front of it, the second during its first 32 MFMAs): 64 live weight registers instead of the 128 of a
// whole-quarter double buffer.  That keeps the kernel at <= 168 VGPRs = three workgroups per CU, so the 625 tiles of a
// 10 000-atom box are all resident and their latency phases overlap.
struct WHalf { f32x4 w[8]; };
__device__ __forceinline__ void load_whalf(const float* __restrict__ Wp, int w, int lane, int half, WHalf& h) {
    const f32x4* W = reinterpret_cast<const f32x4*>(Wp) + (size_t)w * 16 * 64 + lane;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int b = 0; b < 4; ++b) h.w[o * 4 + b] = W[(o * 8 + 4 * half + b) * 64];
}

// acc[o] (features 16 (2 w + o) + 4 g + r of atom a) += W[quarter w][:, K half] * X^T; the two row blocks alternate so that
// the two dependent accumulator chains (40-cycle latency, 32-cycle issue) keep the pipe full
__device__ __forceinline__ void gemm16_half(const WHalf& h, int half, const f32x4 (&XB)[8], f32x4 (&acc)[2], bool skip = false) {
    if (skip) { asm volatile("" ::"v"(h.w[0]), "v"(h.w[7]), "v"(XB[0])); return; }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(h.w[b][r], XB[4 * half + b][r], acc[0]);
            acc[1] = mfma16(h.w[4 + b][r], XB[4 * half + b][r], acc[1]);
        }
}

// ---- split-fp16 form of the same GEMM (reduced-precision edge modes: the node side keeps fp32-grade arithmetic at 3/16 of
// the fp32 matrix time; gamd_f16x3.h has the error analysis) on v_mfma_f32_16x16x32_f16.  K step m covers features
// 32 m .. 32 m + 31; a lane's 8 operand values of it are its chain16 quads XB[2m], XB[2m + 1] (K position 8 g + j <-> feature
// 32 m + 16 (j >> 2) + 4 g + (j & 3)): the C/D layout of one GEMM is still the B operand of the next without a shuffle, and the
// weights are packed to match (pack16_f16x3 in gamd_api.hip): WHalf h.w[o * 4 + mm * 2 + part] = (hi | lo) fragment of row
// block 2 w + o, K step 2 half + mm.  24 MFMAs of 16 cycles per GEMM and wave instead of 64 of 32.
struct XSplit { nf16x8 h[4], l[4]; };
__device__ __forceinline__ void split16(const f32x4 (&XB)[8], XSplit& s) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x4& q = XB[2 * m + (j >> 2)];
            const nf32x2 x = {q[j & 3], q[(j & 3) + 1]};
            const nf16x2 hi = __builtin_convertvector(x, nf16x2);
            const nf32x2 rem = x - __builtin_convertvector(hi, nf32x2);
            const nf16x2 lo = __builtin_convertvector(rem, nf16x2);
            s.h[m][j] = hi[0]; s.h[m][j + 1] = hi[1];
            s.l[m][j] = lo[0]; s.l[m][j + 1] = lo[1];
        }
}
__device__ __forceinline__ void gemm16_half_f16(const WHalf& h, int half, const XSplit& X, f32x4 (&acc)[2], bool skip = false) {
    if (skip) { asm volatile("" ::"v"(h.w[0]), "v"(h.w[7]), "v"(X.h[0])); return; }
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * half + mm;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const nf16x8 wh = __builtin_bit_cast(nf16x8, h.w[o * 4 + mm * 2]), wl = __builtin_bit_cast(nf16x8, h.w[o * 4 + mm * 2 + 1]);
            acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.l[m], acc[o], 0, 0, 0);
            acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, X.h[m], acc[o], 0, 0, 0);
            acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.h[m], acc[o], 0, 0, 0);
        }
    }
}

// One 128x128 GEMM.  DEEP = false (more than two tiles per CU: the fp32 headline size): on entry `wn.h[0]` holds (or has in
// flight) the FIRST K half of this GEMM's weights W; on exit it holds the first half of `next` (the matrix of the GEMM that
// follows; NEXT = false: none) — a half's 32 fp32 MFMAs (1 024 matrix cycles) cover the L2 round trip of the next half, and
// 64 live weight registers keep three workgroups on a CU.  DEEP = true (at most two tiles per CU: every water / small-system
// configuration, and all split-fp16 node GEMMs, whose halves are 12 MFMAs = 200 cycles and hide nothing): `wn` holds BOTH halves
// of W on entry and both halves of `next` on exit, fetched a whole GEMM + exchange ahead (128 live weight registers, two
// workgroups per CU).  The compiler barriers pin the fetches where they are written: hipcc otherwise hoists every load to the
// top of the kernel.  (The (hi | lo) fp16 image of a matrix has the size and the quarter / half structure of the fp32 one.)
struct WFull { WHalf h[2]; };
template <bool DEEP>
__device__ __forceinline__ void load_wfirst(const float* W, int w, int lane, WFull& wn) {
    load_whalf(W, w, lane, 0, wn.h[0]);
    if (DEEP) load_whalf(W, w, lane, 1, wn.h[1]);
}
template <bool NEXT, bool SKIP, bool F16, bool DEEP, typename XT>
__device__ __forceinline__ void gemm16(const float* W, const float* next, WFull& wn, const XT& XB, f32x4 (&acc)[2], int w, int lane) {
    if constexpr (DEEP) {
        const WFull cur = wn;
        if (NEXT) {
            asm volatile("" ::: "memory");
            load_whalf(next, w, lane, 0, wn.h[0]);
            load_whalf(next, w, lane, 1, wn.h[1]);
            asm volatile("" ::: "memory");
        }
        if constexpr (F16) gemm16_half_f16(cur.h[0], 0, XB, acc, SKIP); else gemm16_half(cur.h[0], 0, XB, acc, SKIP);
        if constexpr (F16) gemm16_half_f16(cur.h[1], 1, XB, acc, SKIP); else gemm16_half(cur.h[1], 1, XB, acc, SKIP);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        WHalf cur = wn.h[0];
        asm volatile("" ::: "memory");
        load_whalf(W, w, lane, 1, wn.h[0]);                     // second half: lands during the first half's 32 MFMAs
        asm volatile("" ::: "memory");
        if constexpr (F16) gemm16_half_f16(cur, 0, XB, acc, SKIP); else gemm16_half(cur, 0, XB, acc, SKIP);
        __builtin_amdgcn_sched_barrier(0);
        cur = wn.h[0];
        if (NEXT) {
            asm volatile("" ::: "memory");
            load_whalf(next, w, lane, 0, wn.h[0]);              // next GEMM's first half: lands during the second half + exchange
            asm volatile("" ::: "memory");
        }
        if constexpr (F16) gemm16_half_f16(cur, 1, XB, acc, SKIP); else gemm16_half(cur, 1, XB, acc, SKIP);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// this lane's 2 x 4 floats of a plain row-major [128] row: features 16 (2 w + o) + 4 g + 0..3
__device__ __forceinline__ void load16(const float* __restrict__ row, int w, int g, f32x4 (&v)[2]) {
    v[0] = *reinterpret_cast<const f32x4*>(row + 32 * w + 4 * g);
    v[1] = *reinterpret_cast<const f32x4*>(row + 32 * w + 16 + 4 * g);
}
__device__ __forceinline__ void store16(float* __restrict__ row, int w, int g, const f32x4 (&v)[2]) {
    *reinterpret_cast<f32x4*>(row + 32 * w + 4 * g) = v[0];
    *reinterpret_cast<f32x4*>(row + 32 * w + 16 + 4 * g) = v[1];
}

// every wave contributes its 32 features; afterwards every lane holds its share of the full rows in chain16 layout.  ONE
// workgroup barrier: consecutive exchanges alternate between two buffers, and a wave can only arrive at exchange k + 2 (which
// overwrites the buffer of exchange k) through the barrier of exchange k + 1, which every wave passes after it has read k.
__device__ __forceinline__ void exchange16(float* xbuf, int w, int a, int g, const f32x4 (&mine)[2], f32x4 (&XB)[8]) {
    store16(xbuf + a * XLD, w, g, mine);
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) XB[blk] = *reinterpret_cast<const f32x4*>(xbuf + a * XLD + 16 * blk + 4 * g);
}

// sum over the four lane groups g of an atom (lanes a, a + 16, a + 32, a + 48)
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// NABL (profiling build only, GAMD_NODE_VARIANT; wrong results): 1 = no piece loads (agg = 0: the bound of letting the conv kernels
// write agg), 2 = every weight fragment from one cache-hot kilobyte (the bound of any better weight prefetch), 4 = no GEMMs
// F16: the five GEMMs in split-fp16 (the reduced-precision edge modes; weights packed by pack16_f16x3)
// DEEP: whole-GEMM-ahead weight prefetch, two workgroups per CU (launch_node picks it when every tile is resident that way)
template <int NABL, bool F16, bool DEEP>
__global__ void __launch_bounds__(256, DEEP ? 2 : 3) k_node(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) float xbuf[2][NT * XLD];
    __shared__ __attribute__((aligned(16))) float s_ln[2][GAMD_H];   // LayerNorm weight / bias of pre(l): every lane needs all 128
    __shared__ float obuf[4][NT][3];

    if (a.counters[CNT_OVERFLOW] || a.devflags[DEVFLAG_FROZEN]) return;

    const int lane = threadIdx.x & 63, la = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int atom_raw = blockIdx.x * NT + la;
    const bool valid = atom_raw < a.n;
    const int atom = valid ? atom_raw : a.n - 1;
    const size_t row = (size_t)atom * GAMD_H;

#ifdef GAMD_PROFILING
    long long tm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // compile-time mark indices: tm[] stays in registers (a run-time index sends it to scratch or movrel and skews the marks)
#define NMARK(I) do { if (a.tdbg) tm[I] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NMARK(I) do { } while (0)
#endif
    NMARK(0);                                                      // 0: start
    if (NABL & 2) {       // one hot kilobyte for every weight fragment
        const float* hot = a.pre.wsp;
        a.post.wpep = a.post.wphip = a.pre.wsp = a.pre.wdp = a.pre.wpdp = a.dec_w1p = hot;
    }
    f32x4 XB[8];          // full activation rows (chain16 layout)
    XSplit XS;            // F16: their (hi, lo) fp16 operand images
#define GEMM16(NEXT, W, NXT) do { if constexpr (F16) gemm16<NEXT, (NABL & 4) != 0, true, DEEP>(W, NXT, wn, XS, mine, w, lane); \
                                  else gemm16<NEXT, (NABL & 4) != 0, false, DEEP>(W, NXT, wn, XB, mine, w, lane); } while (0)
#define SPLIT16() do { if constexpr (F16) split16(XB, XS); } while (0)
    f32x4 mine[2];        // this wave's 32 output features
    WFull wn;             // the weights the next MFMAs need (DEEP: a whole GEMM ahead; otherwise one K half ahead in wn.h[0])
    int xb = 0;           // exchange buffer of the next exchange (they alternate)
#define EXCHANGE() do { exchange16(xbuf[xb], w, la, g, mine, XB); xb ^= 1; } while (0)

    if (a.mode != 2 && threadIdx.x < 64) {
        // pre(l)'s LayerNorm parameters -> LDS (visible behind the first exchange barrier): after the exchange every lane
        // normalises its share of the FULL rows, i.e. needs the parameters of 32 features of all four waves' slices
        const float* src = threadIdx.x < 32 ? a.pre.ln_g : a.pre.ln_b;
        *reinterpret_cast<f32x4*>(&s_ln[threadIdx.x >> 5][4 * (threadIdx.x & 31)]) = *reinterpret_cast<const f32x4*>(src + 4 * (threadIdx.x & 31));
    }

    if (a.mode == 0) {
        if (a.node_emb) {
            load16(a.node_emb, w, g, mine);
        } else {
            const float f = a.pos_s[atom].w;                       // node feature (O = 1, H = 0, or the caller's float)
            f32x4 ww[2];
            load16(a.enc_w, w, g, ww);
            load16(a.enc_b, w, g, mine);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[o][r] = f * ww[o][r] + mine[o][r];
        }
        if (valid) store16(a.h_out + row, w, g, mine);
        load_wfirst<DEEP>(a.pre.wsp, w, lane, wn);
    } else {
        // ---- post(l-1): aggregate this wave's slice of the atom's pieces, in order -------------------
        // the three index loads lead: the piece loads depend on them, everything else (P, the residual row, the weights of the
        // first GEMM) is independent and is issued behind them
        const int rp0 = a.row_ptr[atom], dg = a.deg[atom], nax = a.na_excl[atom];
        f32x4 p_in[2], h_res[2];
        load16(a.P_in + row, w, g, p_in);
        load16(a.h_in + row, w, g, h_res);                        // residual: needed after the second GEMM
        asm volatile("" ::: "memory");
        if (DEEP) load_wfirst<true>(a.post.wpep, w, lane, wn);    // lands during the two round trips of the aggregation
        asm volatile("" ::: "memory");
        const int na_incl = nax + ((dg > 0 && (rp0 % GAMD_CHUNK) != 0) ? 1 : 0);
        const int p0 = rp0 / GAMD_CHUNK + na_incl;
        const int np = dg > 0 ? ((rp0 + dg - 1) / GAMD_CHUNK - rp0 / GAMD_CHUNK + 1) : 0;
        mine[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine[1] = mine[0];
        // pieces are fetched in batches of 8 (one memory round trip for the usual 4-6 pieces per atom), summed in piece order
        for (int k0 = 0; !(NABL & 1) && __any(k0 < np); k0 += 8) {
            f32x4 pc[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = (k0 + k < np) ? k0 + k : (np > 0 ? np - 1 : 0);
                load16(a.partial + (size_t)(np > 0 ? p0 + kk : 0) * GAMD_H, w, g, pc[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < np) { mine[0] += pc[k][0]; mine[1] += pc[k][1]; }
        }
        NMARK(1);                                                  // 1: pieces summed
        if (!DEEP) {
            asm volatile("" ::: "memory");                        // the weight fetch stays behind the piece loads (registers)
            load_wfirst<false>(a.post.wpep, w, lane, wn);         // in flight during the exchange
        }
        EXCHANGE();                                               // XB = agg
        SPLIT16();
        NMARK(2);                                                  // 2: exchange 1
        mine[0] = p_in[0]; mine[1] = p_in[1];
        GEMM16(true, a.post.wpep, a.post.wphip);
        NMARK(3);                                                  // 3: GEMM phi_edge
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[o][r] = gamd_silu_hw(mine[o][r]);
        EXCHANGE();                                               // XB = SiLU(P + phi_edge(agg))
        SPLIT16();
        NMARK(4);                                                  // 4: SiLU + exchange 2
        load16(a.post.bphi, w, g, mine);
        GEMM16(true, a.post.wphip, a.mode != 2 ? a.pre.wsp : a.dec_w1p);
        mine[0] += h_res[0]; mine[1] += h_res[1];                 // residual
        if (valid) store16(a.h_out + row, w, g, mine);
        NMARK(5);                                                  // 5: GEMM phi + residual
    }

    EXCHANGE();                                                   // XB = h (mode 0) / h' (modes 1, 2): the rows the next stage reads
    if (a.mode != 2) {
        // ---- pre(l): LayerNorm of the assembled rows, in registers ------------------------------------
        // Every lane holds 32 of its atom's 128 features (the four lane groups g of an atom the four quarters), so the row
        // statistics are two shuffle reductions and no workgroup barrier (rounds 2-4 normalised each wave's 32-feature slice
        // before the exchange: two LDS reductions with a barrier each).  Two-pass form, as torch's.
        // (use_layer_norm=False checkpoints: eval-mode BatchNorm1d is a per-feature affine map; the host folded the running
        //  statistics into ln_g = w / sqrt(var + eps), ln_b = b - mean * ln_g, and the row statistics are not needed)
        float mean = 0.f, rstd = 1.0f;
        if (!a.norm_bn) {
            float ps = 0.f;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) ps += XB[blk][r];
            mean = group_sum(ps) * a.ln_inv_width;
            float pv = 0.f;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = XB[blk][r] - mean; pv += d * d; }
            // zero-padded features (width < 128) each added mean^2 to the sum of squared deviations: taken out again (n_pad = 0: x - 0)
            const float var = (group_sum(pv) - a.ln_n_pad * (mean * mean)) * a.ln_inv_width;
            rstd = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(&s_ln[0][16 * blk + 4 * g]);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&s_ln[1][16 * blk + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r) XB[blk][r] = (XB[blk][r] - mean) * rstd * gg[r] + bb[r];
        }
        // hn: every wave stores its own 32-feature slice (blocks 2 w, 2 w + 1 of the rows it now holds)
        if (valid && !a.hn_perm) {
            f32x4 hs[2];
#pragma unroll
            for (int blk = 0; blk < 8; ++blk)
                if (blk >> 1 == w) hs[blk & 1] = XB[blk];
            store16(a.hn_out + row, w, g, hs);
        }
        if (a.hn_perm && valid) {
            // feature-permuted copy for the row-layout gather of conv_edge_f16x3.hip: position 4 c + j holds feature 32 j + c.
            // Feature 16 blk + 4 g + r = 32 j + c with blk = 2 j + hi, c = 16 hi + 4 g + r: the four j of a (hi, r) sit in ONE
            // lane -> one 16-byte store per (hi, r); wave w writes hi = w >> 1, r = 2 (w & 1), 2 (w & 1) + 1
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (hi == (w >> 1) && (r >> 1) == (w & 1)) {
                        const f32x4 v = {XB[hi][r], XB[2 + hi][r], XB[4 + hi][r], XB[6 + hi][r]};
                        *reinterpret_cast<f32x4*>(a.hn_out + row + 4 * (16 * hi + 4 * g + r)) = v;
                    }
        }
        SPLIT16();
        NMARK(6);                                                  // 6: exchange 3 + LayerNorm
        load16(a.pre.bS, w, g, mine);
        GEMM16(true, a.pre.wsp, a.pre.wdp);
        if (valid) store16(a.S_out + row, w, g, mine);
        NMARK(7);                                                  // 7: GEMM S
        mine[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine[1] = mine[0];
        GEMM16(true, a.pre.wdp, a.pre.wpdp);
        if (valid) store16(a.D_out + row, w, g, mine);
        NMARK(8);                                                  // 8: GEMM D
        load16(a.pre.bP, w, g, mine);
        GEMM16(false, a.pre.wpdp, nullptr);
        if (valid) store16(a.P_out + row, w, g, mine);
        NMARK(9);                                                  // 9: GEMM P
#ifdef GAMD_PROFILING
        if (a.tdbg && a.mode == 1 && lane == 0 && blockIdx.x < 512) {
            long long* o = a.tdbg + ((size_t)blockIdx.x * 4 + w) * 16;
            for (int i = 0; i < 16; ++i) o[i] = tm[i];
        }
#endif
    } else {
        // ---- decoder: Lin(128,128) GELU Lin(128,3); denormalise -------------------------------
        SPLIT16();
        load16(a.dec_b1, w, g, mine);
        GEMM16(false, a.dec_w1p, nullptr);
        float o3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int f0 = 32 * w + 16 * o + 4 * g;
            f32x4 gl;
#pragma unroll
            for (int r = 0; r < 4; ++r) gl[r] = gamd_gelu_hw(mine[o][r]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(a.dec_w2 + c * GAMD_H + f0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o3[c] += ww[r] * gl[r];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o3[c] = group_sum(o3[c]);
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) obuf[w][la][c] = o3[c];
        }
        __syncthreads();
        if (w == 0 && g == 0 && valid) {
            const int orig = a.perm[atom];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ((obuf[0][la][c] + obuf[1][la][c]) + (obuf[2][la][c] + obuf[3][la][c])) + a.dec_b2[c];
                a.forces_norm[3 * (size_t)orig + c] = v;
                if (a.forces) a.forces[3 * (size_t)orig + c] = v * a.scale + a.shift;
                if (!(fabsf(v) <= 3.0e38f)) a.sticky[STICKY_NONFINITE] = 1;      // NaN or inf
            }
        }
    }
}

#undef NMARK
#undef GEMM16
#undef SPLIT16
#undef EXCHANGE

}  // namespace

// Two workgroups per CU hold every tile of up to 2 x 256 x ... tiles at once: then the whole-GEMM-ahead prefetch (128 weight
// registers) costs no occupancy that matters.  More tiles (the 10 000-atom headline: 625) keep the lighter kernel, three per CU.
template <int NABL>
static int launch_node_variant(const NodeArgs& a, int nb, int n_cu, hipStream_t st) {
    const bool deep = nb <= 2 * n_cu;
    if (a.f16x3) {
        if (deep) hipLaunchKernelGGL((k_node<NABL, true, true>), dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_node<NABL, true, false>), dim3(nb), dim3(256), 0, st, a);
    } else {
        if (deep) hipLaunchKernelGGL((k_node<NABL, false, true>), dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_node<NABL, false, false>), dim3(nb), dim3(256), 0, st, a);
    }
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_node(const NodeArgs& a, hipStream_t st) {
    const int nb = (a.n + NT - 1) / NT;
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
#ifdef GAMD_PROFILING
    static int v = -1;
    if (v < 0) { const char* e = getenv("GAMD_NODE_VARIANT"); v = e ? atoi(e) : 0; }
    switch (v) {
        case 1: return launch_node_variant<1>(a, nb, n_cu, st);
        case 2: return launch_node_variant<2>(a, nb, n_cu, st);
        case 3: return launch_node_variant<3>(a, nb, n_cu, st);
        case 4: return launch_node_variant<4>(a, nb, n_cu, st);
        case 7: return launch_node_variant<7>(a, nb, n_cu, st);
        default: break;
    }
#endif
    return launch_node_variant<0>(a, nb, n_cu, st);
}
