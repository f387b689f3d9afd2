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
// Row stride of the exchange buffer in floats.  A ds_write_b128 / ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15,
// 20-27}, ...: MI355X_MICROARCH.md section LDS); lane (a, g) touches dword a * XLD + 4 g (+ const).  With 132 (rounds 2 - 4)
// lanes with equal a + g met in a bank: 1.3e5 - 2.1e5 two-way conflicts per launch (SQ_LDS_BANK_CONFLICT); with 136 every group
// of the exchange covers 16 different 4-bank slots: 6.3e4 left (the small reduction buffers).  (Worth nothing measurable next to
// the kernel's memory round trips; changed because it is free.)
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

// One 128x128 GEMM.  On entry `wn` holds (or has in flight) the FIRST K half of this GEMM's weights W; on exit it holds the
// first half of `next` (the matrix of the GEMM that follows; NEXT = false: none).  The compiler barriers pin the fetches
// where they are written: hipcc otherwise hoists every load to the top of the kernel and pays with 50 more registers.
// (The (hi | lo) fp16 image of a matrix has the size and the quarter / half structure of the fp32 one: same fetches.)
template <bool NEXT, bool SKIP, bool F16, typename XT>
__device__ __forceinline__ void gemm16(const float* W, const float* next, WHalf& wn, const XT& XB, f32x4 (&acc)[2], int w, int lane) {
    WHalf cur = wn;
    asm volatile("" ::: "memory");
    load_whalf(W, w, lane, 1, wn);                          // second half: lands during the first half's 32 MFMAs
    asm volatile("" ::: "memory");
    if constexpr (F16) gemm16_half_f16(cur, 0, XB, acc, SKIP); else gemm16_half(cur, 0, XB, acc, SKIP);
    __builtin_amdgcn_sched_barrier(0);
    cur = wn;
    if (NEXT) {
        asm volatile("" ::: "memory");
        load_whalf(next, w, lane, 0, wn);                   // next GEMM's first half: lands during the second half + exchange
        asm volatile("" ::: "memory");
    }
    if constexpr (F16) gemm16_half_f16(cur, 1, XB, acc, SKIP); else gemm16_half(cur, 1, XB, acc, SKIP);
    __builtin_amdgcn_sched_barrier(0);
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

// every wave contributes its 32 features; afterwards every lane holds its share of the full rows in chain16 layout
__device__ __forceinline__ void exchange16(float* xbuf, int w, int a, int g, const f32x4 (&mine)[2], f32x4 (&XB)[8]) {
    __syncthreads();                                        // previous readers are done
    store16(xbuf + a * XLD, w, g, mine);
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) XB[blk] = *reinterpret_cast<const f32x4*>(xbuf + a * XLD + 16 * blk + 4 * g);
}

// the same 2 x 4 features as fp16 (NodeArgs::tab16: 256-byte rows), natural order (hn) ...
typedef _Float16 nf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nf16x4 cvt16x4(const f32x4& v) {
    const nf16x2 lo = __builtin_convertvector(nf32x2{v[0], v[1]}, nf16x2), hi = __builtin_convertvector(nf32x2{v[2], v[3]}, nf16x2);
    return nf16x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void store16_h(_Float16* __restrict__ row, int w, int g, const f32x4 (&v)[2]) {
    *reinterpret_cast<nf16x4*>(row + 32 * w + 4 * g) = cvt16x4(v[0]);
    *reinterpret_cast<nf16x4*>(row + 32 * w + 16 + 4 * g) = cvt16x4(v[1]);
}
// ... and in the group order of gamd_tab16_pos (S, D): features 32 w + 16 o + 4 g + r  ->  16 (2 w + o) + 8 (g & 1) + 4 (g >> 1) + r
__device__ __forceinline__ void store16_tab(_Float16* __restrict__ row, int w, int g, const f32x4 (&v)[2]) {
    *reinterpret_cast<nf16x4*>(row + 32 * w + 8 * (g & 1) + 4 * (g >> 1)) = cvt16x4(v[0]);
    *reinterpret_cast<nf16x4*>(row + 32 * w + 16 + 8 * (g & 1) + 4 * (g >> 1)) = cvt16x4(v[1]);
}

// sum over the four lane groups g of an atom (lanes a, a + 16, a + 32, a + 48)
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// NABL (profiling build only, GAMD_NODE_VARIANT; wrong results): 1 = no piece loads (agg = 0: the bound of letting the conv kernels
// write agg), 2 = every weight fragment from one cache-hot kilobyte (the bound of any better weight prefetch), 4 = no GEMMs
// F16: the five GEMMs in split-fp16 (the reduced-precision edge modes; weights packed by pack16_f16x3)
template <int NABL, bool F16 = false>
__global__ void __launch_bounds__(256, 3) k_node(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) float xbuf[NT * XLD];
    __shared__ float obuf[4][NT][3];
    __shared__ float red[2][4][NT];

    if (a.counters[CNT_OVERFLOW] || a.devflags[DEVFLAG_FROZEN]) return;
    if (a.mode == 0 && a.l0_gate && a.counters[CNT_REBUILD] == 0) return;      // layer-0 tables of the last rebuild still stand

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
#define GEMM16(NEXT, W, NXT) do { if constexpr (F16) gemm16<NEXT, (NABL & 4) != 0, true>(W, NXT, wn, XS, mine, w, lane); \
                                  else gemm16<NEXT, (NABL & 4) != 0, false>(W, NXT, wn, XB, mine, w, lane); } while (0)
#define SPLIT16() do { if constexpr (F16) split16(XB, XS); } while (0)
    f32x4 mine[2];        // this wave's 32 output features
    WHalf wn;             // the weight half that the next 32 MFMAs need (fetched one half ahead)

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
        load_whalf(a.pre.wsp, w, lane, 0, wn);
    } else {
        // ---- post(l-1): aggregate this wave's slice of the atom's pieces, in order -------------------
        const int rp0 = a.row_ptr[atom], dg = a.deg[atom];
        const int na_incl = a.na_excl[atom] + ((dg > 0 && (rp0 % GAMD_CHUNK) != 0) ? 1 : 0);
        const int p0 = rp0 / GAMD_CHUNK + na_incl;
        const int np = dg > 0 ? ((rp0 + dg - 1) / GAMD_CHUNK - rp0 / GAMD_CHUNK + 1) : 0;
        (void)GAMD_CHK_RANGE(a.sticky, (long long)p0 + np, 0, a.piece_cap, GAMD_CHK_NODE_PIECES);
        mine[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine[1] = mine[0];
        f32x4 p_in[2], h_res[2];
        load16(a.P_in + row, w, g, p_in);                          // in flight during the aggregation
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
        asm volatile("" ::: "memory");                            // the weight fetch stays behind the piece loads (registers)
        load_whalf(a.post.wpep, w, lane, 0, wn);                  // in flight during the exchange
        exchange16(xbuf, w, la, g, mine, XB);                     // XB = agg
        SPLIT16();
        NMARK(2);                                                  // 2: exchange 1
        mine[0] = p_in[0]; mine[1] = p_in[1];
        GEMM16(true, a.post.wpep, a.post.wphip);
        NMARK(3);                                                  // 3: GEMM phi_edge
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[o][r] = gamd_silu_hw(mine[o][r]);
        exchange16(xbuf, w, la, g, mine, XB);                     // XB = SiLU(P + phi_edge(agg))
        SPLIT16();
        NMARK(4);                                                  // 4: SiLU + exchange 2
        load16(a.post.bphi, w, g, mine);
        load16(a.h_in + row, w, g, h_res);                        // residual: lands during the GEMM
        GEMM16(true, a.post.wphip, a.mode != 2 ? a.pre.wsp : a.dec_w1p);
        mine[0] += h_res[0]; mine[1] += h_res[1];                 // residual
        if (valid) store16(a.h_out + row, w, g, mine);
        NMARK(5);                                                  // 5: GEMM phi + residual
    }

    if (a.mode != 2) {
        // ---- pre(l): LayerNorm over the row = reductions over the 4 lane groups and the 4 waves ----
        // (use_layer_norm=False checkpoints: eval-mode BatchNorm1d is a per-feature affine map; the host folded the running
        //  statistics into ln_g = w / sqrt(var + eps), ln_b = b - mean * ln_g, and the row statistics are not needed)
        float mean = 0.f, rstd = 1.0f;
        if (!a.norm_bn) {
            float ps = 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) ps += mine[o][r];
            ps = group_sum(ps);
            if (g == 0) red[0][w][la] = ps;
            __syncthreads();
            mean = ((red[0][0][la] + red[0][1][la]) + (red[0][2][la] + red[0][3][la])) * a.ln_inv_width;
            float pv = 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = mine[o][r] - mean; pv += d * d; }
            pv = group_sum(pv);
            if (g == 0) red[1][w][la] = pv;
            __syncthreads();
            // zero-padded features (width < 128) each added mean^2 to the sum of squared deviations: taken out again (n_pad = 0: x - 0)
            const float var = (((red[1][0][la] + red[1][1][la]) + (red[1][2][la] + red[1][3][la])) - a.ln_n_pad * (mean * mean)) * a.ln_inv_width;
            rstd = 1.0f / sqrtf(var + 1e-5f);
        }
        {
            f32x4 gg[2], bb[2];
            load16(a.pre.ln_g, w, g, gg);
            load16(a.pre.ln_b, w, g, bb);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[o][r] = (mine[o][r] - mean) * rstd * gg[o][r] + bb[o][r];
        }
        if (valid && a.tab16) store16_h(reinterpret_cast<_Float16*>(a.hn_out) + row, w, g, mine);
        else if (valid && !a.hn_perm) store16(a.hn_out + row, w, g, mine);
        exchange16(xbuf, w, la, g, mine, XB);                     // XB = hn
        SPLIT16();
        NMARK(6);                                                  // 6: LayerNorm + exchange 3
        if (a.hn_perm) {
            // feature-permuted copy for the row-layout gather of conv_edge_f16x3.hip, written from the assembled rows
            // in the exchange buffer: position 4 c + j holds feature 32 j + c, one coalesced 16-byte store per (atom, c)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = k * 256 + threadIdx.x, at = idx >> 5, c = idx & 31;
                const float* xr = xbuf + at * XLD + c;
                const f32x4 v = {xr[0], xr[32], xr[64], xr[96]};
                const int atom_k = blockIdx.x * NT + at;
                if (atom_k < a.n) *reinterpret_cast<f32x4*>(a.hn_out + (size_t)atom_k * GAMD_H + 4 * c) = v;
            }
        }
        load16(a.pre.bS, w, g, mine);
        GEMM16(true, a.pre.wsp, a.pre.wdp);
        if (valid && a.tab16) store16_tab(reinterpret_cast<_Float16*>(a.S_out) + row, w, g, mine);
        else if (valid) store16(a.S_out + row, w, g, mine);
        NMARK(7);                                                  // 7: GEMM S
        mine[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine[1] = mine[0];
        GEMM16(true, a.pre.wdp, a.pre.wpdp);
        if (valid && a.tab16) store16_tab(reinterpret_cast<_Float16*>(a.D_out) + row, w, g, mine);
        else if (valid) store16(a.D_out + row, w, g, mine);
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
        exchange16(xbuf, w, la, g, mine, XB);                     // XB = h'
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

}  // namespace

int launch_node(const NodeArgs& a, hipStream_t st) {
    const int nb = (a.n + NT - 1) / NT;
#ifdef GAMD_PROFILING
    static int v = -1;
    if (v < 0) { const char* e = getenv("GAMD_NODE_VARIANT"); v = e ? atoi(e) : 0; }
#define NODE_CASE(V) case V: if (a.f16x3) hipLaunchKernelGGL((k_node<V, true>), dim3(nb), dim3(256), 0, st, a); \
                                 else hipLaunchKernelGGL((k_node<V, false>), dim3(nb), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH(); return 0
    switch (v) {
        NODE_CASE(1); NODE_CASE(2); NODE_CASE(3); NODE_CASE(4); NODE_CASE(7);
        default: break;
    }
#undef NODE_CASE
#endif
    if (a.f16x3) hipLaunchKernelGGL((k_node<0, true>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_node<0, false>), dim3(nb), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
