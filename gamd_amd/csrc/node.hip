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
// N is small (10^4 rows): the kernel is latency-bound, not throughput-bound, so a 32-atom tile is
// split over the 4 waves of a workgroup by OUTPUT feature quarter: wave w computes features
// [32w, 32w+32) of every GEMM (64 MFMAs instead of 256, its 16 KiB weight quarter read straight from
// L2), and the full 128-wide activation row is re-assembled through a 16.5 KiB LDS exchange buffer
// between chained GEMMs.  Activations are in the chain layout of gamd_common.h throughout.
#include "gamd_common.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

constexpr int XLD = GAMD_XLD;                  // padded row stride of the exchange buffer (floats)

__global__ void __launch_bounds__(256) k_node(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) float xbuf[32 * XLD];
    __shared__ float obuf[4][32][3];
    __shared__ float red[2][4][32];

    if (a.counters[CNT_OVERFLOW] || a.devflags[DEVFLAG_FROZEN]) return;

    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int quarter = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int atom_raw = blockIdx.x * GAMD_TILE + slot;
    const bool valid = atom_raw < a.n;
    const int atom = valid ? atom_raw : a.n - 1;
    const size_t row = (size_t)atom * GAMD_H;

    f32x16 X[4];          // full activation row block (chain layout)
    f32x16 mine;          // this wave's output quarter
    WQuarter wa, wb;      // double-buffered weight quarters

    if (a.mode == 0) {
        if (a.node_emb) {
            mine = load_slice(a.node_emb, quarter, half);
        } else {
            const float f = a.pos_s[atom].w;                       // species feature (O=1, H=0)
            const f32x16 w = load_slice(a.enc_w, quarter, half);
            mine = load_slice(a.enc_b, quarter, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = f * w[r] + mine[r];
        }
        if (valid) store_slice(a.h_out + row, quarter, half, mine);
        load_wquarter(a.pre.wsp, quarter, lane, wa);
    } else {
        // ---- post(l-1): aggregate this quarter's slice of the pieces, in order ------------------
        const int rp0 = a.row_ptr[atom], dg = a.deg[atom];
        const int na_incl = a.na_excl[atom] + ((dg > 0 && (rp0 % GAMD_CHUNK) != 0) ? 1 : 0);
        const int p0 = rp0 / GAMD_CHUNK + na_incl;
        const int np = dg > 0 ? ((rp0 + dg - 1) / GAMD_CHUNK - rp0 / GAMD_CHUNK + 1) : 0;
        // pieces are fetched in batches of 8 (one memory round trip for the usual 4-6 pieces per atom) and
        // summed in piece order
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r] = 0.f;
        const f32x16 p_in = load_slice(a.P_in + row, quarter, half);     // also in flight now
        const f32x16 h_res = load_slice(a.h_in + row, quarter, half);
        for (int k0 = 0; __any(k0 < np); k0 += 8) {
            f32x16 pc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = (k0 + k < np) ? k0 + k : (np > 0 ? np - 1 : 0);
                pc[k] = load_slice(a.partial + (size_t)(np > 0 ? p0 + kk : 0) * GAMD_H, quarter, half);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < np) mine += pc[k];
        }
        load_wquarter(a.post.wpep, quarter, lane, wa);            // in flight during the exchange
        exchange(xbuf, quarter, slot, half, mine, X);            // X = agg
        mine = p_in;
        load_wquarter(a.post.wphip, quarter, lane, wb);           // next GEMM's weights behind this one
        gemm_quarter(wa, X, mine);
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r] = gamd_silu_hw(mine[r]);
        exchange(xbuf, quarter, slot, half, mine, X);            // X = SiLU(P + phi_edge(agg))
        mine = load_slice(a.post.bphi, quarter, half);
        if (a.mode != 2) load_wquarter(a.pre.wsp, quarter, lane, wa); else load_wquarter(a.dec_w1p, quarter, lane, wa);
        gemm_quarter(wb, X, mine);
        mine += h_res;                                            // residual
        if (valid) store_slice(a.h_out + row, quarter, half, mine);
    }

    if (a.mode != 2) {
        // ---- pre(l): LayerNorm over the row = two cross-wave reductions of per-atom partial sums ----
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += mine[r];
        ps = gamd_xhalf_sum(ps);
        if (half == 0) red[0][quarter][slot] = ps;
        __syncthreads();
        const float mean = ((red[0][0][slot] + red[0][1][slot]) + (red[0][2][slot] + red[0][3][slot])) * (1.0f / 128.0f);
        float pv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = mine[r] - mean; pv += d * d; }
        pv = gamd_xhalf_sum(pv);
        if (half == 0) red[1][quarter][slot] = pv;
        __syncthreads();
        const float var = ((red[1][0][slot] + red[1][1][slot]) + (red[1][2][slot] + red[1][3][slot])) * (1.0f / 128.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        {
            const f32x16 g = load_slice(a.pre.ln_g, quarter, half), b = load_slice(a.pre.ln_b, quarter, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = (mine[r] - mean) * rstd * g[r] + b[r];
        }
        if (valid && !a.hn_perm) store_slice(a.hn_out + row, quarter, half, mine);
        exchange(xbuf, quarter, slot, half, mine, X);            // X = hn
        if (a.hn_perm) {
            // feature-permuted copy for the row-layout gather of conv_edge_f16x3.hip, written from the assembled rows
            // in the exchange buffer: position 4 c + j holds feature 32 j + c, one coalesced 16-byte store per (atom, c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = k * 256 + threadIdx.x, at = idx >> 5, c = idx & 31;
                const float* xr = xbuf + at * XLD + c;
                const f32x4 v = {xr[0], xr[32], xr[64], xr[96]};
                const int atom_k = blockIdx.x * GAMD_TILE + at;
                if (atom_k < a.n) *reinterpret_cast<f32x4*>(a.hn_out + (size_t)atom_k * GAMD_H + 4 * c) = v;
            }
        }
        mine = load_slice(a.pre.bS, quarter, half);
        load_wquarter(a.pre.wdp, quarter, lane, wb);
        gemm_quarter(wa, X, mine);
        if (valid) store_slice(a.S_out + row, quarter, half, mine);
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r] = 0.f;
        load_wquarter(a.pre.wpdp, quarter, lane, wa);
        gemm_quarter(wb, X, mine);
        if (valid) store_slice(a.D_out + row, quarter, half, mine);
        mine = load_slice(a.pre.bP, quarter, half);
        gemm_quarter(wa, X, mine);
        if (valid) store_slice(a.P_out + row, quarter, half, mine);
    } else {
        // ---- decoder: Lin(128,128) GELU Lin(128,3); denormalise -------------------------------
        exchange(xbuf, quarter, slot, half, mine, X);            // X = h'
        mine = load_slice(a.dec_b1, quarter, half);
        gemm_quarter(wa, X, mine);
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * quarter + 8 * q + 4 * half;
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = gamd_gelu_hw(mine[q * 4 + j]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.dec_w2 + c * GAMD_H + f0);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[c] += w[j] * g[j];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = gamd_xhalf_sum(o[c]);
        if (half == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) obuf[quarter][slot][c] = o[c];
        }
        __syncthreads();
        if (quarter == 0 && half == 0 && valid) {
            const int orig = a.perm[atom];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ((obuf[0][slot][c] + obuf[1][slot][c]) + (obuf[2][slot][c] + obuf[3][slot][c])) + a.dec_b2[c];
                a.forces_norm[3 * (size_t)orig + c] = v;
                if (a.forces) a.forces[3 * (size_t)orig + c] = v * a.scale + a.shift;
                if (!(fabsf(v) <= 3.0e38f)) a.sticky[STICKY_NONFINITE] = 1;      // NaN or inf
            }
        }
    }
}

}  // namespace

int launch_node(const NodeArgs& a0, hipStream_t st) {
    const NodeArgs& a = a0;
    const int nb = (a.n + GAMD_TILE - 1) / GAMD_TILE;
    hipLaunchKernelGGL(k_node, dim3(nb), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
