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
// One wave per 32-atom tile, activations chained through registers (gamd_common.h); the packed
// weight fragments are read straight from L2 (no LDS: nothing is shared between waves here).
#include "gamd_common.h"
#include "gamd_internal.h"

namespace {

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

__global__ void __launch_bounds__(64) k_node(NodeArgs a) {
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int atom_raw = blockIdx.x * GAMD_TILE + slot;
    const bool valid = atom_raw < a.n;
    const int atom = valid ? atom_raw : a.n - 1;
    const size_t row = (size_t)atom * GAMD_H;

    f32x16 X[4], acc[4], hres[4];

    if (a.mode == 0) {
        if (a.node_emb) {
            load_bias_chain(a.node_emb, half, hres);
        } else {
            const float f = a.pos_s[atom].w;                       // species feature (O=1, H=0)
            f32x16 w[4];
            load_bias_chain(a.enc_w, half, w);
            load_bias_chain(a.enc_b, half, hres);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) hres[t][r] = f * w[t][r] + hres[t][r];
        }
    } else {
        // ---- post(l-1): aggregate pieces -------------------------------------------------------
        const int rp0 = a.row_ptr[atom], dg = a.deg[atom];
        const int na_incl = a.na_excl[atom] + ((dg > 0 && (rp0 % GAMD_CHUNK) != 0) ? 1 : 0);
        const int p0 = rp0 / GAMD_CHUNK + na_incl;
        const int np = dg > 0 ? ((rp0 + dg - 1) / GAMD_CHUNK - rp0 / GAMD_CHUNK + 1) : 0;
        zero_acc(X);
        for (int k = 0; __any(k < np); ++k) {
            if (k < np) {
                const float* pr = a.partial + (size_t)(p0 + k) * GAMD_H;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pr + 32 * t + 8 * q + 4 * half);
#pragma unroll
                        for (int j = 0; j < 4; ++j) X[t][q * 4 + j] += v[j];
                    }
            }
        }
        load_row_chain(a.P_in + row, half, acc);
        gemm128<false>((const f32x4*)a.post.wpep, lane, X, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[t][r] = gamd_silu(acc[t][r]);
        load_bias_chain(a.post.bphi, half, acc);
        gemm128<false>((const f32x4*)a.post.wphip, lane, X, acc);
        load_row_chain(a.h_in + row, half, hres);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) hres[t][r] = acc[t][r] + hres[t][r];
    }
    if (valid) store_row_chain(a.h_out + row, half, hres);

    if (a.mode != 2) {
        // ---- pre(l) ----------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 4; ++t) X[t] = hres[t];
        layernorm_chain(X, a.pre.ln_g, a.pre.ln_b, half, 1e-5f);
        if (valid) store_row_chain(a.hn_out + row, half, X);
        load_bias_chain(a.pre.bS, half, acc);
        gemm128<false>((const f32x4*)a.pre.wsp, lane, X, acc);
        if (valid) store_row_chain(a.S_out + row, half, acc);
        zero_acc(acc);
        gemm128<false>((const f32x4*)a.pre.wdp, lane, X, acc);
        if (valid) store_row_chain(a.D_out + row, half, acc);
        load_bias_chain(a.pre.bP, half, acc);
        gemm128<false>((const f32x4*)a.pre.wpdp, lane, X, acc);
        if (valid) store_row_chain(a.P_out + row, half, acc);
    } else {
        // ---- decoder: Lin(128,128) GELU Lin(128,3); denormalise -------------------------------
        load_bias_chain(a.dec_b1, half, acc);
        gemm128<false>((const f32x4*)a.dec_w1p, lane, hres, acc);
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * t + 8 * q + 4 * half;
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = gamd_gelu(acc[t][q * 4 + j]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.dec_w2 + c * GAMD_H + f0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[c] += w[j] * g[j];
                }
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = gamd_xhalf_sum(o[c]) + a.dec_b2[c];
        if (valid && half == 0) {
            const int orig = a.perm[atom];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.forces_norm[3 * (size_t)orig + c] = o[c];
                if (a.forces) a.forces[3 * (size_t)orig + c] = o[c] * a.scale + a.shift;
            }
        }
    }
}

}  // namespace

int launch_node(const NodeArgs& a, hipStream_t st) {
    const int nb = (a.n + GAMD_TILE - 1) / GAMD_TILE;
    hipLaunchKernelGGL(k_node, dim3(nb), dim3(64), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
