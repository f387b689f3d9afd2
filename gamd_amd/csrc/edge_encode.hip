// edge_encode.hip — fused edge-feature construction + edge encoder MLP + LayerNorm.
//
// Replaces (reference, code/nn_module.py):
//   :603-634  calc_edge_feat   (gather, min-image, norm, unit vector, standardise, 40 RBFs, concat)
//   :510-511  bond flag as 45th feature (water)
//   :646      edge_layer_norm(edge_encoder(feat))   MLP 44|45 -> 128 -> 128 -> 128, GELU(erf)
// The [E,44] feature matrix is never materialised: each lane builds its share of the features in
// registers, already in MFMA operand order, and the three GEMMs chain through the register file
// (gamd_common.h).  All three weight matrices stay resident in LDS (152 KiB) for the whole
// persistent kernel, so the main loop has no barriers.  Output: e in fragment order (e_frag), the
// layout the conv-layer kernel loads with perfectly coalesced 16-byte reads.
#include "gamd_common.h"
#include "gamd_internal.h"
#include <cstdlib>

namespace {

constexpr int ENC_W1_FLOATS = 4 * 6 * 64 * 4;              // K padded to 48 (24 MFMA steps)
constexpr int ENC_LDS_FLOATS = ENC_W1_FLOATS + 2 * GAMD_WFRAG_FLOATS + 5 * 128 + 64;

// Edge features of nn_module.py:603-634 (+ the bond flag of :510-511) for this lane's edge, already in the operand order of
// the first GEMM: MFMA K step s covers features (2 s, 2 s + 1), lanes 0-31 supply the even one, lanes 32-63 the odd one.
//   features: 0-2 unit vector, 3 standardised length d, 4-43 RBFs of d, 44 bond flag (NFEAT == 45)
template <int NFEAT, int ABL, typename CPtr>
__device__ __forceinline__ void edge_features(const EncArgs& a, CPtr cen, int src, int dst, const float4& ps, const float4& pd,
                                              int half, float (&F)[24]) {
    // No floating-point contraction in here: which multiply-add pairs hipcc fuses depends on the kernel the function is inlined
    // into, and k_edge_encode / k_edge_encode_small must produce the same bits (the host switches between them by edge count).
    // Fused operations are written out (fmaf) where they are meant.
#pragma clang fp contract(off)
    // nn_module.py:615-624
    const BoxDims B = gamd_edge_box(a, dst);
    const float rx = gamd_min_image_wrapped(ps.x - pd.x, B.bx, B.hx);
    const float ry = gamd_min_image_wrapped(ps.y - pd.y, B.by, B.hy);
    const float rz = gamd_min_image_wrapped(ps.z - pd.z, B.bz, B.hz);
    const float nrm = sqrtf((rx * rx + ry * ry) + rz * rz);
    const float den = nrm + 1e-8f;
    const float d = (nrm - a.length_mean) / a.length_std;          // :630
    F[0] = half ? ry / den : rx / den;
    F[1] = half ? d : rz / den;
    if (!(ABL & 8) && a.rbf.uniform) {
        gamd_rbf_chains(d, half, a.gamma * -1.4426950408889634f, a.rbf, F);
    } else {
#pragma unroll
        for (int s = 2; s < 22; ++s) {
            const float radial = d - cen[2 * (s - 2) + half];           // :261-263
            F[s] = (ABL & 8) ? radial : __builtin_amdgcn_exp2f((a.gamma * -1.4426950408889634f) * (radial * radial));
        }
    }
    F[22] = 0.f; F[23] = 0.f;
    if (NFEAT == 45) {
        // bond_graph.has_edges_between(centre, neigh), nn_module.py:510
        float flag = 0.f;
        if (a.bond_nbr) {
            const int io = a.perm[dst], jo = a.perm[src];
            const int4 nb = *reinterpret_cast<const int4*>(a.bond_nbr + 4 * (size_t)io);
            flag = (nb.x == jo || nb.y == jo || nb.z == jo || nb.w == jo) ? 1.f : 0.f;
        }
        F[22] = half ? 0.f : flag;
    }
}

// X = GELU(acc) on a 32 x 128 block.  ABL bit 1 (profiling build): x/2 instead (timing ablation, wrong results); bit 16:
// the scalar form
template <int ABL>
__device__ __forceinline__ void gelu_block(const f32x16 (&acc)[4], f32x16 (&X)[4], const GeluCoef& k) {
    if (ABL & 32) __builtin_amdgcn_s_setprio(1);           // ABL 32: the block at priority 1 (conv_edge_bf16.hip's finding, tried here)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            if (ABL & 1) { X[t][r] = acc[t][r] * 0.5f; X[t][r + 1] = acc[t][r + 1] * 0.5f; }
            else if (ABL & 16) { X[t][r] = gamd_gelu_hw(acc[t][r]); X[t][r + 1] = gamd_gelu_hw(acc[t][r + 1]); }
            else {
                const gelu_f2 y = gelu_pair(gelu_f2{acc[t][r], acc[t][r + 1]}, k);
                X[t][r] = y[0]; X[t][r + 1] = y[1];
            }
        }
    if (ABL & 32) __builtin_amdgcn_s_setprio(0);
}

template <int NFEAT, int ABL>
__global__ void __launch_bounds__(512, 2) k_edge_encode(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1 = lds;
    float* w2 = w1 + ENC_W1_FLOATS;
    float* w3 = w2 + GAMD_WFRAG_FLOATS;
    float* vb1 = w3 + GAMD_WFRAG_FLOATS;
    float* vb2 = vb1 + 128;
    float* vb3 = vb2 + 128;
    float* vg = vb3 + 128;
    float* vbeta = vg + 128;
    float* cen = vbeta + 128;

    const int tid = threadIdx.x;
    const GeluCoef gk = gelu_coef();
    for (int i = tid; i < ENC_W1_FLOATS / 4; i += 512) ((f32x4*)w1)[i] = ((const f32x4*)a.w1p)[i];
    for (int i = tid; i < GAMD_WFRAG_FLOATS / 4; i += 512) {
        ((f32x4*)w2)[i] = ((const f32x4*)a.w2p)[i];
        ((f32x4*)w3)[i] = ((const f32x4*)a.w3p)[i];
    }
    if (tid < 128) {
        vb1[tid] = a.b1[tid]; vb2[tid] = a.b2[tid]; vb3[tid] = a.b3[tid];
        vg[tid] = a.ln_g[tid]; vbeta[tid] = a.ln_b[tid];
    }
    if (tid < 40) cen[tid] = a.centers[tid];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, slot = lane & 31, half = lane >> 5;
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    // work unit = 4 tiles (one per SIMD); waves 0-3 and 4-7 take successive units of the workgroup's list, which
    // balances the launch to half a round (the two waves of a SIMD share its matrix pipe)
    const int n_units = (n_tiles + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    const int n_iter = first < end ? ((end - first + step - 1) / step + 1) / 2 : 0;
    auto tile_of = [&](int it) {
        const int u = first + (2 * it + (wave >> 2)) * step;
        return (it < n_iter && u < end) ? u * 4 + (wave & 3) : n_tiles;
    };

    constexpr int KSTEPS = (NFEAT + 1) / 2;      // 22 (LJ) or 23 (water + bond flag)

    // geometry of the first tile; every iteration then prefetches the next tile's indices and positions
    // (a three-deep dependent load chain) behind the current tile's GEMMs
    int E32 = (int)E;
    // Unconditional loads from a clamped index (E >= 1 whenever the loop runs), the validity applied where the value is
    // used: "r = 0; if (ok) r = load" makes hipcc guard the re-initialisation of r against the previous load into it —
    // with the stores of e in flight that guard is an s_waitcnt vmcnt(0) at the top of every tile.
    auto fetch_idx = [&](int tile, int& src, int& dst) -> bool {
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool ok = tile < n_tiles && x < E32;
        const int xc = ok ? x : 0;
        src = GAMD_CHK_RANGE(a.sticky, a.col[xc], 0, a.zero_row, GAMD_CHK_ENC_SRC); dst = GAMD_CHK_RANGE(a.sticky, a.erow[xc], 0, a.zero_row, GAMD_CHK_ENC_DST);
        return ok;
    };
    int src_c, dst_c;
    {
        const bool ok = fetch_idx(tile_of(0), src_c, dst_c);
        src_c = ok ? src_c : 0; dst_c = ok ? dst_c : 0;
    }
    float4 ps = a.pos_s[src_c], pd = a.pos_s[dst_c];

    for (int it = 0; it < n_iter; ++it) {
        const int tile = tile_of(it);
        int src_n, dst_n;
        const bool ok_n = fetch_idx(tile_of(it + 1), src_n, dst_n);
        if (tile >= n_tiles) { asm volatile("" ::"v"(src_n), "v"(dst_n)); continue; }      // (consumed on this path too)
        const long long x = (long long)tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = x < E;
        const int src = src_c, dst = dst_c;
        float F[24];
        if (ABL & 128) __builtin_amdgcn_s_setprio(1);
        edge_features<NFEAT, ABL>(a, cen, src, dst, ps, pd, half, F);
        if (ABL & 128) { asm volatile("" : "+v"(F[0]), "+v"(F[23])); __builtin_amdgcn_s_setprio(0); }
        if (a.feat_dbg && valid) {
#pragma unroll
            for (int s = 0; s < 24; ++s) a.feat_dbg[x * 48 + 2 * s + half] = F[s];
        }

        // ---- GEMM 1: [128 x NFEAT] ----
        f32x16 acc[4], X[4];
        load_bias_chain(vb1, half, acc);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (4 * g >= KSTEPS) break;
                const f32x4 w = ((const f32x4*)w1)[(tp * 6 + g) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * g + j < KSTEPS) acc[tp] = mfma32(w[j], F[4 * g + j], acc[tp]);
            }
        }
        gelu_block<ABL>(acc, X, gk);
        // ---- GEMM 2 ----
        load_bias_chain(vb2, half, acc);
        gemm128<false>((const f32x4*)w2, lane, X, acc);
        gelu_block<ABL>(acc, X, gk);
        // next tile's positions (its indices were fetched at the top of this iteration); consumed at the top of the next
        // iteration.  Issued HERE and waited for just before the stores of e below: with loads and stores both in flight hipcc
        // can only wait with vmcnt(0), i.e. a wait for these two loads placed behind the stores would sit out the write
        // latency of the whole 16 KiB tile.
        src_c = ok_n ? src_n : 0; dst_c = ok_n ? dst_n : 0;
        ps = a.pos_s[src_c]; pd = a.pos_s[dst_c];
        // ---- GEMM 3 + LayerNorm ----
        load_bias_chain(vb3, half, acc);
        gemm128<false>((const f32x4*)w3, lane, X, acc);
        if (ABL & 64) __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 2)) layernorm_chain_centered(acc, vg, vbeta, half, 1e-5f, a.ln_inv_width);     // W3, b3 arrive centred
        if (ABL & 64) { asm volatile("" : "+v"(acc[0]), "+v"(acc[3])); __builtin_amdgcn_s_setprio(0); }
        if (a.self_loop) {
            // self_loop_mode 1: the last edge of every row is the loop an in-place add_self_loop would have appended AFTER
            // edata['e'] was set (nn_module.py:649-652): its embedding is DGL's zero fill, not an encoded feature row
            if (valid && gamd_is_appended_loop(a, x, src, dst)) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            }
        }
        // ---- store e fragment: 16 x 1 KiB coalesced ----
        asm volatile("" ::"v"(ps.x), "v"(ps.y), "v"(ps.z), "v"(ps.w), "v"(pd.x), "v"(pd.y), "v"(pd.z), "v"(pd.w));
        f32x4* out = (f32x4*)a.e_frag + (size_t)tile * 16 * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[t][q * 4 + j];
                if (!(ABL & 4) || v[0] == 123.456f) out[(t * 4 + q) * 64 + lane] = v;
            }
    }
}

// ---- small edge counts (the reference's own drivers: 258 LJ atoms, 6 000 edges = 190 tiles) -----------------------------
// The persistent kernel above spends ~10 us staging 152 KiB of weights into LDS per workgroup before its first tile: too
// long when there is less than one tile per workgroup.  Here one 32-edge tile is shared by the four waves of a 256-thread
// workgroup (conv_edge_small.hip's scheme): wave w computes output features [32 w, 32 w + 32) of each of the three GEMMs
// from its weight quarter read straight from L2, applies GELU to its 16 values per lane (a quarter of the VALU work per
// wave), and the 128-wide rows are re-assembled through LDS between the GEMMs.  Same operations in the same order per
// output element as k_edge_encode (including the LayerNorm sums: per 32-feature block, then a fixed tree): bit-identical.
template <int NFEAT>
__global__ void __launch_bounds__(256) k_edge_encode_small(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    __shared__ __attribute__((aligned(16))) float xbuf[32 * GAMD_XLD];
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int quarter = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    constexpr int KSTEPS = (NFEAT + 1) / 2;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long x = (long long)tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = x < E;
        const int src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_ENC_SRC) : 0, dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_ENC_DST) : 0;
        const float4 ps = a.pos_s[src], pd = a.pos_s[dst];
        f32x4 w1[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) w1[g] = ((const f32x4*)a.w1p)[(quarter * 6 + g) * 64 + lane];
        WQuarter wa, wb;
        load_wquarter(a.w2p, quarter, lane, wa);
        float F[24];
        edge_features<NFEAT, 0>(a, a.centers, src, dst, ps, pd, half, F);
        if (a.feat_dbg && valid && quarter == 0) {
#pragma unroll
            for (int s = 0; s < 24; ++s) a.feat_dbg[x * 48 + 2 * s + half] = F[s];
        }
        // GEMM 1 + GELU
        f32x16 acc = load_slice(a.b1, quarter, half), X[4];
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (4 * g >= KSTEPS) break;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * g + j < KSTEPS) acc = mfma32(w1[g][j], F[4 * g + j], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_gelu_hw(acc[r]);
        exchange(xbuf, quarter, slot, half, acc, X);
        // GEMM 2 + GELU
        acc = load_slice(a.b2, quarter, half);
        load_wquarter(a.w3p, quarter, lane, wb);
        gemm_quarter<false>(wa, X, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_gelu_hw(acc[r]);
        exchange(xbuf, quarter, slot, half, acc, X);
        // GEMM 3 (rows arrive centred) + LayerNorm
        acc = load_slice(a.b3, quarter, half);
        gemm_quarter<false>(wb, X, acc);
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v = fmaf(acc[r], acc[r], v);
        __syncthreads();                                   // previous tile's readers of red[] are done
        red[quarter][lane] = v;
        __syncthreads();
        const float var = gamd_xhalf_sum((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * a.ln_inv_width;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const f32x16 g = load_slice(a.ln_g, quarter, half), b = load_slice(a.ln_b, quarter, half);
        const bool zero_row = valid && gamd_is_appended_loop(a, x, src, dst);      // appended loop: e = 0
        f32x4* out = (f32x4*)a.e_frag + (size_t)tile * 16 * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = zero_row ? 0.f : fmaf(acc[q * 4 + j] * rstd, g[q * 4 + j], b[q * 4 + j]);
            out[(quarter * 4 + q) * 64 + lane] = o;
        }
    }
}

}  // namespace

int launch_edge_encode_small(const EncArgs& a, int n_blocks, hipStream_t st) {
    if (a.n_feat == 44) hipLaunchKernelGGL((k_edge_encode_small<44>), dim3(n_blocks), dim3(256), 0, st, a);
    else if (a.n_feat == 45) hipLaunchKernelGGL((k_edge_encode_small<45>), dim3(n_blocks), dim3(256), 0, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}

template <int ABL>
static int launch_abl(const EncArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * ENC_LDS_FLOATS;
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_edge_encode<44, ABL>, k_edge_encode<45, ABL>)) return e;
    if (a.n_feat == 44) hipLaunchKernelGGL((k_edge_encode<44, ABL>), dim3(n_blocks), dim3(512), lds, st, a);
    else if (a.n_feat == 45) hipLaunchKernelGGL((k_edge_encode<45, ABL>), dim3(n_blocks), dim3(512), lds, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}

// Production: the GELU blocks and the feature construction (the dense vector stretches of a tile) at s_setprio 1.  The two waves
// of a SIMD share its vector issue; a wave inside such a stretch that keeps losing issue slots to its partner's scattered
// vector instructions finishes later and delays the partner's next stretch as well (found on k_conv_edge_bf16's SiLU blocks,
// round 6).  Same-box at C2: 418 -> 396 us, 0.74 -> 0.78 of the fp32 matrix peak, bit-identical e; LayerNorm at priority as
// well: 408 (not kept).  GAMD_ENC_VARIANT=0 (profiling build) is the round-5 kernel for the A/B.
constexpr int ENC_PRODUCTION = 32 | 128;
int launch_edge_encode(const EncArgs& a, int n_blocks, hipStream_t st) {
#ifdef GAMD_PROFILING
    // timing ablations (wrong results by construction): compiled into libgamd_hip_prof.so only
    static int v = -1;
    if (v < 0) { const char* s = getenv("GAMD_ENC_VARIANT"); v = s ? atoi(s) : ENC_PRODUCTION; }
    switch (v) {
        case 0: return launch_abl<0>(a, n_blocks, st);
        case 1: return launch_abl<1>(a, n_blocks, st);
        case 2: return launch_abl<2>(a, n_blocks, st);
        case 4: return launch_abl<4>(a, n_blocks, st);
        case 8: return launch_abl<8>(a, n_blocks, st);
        case 15: return launch_abl<15>(a, n_blocks, st);
        case 16: return launch_abl<16>(a, n_blocks, st);      // scalar GELU (round-2 form; same bits)
        case 32: return launch_abl<32>(a, n_blocks, st);      // GELU blocks at s_setprio 1 (same bits)
        case 96: return launch_abl<96>(a, n_blocks, st);      // + LayerNorm
        case 160: return launch_abl<160>(a, n_blocks, st);    // GELU + feature construction
        case 224: return launch_abl<224>(a, n_blocks, st);    // all three
        default: break;
    }
#endif
    return launch_abl<ENC_PRODUCTION>(a, n_blocks, st);
}
