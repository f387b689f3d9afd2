// edge_encode_bf16.hip — bf16-MFMA variant of the fused edge-feature + edge-encoder kernel (config 5).
//
// Same math as edge_encode.hip (nn_module.py:603-634, :646): features built in fp32 registers, rounded to
// bf16 as MFMA operands (v_mfma_f32_32x32x16_bf16, fp32 accumulate), GELU and LayerNorm in fp32; `e` is
// written as bf16 fragments (8 KiB per 32-edge tile) in exactly the operand order conv_edge_bf16.hip loads.
// K of the first GEMM is padded to 48 = 3 MFMA steps; step s, lane (slot, half) supplies features
// 16s + 8half + 0..7.
#include "gamd_bf16.h"
#include "gamd_internal.h"

namespace {

constexpr int ENCB_W1_BYTES = 4 * 3 * 64 * 16;                       // 12 KiB
constexpr int ENCB_LDS_BYTES = ENCB_W1_BYTES + 2 * GAMD_WFRAG_BF16_BYTES + (5 * 128 + 64) * 4;

// GELU of a 32 x 128 block + rounding to the bf16 operands of the next GEMM, two elements at a time (gelu_pair: the Horner chain
// of the exponent polynomial as packed instructions, per element the operations of gamd_gelu_hw: same bits as the scalar form)
__device__ __forceinline__ void gelu_pack_bf16(const f32x16 (&X)[4], bf16x8 (&P)[4][2], const GeluCoef& k) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            gamd_u32x4 w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gelu_f2 y = gelu_pair(gelu_f2{X[t][8 * u + 2 * q], X[t][8 * u + 2 * q + 1]}, k);
                w[q] = gamd_pk_bf16(y[0], y[1]);
            }
            P[t][u] = __builtin_bit_cast(bf16x8, w);
        }
}

template <int NFEAT>
__global__ void __launch_bounds__(512, 2) k_edge_encode_bf16(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const bf16x8* W1 = reinterpret_cast<const bf16x8*>(ldsb);
    const bf16x8* W2 = reinterpret_cast<const bf16x8*>(ldsb + ENCB_W1_BYTES);
    const bf16x8* W3 = W2 + 2048;
    float* vb1 = reinterpret_cast<float*>(ldsb + ENCB_W1_BYTES + 2 * GAMD_WFRAG_BF16_BYTES);
    float* vb2 = vb1 + 128;
    float* vb3 = vb2 + 128;
    float* vg = vb3 + 128;
    float* vbeta = vg + 128;
    float* cen = vbeta + 128;

    const int tid = threadIdx.x;
    {
        f32x4* d = reinterpret_cast<f32x4*>(ldsb);
        const f32x4* s1 = reinterpret_cast<const f32x4*>(a.w1p);
        const f32x4* s2 = reinterpret_cast<const f32x4*>(a.w2p);
        const f32x4* s3 = reinterpret_cast<const f32x4*>(a.w3p);
        for (int i = tid; i < ENCB_W1_BYTES / 16; i += 512) d[i] = s1[i];
        for (int i = tid; i < 2048; i += 512) { d[ENCB_W1_BYTES / 16 + i] = s2[i]; d[ENCB_W1_BYTES / 16 + 2048 + i] = s3[i]; }
        if (tid < 128) { vb1[tid] = a.b1[tid]; vb2[tid] = a.b2[tid]; vb3[tid] = a.b3[tid]; vg[tid] = a.ln_g[tid]; vbeta[tid] = a.ln_b[tid]; }
        if (tid < 40) cen[tid] = a.centers[tid];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, slot = lane & 31, half = lane >> 5;
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_wg_tiles = (n_tiles + 7) / 8;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    bf16x8* efrag = reinterpret_cast<bf16x8*>(a.e_frag);
    const float gexp = a.gamma * -1.4426950408889634f;
    const GeluCoef gk = gelu_coef();

    for (int wt = first; wt < end; wt += step) {
        const int tile = wt * 8 + wave;
        if (tile >= n_tiles) continue;
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = x < E;
        const int src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_ENC_SRC) : 0;
        const int dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_ENC_DST) : 0;
        const float4 ps = a.pos_s[src], pd = a.pos_s[dst];
        const BoxDims B = gamd_edge_box(a, dst);
        const float rx = gamd_min_image_wrapped(ps.x - pd.x, B.bx, B.hx);
        const float ry = gamd_min_image_wrapped(ps.y - pd.y, B.by, B.hy);
        const float rz = gamd_min_image_wrapped(ps.z - pd.z, B.bz, B.hz);
        const float nrm = sqrtf((rx * rx + ry * ry) + rz * rz);
        const float den = nrm + 1e-8f;
        const float d = (nrm - a.length_mean) / a.length_std;
        float bond = 0.f;
        if (NFEAT == 45 && a.bond_nbr) {
            const int io = a.perm[dst], jo = a.perm[src];
            const int4 nb = *reinterpret_cast<const int4*>(a.bond_nbr + 4 * (size_t)io);
            bond = (nb.x == jo || nb.y == jo || nb.z == jo || nb.w == jo) ? 1.f : 0.f;
        }
        // features 16s + 8half + j, j = 0..7 (fp32), then packed per step
        bf16x8 F[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * half + j;
                float v;
                if (s == 0 && j < 4 && half == 0) v = j == 0 ? rx / den : (j == 1 ? ry / den : (j == 2 ? rz / den : d));
                else if (k < 44) { const float radial = d - cen[k - 4]; v = __builtin_amdgcn_exp2f(gexp * (radial * radial)); }
                else v = (k == 44 && NFEAT == 45) ? bond : 0.f;
                f[j] = v;
                if (a.feat_dbg && valid) a.feat_dbg[(size_t)x * 48 + k] = v;
            }
            gamd_u32x4 w;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = gamd_pk_bf16(f[2 * q], f[2 * q + 1]);
            F[s] = __builtin_bit_cast(bf16x8, w);
        }
        // GEMM 1 (K = 48)
        f32x16 acc[4];
        bf16x8 P[4][2];
        load_bias_chain(vb1, half, acc);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[(tp * 3 + s) * 64 + lane], F[s], acc[tp], 0, 0, 0);
        gelu_pack_bf16(acc, P, gk);
        // GEMM 2
        load_bias_chain(vb2, half, acc);
        gemm128_bf16<false>(W2, lane, P, acc);
        gelu_pack_bf16(acc, P, gk);
        // GEMM 3 + LayerNorm (fp32)
        load_bias_chain(vb3, half, acc);
        gemm128_bf16<false>(W3, lane, P, acc);
        layernorm_chain(acc, vg, vbeta, half, 1e-5f);
        pack_chain_bf16(acc, P);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) efrag[((size_t)tile * 8 + t * 2 + u) * 64 + lane] = P[t][u];
    }
}

}  // namespace

int launch_edge_encode_bf16(const EncArgs& a, int n_blocks, hipStream_t st) {
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)ENCB_LDS_BYTES, k_edge_encode_bf16<44>, k_edge_encode_bf16<45>)) return e;
    if (a.n_feat == 44) hipLaunchKernelGGL(k_edge_encode_bf16<44>, dim3(n_blocks), dim3(512), ENCB_LDS_BYTES, st, a);
    else if (a.n_feat == 45) hipLaunchKernelGGL(k_edge_encode_bf16<45>, dim3(n_blocks), dim3(512), ENCB_LDS_BYTES, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}
