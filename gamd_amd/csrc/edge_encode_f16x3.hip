// edge_encode_f16x3.hip — the fused edge-feature + edge-encoder kernel (edge_encode.hip; nn_module.py:603-634, :646)
// with its three GEMMs on the fp16 matrix pipe via operand splitting (gamd_f16x3.h): fp32-grade results at 3/16 of
// the fp32 matrix time.  Features, GELU(erf) and LayerNorm are fp32 as before.  All three weight matrices stay
// resident in LDS as [hi | lo] fp16 fragment images (24 + 64 + 64 KiB, the footprint of the fp32 kernel), so the
// main loop has no barriers.  `e` is written already split, [tile][t][u][hi|lo][lane][8 halves] (16 KiB per tile,
// the size of the fp32 fragment), which is the operand order conv_edge_f16x3.hip loads.
// K of the first GEMM is padded to 48 = 3 MFMA steps; step s, lane (slot, half) supplies features 16s + 8half + 0..7.
#include "gamd_f16x3.h"
#include "gamd_internal.h"

namespace {

constexpr int ENCH_W1_PART = 4 * 3 * 64;                              // f16x8 entries per part (12 KiB)
constexpr int ENCH_LDS_BYTES = 2 * ENCH_W1_PART * 16 + 2 * 65536 + (5 * 128 + 64) * 4;

template <int NFEAT>
__global__ void __launch_bounds__(512, 2) k_edge_encode_f16x3(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const f16x8* W1 = reinterpret_cast<const f16x8*>(ldsb);                       // [hi | lo], 2 x 12 KiB
    const f16x8* W2 = W1 + 2 * ENCH_W1_PART;                                      // [hi | lo], 2 x 32 KiB
    const f16x8* W3 = W2 + 4096;
    float* vb1 = reinterpret_cast<float*>(ldsb + 2 * ENCH_W1_PART * 16 + 2 * 65536);
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
        for (int i = tid; i < 2 * ENCH_W1_PART; i += 512) d[i] = s1[i];
        for (int i = tid; i < 4096; i += 512) { d[2 * ENCH_W1_PART + i] = s2[i]; d[2 * ENCH_W1_PART + 4096 + i] = s3[i]; }
        if (tid < 128) { vb1[tid] = a.b1[tid]; vb2[tid] = a.b2[tid]; vb3[tid] = a.b3[tid]; vg[tid] = a.ln_g[tid]; vbeta[tid] = a.ln_b[tid]; }
        if (tid < 40) cen[tid] = a.centers[tid];
    }
    __syncthreads();

    const int lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_units = (n_tiles + 3) / 4;                   // 4-tile work units, see edge_encode.hip
    int first, end, step;
    gamd_xcd_range(n_units, blockIdx.x, gridDim.x, first, end, step);
    const int n_iter = first < end ? ((end - first + step - 1) / step + 1) / 2 : 0;
    f16x8* efrag = reinterpret_cast<f16x8*>(a.e_frag);
    const float gexp = a.gamma * -1.4426950408889634f;

    for (int it = 0; it < n_iter; ++it) {
        const int unit = first + (2 * it + (wave >> 2)) * step;
        const int tile = unit * 4 + (wave & 3);
        if (unit >= end || tile >= n_tiles) continue;
        asm volatile("" ::: "memory");                       // keep loop-invariant LDS reads inside the loop
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
        // ---- GEMM 1 (K = 48): features 16s + 8half + j in fp32, split per step ----
        f32x16 acc[4], X[4];
        load_bias_chain(vb1, half, acc);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            f32x16 fv;                                       // only the first 8 entries are used
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * half + j;
                float v;
                if (s == 0 && j < 4 && half == 0) v = j == 0 ? rx / den : (j == 1 ? ry / den : (j == 2 ? rz / den : d));
                else if (k < 44) { const float radial = d - cen[k - 4]; v = __builtin_amdgcn_exp2f(gexp * (radial * radial)); }
                else v = (k == 44 && NFEAT == 45) ? bond : 0.f;
                fv[j] = v;
                if (a.feat_dbg && valid) a.feat_dbg[(size_t)x * 48 + k] = v;
            }
#pragma unroll
            for (int j = 8; j < 16; ++j) fv[j] = 0.f;
            f16x8 fh, fl;
            gamd_split8(fv, 0, fh, fl);
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const f16x8 wh = W1[(tp * 3 + s) * 64 + lane], wl = W1[ENCH_W1_PART + (tp * 3 + s) * 64 + lane];
                acc[tp] = mfma_f16(wh, fl, acc[tp]);
                acc[tp] = mfma_f16(wl, fh, acc[tp]);
                acc[tp] = mfma_f16(wh, fh, acc[tp]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[t][r] = gamd_gelu_hw(acc[t][r]);
        // ---- GEMM 2 ----
        load_bias_chain(vb2, half, acc);
        gemm128_f16x3<false>(W2, lane, X, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[t][r] = gamd_gelu_hw(acc[t][r]);
        // ---- GEMM 3 + LayerNorm (fp32) ----
        load_bias_chain(vb3, half, acc);
        gemm128_f16x3<false>(W3, lane, X, acc);
        layernorm_chain(acc, vg, vbeta, half, 1e-5f);
        // ---- store e already split: 16 x 1 KiB coalesced ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f16x8 eh, el;
                gamd_split8(acc[t], u, eh, el);
                efrag[(((size_t)tile * 8 + t * 2 + u) * 2 + 0) * 64 + lane] = eh;
                efrag[(((size_t)tile * 8 + t * 2 + u) * 2 + 1) * 64 + lane] = el;
            }
    }
}

}  // namespace

int launch_edge_encode_f16x3(const EncArgs& a, int n_blocks, hipStream_t st) {
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)ENCH_LDS_BYTES, k_edge_encode_f16x3<44>, k_edge_encode_f16x3<45>)) return e;
    if (a.n_feat == 44) hipLaunchKernelGGL(k_edge_encode_f16x3<44>, dim3(n_blocks), dim3(512), ENCH_LDS_BYTES, st, a);
    else if (a.n_feat == 45) hipLaunchKernelGGL(k_edge_encode_f16x3<45>, dim3(n_blocks), dim3(512), ENCH_LDS_BYTES, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}
