// wide_d.hip — the generic-width path for hidden_dim above 128.
//
// `hidden_dim` is the inner width D of the reference's MLPs (nn_module.py:21-60): the edge encoder is F -> D -> D -> Eh
// (:306-317), a conv layer's edge_affine ends in D, src/dst_affine, phi_dst and phi_edge map H -> D, theta_edge is
// SiLU Lin(D, D) SiLU Lin(D, H), phi is SiLU Lin(D, H) (:95-106) and the decoder is Lin(H, D) GELU Lin(D, 3) (:320).  Every
// shipped configuration has D = 128 (one block); the kernels here serve 128 < D <= 256, i.e. DT = 2 blocks of 128 (the
// host zero-pads in-between widths, padded features are exact zeros through every Linear / SiLU / GELU).  They are the
// DT-block siblings of wide.hip's three kernels — same chain layout, same fragment order, same pieces — with every D-wide
// operand two K blocks accumulated into one output and every D-wide result two output blocks.  fp32 only; one wave per
// SIMD (4-wave workgroups, the whole register file per wave: a tile's D-wide activations are 128 registers each).
#include "gamd_common.h"
#include "gamd_internal.h"
#include "gamd_wide.h"

namespace {

constexpr int ENC_W1_BLOCK_FLOATS = 4 * 6 * 64 * 4;      // one 128-output block of the encoder's first Linear (K padded to 48)

__device__ __forceinline__ void wd_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Two 64 KiB slots at the start of LDS, weight blocks streamed through them in the order the kernel uses them: phase g reads
// slot g & 1 while the block of phase g + 1 arrives in the other one (the scheme of k_conv_edge_wide, 4 waves here).
template <int NP>
struct BlockRing {
    const float* blocks;
    float* lds;
    int wave;
    unsigned lane16;
    unsigned g = 0;
    int blk = 0;
    __device__ __forceinline__ void prime() { gamd_stage_weight_raw_contig<4>(blocks, lds, wave, lane16); }
    __device__ __forceinline__ const f32x4* begin() {
        const int nb = (blk + 1 == NP) ? 0 : blk + 1;
        gamd_stage_weight_raw_contig<4>(blocks + (size_t)nb * GAMD_WFRAG_FLOATS, lds + ((g + 1) & 1u) * GAMD_WFRAG_FLOATS, wave, lane16);
        unsigned off = (g & 1u) * (unsigned)(GAMD_WFRAG_FLOATS * sizeof(float));
        asm volatile("" : "+s"(off));                     // one base register + immediate offsets (see k_conv_edge_wide)
        return (const f32x4*)((const char*)lds + off);
    }
    __device__ __forceinline__ void end() { wd_barrier(); ++g; blk = (blk + 1 == NP) ? 0 : blk + 1; }
};

// ================================================================================================
// edge features + edge encoder F -> D -> D -> Eh (GELU) + LayerNorm(Eh)
//   blocks at a.w1p: [W1, DT blocks of 24 KiB in one 64 KiB slot] | W2[db'][db] (DT x DT) | W3[ob][db] (EHT x DT)
// ================================================================================================
template <int NFEAT, int EHT, int DT>
__global__ void __launch_bounds__(256, 1) k_edge_encode_wide_d(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;
    constexpr int EH = 128 * EHT, DP = 128 * DT;
    constexpr int NP = 1 + DT * DT + EHT * DT;
    constexpr bool EXPAND = NFEAT >= 44;
    constexpr int KSTEPS = (NFEAT + 1) / 2;
    static_assert(DT * ENC_W1_BLOCK_FLOATS <= GAMD_WFRAG_FLOATS, "first Linear fits one slot");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* vb1 = lds + 2 * GAMD_WFRAG_FLOATS;
    float* vb2 = vb1 + DP;
    float* vb3 = vb2 + DP;
    float* vg = vb3 + EH;
    float* vbeta = vg + EH;
    float* cen = vbeta + EH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    const int n_wg_tiles = (n_tiles + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;

    for (int i = tid; i < DP; i += 256) { vb1[i] = a.b1[i]; vb2[i] = a.b2[i]; }
    for (int i = tid; i < EH; i += 256) { vb3[i] = a.b3[i]; vg[i] = a.ln_g[i]; vbeta[i] = a.ln_b[i]; }
    if (EXPAND && tid < 40) cen[tid] = a.centers[tid];
    BlockRing<NP> ring{a.w1p, lds, wave, (unsigned)lane * 16u};
    ring.prime();
    wd_barrier();

    for (int wt = first; wt < end; wt += step) {          // uniform over the workgroup (barriers inside)
        asm volatile("" ::: "memory");
        const int tile = wt * 4 + wave;
        const bool active = tile < n_tiles;
        const long long x = (long long)tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = active && x < E;
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
        float F[24];
#pragma unroll
        for (int s = 0; s < 24; ++s) F[s] = 0.f;
        F[0] = half ? ry / den : rx / den;
        F[1] = half ? d : rz / den;
        if (EXPAND && a.rbf.uniform) {
            gamd_rbf_chains(d, half, a.gamma * -1.4426950408889634f, a.rbf, F);
        } else if (EXPAND) {
#pragma unroll
            for (int s = 2; s < 22; ++s) {
                const float radial = d - cen[2 * (s - 2) + half];
                F[s] = __builtin_amdgcn_exp2f((a.gamma * -1.4426950408889634f) * (radial * radial));
            }
        }
        if (NFEAT & 1) {                                   // bond flag is the last feature (even index)
            float flag = 0.f;
            if (a.bond_nbr) {
                const int io = a.perm[dst], jo = a.perm[src];
                const int4 nb = *reinterpret_cast<const int4*>(a.bond_nbr + 4 * (size_t)io);
                flag = (nb.x == jo || nb.y == jo || nb.z == jo || nb.w == jo) ? 1.f : 0.f;
            }
            F[(NFEAT - 1) / 2] = half ? 0.f : flag;
        }
        if (a.feat_dbg && valid) {
#pragma unroll
            for (int s = 0; s < 24; ++s) a.feat_dbg[x * 48 + 2 * s + half] = F[s];
        }

        // ---- X1 = GELU(W1 F + b1), D wide ---------------------------------------------------------
        f32x16 X1[DT][4];
        {
            const f32x4* W = ring.begin();
#pragma unroll
            for (int db = 0; db < DT; ++db) {
                load_bias_chain(vb1 + 128 * db, half, X1[db]);
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        if (4 * g >= KSTEPS) break;
                        const f32x4 w = W[db * (ENC_W1_BLOCK_FLOATS / 4) + (tp * 6 + g) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4 * g + j < KSTEPS) X1[db][tp] = mfma32(w[j], F[4 * g + j], X1[db][tp]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) X1[db][t][r] = gamd_gelu_hw(X1[db][t][r]);
            }
            ring.end();
        }
        // ---- X2 = GELU(W2 X1 + b2), D wide --------------------------------------------------------
        f32x16 X2[DT][4];
#pragma unroll
        for (int ob = 0; ob < DT; ++ob) {
            load_bias_chain(vb2 + 128 * ob, half, X2[ob]);
#pragma unroll
            for (int db = 0; db < DT; ++db) {
                const f32x4* W = ring.begin();
                if (active) gemm128<false>(W, lane, X1[db], X2[ob]);
                ring.end();
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) X2[ob][t][r] = gamd_gelu_hw(X2[ob][t][r]);
        }
        // ---- Y = W3 X2 + b3, Eh wide ---------------------------------------------------------------
        f32x16 Y[EHT][4];
#pragma unroll
        for (int ob = 0; ob < EHT; ++ob) {
            load_bias_chain(vb3 + 128 * ob, half, Y[ob]);
#pragma unroll
            for (int db = 0; db < DT; ++db) {
                const f32x4* W = ring.begin();
                if (active) gemm128<false>(W, lane, X2[db], Y[ob]);
                ring.end();
            }
        }
        // LayerNorm over Eh features (torch: biased variance, eps inside the sqrt)
        float s1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < EHT; ++ob)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s1 += Y[ob][t][r];
        const float mean = gamd_xhalf_sum(s1) * a.ln_inv_width;
        float s2 = 0.f;
#pragma unroll
        for (int ob = 0; ob < EHT; ++ob)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float dd = Y[ob][t][r] - mean; s2 += dd * dd; }
        // (zero-padded features each added mean^2 to s2: taken out again)
        const float rstd = 1.0f / sqrtf((gamd_xhalf_sum(s2) - a.ln_n_pad * (mean * mean)) * a.ln_inv_width + 1e-5f);
        // self_loop_mode 1: the appended loop (last edge of its row) carries DGL's zero-filled embedding (nn_module.py:364)
        const bool zero_row = valid && gamd_is_appended_loop(a, x, src, dst);
        if (active) {
            f32x4* out = (f32x4*)a.e_frag + (size_t)tile * EHT * 16 * 64;
#pragma unroll
            for (int ob = 0; ob < EHT; ++ob)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 128 * ob + 32 * t + 8 * q + 4 * half;
                        const f32x4 g = *reinterpret_cast<const f32x4*>(&vg[f0]);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(&vbeta[f0]);
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = zero_row ? 0.f : (Y[ob][t][q * 4 + j] - mean) * rstd * g[j] + b[j];
                        out[((ob * 4 + t) * 4 + q) * 64 + lane] = v;
                    }
        }
    }
}

// ================================================================================================
// conv layer, edge side (nn_module.py:135-142) for Eh = 128 EHT, H = 128 HT, hidden_dim = 128 DT
//   blocks at a.w1p:  W1[:, kb] (EHT) | W2[db, :] (DT) | W3[db', db] (DT x DT) | W4[ob, db] (HT x DT)
//   edge_affine's inner width stays 128 (MLP's default, nn_module.py:25,95); S, D rows are 128 DT wide
// ================================================================================================
template <int EHT, int HT, int DT>
__global__ void __launch_bounds__(256, 1) k_conv_edge_wide_d(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;
    constexpr int NP = EHT + DT + DT * DT + HT * DT;
    constexpr int H = 128 * HT, DP = 128 * DT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* vb1 = lds + 2 * GAMD_WFRAG_FLOATS;
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + DP;

    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_wg_tiles = (n_tiles + 3) / 4;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;

    if (tid < 128) vb1[tid] = a.b1[tid];
    for (int i = tid; i < DP; i += 256) vb3[i] = a.b3[i];
    for (int i = tid; i < H; i += 256) vb4[i] = a.b4[i];
    BlockRing<NP> ring{a.w1p, lds, wave, (unsigned)lane * 16u};
    ring.prime();
    wd_barrier();

    for (int wt = first; wt < end; wt += step) {
        asm volatile("" ::: "memory");
        const int tile = wt * 4 + wave;
        const bool active = tile < n_tiles;
        const int x = tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = active && x < E;
        const int src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_CONV_SRC) : 0;
        const int dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_CONV_DST) : 0;
        const int x0 = tile * GAMD_TILE + 16 * half;
        int nvalid = E - x0;
        nvalid = !active ? 0 : (nvalid >= 16 ? 16 : (nvalid <= 0 ? 0 : nvalid));
        unsigned mask = 0;
        int p0 = 0;
        if (active) { mask = a.chunk_mask[tile * 2 + half]; p0 = GAMD_CHK_RANGE(a.sticky, a.chunk_piece[tile * 2 + half], 0, a.piece_cap - 17, GAMD_CHK_PIECE); }

        f32x16 T1[4], X[4];
        // ---- T1 = SiLU(W1 e + b1), K = Eh, 128 wide -----------------------------------------------
        load_bias_chain(vb1, half, T1);
#pragma unroll
        for (int kb = 0; kb < EHT; ++kb) {
            const f32x4* W = ring.begin();
            if (active) {
                const f32x4* ef = (const f32x4*)a.e_frag + ((size_t)tile * EHT + kb) * 16 * 64;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = ef[(t * 4 + q) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j) X[t][q * 4 + j] = v[j];
                    }
                gemm128<false>(W, lane, X, T1);
            }
            ring.end();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) T1[t][r] = gamd_silu_hw(T1[t][r]);
        // ---- T3[db] = SiLU(W2[db] T1 + S[src][db] + D[dst][db]) ------------------------------------
        f32x16 T3[DT][4];
#pragma unroll
        for (int db = 0; db < DT; ++db) {
            const f32x4* W = ring.begin();
            if (active) {
                load_row_chain(a.S + (size_t)src * DP + 128 * db, half, T3[db]);
                load_row_chain(a.D + (size_t)dst * DP + 128 * db, half, X);
#pragma unroll
                for (int t = 0; t < 4; ++t) T3[db][t] += X[t];
                gemm128<false>(W, lane, T1, T3[db]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) T3[db][t][r] = gamd_silu_hw(T3[db][t][r]);
            }
            ring.end();
        }
        // ---- T4[ob] = SiLU(sum_db W3[ob][db] T3[db] + b3[ob]) --------------------------------------
        f32x16 T4[DT][4];
#pragma unroll
        for (int ob = 0; ob < DT; ++ob) {
            load_bias_chain(vb3 + 128 * ob, half, T4[ob]);
#pragma unroll
            for (int db = 0; db < DT; ++db) {
                const f32x4* W = ring.begin();
                if (active) gemm128<false>(W, lane, T3[db], T4[ob]);
                ring.end();
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) T4[ob][t][r] = gamd_silu_hw(T4[ob][t][r]);
        }
        // ---- e_emb block ob = sum_db T4[db] W4[ob][db]^T + b4 (F2), message with hn[src], segment sum
#pragma unroll
        for (int ob = 0; ob < HT; ++ob) {
            f32x16 U[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const float b = vb4[128 * ob + 32 * tp + slot];
#pragma unroll
                for (int r = 0; r < 16; ++r) U[tp][r] = b;
            }
#pragma unroll
            for (int db = 0; db < DT; ++db) {
                const f32x4* W = ring.begin();
                if (active) {
                    if (db == DT - 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
                            const int s = __shfl(src, rho, 64);
                            const float* hrow = a.hn + (size_t)s * H + 128 * ob + slot;
#pragma unroll
                            for (int tp = 0; tp < 4; ++tp) X[tp][r] = hrow[32 * tp];
                        }
                    }
                    gemm128<true>(W, lane, T4[db], U);
                    if (db == DT - 1) {
                        if (a.emb_out) {                       // update_edge_emb: e_emb rows for launch_edge_update (edge x0 + r)
                            float* er = a.emb_out + (size_t)x0 * H + 128 * ob + slot;
#pragma unroll
                            for (int r = 0; r < 16; ++r)
#pragma unroll
                                for (int tp = 0; tp < 4; ++tp) er[(size_t)r * H + 32 * tp] = U[tp][r];
                        }
                        const unsigned keep_bits = ~(mask << 1);
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                U[tp][r] = gamd_msg_acc((r < nvalid) ? X[tp][r] : 0.f, U[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? U[tp][r - 1] : 0.f);
                        unsigned ends = mask;
                        if (nvalid > 0 && !((mask >> (nvalid - 1)) & 1u)) ends |= 1u << (nvalid - 1);
                        int p = p0;
                        while (__any(ends != 0)) {
                            if (ends != 0) {
                                const int r = __builtin_ctz(ends);
                                ends &= ends - 1;
#pragma unroll
                                for (int tp = 0; tp < 4; ++tp) {
                                    float v = U[tp][0];
#pragma unroll
                                    for (int k = 1; k < 16; ++k) v = (r == k) ? U[tp][k] : v;
                                    a.partial[(size_t)p * H + 128 * ob + 32 * tp + slot] = v;
                                }
                                ++p;
                            }
                        }
                    }
                }
                ring.end();
            }
        }
    }
}

// ================================================================================================
// node side (wide.hip's k_node_wide) with D = 128 DT: S, D, P rows and the decoder's inner layer are DT blocks
//   step order: phi_edge (db, kb) | phi (ob, db) | then S (db, kb) | D (db, kb) | P (db, kb), or the decoder's (db, kb)
// ================================================================================================

template <int HT, int DT>
__global__ void __launch_bounds__(256) k_node_wide_d(NodeArgs a) {
    constexpr int H = 128 * HT, DP = 128 * DT, NB = HT * DT;
    constexpr int XLDW = 128 * (HT > DT ? HT : DT) + 4;
    __shared__ __attribute__((aligned(16))) float xbuf[32 * XLDW];
    __shared__ float obuf[4][32][3];
    __shared__ float red[2][4][32];

    if (a.counters[CNT_OVERFLOW] || a.devflags[DEVFLAG_FROZEN]) return;
    if (a.mode == 0 && a.l0_gate && a.counters[CNT_REBUILD] == 0) return;      // layer-0 tables of the last rebuild still stand (node.hip)

    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int quarter = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int atom_raw = blockIdx.x * GAMD_TILE + slot;
    const bool valid = atom_raw < a.n;
    const int atom = valid ? atom_raw : a.n - 1;
    const size_t rowH = (size_t)atom * H, rowD = (size_t)atom * DP;

    f32x16 XH[HT][4];
    f32x16 mine[HT];
    WQ wqa, wqb;                                             // weight quarters one block-GEMM ahead (see k_node_wide)
    constexpr size_t BLK = GAMD_WFRAG_FLOATS;
    auto wptr = [&](int st) -> const float* {
        if (st < NB) return a.post.wpep + (size_t)st * BLK;
        if (st < 2 * NB) return a.post.wphip + (size_t)(st - NB) * BLK;
        const int i = st - 2 * NB;
        if (a.mode == 2) return i < NB ? a.dec_w1p + (size_t)i * BLK : nullptr;
        if (i < NB) return a.pre.wsp + (size_t)i * BLK;
        if (i < 2 * NB) return a.pre.wdp + (size_t)(i - NB) * BLK;
        return i < 3 * NB ? a.pre.wpdp + (size_t)(i - 2 * NB) * BLK : nullptr;
    };
    auto step = [&](int st, const f32x16 (&Xb)[4], f32x16& acc) {
        WQ& cur = (st & 1) ? wqb : wqa;
        WQ& nxt = (st & 1) ? wqa : wqb;
        const float* np_ = wptr(st + 1);
        asm volatile("" ::: "memory");
        if (np_) wq_load(np_, quarter, lane, nxt);
        asm volatile("" ::: "memory");
        wq_gemm(cur, Xb, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    wq_load(a.mode == 0 ? wptr(2 * NB) : wptr(0), quarter, lane, wqa);        // (2 NB is even: buffer a)

    if (a.mode == 0) {
#pragma unroll
        for (int b = 0; b < HT; ++b) {
            if (a.node_emb) {
                mine[b] = load_slice(a.node_emb + 128 * b, quarter, half);
            } else {
                const float f = a.pos_s[atom].w;
                const f32x16 w = load_slice(a.enc_w + 128 * b, quarter, half);
                mine[b] = load_slice(a.enc_b + 128 * b, quarter, half);
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[b][r] = f * w[r] + mine[b][r];
            }
            if (valid) store_slice(a.h_out + rowH + 128 * b, quarter, half, mine[b]);
        }
    } else {
        // ---- post(l-1): agg = sum of this atom's pieces, in order -------------------------------
        const int rp0 = a.row_ptr[atom], dg = a.deg[atom];
        const int na_incl = a.na_excl[atom] + ((dg > 0 && (rp0 % GAMD_CHUNK) != 0) ? 1 : 0);
        const int p0 = rp0 / GAMD_CHUNK + na_incl;
        const int np = dg > 0 ? ((rp0 + dg - 1) / GAMD_CHUNK - rp0 / GAMD_CHUNK + 1) : 0;
        (void)GAMD_CHK_RANGE(a.sticky, (long long)p0 + np, 0, a.piece_cap, GAMD_CHK_NODE_PIECES);
#pragma unroll
        for (int b = 0; b < HT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[b][r] = 0.f;
        f32x16 h_res[HT];
#pragma unroll
        for (int b = 0; b < HT; ++b) h_res[b] = load_slice(a.h_in + rowH + 128 * b, quarter, half);
        for (int k0 = 0; __any(k0 < np); k0 += 4) {
            f32x16 pc[4][HT];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kk = (k0 + k < np) ? k0 + k : (np > 0 ? np - 1 : 0);
                const float* prow = a.partial + (size_t)(np > 0 ? p0 + kk : 0) * H;
#pragma unroll
                for (int b = 0; b < HT; ++b) pc[k][b] = load_slice(prow + 128 * b, quarter, half);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k0 + k < np) {
#pragma unroll
                    for (int b = 0; b < HT; ++b) mine[b] += pc[k][b];
                }
        }
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, XH);         // XH = agg
        f32x16 accD[DT];
#pragma unroll
        for (int db = 0; db < DT; ++db) {                                     // phi_edge: H -> D
            accD[db] = load_slice(a.P_in + rowD + 128 * db, quarter, half);
#pragma unroll
            for (int kb = 0; kb < HT; ++kb) step(db * HT + kb, XH[kb], accD[db]);
#pragma unroll
            for (int r = 0; r < 16; ++r) accD[db][r] = gamd_silu_hw(accD[db][r]);
        }
        f32x16 XD[DT][4];
        exchange_blocks<DT, XLDW>(xbuf, quarter, slot, half, accD, XD);         // XD = SiLU(P + phi_edge(agg))
#pragma unroll
        for (int ob = 0; ob < HT; ++ob) {                                     // phi: D -> H, residual
            mine[ob] = load_slice(a.post.bphi + 128 * ob, quarter, half);
#pragma unroll
            for (int db = 0; db < DT; ++db) step(NB + ob * DT + db, XD[db], mine[ob]);
            mine[ob] += h_res[ob];
            if (valid) store_slice(a.h_out + rowH + 128 * ob, quarter, half, mine[ob]);
        }
    }

    if (a.mode != 2) {
        // ---- pre(l): LayerNorm over H (or the folded BatchNorm), then S, D, P (H -> D each) ------
        float mean = 0.f, rstd = 1.0f;
        if (!a.norm_bn) {
            float ps = 0.f;
#pragma unroll
            for (int b = 0; b < HT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += mine[b][r];
            ps = gamd_xhalf_sum(ps);
            if (half == 0) red[0][quarter][slot] = ps;
            __syncthreads();
            mean = ((red[0][0][slot] + red[0][1][slot]) + (red[0][2][slot] + red[0][3][slot])) * a.ln_inv_width;
            float pv = 0.f;
#pragma unroll
            for (int b = 0; b < HT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = mine[b][r] - mean; pv += d * d; }
            pv = gamd_xhalf_sum(pv);
            if (half == 0) red[1][quarter][slot] = pv;
            __syncthreads();
            const float var = (((red[1][0][slot] + red[1][1][slot]) + (red[1][2][slot] + red[1][3][slot])) - a.ln_n_pad * (mean * mean)) * a.ln_inv_width;
            rstd = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int b = 0; b < HT; ++b) {
            const f32x16 g = load_slice(a.pre.ln_g + 128 * b, quarter, half), be = load_slice(a.pre.ln_b + 128 * b, quarter, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[b][r] = (mine[b][r] - mean) * rstd * g[r] + be[r];
            if (valid) store_slice(a.hn_out + rowH + 128 * b, quarter, half, mine[b]);
        }
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, XH);         // XH = hn
#pragma unroll
        for (int db = 0; db < DT; ++db) {
            f32x16 acc = load_slice(a.pre.bS + 128 * db, quarter, half);
#pragma unroll
            for (int kb = 0; kb < HT; ++kb) step(2 * NB + db * HT + kb, XH[kb], acc);
            if (valid) store_slice(a.S_out + rowD + 128 * db, quarter, half, acc);
        }
#pragma unroll
        for (int db = 0; db < DT; ++db) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < HT; ++kb) step(3 * NB + db * HT + kb, XH[kb], acc);
            if (valid) store_slice(a.D_out + rowD + 128 * db, quarter, half, acc);
        }
#pragma unroll
        for (int db = 0; db < DT; ++db) {
            f32x16 acc = load_slice(a.pre.bP + 128 * db, quarter, half);
#pragma unroll
            for (int kb = 0; kb < HT; ++kb) step(4 * NB + db * HT + kb, XH[kb], acc);
            if (valid) store_slice(a.P_out + rowD + 128 * db, quarter, half, acc);
        }
    } else {
        // ---- decoder: Lin(H, D) GELU Lin(D, 3); denormalise ---------------------------------------
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, XH);         // XH = h'
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < DT; ++db) {
            f32x16 acc = load_slice(a.dec_b1 + 128 * db, quarter, half);
#pragma unroll
            for (int kb = 0; kb < HT; ++kb) step(2 * NB + db * HT + kb, XH[kb], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 128 * db + 32 * quarter + 8 * q + 4 * half;
                f32x4 gl;
#pragma unroll
                for (int j = 0; j < 4; ++j) gl[j] = gamd_gelu_hw(acc[q * 4 + j]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.dec_w2 + c * DP + f0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[c] += w[j] * gl[j];
                }
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

template <int NFEAT, int EHT, int DT>
int enc_launch_d(const EncArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * (2 * GAMD_WFRAG_FLOATS + 2 * 128 * DT + 3 * 128 * EHT + 64);
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_edge_encode_wide_d<NFEAT, EHT, DT>)) return e;
    hipLaunchKernelGGL((k_edge_encode_wide_d<NFEAT, EHT, DT>), dim3(n_blocks), dim3(256), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

template <int EHT, int HT, int DT>
int conv_launch_d(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * (2 * GAMD_WFRAG_FLOATS + 128 + 128 * DT + 128 * HT);
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_conv_edge_wide_d<EHT, HT, DT>)) return e;
    hipLaunchKernelGGL((k_conv_edge_wide_d<EHT, HT, DT>), dim3(n_blocks), dim3(256), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_edge_encode_wide_d(const EncArgs& a, int eht, int dt, int n_blocks, hipStream_t st) {
    if (dt != 2 || a.e_format != 0) return -22;
    if (eht == 1) {
        switch (a.n_feat) {
            case 4: return enc_launch_d<4, 1, 2>(a, n_blocks, st);
            case 5: return enc_launch_d<5, 1, 2>(a, n_blocks, st);
            case 44: return enc_launch_d<44, 1, 2>(a, n_blocks, st);
            case 45: return enc_launch_d<45, 1, 2>(a, n_blocks, st);
        }
    } else if (eht == 2) {
        switch (a.n_feat) {
            case 4: return enc_launch_d<4, 2, 2>(a, n_blocks, st);
            case 5: return enc_launch_d<5, 2, 2>(a, n_blocks, st);
            case 44: return enc_launch_d<44, 2, 2>(a, n_blocks, st);
            case 45: return enc_launch_d<45, 2, 2>(a, n_blocks, st);
        }
    }
    return -22;
}

int launch_conv_edge_wide_d(const ConvEdgeArgs& a, int eht, int ht, int dt, int n_blocks, hipStream_t st) {
    if (dt != 2) return -22;
    if (eht == 1 && ht == 1) return conv_launch_d<1, 1, 2>(a, n_blocks, st);
    if (eht == 1 && ht == 2) return conv_launch_d<1, 2, 2>(a, n_blocks, st);
    if (eht == 2 && ht == 1) return conv_launch_d<2, 1, 2>(a, n_blocks, st);
    if (eht == 2 && ht == 2) return conv_launch_d<2, 2, 2>(a, n_blocks, st);
    return -22;
}

int launch_node_wide_d(const NodeArgs& a, int ht, int dt, hipStream_t st) {
    if (dt != 2 || a.f16x3) return -22;
    const int nb = (a.n + GAMD_TILE - 1) / GAMD_TILE;
    if (ht == 1) hipLaunchKernelGGL((k_node_wide_d<1, 2>), dim3(nb), dim3(256), 0, st, a);
    else if (ht == 2) hipLaunchKernelGGL((k_node_wide_d<2, 2>), dim3(nb), dim3(256), 0, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}
