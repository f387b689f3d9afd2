// wide.hip — the same path for widths other than 128/128/128.
//
// The DFT-water configuration of the reference (WaterMDDynamicBoxNet, nn_module.py:266-320, driven by
// water/test_script/test_nosehoover_hb.py:69-81) has node width H = encoding_size = 256, edge-embedding
// width Eh = 256, hidden_dim = 128, 5 conv layers, and may switch the RBF expansion off
// (`expand_edge=False`: 4 | 5 edge features, nn_module.py:322-336).  The kernels here are the generic-width
// siblings of edge_encode.hip / conv_edge.hip / node.hip:
//
//   H = 128*HT, Eh = 128*EHT (HT, EHT in {1, 2}), hidden_dim = 128.
//
// Every Linear is decomposed into 128x128 blocks in the fragment order of gamd_common.h, so the chain
// layout, the F1/F2 orientations and the partial-sum pieces carry over unchanged; a 256-wide operand is
// two K blocks accumulated into one output, a 256-wide result is two output blocks.  These kernels favour
// simplicity over the last 20 % (plain workgroup barriers, no cross-barrier prefetch): the shipped wide
// configuration is a 774-atom box where every launch is latency-bound anyway.
#include "gamd_common.h"
#include "gamd_bf16.h"
#include "gamd_f16x3.h"
#include "gamd_internal.h"
#include "gamd_wide.h"

namespace {

constexpr int WIDE_ENC_W1_FLOATS = 4 * 6 * 64 * 4;

// all of this wave's LDS DMA and loads have landed, then the workgroup meets
__device__ __forceinline__ void wide_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ================================================================================================
// edge features + edge encoder + LayerNorm(Eh)
//   MLP NFEAT -> 128 -> 128 -> Eh (GELU), nn_module.py:306-317; features nn_module.py:322-336 / :603-634
// ================================================================================================
// Waves per workgroup of the encoder: 8 (two per SIMD, 256 registers each) for Eh = 128; for Eh = 256 the LayerNorm needs both
// 128-wide output blocks of a tile at once — X + Y[0] + Y[1] = 192 registers next to the weight fragments in flight — which
// does not fit 256 registers (66 spilled registers in round 3): 4 waves, one per SIMD with the whole register file, no scratch.
template <int EHT> struct EncWaves { static constexpr int value = EHT == 2 ? 4 : 8; };

// LP: the reduced-precision edge modes (a.e_format 1 / 2): the two 128-wide GEMMs in split-fp16, e written in operand form
template <int NFEAT, int EHT, bool LP = false>
__global__ void __launch_bounds__(64 * EncWaves<EHT>::value) k_edge_encode_wide(EncArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    static_assert(EHT == 1 || EHT == 2, "edge embedding width 128 or 256");
    constexpr int EH = 128 * EHT;
    constexpr int NW = EncWaves<EHT>::value, NT = 64 * NW;
    constexpr bool EXPAND = NFEAT >= 44;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1 = lds;
    float* w2 = w1 + WIDE_ENC_W1_FLOATS;
    float* ws = w2 + GAMD_WFRAG_FLOATS;              // W3 block slot
    float* vb1 = ws + GAMD_WFRAG_FLOATS;
    float* vb2 = vb1 + 128;
    float* vb3 = vb2 + 128;
    float* vg = vb3 + EH;
    float* vbeta = vg + EH;
    float* cen = vbeta + EH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    for (int i = tid; i < WIDE_ENC_W1_FLOATS / 4; i += NT) ((f32x4*)w1)[i] = ((const f32x4*)a.w1p)[i];
    for (int i = tid; i < GAMD_WFRAG_FLOATS / 4; i += NT) ((f32x4*)w2)[i] = ((const f32x4*)a.w2p)[i];
    if (EHT == 1)
        for (int i = tid; i < GAMD_WFRAG_FLOATS / 4; i += NT) ((f32x4*)ws)[i] = ((const f32x4*)a.w3p)[i];
    if (tid < 128) { vb1[tid] = a.b1[tid]; vb2[tid] = a.b2[tid]; }
    for (int i = tid; i < EH; i += NT) { vb3[i] = a.b3[i]; vg[i] = a.ln_g[i]; vbeta[i] = a.ln_b[i]; }
    if (EXPAND && tid < 40) cen[tid] = a.centers[tid];
    __syncthreads();

    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    const int n_wg_tiles = (n_tiles + NW - 1) / NW;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    constexpr int KSTEPS = (NFEAT + 1) / 2;

    for (int wt = first; wt < end; wt += step) {          // uniform over the workgroup (barriers inside)
        asm volatile("" ::: "memory");
        const int tile = wt * NW + wave;
        const bool active = tile < n_tiles;
        const long long x = (long long)tile * GAMD_TILE + gamd_pi(slot);
        const bool valid = active && x < E;
        const int src = valid ? GAMD_CHK_RANGE(a.sticky, a.col[x], 0, a.zero_row, GAMD_CHK_ENC_SRC) : 0, dst = valid ? GAMD_CHK_RANGE(a.sticky, a.erow[x], 0, a.zero_row, GAMD_CHK_ENC_DST) : 0;
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
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[t][r] = gamd_gelu_hw(acc[t][r]);
        load_bias_chain(vb2, half, acc);
        // (reduced-precision edge modes, a.e_format != 0: the two 128-wide GEMMs of the encoder in split-fp16 -- fp32-grade at 3/16
        //  of the fp32 matrix time; W2 / W3 blocks arrive as (hi | lo) fp16 images of the same size; the K = 48 first layer stays fp32)
        if (LP) gemm128_f16x3<false>((const f16x8*)w2, lane, X, acc);
        else gemm128<false>((const f32x4*)w2, lane, X, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[t][r] = gamd_gelu_hw(acc[t][r]);

        f32x16 Y[EHT][4];
#pragma unroll
        for (int ob = 0; ob < EHT; ++ob) {
            if (EHT > 1) {
                wide_barrier();                            // previous readers of the slot are done
                gamd_stage_weight<NW>(a.w3p + (size_t)ob * GAMD_WFRAG_FLOATS, ws, wave, lane16);
                wide_barrier();
            }
            load_bias_chain(vb3 + 128 * ob, half, Y[ob]);
            unsigned ws_off = (unsigned)((WIDE_ENC_W1_FLOATS + GAMD_WFRAG_FLOATS) * sizeof(float));
            asm volatile("" : "+s"(ws_off));                 // one base register + immediate offsets (see k_conv_edge_wide)
            if (LP) gemm128_f16x3<false>((const f16x8*)((const char*)lds + ws_off), lane, X, Y[ob]);
            else gemm128<false>((const f32x4*)((const char*)lds + ws_off), lane, X, Y[ob]);
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
                for (int t = 0; t < 4; ++t) {
                    f32x16 nv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 128 * ob + 32 * t + 8 * q + 4 * half;
                        const f32x4 g = *reinterpret_cast<const f32x4*>(&vg[f0]);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(&vbeta[f0]);
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = zero_row ? 0.f : (Y[ob][t][q * 4 + j] - mean) * rstd * g[j] + b[j];
                            nv[q * 4 + j] = v[j];
                        }
                        if (!LP) out[((ob * 4 + t) * 4 + q) * 64 + lane] = v;
                    }
                    if (LP && a.e_format == 1) {
                        // bf16 edge MLP (wide_lp.hip): e as bf16 fragments, [tile][block][t][u][lane][8 values]
                        bf16x8* efrag = reinterpret_cast<bf16x8*>(a.e_frag);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            gamd_u32x4 w;
#pragma unroll
                            for (int k = 0; k < 4; ++k) w[k] = gamd_pk_bf16(nv[8 * u + 2 * k], nv[8 * u + 2 * k + 1]);
                            efrag[(((size_t)tile * EHT + ob) * 8 + t * 2 + u) * 64 + lane] = __builtin_bit_cast(bf16x8, w);
                        }
                    }
                    if (LP && a.e_format == 2) {
                        // split-fp16 edge MLP (wide_lp.hip / conv_edge_f16x3.hip): e already split into (hi, lo) fp16 operand
                        // images, [tile][block][t][u][hi | lo][lane][8 halves] -- the layout edge_encode_f16x3.hip writes
                        f16x8* efrag = reinterpret_cast<f16x8*>(a.e_frag);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            f16x8 eh, el;
                            gamd_split8(nv, u, eh, el);
                            efrag[((((size_t)tile * EHT + ob) * 8 + t * 2 + u) * 2 + 0) * 64 + lane] = eh;
                            efrag[((((size_t)tile * EHT + ob) * 8 + t * 2 + u) * 2 + 1) * 64 + lane] = el;
                        }
                    }
                }
        }
    }
}

// ================================================================================================
// conv layer, edge side (nn_module.py:135-142) for Eh = 128*EHT, H = 128*HT
//   weight blocks (64 KiB each, contiguous at a.w1p):  W1[:, kb] (EHT) | W2 | W3 | W4[ob, :] (HT)
// ================================================================================================
template <int EHT, int HT>
__global__ void __launch_bounds__(512, 2) k_conv_edge_wide(ConvEdgeArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;          // frozen run: nothing to compute until the host has regrown and resumed
    constexpr int NP = EHT + 2 + HT;
    constexpr int H = 128 * HT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* vb1 = lds + 2 * GAMD_WFRAG_FLOATS;
    float* vb3 = vb1 + 128;
    float* vb4 = vb3 + 128;

    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    int E = a.counters[CNT_E];
    if ((long long)E > a.e_cap) E = (int)a.e_cap;
    const int n_tiles = (E + GAMD_TILE - 1) / GAMD_TILE;
    const int n_wg_tiles = (n_tiles + 7) / 8;
    int first, end, step;
    gamd_xcd_range(n_wg_tiles, blockIdx.x, gridDim.x, first, end, step);
    if (first >= end) return;

    if (tid < 128) { vb1[tid] = a.b1[tid]; vb3[tid] = a.b3[tid]; }
    if (tid < H) vb4[tid] = a.b4[tid];
    gamd_stage_weight_raw_contig<8>(a.w1p, lds, wave, lane16);
    wide_barrier();

    unsigned g = 0;                    // running phase counter: block g % NP sits in slot g & 1
    int blk = 0;
    // start the DMA of the next block into the other slot, hand back this phase's slot.  The copy is issued from inline
    // assembly: a compiler-tracked global_load_lds turns the wait of this phase's first weight read into vmcnt(0), i.e. every
    // phase would sit out the copy it has just issued for the NEXT one (conv_edge.hip); its landing is guaranteed by the
    // vmcnt(0) of wide_barrier at the end of the phase.
    auto begin_phase = [&]() -> const f32x4* {
        const int nb = (blk + 1 == NP) ? 0 : blk + 1;
        gamd_stage_weight_raw_contig<8>(a.w1p + (size_t)nb * GAMD_WFRAG_FLOATS, lds + ((g + 1) & 1u) * GAMD_WFRAG_FLOATS, wave, lane16);
        // The slot offset stays a run-time value: with an even number of phases per tile (EHT + 2 + HT = 4 or 6) the slot of
        // every phase is a compile-time constant, and hipcc then materialises one address register per 1 KiB fragment of the
        // upper slot (its byte offsets >= 65 536 do not fit the 16-bit offset field of ds_read_b128): ~30 loop-invariant
        // registers, 8-14 of them spilled to scratch in round 3.  One base register + immediate offsets instead.
        unsigned off = (g & 1u) * (unsigned)(GAMD_WFRAG_FLOATS * sizeof(float));
        asm volatile("" : "+s"(off));
        return (const f32x4*)((const char*)lds + off);
    };
    auto end_phase = [&]() { wide_barrier(); ++g; blk = (blk + 1 == NP) ? 0 : blk + 1; };

    for (int wt = first; wt < end; wt += step) {
        asm volatile("" ::: "memory");
        const int tile = wt * 8 + wave;
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

        f32x16 T[4], U[4], X[4];
        // ---- T = SiLU(W1 e + b1), K = Eh --------------------------------------------------------
        load_bias_chain(vb1, half, T);
#pragma unroll
        for (int kb = 0; kb < EHT; ++kb) {
            const f32x4* W = begin_phase();
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
                gemm128<false>(W, lane, X, T);
            }
            end_phase();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[t][r] = gamd_silu_hw(T[t][r]);
        // ---- U = SiLU(W2 T + S[src] + D[dst]) ---------------------------------------------------
        {
            const f32x4* W = begin_phase();
            if (active) {
                load_row_chain(a.S + (size_t)src * 128, half, U);
                load_row_chain(a.D + (size_t)dst * 128, half, X);
#pragma unroll
                for (int t = 0; t < 4; ++t) U[t] += X[t];
                gemm128<false>(W, lane, T, U);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) U[t][r] = gamd_silu_hw(U[t][r]);
            }
            end_phase();
        }
        // ---- T = SiLU(W3 U + b3) ----------------------------------------------------------------
        {
            const f32x4* W = begin_phase();
            if (active) {
                load_bias_chain(vb3, half, T);
                gemm128<false>(W, lane, U, T);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[t][r] = gamd_silu_hw(T[t][r]);
            }
            end_phase();
        }
        // ---- e_emb block ob = T W4[ob]^T + b4 (F2), message with hn[src], segment sum -----------
#pragma unroll
        for (int ob = 0; ob < HT; ++ob) {
            const f32x4* W = begin_phase();
            if (active) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int s = __shfl(src, rho, 64);
                    const float* hrow = a.hn + (size_t)s * H + 128 * ob + slot;
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) X[tp][r] = hrow[32 * tp];
                }
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const float b = vb4[128 * ob + 32 * tp + slot];
#pragma unroll
                    for (int r = 0; r < 16; ++r) U[tp][r] = b;
                }
                gemm128<true>(W, lane, T, U);
                if (a.emb_out) {                           // update_edge_emb: e_emb rows for launch_edge_update (edge x0 + r)
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
                    for (int r = 0; r < 16; ++r) {
                        U[tp][r] = gamd_msg_acc((r < nvalid) ? X[tp][r] : 0.f, U[tp][r], (r > 0 && ((keep_bits >> r) & 1u)) ? U[tp][r - 1] : 0.f);
                    }
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
            end_phase();
        }
    }
}

// ================================================================================================
// update_edge_emb (nn_module.py:140-146): e for the next layers = edge_layer_norm(e_emb), from the conv kernel's [E][H]
// rows back into the encoder's fragment order (lane = edge gamd_pi(slot), registers = features 4 half + j of each 8-block).
// One wave per 32-edge tile; a lane reads the half of its edge's row it will write, the other half's sums arrive by DPP.
// A dead branch in every shipped configuration (SURVEY a-9): written for correctness, two extra passes over E x H floats.
// ================================================================================================
template <int HT>
__global__ void __launch_bounds__(256) k_edge_update(EdgeUpdateArgs a) {
    if (a.devflags[DEVFLAG_FROZEN]) return;
    constexpr int H = 128 * HT;
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const int n_tiles = (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const long long x = (long long)tile * GAMD_TILE + gamd_pi(slot);      // rows up to the end of the last tile exist
        const f32x4* row = reinterpret_cast<const f32x4*>(a.emb + (size_t)x * H) + half;
        f32x4 v[HT * 16];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < HT * 16; ++i) {                                   // i = (ob*4 + t)*4 + q: features 8 i + 4 half + j
            v[i] = row[2 * i];
            s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = gamd_xhalf_sum(s1) * a.ln_inv_width;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < HT * 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; s2 += d * d; }
        const float rstd = 1.0f / sqrtf((gamd_xhalf_sum(s2) - a.ln_n_pad * (mean * mean)) * a.ln_inv_width + 1e-5f);
        f32x4* out = (f32x4*)a.e_frag_out + (size_t)tile * HT * 16 * 64;
#pragma unroll
        for (int i = 0; i < HT * 16; ++i) {
            const f32x4 g = reinterpret_cast<const f32x4*>(a.ln_g)[2 * i + half], b = reinterpret_cast<const f32x4*>(a.ln_b)[2 * i + half];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
            out[i * 64 + lane] = o;
        }
    }
}

// ================================================================================================
// node side (see node.hip) for H = 128*HT: 32-atom tile over 4 waves by output quarter of each block
// ================================================================================================
// split-fp16 form (the reduced-precision edge mode of the generic widths, wide_lp.hip): the block is a (hi | lo) fp16 image
// (pack128_f16x3), a quarter = 8 + 8 fragments of 16 bytes per lane; 24 MFMAs of 32 cycles per block GEMM instead of 64 of 64.
// The activations are split per K step on the fly (gamd_split8): 160 VALU per block GEMM, nothing next to the MFMAs saved.
__device__ __forceinline__ void wq_load_f16(const float* __restrict__ Wp, int quarter, int lane, WQ& o) {
    const f32x4* W = reinterpret_cast<const f32x4*>(Wp) + (size_t)quarter * 8 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) { o.w[i] = W[i * 64]; o.w[8 + i] = W[2048 + i * 64]; }
}
__device__ __forceinline__ void wq_gemm_f16(const WQ& wq, const f32x16 (&X)[4], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 xh, xl;
            gamd_split8(X[t], u, xh, xl);
            const f16x8 wh = __builtin_bit_cast(f16x8, wq.w[t * 2 + u]), wl = __builtin_bit_cast(f16x8, wq.w[8 + t * 2 + u]);
            acc = mfma_f16(wh, xl, acc);
            acc = mfma_f16(wl, xh, acc);
            acc = mfma_f16(wh, xh, acc);
        }
}
template <int HT, bool F16 = false>
__global__ void __launch_bounds__(256) k_node_wide(NodeArgs a) {
    constexpr int H = 128 * HT;
    constexpr int XLDW = H + 4;
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
    const size_t rowH = (size_t)atom * H, rowD = (size_t)atom * 128;

    f32x16 X[HT][4];
    f32x16 mine[HT];
    // Weight quarters travel one block-GEMM ahead: step s of the kernel's GEMM sequence (phi_edge kb | phi ob | S kb | D kb |
    // P kb, or the decoder's kb) runs from buffer s & 1 while the quarter of step s + 1 is fetched from L2 into the other
    // one — also across the exchanges and the LayerNorm between the stages.  (Fetched right in front of its GEMM, as in
    // rounds 1-2, every one of the 5 HT block GEMMs sat out an L2 round trip: 39 us per launch on the 774-atom system.)
    WQ wqa, wqb;
    constexpr size_t BLK = GAMD_WFRAG_FLOATS;
    auto wptr = [&](int st) -> const float* {               // weight block of step st (st >= 2 HT: pre(l) or decoder)
        if (st < HT) return a.post.wpep + (size_t)st * BLK;
        if (st < 2 * HT) return a.post.wphip + (size_t)(st - HT) * BLK;
        const int i = st - 2 * HT;
        if (a.mode == 2) return i < HT ? a.dec_w1p + (size_t)i * BLK : nullptr;
        if (i < HT) return a.pre.wsp + (size_t)i * BLK;
        if (i < 2 * HT) return a.pre.wdp + (size_t)(i - HT) * BLK;
        return i < 3 * HT ? a.pre.wpdp + (size_t)(i - 2 * HT) * BLK : nullptr;
    };
    // step st: fetch step st + 1's quarter, then the 64 MFMAs of this one
    auto step = [&](int st, const f32x16 (&Xb)[4], f32x16& acc) {
        WQ& cur = (st & 1) ? wqb : wqa;
        WQ& nxt = (st & 1) ? wqa : wqb;
        const float* np_ = wptr(st + 1);
        asm volatile("" ::: "memory");
        if (np_) { if (F16) wq_load_f16(np_, quarter, lane, nxt); else wq_load(np_, quarter, lane, nxt); }
        asm volatile("" ::: "memory");
        if (F16) wq_gemm_f16(cur, Xb, acc); else wq_gemm(cur, Xb, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        const float* w0 = a.mode == 0 ? wptr(2 * HT) : wptr(0);          // (2 HT is even: buffer a)
        if (F16) wq_load_f16(w0, quarter, lane, wqa); else wq_load(w0, quarter, lane, wqa);
    }

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
        f32x16 acc = load_slice(a.P_in + rowD, quarter, half);
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
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, X);          // X = agg
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) step(kb, X[kb], acc);                 // phi_edge: H -> 128
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gamd_silu_hw(acc[r]);
        f32x16 one[1] = {acc};
        f32x16 X1[1][4];
        exchange_blocks<1, XLDW>(xbuf, quarter, slot, half, one, X1);           // X1 = SiLU(P + phi_edge(agg))
#pragma unroll
        for (int ob = 0; ob < HT; ++ob) {                                     // phi: 128 -> H, residual
            mine[ob] = load_slice(a.post.bphi + 128 * ob, quarter, half);
            step(HT + ob, X1[0], mine[ob]);
            mine[ob] += h_res[ob];
            if (valid) store_slice(a.h_out + rowH + 128 * ob, quarter, half, mine[ob]);
        }
    }

    if (a.mode != 2) {
        // ---- pre(l): LayerNorm over H, then S, D, P (H -> 128 each) ------------------------------
        // (norm_bn: eval-mode BatchNorm1d folded by the host into ln_g / ln_b, no row statistics -- see node.hip)
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
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, X);          // X = hn
        f32x16 acc = load_slice(a.pre.bS, quarter, half);
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) step(2 * HT + kb, X[kb], acc);
        if (valid) store_slice(a.S_out + rowD, quarter, half, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) step(3 * HT + kb, X[kb], acc);
        if (valid) store_slice(a.D_out + rowD, quarter, half, acc);
        acc = load_slice(a.pre.bP, quarter, half);
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) step(4 * HT + kb, X[kb], acc);
        if (valid) store_slice(a.P_out + rowD, quarter, half, acc);
    } else {
        // ---- decoder: Lin(H,128) GELU Lin(128,3); denormalise ------------------------------------
        exchange_blocks<HT, XLDW>(xbuf, quarter, slot, half, mine, X);          // X = h'
        f32x16 acc = load_slice(a.dec_b1, quarter, half);
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) step(2 * HT + kb, X[kb], acc);
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * quarter + 8 * q + 4 * half;
            f32x4 gl;
#pragma unroll
            for (int j = 0; j < 4; ++j) gl[j] = gamd_gelu_hw(acc[q * 4 + j]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.dec_w2 + c * 128 + f0);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[c] += w[j] * gl[j];
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

template <int NFEAT, int EHT, bool LP>
int enc_launch2(const EncArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * (WIDE_ENC_W1_FLOATS + 2 * GAMD_WFRAG_FLOATS + 256 + 3 * 128 * EHT + 64);
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_edge_encode_wide<NFEAT, EHT, LP>)) return e;
    hipLaunchKernelGGL((k_edge_encode_wide<NFEAT, EHT, LP>), dim3(n_blocks), dim3(64 * EncWaves<EHT>::value), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
template <int NFEAT, int EHT>
int enc_launch(const EncArgs& a, int n_blocks, hipStream_t st) {
    return a.e_format != 0 ? enc_launch2<NFEAT, EHT, true>(a, n_blocks, st) : enc_launch2<NFEAT, EHT, false>(a, n_blocks, st);
}

template <int EHT, int HT>
int conv_launch(const ConvEdgeArgs& a, int n_blocks, hipStream_t st) {
    const size_t lds = sizeof(float) * (2 * GAMD_WFRAG_FLOATS + 256 + 128 * HT);
    static PerDeviceOnce once;
    if (int e = gamd_allow_dynamic_lds(once, (int)lds, k_conv_edge_wide<EHT, HT>)) return e;
    hipLaunchKernelGGL((k_conv_edge_wide<EHT, HT>), dim3(n_blocks), dim3(512), lds, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_edge_encode_wide(const EncArgs& a, int eht, int n_blocks, hipStream_t st) {
    if (eht == 1) {
        switch (a.n_feat) {
            case 4: return enc_launch<4, 1>(a, n_blocks, st);
            case 5: return enc_launch<5, 1>(a, n_blocks, st);
            case 44: return enc_launch<44, 1>(a, n_blocks, st);
            case 45: return enc_launch<45, 1>(a, n_blocks, st);
        }
    } else if (eht == 2) {
        switch (a.n_feat) {
            case 4: return enc_launch<4, 2>(a, n_blocks, st);
            case 5: return enc_launch<5, 2>(a, n_blocks, st);
            case 44: return enc_launch<44, 2>(a, n_blocks, st);
            case 45: return enc_launch<45, 2>(a, n_blocks, st);
        }
    }
    return -22;
}

int launch_conv_edge_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st) {
    if (eht == 1 && ht == 1) return conv_launch<1, 1>(a, n_blocks, st);
    if (eht == 1 && ht == 2) return conv_launch<1, 2>(a, n_blocks, st);
    if (eht == 2 && ht == 1) return conv_launch<2, 1>(a, n_blocks, st);
    if (eht == 2 && ht == 2) return conv_launch<2, 2>(a, n_blocks, st);
    return -22;
}

int launch_edge_update(const EdgeUpdateArgs& a, int ht, int n_blocks, hipStream_t st) {
    if (ht == 1) hipLaunchKernelGGL(k_edge_update<1>, dim3(n_blocks), dim3(256), 0, st, a);
    else if (ht == 2) hipLaunchKernelGGL(k_edge_update<2>, dim3(n_blocks), dim3(256), 0, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_node_wide(const NodeArgs& a, int ht, hipStream_t st) {
    const int nb = (a.n + GAMD_TILE - 1) / GAMD_TILE;
    if (ht == 1 && !a.f16x3) hipLaunchKernelGGL((k_node_wide<1, false>), dim3(nb), dim3(256), 0, st, a);
    else if (ht == 2 && !a.f16x3) hipLaunchKernelGGL((k_node_wide<2, false>), dim3(nb), dim3(256), 0, st, a);
    else if (ht == 1) hipLaunchKernelGGL((k_node_wide<1, true>), dim3(nb), dim3(256), 0, st, a);
    else if (ht == 2) hipLaunchKernelGGL((k_node_wide<2, true>), dim3(nb), dim3(256), 0, st, a);
    else return -22;
    GAMD_CHECK_LAUNCH();
    return 0;
}
