// ws_chain_probe.hip — ceiling of a WEIGHT-STATIONARY conv-edge GEMM chain on gfx950.
//
// The shipped k_conv_edge streams the layer's four 64 KiB matrices L2 -> LDS (2-slot ring, barrier per GEMM) and every wave
// reads the whole matrix from LDS per tile.  Here the roles are swapped: one wave per SIMD (512 registers), wave w keeps
// output block w (32 features x 128 K = 64 registers) of ALL FOUR matrices in accumulation registers for the whole launch,
// and the 32-edge x 128-feature activation tiles travel through LDS in fragment order (the e_frag layout): per (tile, stage)
// job a wave reads the 16 KiB input tile (16 ds_read_b128, rolling: block t of the next job is fetched as soon as block t
// of the current one has been consumed), runs 64 MFMAs, applies SiLU to its 16 output registers and writes them back (4
// ds_write_b128).  Tiles are processed in batches of four, stage by stage, with a workgroup barrier every two jobs; no
// weight traffic at all after the prologue, work granularity = one tile per CU (launch tail 0.6 % instead of 3 % at C2).
// The post-op of job k (SiLU, LDS write) is issued inside the MFMA stream of job k + 1 (two accumulator sets), so the
// matrix pipe never drains at a job boundary.
//
// mode bits: 1 = e tiles streamed from global memory into LDS with global_load_lds (16 KiB per tile);
//            2 = S[src] / D[dst] quarter gathers for stage 1 and hn[src] gathers for stage 3 (random rows of a 10 000 x 128 table)
//            4 = SiLU -> multiply; 8 = no barriers (timing only)
#include "../gamd_common.h"
#include <cstdio>
#include <vector>
#include <type_traits>

__device__ __forceinline__ float silu_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr int G = 4;                                   // tiles per batch
constexpr int TILE_FLOATS = 32 * 128;                  // 16 KiB

__device__ __forceinline__ void read_block(const float* xin, int t, int lane, f32x16& x) {
    const f32x4* p = reinterpret_cast<const f32x4*>(xin) + t * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = p[q * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[q * 4 + j] = v[j];
    }
}

// MFMA with the weight operand in an ACCUMULATION register (hipcc never allocates the A / B operands of an MFMA builtin to
// AGPRs: with builtins the 256 weight registers are parked in AGPRs and copied back with ~36 v_accvgpr_read per job).
// The hazard recogniser cannot see inside inline assembly: a result is touched by VALU / LDS only behind >= 16 further
// MFMAs (guard() pins that point), far beyond the 18 wait states an XDL write needs.
template <bool F2>
__device__ __forceinline__ void mfma_w(f32x16& acc, float w, float x) {
    if (F2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %1, %0" : "+v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
// first MFMA of a chain: C-in from other registers (bias / gathered rows), no copy
template <bool F2>
__device__ __forceinline__ void mfma_w_first(f32x16& acc, float w, float x, const f32x16& cin) {
    if (F2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %1, %3" : "=&v"(acc) : "a"(w), "v"(x), "v"(cin));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(x), "v"(cin));
}
__device__ __forceinline__ void guard(f32x16& acc) { asm volatile("" : "+v"(acc)); }

__device__ __forceinline__ void write_quarter(float* xout, int w, int lane, const f32x16& acc) {
    f32x4* p = reinterpret_cast<f32x4*>(xout) + w * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[q * 4 + j];
        p[q * 64] = v;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 1) k_ws(const float* __restrict__ W4, const float* __restrict__ table,
                                               const float* __restrict__ estream, float* __restrict__ out, int n_batches) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* setA = lds;                                  // stage 0 / 2 input
    float* setB = lds + G * TILE_FLOATS;                // stage 1 / 3 input
    const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;

    WQuarter wq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) load_wquarter(W4 + (size_t)m * GAMD_WFRAG_FLOATS, w, lane, wq[m]);
    f32x16 b1, b3, b4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { b1[r] = 0.01f * (r + half); b3[r] = -0.01f * (r + w); b4[r] = 0.02f * slot; }

    for (int i = tid; i < 2 * G * TILE_FLOATS; i += 256) lds[i] = 0.001f * ((i * 7 + blockIdx.x) % 97) - 0.04f;
    __syncthreads();

    auto dma_tile = [&](int batch, int i) {            // this wave's quarter (4 KiB) of e tile -> setA[i]
        const float* g = estream + ((size_t)((blockIdx.x * 64 + (batch & 63)) * G + i)) * TILE_FLOATS;
        const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(setA + i * TILE_FLOATS);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int chunk = w * 4 + k;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         ::"v"(lane16), "s"(reinterpret_cast<const char*>(g) + chunk * 1024), "s"(l0 + chunk * 1024u) : "memory");
        }
    };

    f32x16 X[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) read_block(setA, t, lane, X[t]);
    f32x16 acc[2];
    acc[1] = b4;
    float chk = 0.f;
    unsigned rng = blockIdx.x * 2654435761u + 12345u;
    f32x16 sq[2], dq, hq[2];
    sq[0] = sq[1] = dq = hq[0] = hq[1] = b1;

    for (int b = 0; b < n_batches; ++b) {
        // per-tile source / destination rows of this lane's edge (stand-in for col / erow)
        unsigned src[G], dst[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            rng = rng * 1664525u + 1013904223u;
            const unsigned base = (rng >> 8) % 9000u;                     // neighbours are spatially close: rows near a base
            src[i] = base + ((slot * 37u + i * 11u) % 600u);
            dst[i] = base + (slot >> 3);
        }
        auto gather_s = [&](int i) { if (MODE & 2) sq[i & 1] = load_slice(table + (size_t)src[i] * 128, w, half); };
        auto gather_d = [&](int i) { if (MODE & 2) dq = load_slice(table + (size_t)dst[i] * 128, w, half); };
        auto gather_hn = [&](int i) {
            if (MODE & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rho = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const unsigned s = __shfl(src[i], rho, 64);
                    hq[i & 1][r] = table[(size_t)s * 128 + 32 * w + slot];
                }
            }
        };
        // post-op of job (s, i) whose raw result is in a: SiLU + LDS write (stages 0-2), message + segment sum (stage 3)
        auto post = [&](int s, int i, f32x16& a) {
            float* xout = (s & 1) ? setA : setB;
            if (s < 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (s == 1 && (MODE & 2)) ? a[r] + sq[i & 1][r] : a[r];
                    a[r] = (MODE & 4) ? v * 0.5f : silu_hw(v);
                }
                write_quarter(xout + i * TILE_FLOATS, w, lane, a);
            } else {
                float run = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { run = ((r & 3) ? run : 0.f) + a[r] * ((MODE & 2) ? hq[i & 1][r] : 1.0f); a[r] = run; }
                chk += a[3] + a[7] + a[11] + a[15];
            }
        };
        auto job = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int s = k >> 2, i = k & 3;
            constexpr int kp = (k + 15) & 15, sp = kp >> 2, ip = kp & 3;      // previous job (post-op pending in acc[kp & 1])
            const float* xin = (s & 1) ? setB : setA;
            const float* xin_next = (s & 1) ? setA : setB;              // input set of stage s + 1 (next batch's stage 0 -> setA)
            const float* nx = (i + 1 < G) ? xin + (i + 1) * TILE_FLOATS : xin_next;
            f32x16& a = acc[k & 1];
            f32x16& ap = acc[kp & 1];
            const WQuarter& wk = wq[s];
            // gathers: D[dst] of the next stage-1 job (its C-in), S[src] for its post-op, hn[src] for the next stage-3 post-op
            if (s == 0 && i == G - 1) { gather_d(0); gather_s(0); }
            if (s == 1 && i + 1 < G) gather_s(i + 1);
            if (s == 2 && i == G - 1) gather_hn(0);
            if (s == 3 && i + 1 < G) gather_hn(i + 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float wv = wk.w[t * 4 + q][j], xv = X[t][q * 4 + j];
                        if (t == 0 && q == 0 && j == 0) {
                            const f32x16& cin = s == 0 ? b1 : (s == 1 ? ((MODE & 2) ? dq : b3) : (s == 2 ? b3 : b4));
                            if (s == 3) mfma_w_first<true>(a, wv, xv, cin); else mfma_w_first<false>(a, wv, xv, cin);
                        } else {
                            if (s == 3) mfma_w<true>(a, wv, xv); else mfma_w<false>(a, wv, xv);
                        }
                    }
                if (t == 0) {
                    guard(ap);                       // the previous job's result: 16 MFMAs old
                    if (s == 1 && i + 1 < G) gather_d(i + 1);          // dq is free again (consumed as C-in by the first MFMA)
                    post(sp, ip, ap);
                }
                if (t == 1 && (i == 0 || i == 2)) {
                    // the writes of the two previous jobs (and this job's block-0 prefetch) are >= 16 MFMAs old
                    if ((MODE & 1) && ((s == 0 && i == 0) || (s == 3 && i == 2))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // e tiles issued two barriers ago have landed
                    if (!(MODE & 8)) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    if ((MODE & 1) && s == 3) { dma_tile(b + 1, i); dma_tile(b + 1, i + 1); }
                }
                read_block(nx, t, lane, X[t]);
            }
        };
#define JOB(K) job(std::integral_constant<int, K>{})
        JOB(0); JOB(1); JOB(2); JOB(3); JOB(4); JOB(5); JOB(6); JOB(7);
        JOB(8); JOB(9); JOB(10); JOB(11); JOB(12); JOB(13); JOB(14); JOB(15);
#undef JOB
    }
    float s = chk;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += X[t][r] + acc[0][r] + acc[1][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
double run(const float* dW, const float* dT, const float* dE, float* dOut, int n_batches) {
    const size_t ldsb = sizeof(float) * 2 * G * TILE_FLOATS;
    hipFuncSetAttribute((const void*)k_ws<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_ws<MODE><<<256, 256, ldsb>>>(dW, dT, dE, dOut, 2);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 0; }
    hipEventRecord(e0);
    k_ws<MODE><<<256, 256, ldsb>>>(dW, dT, dE, dOut, n_batches);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * n_batches * G * 4.0 * 256 * 4096.0;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<float> W(4 * GAMD_WFRAG_FLOATS);
    for (size_t i = 0; i < W.size(); ++i) W[i] = ((i * 2654435761u) % 1000) * 1e-5f - 0.005f;
    std::vector<float> T(10000 * 128);
    for (size_t i = 0; i < T.size(); ++i) T[i] = ((i * 40503u) % 1000) * 1e-3f - 0.5f;
    const size_t stream_floats = (size_t)256 * 64 * G * TILE_FLOATS;      // 1 GiB of e tiles
    float *dW, *dT, *dE, *dOut;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dT, T.size() * 4); hipMalloc(&dE, stream_floats * 4); hipMalloc(&dOut, 256 * 256 * 4);
    hipMemset(dE, 0, stream_floats * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice);
    const int nb = 20;            // 80 tiles per workgroup (C2: 77.5)
    for (int rep = 0; rep < 3; ++rep) {
        printf("ws mode0 chain only                               : %.1f TF\n", run<0>(dW, dT, dE, dOut, nb));
        printf("ws mode4 chain, SiLU -> multiply                  : %.1f TF\n", run<4>(dW, dT, dE, dOut, nb));
        printf("ws mode8 chain, no barriers                       : %.1f TF\n", run<8>(dW, dT, dE, dOut, nb));
        printf("ws mode1 + e tiles streamed into LDS              : %.1f TF\n", run<1>(dW, dT, dE, dOut, nb));
        printf("ws mode2 + S/D/hn gathers                         : %.1f TF\n", run<2>(dW, dT, dE, dOut, nb));
        printf("ws mode3 + both                                   : %.1f TF\n", run<3>(dW, dT, dE, dOut, nb));
    }
    return 0;
}
