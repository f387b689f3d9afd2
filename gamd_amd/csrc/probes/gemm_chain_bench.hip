// Probe: where does the conv-edge kernel lose MFMA throughput?  Times building blocks in isolation.
//   mode 0: pure MFMA, 4 independent accumulators, operands in registers          (chip ceiling)
//   mode 1: gemm128 from LDS-resident packed weights, chained, no barrier, no activation
//   mode 2: mode 1 + hardware SiLU after every GEMM
//   mode 3: mode 2 + one __syncthreads and one 64 KiB global_load_lds restage per GEMM (the kernel's phase)
//   mode 4: mode 3 without SiLU
#include "../gamd_common.h"
#include <cstdio>
#include <vector>

__device__ __forceinline__ float silu_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// explicit one-group-ahead prefetch of the weight fragment
template <bool F2, typename WPtr>
__device__ __forceinline__ void gemm128_pf(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4]) {
    f32x4 w = W[lane];
#pragma unroll
    for (int g = 0; g < 64; ++g) {
        const int tp = g >> 4, t = (g >> 2) & 3, q = g & 3;
        f32x4 wn = w;
        if (g + 1 < 64) wn = W[(g + 1) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tp] = mfma32(w[j], X[t][q * 4 + j], acc[tp]);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        w = wn;
    }
}

// software-pipelined post-op as in conv_edge.hip
template <bool F2, typename WPtr, typename Post>
__device__ __forceinline__ void gemm128_post(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4], Post post) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = W[((tp * 4 + t) * 4 + q) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[tp] = mfma32(w[j], X[t][q * 4 + j], acc[tp]);
                if (tp > 0) post(tp - 1, t * 4 + q);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

// pipelined post-op + explicit one-group-ahead weight prefetch
template <bool F2, typename WPtr, typename Post>
__device__ __forceinline__ void gemm128_post_pf(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4], Post post) {
    f32x4 w = W[lane];
#pragma unroll
    for (int g = 0; g < 64; ++g) {
        const int tp = g >> 4, t = (g >> 2) & 3, q = g & 3;
        f32x4 wn = w;
        if (g + 1 < 64) wn = W[(g + 1) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tp] = mfma32(w[j], X[t][q * 4 + j], acc[tp]);
        if (tp > 0) post(tp - 1, t * 4 + q);
        w = wn;
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const float* __restrict__ W, float* __restrict__ out, int iters, const float* __restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    (void)G;
    f32x16 X[4], acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { X[t][r] = 0.001f * (lane + r + t); acc[t][r] = 0.f; }
    if (MODE == 0) {
        float a = 0.5f + lane * 1e-3f, b = 0.25f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma32(a, b, acc[t]);
            }
        }
    } else {
        float* buf0 = lds; float* buf1 = lds + GAMD_WFRAG_FLOATS;
        for (int i = tid; i < GAMD_WFRAG_FLOATS / 4; i += 512) { ((f32x4*)buf0)[i] = ((const f32x4*)W)[i]; ((f32x4*)buf1)[i] = ((const f32x4*)W)[i]; }
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            float* cur = (it & 1) ? buf1 : buf0;
            float* nxt = (it & 1) ? buf0 : buf1;
            if (MODE == 3 || MODE == 4 || MODE == 6 || MODE == 9 || MODE == 11 || MODE == 12) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int chunk = kk * 8 + wave;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + chunk * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(nxt + chunk * 256), 16, 0, 0);
                }
            }
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
            if (MODE == 11 || MODE == 12) {
                // mode 9 + per-GEMM streamed loads consumed by the post-op (like e / S / D): 16 x b128 per lane
                const size_t tilebase = ((size_t)(blockIdx.x * 8 + wave) * 64 + (it & 63)) * 4096;
                f32x16 S[4];
                const f32x4* gp = (const f32x4*)(G + tilebase);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = gp[(t * 4 + q) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j) S[t][q * 4 + j] = v[j];
                    }
                if (MODE == 12) {      // plus 64 dword gathers (hn-like), row chosen pseudo-randomly
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = ((it * 16 + r) * 2654435761u + blockIdx.x * 97u) % 10000u;
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) S[tp][r] += G[(size_t)row * 128 + 32 * tp + (lane & 31)];
                    }
                }
                gemm128_post<false>((const f32x4*)cur, lane, X, acc, [&](int tp, int g) { acc[tp][g] = silu_hw(acc[tp][g] + S[tp][g]); });
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
                __syncthreads();
                continue;
            }
            if (MODE == 10) {
                gemm128_post_pf<false>((const f32x4*)cur, lane, X, acc, [&](int tp, int g) { acc[tp][g] = silu_hw(acc[tp][g]); });
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
                continue;
            }
            if (MODE == 8 || MODE == 9) {
                gemm128_post<false>((const f32x4*)cur, lane, X, acc, [&](int tp, int g) { acc[tp][g] = silu_hw(acc[tp][g]); });
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
                if (MODE == 9) __syncthreads();
                continue;
            }
            if (MODE == 7) gemm128<true>((const f32x4*)cur, lane, X, acc);
            else if (MODE >= 5) gemm128_pf<false>((const f32x4*)cur, lane, X, acc);
            else gemm128<false>((const f32x4*)cur, lane, X, acc);
            if (MODE == 2 || MODE == 3 || MODE == 6) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) X[t][r] = silu_hw(acc[t][r]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t] * 0.01f;
            }
            if (MODE == 3 || MODE == 4 || MODE == 6) __syncthreads();
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r] + X[t][r];
    out[blockIdx.x * 512 + tid] = s;
}

static float* g_stream = nullptr;
template <int MODE>
double run(const float* dW, float* dOut, int iters, int threads = 512) {
    const float* dG = g_stream;
    const size_t ldsb = sizeof(float) * 2 * GAMD_WFRAG_FLOATS;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads, ldsb>>>(dW, dOut, 4, dG);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads, ldsb>>>(dW, dOut, iters, dG);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * (threads / 64) * iters * 256 * 4096.0;   // blocks*waves*iters*MFMAs*flop
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<float> W(GAMD_WFRAG_FLOATS);
    for (size_t i = 0; i < W.size(); ++i) W[i] = ((i * 2654435761u) % 1000) * 1e-5f - 0.005f;
    float *dW, *dOut; hipMalloc(&dW, W.size() * 4); hipMalloc(&dOut, 256 * 512 * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const int iters = 400;
    float* dG; hipMalloc(&dG, (size_t)256 * 8 * 64 * 4096 * 4);     // 2 GiB streamed region
    hipMemset(dG, 0, (size_t)256 * 8 * 64 * 4096 * 4);
    g_stream = dG;
    for (int rep = 0; rep < 2; ++rep) {
        printf("mode0 pure mfma        : %.1f TF\n", run<0>(dW, dOut, iters));
        printf("mode1 gemm128 from LDS : %.1f TF\n", run<1>(dW, dOut, iters));
        printf("mode2 + silu           : %.1f TF\n", run<2>(dW, dOut, iters));
        printf("mode3 + barrier+stage  : %.1f TF\n", run<3>(dW, dOut, iters));
        printf("mode4 barrier+stage, no silu: %.1f TF\n", run<4>(dW, dOut, iters));
        printf("mode5 prefetch gemm (8 waves)        : %.1f TF\n", run<5>(dW, dOut, iters));
        printf("mode6 prefetch + silu + barrier+stage: %.1f TF\n", run<6>(dW, dOut, iters));
        printf("mode7 gemm128 F2 orientation (8 waves): %.1f TF\n", run<7>(dW, dOut, iters));
        printf("mode8 pipelined silu post-op, no barrier (8 waves): %.1f TF\n", run<8>(dW, dOut, iters));
        printf("mode9 pipelined silu + barrier + stage (8 waves) : %.1f TF\n", run<9>(dW, dOut, iters));
        printf("mode8 pipelined silu post-op (4 waves)           : %.1f TF\n", run<8>(dW, dOut, iters, 256));
        printf("mode10 pipelined silu + w prefetch (8 waves)     : %.1f TF\n", run<10>(dW, dOut, iters));
        printf("mode10 pipelined silu + w prefetch (4 waves)     : %.1f TF\n", run<10>(dW, dOut, iters, 256));
        printf("mode11 hooks+barrier+stage + 16 KB streamed load per GEMM: %.1f TF\n", run<11>(dW, dOut, iters));
        printf("mode12 mode11 + 64 dword gathers per GEMM               : %.1f TF\n", run<12>(dW, dOut, iters));
        printf("mode1 4 waves/CU (1 per SIMD)        : %.1f TF\n", run<1>(dW, dOut, iters, 256));
        printf("mode5 4 waves/CU prefetch            : %.1f TF\n", run<5>(dW, dOut, iters, 256));
        printf("mode2 4 waves/CU + silu              : %.1f TF\n", run<2>(dW, dOut, iters, 256));
    }
    return 0;
}
