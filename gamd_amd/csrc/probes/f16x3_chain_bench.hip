// Probe: cost structure of the split-fp16 GEMM chain (gamd_f16x3.h) on one CU: cycles per 32x128x128 GEMM per wave.
//   mode 0: pure v_mfma_f32_32x32x16_f16, 4 independent accumulators, register operands (96 MFMAs per "GEMM")
//   mode 1: tp-outer GEMM from LDS operand images, no activation (operand set reused)
//   mode 2: VALU only: SiLU + split of 64 elements per lane per "GEMM"
//   mode 3: mode 1 with the SiLU + split post-op pipelined between the MFMAs (the kernel's inner loop)
//   mode 4: mode 1 followed by a trailing SiLU + split
//   mode 5: mode 3 + one barrier and one 64 KiB global_load_lds restage per GEMM (the kernel's phase)
//   mode 6: t,u-outer GEMM with on-the-fly split of an fp32 block, trailing SiLU
#include "../gamd_f16x3.h"
#include <cstdio>
#include <vector>

// (OpSet: gamd_f16x3.h)

__device__ __forceinline__ void put_pair(OpSet& P, int t, int r0, float x0, float x1) {
    const gamd_f32x2_t x = {x0, x1};
    const gamd_f16x2 h = __builtin_convertvector(x, gamd_f16x2);
    const gamd_f32x2_t rem = x - __builtin_convertvector(h, gamd_f32x2_t);
    const gamd_f16x2 l = __builtin_convertvector(rem, gamd_f16x2);
    const int u = r0 >> 3, d = (r0 & 7) >> 1;
    P.w[t][u][0][d] = __builtin_bit_cast(unsigned, h);
    P.w[t][u][1][d] = __builtin_bit_cast(unsigned, l);
}

// the same with the instruction order pinned: every MFMA is followed by a slice of the previous tile's post-op VALU
// work (a wave issues in order: VALU ops queued behind a stalled dependent MFMA cannot use the idle cycles)
template <int NV, typename Post>
__device__ __forceinline__ void gemm_post_sched(const f16x8* W, int lane, const OpSet& P, f32x16 (&acc)[4], Post post) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f16x8 wh = W[((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 wl = W[2048 + ((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 xh = __builtin_bit_cast(f16x8, P.w[t][u][0]), xl = __builtin_bit_cast(f16x8, P.w[t][u][1]);
                acc[tp] = mfma_f16(wh, xl, acc[tp]);
                acc[tp] = mfma_f16(wl, xh, acc[tp]);
                acc[tp] = mfma_f16(wh, xh, acc[tp]);
                if (tp > 0) {
                    post(tp - 1, 2 * (t * 2 + u));
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                    }
                }
            }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) post(3, 2 * k);
}

// explicit DEPTH-step-ahead register prefetch of the weight fragments (no post-op)
template <int DEPTH>
__device__ __forceinline__ void gemm_prefetch(const f16x8* W, int lane, const OpSet& P, f32x16 (&acc)[4]) {
    f16x8 wh[DEPTH + 1], wl[DEPTH + 1];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { wh[d] = W[d * 64 + lane]; wl[d] = W[2048 + d * 64 + lane]; }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int tp = i >> 3, t = (i >> 1) & 3, u = i & 1;
        if (i + DEPTH < 32) {
            wh[(i + DEPTH) % (DEPTH + 1)] = W[(i + DEPTH) * 64 + lane];
            wl[(i + DEPTH) % (DEPTH + 1)] = W[2048 + (i + DEPTH) * 64 + lane];
        }
        const f16x8 xh = __builtin_bit_cast(f16x8, P.w[t][u][0]), xl = __builtin_bit_cast(f16x8, P.w[t][u][1]);
        const f16x8 a = wh[i % (DEPTH + 1)], b = wl[i % (DEPTH + 1)];
        acc[tp] = mfma_f16(a, xl, acc[tp]);
        acc[tp] = mfma_f16(b, xh, acc[tp]);
        acc[tp] = mfma_f16(a, xh, acc[tp]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two ds_reads of step i + DEPTH
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);      // then this step's MFMAs
    }
}

template <typename Post>
__device__ __forceinline__ void gemm_post(const f16x8* W, int lane, const OpSet& P, f32x16 (&acc)[4], Post post, bool pipelined) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f16x8 wh = W[((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 wl = W[2048 + ((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 xh = __builtin_bit_cast(f16x8, P.w[t][u][0]), xl = __builtin_bit_cast(f16x8, P.w[t][u][1]);
                acc[tp] = mfma_f16(wh, xl, acc[tp]);
                acc[tp] = mfma_f16(wl, xh, acc[tp]);
                acc[tp] = mfma_f16(wh, xh, acc[tp]);
                if (pipelined && tp > 0) post(tp - 1, 2 * (t * 2 + u));
            }
    }
    if (pipelined) {
#pragma unroll
        for (int k = 0; k < 8; ++k) post(3, 2 * k);
    }
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const float* __restrict__ Wg, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[4], X[4];
    OpSet PA, PB;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; X[t][r] = 0.001f * (lane + r + t); }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; r += 2) { put_pair(PA, t, r, X[t][r], X[t][r + 1]); put_pair(PB, t, r, X[t][r + 1], X[t][r]); }
    float* buf0 = lds; float* buf1 = lds + GAMD_WFRAG_FLOATS;
    for (int i = tid; i < GAMD_WFRAG_FLOATS / 4; i += blockDim.x) { ((f32x4*)buf0)[i] = ((const f32x4*)Wg)[i]; ((f32x4*)buf1)[i] = ((const f32x4*)Wg)[i]; }
    __syncthreads();
    if (MODE == 0) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.5f + lane * 1e-3f); b[j] = (_Float16)0.25f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 24; ++kk)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_f16(a, b, acc[t]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            float* cur = (it & 1) ? buf1 : buf0;
            float* nxt = (it & 1) ? buf0 : buf1;
            OpSet& Pin = PA;
            if (MODE == 5) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int chunk = kk * 8 + wave;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wg + chunk * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(nxt + chunk * 256), 16, 0, 0);
                }
            }
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
            if (MODE == 1) {
                gemm_post((const f16x8*)cur, lane, Pin, acc, [&](int, int) {}, false);
                PA.w[0][0][0][0] ^= __builtin_bit_cast(unsigned, acc[0][0]) & 1u;      // keep a dependence
            } else if (MODE == 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) put_pair(PB, t, r, gamd_silu_hw(X[t][r]), gamd_silu_hw(X[t][r + 1]));
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t][0] += __builtin_bit_cast(float, PB.w[t][0][1][0]) * 1e-30f;
            } else if (MODE == 3 || MODE == 5) {
                gemm_post((const f16x8*)cur, lane, PA, acc, [&](int tp, int r0) {
                    put_pair(PB, tp, r0, gamd_silu_hw(acc[tp][r0]), gamd_silu_hw(acc[tp][r0 + 1]));
                }, true);
                for (int t = 0; t < 4; ++t) for (int u = 0; u < 2; ++u) for (int p = 0; p < 2; ++p) PA.w[t][u][p] = PB.w[t][u][p];
                if (MODE == 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            } else if (MODE == 7 || MODE == 8 || MODE == 9) {
                constexpr int NV = MODE == 7 ? 4 : (MODE == 8 ? 6 : 8);
                gemm_post_sched<NV>((const f16x8*)cur, lane, PA, acc, [&](int tp, int r0) {
                    put_pair(PB, tp, r0, gamd_silu_hw(acc[tp][r0]), gamd_silu_hw(acc[tp][r0 + 1]));
                });
                for (int t = 0; t < 4; ++t) for (int u = 0; u < 2; ++u) for (int p = 0; p < 2; ++p) PA.w[t][u][p] = PB.w[t][u][p];
            } else if (MODE == 10 || MODE == 11 || MODE == 12) {
                constexpr int DEPTH = MODE == 10 ? 1 : (MODE == 11 ? 2 : 4);
                gemm_prefetch<DEPTH>((const f16x8*)cur, lane, Pin, acc);
                PA.w[0][0][0][0] ^= __builtin_bit_cast(unsigned, acc[0][0]) & 1u;
            } else if (MODE == 4) {
                gemm_post((const f16x8*)cur, lane, PA, acc, [&](int, int) {}, false);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) put_pair(PA, t, r, gamd_silu_hw(acc[t][r]), gamd_silu_hw(acc[t][r + 1]));
            } else if (MODE == 6) {
                gemm128_f16x3<false>((const f16x8*)cur, lane, X, acc);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) X[t][r] = gamd_silu_hw(acc[t][r]);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r] + X[t][r];
    s += __builtin_bit_cast(float, PA.w[1][1][1][1]) + __builtin_bit_cast(float, PB.w[2][0][0][3]);
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, const float* dW, float* dOut, int iters, int threads) {
    const size_t ldsb = sizeof(float) * 2 * GAMD_WFRAG_FLOATS;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads, ldsb>>>(dW, dOut, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads, ldsb>>>(dW, dOut, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_per_gemm = ms * 1e3 / iters;                         // wall time per GEMM round (all waves of a CU)
    const double eq_tf = 256.0 * (threads / 64) * iters * 2.0 * 32 * 128 * 128 / (ms * 1e-3) / 1e12;   // fp32-equivalent
    printf("%-58s %d waves/CU: %7.3f us per GEMM round  = %6.0f cycles @2.4GHz   %7.1f TF fp32-equivalent\n", name, threads / 64,
           us_per_gemm, us_per_gemm * 2400.0, eq_tf);
}

int main() {
    std::vector<float> W(GAMD_WFRAG_FLOATS);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 1e-3f;
    float *dW, *dOut; hipMalloc(&dW, W.size() * 4); hipMalloc(&dOut, 256 * 512 * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        run<0>("mode0 pure f16 MFMA (96 per GEMM)", dW, dOut, iters, threads);
        run<1>("mode1 tp-outer GEMM from LDS, no activation", dW, dOut, iters, threads);
        run<2>("mode2 VALU only: SiLU + split of 64 elements", dW, dOut, iters, threads);
        run<3>("mode3 GEMM + pipelined SiLU/split post-op", dW, dOut, iters, threads);
        run<4>("mode4 GEMM + trailing SiLU/split", dW, dOut, iters, threads);
        run<5>("mode5 mode3 + barrier + 64 KiB restage per GEMM", dW, dOut, iters, threads);
        run<6>("mode6 t,u-outer GEMM, on-the-fly split, trailing SiLU", dW, dOut, iters, threads);
        run<10>("mode10 mode1 with explicit 1-step weight prefetch", dW, dOut, iters, threads);
        run<11>("mode11 mode1 with explicit 2-step weight prefetch", dW, dOut, iters, threads);
        run<12>("mode12 mode1 with explicit 4-step weight prefetch", dW, dOut, iters, threads);
        run<7>("mode7 mode3 with pinned order: MFMA + 4 VALU", dW, dOut, iters, threads);
        run<8>("mode8 mode3 with pinned order: MFMA + 6 VALU", dW, dOut, iters, threads);
        run<9>("mode9 mode3 with pinned order: MFMA + 8 VALU", dW, dOut, iters, threads);
    }
    return 0;
}
