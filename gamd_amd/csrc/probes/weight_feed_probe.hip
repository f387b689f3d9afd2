// Probe: how should the conv-edge kernel's GEMM chain be fed with its four 64 KiB weight matrices?
// The shipped kernel streams them L2 -> LDS into a 2-slot ring with one workgroup barrier per GEMM phase; the
// barrier-locked phases are what keeps it at 0.76-0.78 of the fp32 matrix peak.  This probe times the chain of four
// 128x128 GEMMs + SiLU per 32-row tile (the kernel's arithmetic, no gathers) under different weight feeds:
//   mode 0  LDS ring, all 8 waves issue the DMA, barrier per GEMM                         (the shipped structure)
//   mode 1  LDS ring, only waves 4-7 issue the DMA (16 x 1 KiB each)                     (asymmetric DMA)
//   mode 2  no LDS: every wave reads its weight fragments straight from L2 with global_load_dwordx4, DEPTH fragments
//           ahead, no barrier at all (free-running waves), 8 waves per CU
//   mode 3  the same with 4 waves per CU (one per SIMD)
//   mode 4  hybrid: W1, W2 resident in LDS (128 KiB, loaded once), W3, W4 straight from L2; no barrier
//   mode 8  LDS ring fed as in mode 0, but per-slot ready / done counters in LDS instead of workgroup barriers
// FLOP/s counts 4 x 256 MFMAs per tile.
#include "../gamd_common.h"
#include <cstdio>
#include <vector>

__device__ __forceinline__ float silu_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// LDS-fed GEMM with the software-pipelined post-op of conv_edge.hip
template <typename WPtr, typename Post>
__device__ __forceinline__ void gemm_lds(WPtr W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4], Post post) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = W[((tp * 4 + t) * 4 + q) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[tp] = mfma32(w[j], X[t][q * 4 + j], acc[tp]);
                if (tp > 0) post(tp - 1, t * 4 + q);
            }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

// L2-fed GEMM: w[] holds fragments 0..D-1 of THIS matrix on entry and fragments 0..D-1 of the NEXT one on exit, so
// the load stream never drains at a GEMM boundary
template <int D, typename Post>
__device__ __forceinline__ void gemm_l2(const f32x4* __restrict__ W, const f32x4* __restrict__ Wnext, int lane, f32x4 (&w)[D],
                                        const f32x16 (&X)[4], f32x16 (&acc)[4], Post post) {
    asm volatile("" : "+v"(lane));              // opaque per call: the loads are loop-invariant and would be hoisted + spilled
#pragma unroll
    for (int g = 0; g < 64; ++g) {
        const int tp = g >> 4, t = (g >> 2) & 3, q = g & 3;
        const f32x4 cur = w[g % D];
        w[g % D] = (g + D < 64) ? W[(g + D) * 64 + lane] : Wnext[(g + D - 64) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tp] = mfma32(cur[j], X[t][q * 4 + j], acc[tp]);
        if (tp > 0) post(tp - 1, t * 4 + q);
        __builtin_amdgcn_sched_barrier(0);      // keep the load of fragment g + D here (hipcc would hoist all 64 and spill)
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

// LDS-fed GEMM over TWO tiles held by one wave: every weight fragment read feeds 8 MFMAs
template <typename WPtr, typename Post>
__device__ __forceinline__ void gemm_lds2(WPtr W, int lane, const f32x16 (&X0)[4], const f32x16 (&X1)[4], f32x16 (&a0)[4],
                                          f32x16 (&a1)[4], Post post) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = W[((tp * 4 + t) * 4 + q) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) a0[tp] = mfma32(w[j], X0[t][q * 4 + j], a0[tp]);
#pragma unroll
                for (int j = 0; j < 4; ++j) a1[tp] = mfma32(w[j], X1[t][q * 4 + j], a1[tp]);
                if (tp > 0) post(tp - 1, t * 4 + q);
            }
#pragma unroll
    for (int g = 0; g < 16; ++g) post(3, g);
}

template <int MODE, int D, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k(const float* __restrict__ W4, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x16 X[4], acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { X[t][r] = 0.001f * (lane + r + t); acc[t][r] = 0.f; }
    auto Wm = [&](int m) { return W4 + (size_t)(m & 3) * GAMD_WFRAG_FLOATS; };
    auto silu_post = [&](int tp, int g) { acc[tp][g] = silu_hw(acc[tp][g]); };

    if (MODE == 5 || MODE == 6) {
        // one wave per SIMD, two tiles per wave, LDS ring + barrier per GEMM; mode 6 adds per GEMM and tile a 16 KB streamed
        // read consumed by the post-op plus 64 dword gathers (the kernel's e / S / D / hn traffic)
        float* buf[2] = {lds, lds + GAMD_WFRAG_FLOATS};
        f32x16 Y[4], acc1[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { Y[t][r] = 0.002f * (lane + r - t); acc1[t][r] = 0.f; }
        gamd_stage_weight<4>(Wm(0), buf[0], wave, lane16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                gamd_stage_weight<4>(Wm(m + 1), buf[(m + 1) & 1], wave, lane16);
                for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { acc[t][r] = 0.01f; acc1[t][r] = 0.02f; }
                if (MODE == 6) {
                    f32x16 S0[4], S1[4];
                    const size_t tb = ((size_t)(blockIdx.x * 4 + wave) * 64 + ((it * 4 + m) & 63)) * 8192;
                    const f32x4* gp = (const f32x4*)(W4 + 4 * GAMD_WFRAG_FLOATS + tb);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v0 = gp[(t * 4 + q) * 64 + lane], v1 = gp[1024 + (t * 4 + q) * 64 + lane];
#pragma unroll
                            for (int j = 0; j < 4; ++j) { S0[t][q * 4 + j] = v0[j]; S1[t][q * 4 + j] = v1[j]; }
                        }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = ((it * 16 + r) * 2654435761u + blockIdx.x * 97u + m) % 10000u;
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            S0[tp][r] += W4[4 * GAMD_WFRAG_FLOATS + (size_t)row * 128 + 32 * tp + (lane & 31)];
                            S1[tp][r] += W4[4 * GAMD_WFRAG_FLOATS + (size_t)(row + 77) * 128 + 32 * tp + (lane & 31)];
                        }
                    }
                    gemm_lds2((const f32x4*)buf[m & 1], lane, X, Y, acc, acc1, [&](int tp, int g) {
                        acc[tp][g] = silu_hw(acc[tp][g] + S0[tp][g]); acc1[tp][g] = silu_hw(acc1[tp][g] + S1[tp][g]); });
                } else {
                    gemm_lds2((const f32x4*)buf[m & 1], lane, X, Y, acc, acc1, [&](int tp, int g) {
                        acc[tp][g] = silu_hw(acc[tp][g]); acc1[tp][g] = silu_hw(acc1[tp][g]); });
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) { X[t] = acc[t]; Y[t] = acc1[t]; }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        float s1 = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s1 += acc1[t][r] + Y[t][r];
        out[blockIdx.x * 512 + tid + 256] = s1;
    } else if (MODE == 0 || MODE == 1 || MODE == 7) {
        float* buf[2] = {lds, lds + GAMD_WFRAG_FLOATS};
        auto stage = [&](int m, float* dst) {
            if (MODE == 1) { if (wave >= 4) gamd_stage_weight<4>(Wm(m), dst, wave & 3, lane16); }
            else gamd_stage_weight<8>(Wm(m), dst, wave, lane16);
        };
        stage(0, buf[0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                stage(m + 1, buf[(m + 1) & 1]);
                for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
                if (MODE == 7) {
                    f32x16 S0[4];
                    const size_t tb = ((size_t)(blockIdx.x * 8 + wave) * 64 + ((it * 4 + m) & 63)) * 4096;
                    const f32x4* gp = (const f32x4*)(W4 + 4 * GAMD_WFRAG_FLOATS + tb);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v0 = gp[(t * 4 + q) * 64 + lane];
#pragma unroll
                            for (int j = 0; j < 4; ++j) S0[t][q * 4 + j] = v0[j];
                        }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = ((it * 16 + r) * 2654435761u + blockIdx.x * 97u + m) % 10000u;
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) S0[tp][r] += W4[4 * GAMD_WFRAG_FLOATS + (size_t)row * 128 + 32 * tp + (lane & 31)];
                    }
                    gemm_lds((const f32x4*)buf[m & 1], lane, X, acc, [&](int tp, int g) { acc[tp][g] = silu_hw(acc[tp][g] + S0[tp][g]); });
                } else
                gemm_lds((const f32x4*)buf[m & 1], lane, X, acc, silu_post);
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
    } else if (MODE == 8) {
        // LDS ring WITHOUT workgroup barriers: per slot a "ready" counter (waves whose share of the copy has landed) and a
        // "done" counter (waves that have finished reading it).  A wave starts phase k when ready[k & 1] says all eight shares
        // of that matrix are in; it issues its share of phase k + 1's matrix as soon as done[] says everybody has left that
        // slot (checked at the phase start and again at every output-tile boundary), and announces it half a GEMM later.
        // Waves may drift by most of a phase instead of meeting four times per tile.
        auto slot_ptr = [&](int sl) { return lds + sl * GAMD_WFRAG_FLOATS; };
        int* cnt = reinterpret_cast<int*>(lds + 2 * GAMD_WFRAG_FLOATS);      // ready[0], ready[1], done[0], done[1]
        if (tid < 4) cnt[tid] = 0;
        __syncthreads();
        auto poll = [&](int idx, int target) {
            int spins = 0;
            while (__hip_atomic_load(cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(1);
        };
        auto bump = [&](int idx) {
            if (lane == 0) __hip_atomic_fetch_add(cnt + idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        gamd_stage_weight_raw<8>(Wm(0), slot_ptr(0), wave, lane16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bump(0);
        const int n_phase = 4 * iters;
        for (int k = 0; k < n_phase; ++k) {
            const int slot = k & 1, nslot = slot ^ 1;
            poll(slot, 8 * ((k >> 1) + 1));                                   // weights of phase k are in
            // slot nslot was last read in phase k - 1: its ((k - 1) >> 1)-th use
            const int free_target = k >= 1 ? 8 * (((k - 1) >> 1) + 1) : 0;
            bool issued = false, told = false;
            auto try_issue = [&]() {
                if (!issued && __hip_atomic_load(cnt + 2 + nslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= free_target) {
                    gamd_stage_weight_raw<8>(Wm(k + 1), slot_ptr(nslot), wave, lane16);
                    issued = true;
                }
            };
            try_issue();
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
            gemm_lds((const f32x4*)slot_ptr(slot), lane, X, acc, [&](int tp, int g) {
                if (g == 0) {                                                 // output-tile boundary: tp + 1 is starting
                    if (!issued) try_issue();
                    else if (!told && tp >= 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bump(nslot); told = true; }
                }
                acc[tp][g] = silu_hw(acc[tp][g]);
            });
#pragma unroll
            for (int t = 0; t < 4; ++t) X[t] = acc[t];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bump(2 + slot);                                                   // done with this slot
            if (!issued) { poll(2 + nslot, free_target); try_issue(); }
            if (!told) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bump(nslot); }
        }
    } else if (MODE == 2 || MODE == 3) {
        f32x4 w[D];
#pragma unroll
        for (int d = 0; d < D; ++d) w[d] = ((const f32x4*)Wm(0))[d * 64 + lane];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
                gemm_l2<D>((const f32x4*)Wm(m), (const f32x4*)Wm(m + 1), lane, w, X, acc, silu_post);
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
            }
    } else {
        for (int i = tid; i < 2 * GAMD_WFRAG_FLOATS / 4; i += blockDim.x) ((f32x4*)lds)[i] = ((const f32x4*)W4)[i];
        __syncthreads();
        f32x4 w[D];
#pragma unroll
        for (int d = 0; d < D; ++d) w[d] = ((const f32x4*)Wm(2))[d * 64 + lane];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
                gemm_lds((const f32x4*)(lds + m * GAMD_WFRAG_FLOATS), lane, X, acc, silu_post);
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
            }
#pragma unroll
            for (int m = 2; m < 4; ++m) {
                for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.01f;
                gemm_l2<D>((const f32x4*)Wm(m), (const f32x4*)Wm(m == 2 ? 3 : 2), lane, w, X, acc, silu_post);
#pragma unroll
                for (int t = 0; t < 4; ++t) X[t] = acc[t];
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r] + X[t][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int D, int threads>
double run(const float* dW, float* dOut, int iters) {
    const size_t ldsb = sizeof(float) * 2 * GAMD_WFRAG_FLOATS + 64;
    hipFuncSetAttribute((const void*)k<MODE, D, threads>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, D, threads><<<256, threads, ldsb>>>(dW, dOut, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, D, threads><<<256, threads, ldsb>>>(dW, dOut, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * (threads / 64) * iters * 4.0 * 256 * 4096.0 * ((MODE == 5 || MODE == 6) ? 2.0 : 1.0);
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<float> W(4 * GAMD_WFRAG_FLOATS);
    for (size_t i = 0; i < W.size(); ++i) W[i] = ((i * 2654435761u) % 1000) * 1e-5f - 0.005f;
    // 4 weight matrices, then a 512 MiB region the streamed / gathered reads of modes 6 and 7 walk through
    const size_t stream_floats = (size_t)256 * 8 * 64 * 4096;
    float *dW, *dOut; hipMalloc(&dW, (W.size() + stream_floats) * 4); hipMalloc(&dOut, 256 * 1024 * 4);
    hipMemset(dW, 0, (W.size() + stream_floats) * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const int iters = 100;      // tiles per wave
    for (int rep = 0; rep < 2; ++rep) {
        printf("mode0 LDS ring, symmetric DMA, barrier per GEMM (8 waves)  : %.1f TF\n", run<0, 4, 512>(dW, dOut, iters));
        printf("mode1 LDS ring, DMA by waves 4-7 only                      : %.1f TF\n", run<1, 4, 512>(dW, dOut, iters));
        printf("mode2 weights straight from L2, depth 4, no barrier (8 w)  : %.1f TF\n", run<2, 4, 512>(dW, dOut, iters));
        printf("mode2 weights straight from L2, depth 8, no barrier (8 w)  : %.1f TF\n", run<2, 8, 512>(dW, dOut, iters));
        printf("mode3 weights straight from L2, depth 8 (4 waves)          : %.1f TF\n", run<3, 8, 256>(dW, dOut, iters));
        printf("mode3 weights straight from L2, depth 16 (4 waves)         : %.1f TF\n", run<3, 16, 256>(dW, dOut, iters));
        printf("mode4 W1,W2 LDS-resident + W3,W4 from L2 depth 4 (8 waves) : %.1f TF\n", run<4, 4, 512>(dW, dOut, iters));
        printf("mode4 W1,W2 LDS-resident + W3,W4 from L2 depth 8 (4 waves) : %.1f TF\n", run<4, 8, 256>(dW, dOut, iters));
        printf("mode5 LDS ring, 4 waves x 2 tiles each, barrier per GEMM       : %.1f TF\n", run<5, 4, 256>(dW, dOut, iters / 2));
        printf("mode8 LDS ring, ready/done counters instead of barriers (8 w)  : %.1f TF\n", run<8, 4, 512>(dW, dOut, iters));
        printf("mode7 mode0 + 16 KB streamed + 64 gathers per GEMM (8 w x 1 t) : %.1f TF\n", run<7, 4, 512>(dW, dOut, iters));
        printf("mode6 mode5 + the same traffic per tile (4 waves x 2 tiles)    : %.1f TF\n", run<6, 4, 256>(dW, dOut, iters / 2));
    }
    return 0;
}
