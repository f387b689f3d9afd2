// mfma_issue_probe.hip — how long does a wave take per v_mfma_f32_32x32x2_f32 when consecutive MFMAs
//   mode 0: all accumulate into ONE tile (each depends on the previous one: the pattern of the GEMM chains),
//   mode 1: alternate between TWO independent tiles, mode 2: round-robin over FOUR,
// with 1 or 2 waves per SIMD, and with / without a VALU instruction between MFMAs?  Prints cycles per MFMA per SIMD
// (s_memtime) — 64 is the pipe's peak for this instruction.  GPU box only; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int VALU>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, w = 1.0f + threadIdx.x * 1e-4f, v = 0.5f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v2 = {0.5f, 0.25f}, c2 = {1.0001f, 0.9999f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a0, 0, 0, 0);
            if (MODE == 1) { if (k & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a0, 0, 0, 0); }
            if (MODE == 2) {
                if ((k & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a0, 0, 0, 0);
                if ((k & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a1, 0, 0, 0);
                if ((k & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a2, 0, 0, 0);
                if ((k & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a3, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < VALU % 100; ++j) {
                // VALU 1xx: transcendental instructions (alternating v_exp_f32 / v_rcp_f32) instead of v_fma_f32;
                // VALU 2xx / 3xx: packed v_pk_mul_f32 / v_pk_fma_f32 (two elements per instruction)
                if (VALU >= 300) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(c2)); }
                else if (VALU >= 200) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v2) : "v"(c2)); }
                else if (VALU >= 100) { v = (j & 1) ? __builtin_amdgcn_rcpf(v) : __builtin_amdgcn_exp2f(v); asm volatile("" : "+v"(v)); }
                else { v = __builtin_fmaf(v, 1.0001f, 0.25f); asm volatile("" : "+v"(v)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = v + v2[0] + v2[1];
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int VALU>
void run(int threads, const char* name) {
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    hipLaunchKernelGGL((k<MODE, VALU>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipLaunchKernelGGL((k<MODE, VALU>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto c : h) sum += (double)c;
    const double per_wave = sum / h.size() / (iters * 16.0);
    const int waves_per_simd = threads / 256;
    printf("%-34s waves/SIMD %d  VALU/MFMA %3d : %7.2f ticks per MFMA per wave = %7.2f per SIMD slot\n", name, waves_per_simd, VALU,
           per_wave, per_wave / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>(256, "one accumulator (dependent)");
    run<1, 0>(256, "two accumulators");
    run<2, 0>(256, "four accumulators");
    run<0, 0>(512, "one accumulator (dependent)");
    run<1, 0>(512, "two accumulators");
    run<0, 2>(256, "one accumulator + VALU");
    run<0, 6>(256, "one accumulator + VALU");
    run<0, 12>(256, "one accumulator + VALU");
    run<1, 6>(256, "two accumulators + VALU");
    run<0, 6>(512, "one accumulator + VALU");
    run<0, 12>(512, "one accumulator + VALU");
    run<1, 6>(512, "two accumulators + VALU");
    run<0, 102>(256, "one accumulator + 2 exp/rcp");
    run<0, 106>(256, "one accumulator + 6 exp/rcp");
    run<0, 106>(512, "one accumulator + 6 exp/rcp");
    run<0, 112>(512, "one accumulator + 12 exp/rcp");
    run<0, 206>(512, "one accumulator + 6 pk_mul");
    run<0, 212>(512, "one accumulator + 12 pk_mul");
    run<0, 306>(512, "one accumulator + 6 pk_fma");
    run<0, 312>(512, "one accumulator + 12 pk_fma");
    return 0;
}
