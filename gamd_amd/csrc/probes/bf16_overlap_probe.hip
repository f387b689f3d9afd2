// bf16_overlap_probe.hip — can the bf16 matrix pipe and the vector ALU of ONE SIMD be kept busy at the same time by the waves
// that share it, and what does the arbitration between those waves cost?  (k_conv_edge_bf16: per 32-edge tile a wave runs 4 x 32
// v_mfma_f32_32x32x16_bf16 and 3 SiLU blocks of 128 v_exp_f32 + 128 v_rcp_f32 + ~100 simple instructions; two waves per SIMD
// take about the SUM of matrix time and vector time per tile pair.)
//
//   part A (roles): W waves per SIMD; waves [0, W/2) of every SIMD run only MFMAs, the others only SiLU blocks (or the other
//                   way round): time of each role alone and of both together.
//   part B (alternating, the kernel's pattern): every wave alternates 32 MFMAs and one SiLU block; 1, 2, 3, 4 waves per SIMD;
//                   with s_setprio raised in the SiLU block / in the MFMA block / for the younger half.
// Prints SIMD-cycles per (32 MFMAs + 1 SiLU block) unit at 2.4 GHz.  GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma32(f32x16 (&acc)[4], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
// one SiLU block of the kernel: 64 elements per lane, per pair: pk_mul, 2 exp, pk_add, 2 rcp, pk_mul, cvt_pk
template <int TRANS>
__device__ __forceinline__ void silu_block(float (&v)[16], float c) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(v[j]));
            if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(t)); else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(t));
            asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
            if (TRANS) asm volatile("v_rcp_f32 %0, %0" : "+v"(t)); else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(t));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(t));
        }
}

// MODE bit 0: MFMA role active, bit 1: SiLU role active; SWAP: the older half (waves 0 .. W/2-1) takes the SiLU role
template <int MODE, int SWAP, int THREADS>
__global__ void __launch_bounds__(THREADS) k_roles(float* out, int iters) {
    const int wave = threadIdx.x >> 6, nw = THREADS / 64;
    const bool first_half = wave < nw / 2;             // waves w and w + nw/2 share a SIMD (4 SIMDs, round robin)
    const bool mfma_role = SWAP ? !first_half : first_half;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 2, 2, 3, 3, 4, 4};
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.5f + 0.01f * j + 1e-4f * threadIdx.x;
    if (mfma_role) {
        if (MODE & 1) for (int i = 0; i < iters; ++i) { mfma32(acc, a, b); asm volatile("" : "+v"(a)); }
    } else {
        if (MODE & 2) for (int i = 0; i < iters; ++i) silu_block<1>(v, -1.44f);
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

// PRIO: 0 none, 1 setprio 1 inside the SiLU block, 2 inside the MFMA block, 3 static prio 1 for the younger half
template <int PRIO, int TRANS, int THREADS>
__global__ void __launch_bounds__(THREADS) k_alt(float* out, int iters, int stagger) {
    const int wave = threadIdx.x >> 6, nw = THREADS / 64;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 2, 2, 3, 3, 4, 4};
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.5f + 0.01f * j + 1e-4f * threadIdx.x;
    if (PRIO == 3 && wave >= nw / 2) __builtin_amdgcn_s_setprio(1);
    if (stagger && wave >= 4) silu_block<TRANS>(v, -1.44f);          // de-phase the partner
    for (int i = 0; i < iters; ++i) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        mfma32(acc, a, b);
        asm volatile("" : "+v"(a));
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        // the SiLU consumes the accumulators (dependency as in the kernel)
        v[0] += acc[0][0]; v[5] += acc[1][3]; v[10] += acc[2][7]; v[15] += acc[3][15];
        silu_block<TRANS>(v, -1.44f);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

// part C: the MFMA block of part B with its weight fragments read from LDS through an 8-deep register ring (gemm128_bf16_pf of
// gamd_bf16.h: the kernel's GEMM phase), RING = 0: operands from registers as in part B
template <int D>
__device__ __forceinline__ void mfma32_lds(const bf16x8* W, int lane, f32x16 (&acc)[4], const bf16x8& a) {
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) w[i] = W[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % D], a, acc[i & 3], 0, 0, 0);
        if (i + D < 32) w[i % D] = W[(i + D) * 64 + lane];
    }
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int i = 0; i < 32 - D; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, D, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// INTER: the SiLU block is issued in 32 slices of 10 instructions, one slice behind every MFMA (the "lazy" structure: VALU
// of the previous output block between the MFMAs of this one, inside ONE wave)
template <int LDS, int INTER, int THREADS>
__global__ void __launch_bounds__(THREADS) k_alt2(float* out, int iters) {
    __shared__ bf16x8 Wl[32 * 64];
    for (int i = threadIdx.x; i < 32 * 64; i += THREADS) Wl[i] = bf16x8{1, 2, 3, 4, 5, 6, 7, (short)i};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 2, 2, 3, 3, 4, 4};
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.5f + 0.01f * j + 1e-4f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (INTER) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const bf16x8 w = LDS ? Wl[m * 64 + lane] : b;
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 2; ++k) {          // two elements = 10 instructions per MFMA: 64 elements per 32 MFMAs
                    const int j = (2 * m + k) & 15;
                    float t;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(-1.44f), "v"(v[j]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(t));
                }
            }
            asm volatile("" : "+v"(a));
        } else {
            if (LDS) mfma32_lds<8>(Wl, lane, acc, a); else mfma32(acc, a, b);
            asm volatile("" : "+v"(a));
            v[0] += acc[0][0]; v[5] += acc[1][3]; v[10] += acc[2][7]; v[15] += acc[3][15];
            silu_block<1>(v, -1.44f);
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <typename F> static double time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(2000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE, int SWAP, int THREADS> void run_roles(float* out, const char* name) {
    const double ms = time_ms([&](int it) { hipLaunchKernelGGL((k_roles<MODE, SWAP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, it); });
    const int per_simd = THREADS / 512;                 // waves of each role per SIMD
    printf("roles  %-52s %d+%d waves/SIMD: %7.1f cycles per unit (32 MFMA = 1024 matrix cycles | 1 SiLU block = 320 VALU)\n", name,
           per_simd, per_simd, ms * 1e-3 * 2.4e9 / (2000.0 * per_simd));
}
template <int PRIO, int TRANS, int THREADS> void run_alt(float* out, const char* name, int stagger) {
    const double ms = time_ms([&](int it) { hipLaunchKernelGGL((k_alt<PRIO, TRANS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, it, stagger); });
    const int wps = THREADS / 256;
    printf("alt    %-52s %d waves/SIMD: %7.1f SIMD-cycles per unit (32 MFMA + 1 SiLU block)\n", name, wps, ms * 1e-3 * 2.4e9 / (2000.0 * wps));
}

template <int LDS, int INTER, int THREADS> void run_alt2(float* out, const char* name) {
    const double ms = time_ms([&](int it) { hipLaunchKernelGGL((k_alt2<LDS, INTER, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, it); });
    const int wps = THREADS / 256;
    printf("alt2   %-52s %d waves/SIMD: %7.1f SIMD-cycles per unit (32 MFMA + 1 SiLU block)\n", name, wps, ms * 1e-3 * 2.4e9 / (2000.0 * wps));
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 256 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        run_roles<1, 0, 512>(out, "MFMA role alone (older half)");
        run_roles<2, 0, 512>(out, "SiLU role alone (younger half)");
        run_roles<3, 0, 512>(out, "both: older = MFMA, younger = SiLU");
        run_roles<3, 1, 512>(out, "both: older = SiLU, younger = MFMA");
        run_roles<3, 0, 1024>(out, "both, 2 + 2 waves per SIMD");
        run_alt<0, 1, 256>(out, "alternating", 0);
        run_alt<0, 1, 512>(out, "alternating", 0);
        run_alt<0, 1, 512>(out, "alternating, partner de-phased", 1);
        run_alt<1, 1, 512>(out, "alternating, prio 1 in SiLU", 0);
        run_alt<2, 1, 512>(out, "alternating, prio 1 in MFMA", 0);
        run_alt<3, 1, 512>(out, "alternating, static prio 1 for the younger half", 0);
        run_alt<0, 1, 768>(out, "alternating", 0);
        run_alt<0, 1, 1024>(out, "alternating", 0);
        run_alt<0, 0, 256>(out, "alternating, transcendentals -> v_mul", 0);
        run_alt<0, 0, 512>(out, "alternating, transcendentals -> v_mul", 0);
        run_alt<0, 0, 1024>(out, "alternating, transcendentals -> v_mul", 0);
        run_alt2<0, 0, 256>(out, "alternating (register operands)");
        run_alt2<0, 0, 512>(out, "alternating (register operands)");
        run_alt2<1, 0, 256>(out, "alternating, weights from LDS through an 8-deep ring");
        run_alt2<1, 0, 512>(out, "alternating, weights from LDS through an 8-deep ring");
        run_alt2<0, 1, 256>(out, "interleaved in one wave: 10 VALU behind every MFMA");
        run_alt2<0, 1, 512>(out, "interleaved in one wave: 10 VALU behind every MFMA");
        run_alt2<1, 1, 256>(out, "interleaved, weights from LDS (compiler-scheduled)");
        run_alt2<1, 1, 512>(out, "interleaved, weights from LDS (compiler-scheduled)");
    }
    (void)hipFree(out);
    return 0;
}
