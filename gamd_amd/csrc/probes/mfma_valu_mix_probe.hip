// mfma_valu_mix_probe.hip — what do the post-ops of the conv kernel cost next to a dependent fp32 MFMA chain, with one and
// with two waves per SIMD?  Per 8 MFMAs (asm, one accumulator) a block of INDEPENDENT vector instructions:
//   mode 0: none; 1: 32 v_mul_f32; 2: 16 v_exp_f32 + 16 v_rcp_f32; 3: eight SiLUs as the kernel issues them (8 v_mul, 8 v_exp,
//   8 v_add, 8 v_rcp, 4 v_pk_mul); 4: the same eight SiLUs with the reciprocal by magic-constant guess + 3 Newton steps
//   (8 v_mul, 8 v_exp, 8 v_add, 8 v_sub (guess), 24 x 2 v_fma, 4 v_pk_mul: one transcendental per element instead of two).
// Prints cycles per MFMA per SIMD slot from HIP-event time at the measured clock.  GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF8 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t" \
            "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k(float* out, int iters) {
    f32x16 a0 = {0};
    float x = threadIdx.x * 1e-3f, w = 1.0f + threadIdx.x * 1e-4f;
    float v[8], t[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.1f * i + 1e-3f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            asm volatile(MF8 : "+v"(a0) : "v"(w), "v"(x));
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(w));
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j]));
            } else if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t[j]) : "v"(v[j]));
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(t[j]));
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t[j]));
                if (MODE == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_rcp_f32 %0, %0" : "+v"(t[j]));
                } else {
                    float r[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_sub_u32 %0, 0x7ef311c7, %1" : "=v"(r[j]) : "v"(t[j]));
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        float e[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, -%1, %2, 1.0" : "=v"(e[j]) : "v"(t[j]), "v"(r[j]));
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(r[j]) : "v"(e[j]));
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = r[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j += 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(float2*)&v[j]) : "v"(*(float2*)&t[j]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += a0[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int MODE, int THREADS> void run(const char* name) {
    const int blocks = 256, iters = 4000;
    float* out; (void)hipMalloc(&out, 4 * blocks * THREADS);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, 100);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * (THREADS / 256);
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;        // at 2.4 GHz
    printf("%-58s waves/SIMD %d: %6.2f cycles per MFMA per SIMD slot (64 = matrix pipe full) -> block costs %6.1f cycles per 8 MFMAs\n",
           name, THREADS / 256, cyc, (cyc - 64.0) * 8);
    (void)hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 256>("MFMA only"); run<1, 256>("+ 32 v_mul"); run<2, 256>("+ 16 v_exp + 16 v_rcp"); run<3, 256>("+ 8 SiLU (exp + rcp)"); run<4, 256>("+ 8 SiLU (exp + Newton rcp)");
        run<0, 512>("MFMA only"); run<1, 512>("+ 32 v_mul"); run<2, 512>("+ 16 v_exp + 16 v_rcp"); run<3, 512>("+ 8 SiLU (exp + rcp)"); run<4, 512>("+ 8 SiLU (exp + Newton rcp)");
    }
    return 0;
}
