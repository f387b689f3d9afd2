#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Does an LDS / VMEM load returning into VGPRs cost MFMA issue time?  Per 4 dependent MFMAs:
//  MODE 0: nothing; 1: one ds_read_b128 (result unused); 2: two ds_read_b128; 3: one global_load_dwordx4 (L1/L2-hot);
//  4: one ds_read_b128 whose result feeds the NEXT group's MFMAs (1 group ahead); 5: same, 2 groups ahead (double buffer)
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k(const float* __restrict__ g, float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += THREADS) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0};
    float x = threadIdx.x * 1e-3f;
    const int lane = threadIdx.x & 63;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + lane * 16u;
    f32x4 w0 = {1, 1, 1, 1}, w1 = {1, 1, 1, 1}, d0, d1;
    const f32x4* gp = (const f32x4*)g + lane;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 1 || MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d0) : "v"(la), "n"((k & 15) * 1024));
            if (MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d1) : "v"(la), "n"((k & 15) * 1024 + 16384));
            if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d0) : "v"(gp), "n"((k & 3) * 1024));
            if (MODE == 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %3, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %4, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %5, %2, %0"
                             : "+v"(a0) : "v"(w0[0]), "v"(x), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w0) : "v"(la), "n"((k & 15) * 1024));
            } else if (MODE == 5) {
                f32x4& cur = (k & 1) ? w1 : w0;
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %3, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %4, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %5, %2, %0"
                             : "+v"(a0) : "v"(cur[0]), "v"(x), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cur) : "v"(la), "n"((k & 15) * 1024));
            } else {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
                             : "+v"(a0) : "v"(w0[0]), "v"(x));
            }
        }
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)");
        if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = w0[0] + w1[0];
    if (MODE == 1 || MODE == 2 || MODE == 3) s += d0[0];
    if (MODE == 2) s += d1[0];
    for (int i = 0; i < 16; ++i) s += a0[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}
template <int MODE, int THREADS> void run(const float* g, const char* name) {
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4 * blocks * THREADS); (void)hipMalloc(&cyc, 8 * blocks * 8);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, g, out, cyc, 10);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, g, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * (THREADS / 64));
    std::vector<long long> all(blocks * 8);
    (void)hipMemcpy(all.data(), cyc, 8 * all.size(), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < THREADS / 64; ++w) { sum += (double)all[b * 8 + w]; ++n; }
    const double per = sum / n / (iters * 64.0);
    const double tf = 256.0 * (THREADS / 64) * iters * 64.0 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-62s waves/SIMD %d: %7.2f ticks/MFMA/wave = %6.2f per slot, %.1f TF, tick rate %.3f GHz\n", name, THREADS / 256, per, per / (THREADS / 256), tf,
           sum / n / (ms * 1e-3) / 1e9);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    float* g; (void)hipMalloc(&g, 1 << 20); (void)hipMemset(g, 0, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 256>(g, "4 MFMA");
        run<1, 256>(g, "4 MFMA + 1 ds_read_b128 (unused)");
        run<2, 256>(g, "4 MFMA + 2 ds_read_b128 (unused)");
        run<3, 256>(g, "4 MFMA + 1 global_load_dwordx4 (unused, cache-hot)");
        run<4, 256>(g, "4 MFMA fed by ds_read_b128 one group ahead (single buffer)");
        run<5, 256>(g, "4 MFMA fed by ds_read_b128 two groups ahead (double buffer)");
        run<0, 512>(g, "4 MFMA");
        run<1, 512>(g, "4 MFMA + 1 ds_read_b128 (unused)");
        run<2, 512>(g, "4 MFMA + 2 ds_read_b128 (unused)");
        run<3, 512>(g, "4 MFMA + 1 global_load_dwordx4 (unused, cache-hot)");
        run<4, 512>(g, "4 MFMA fed by ds_read_b128 one group ahead (single buffer)");
        run<5, 512>(g, "4 MFMA fed by ds_read_b128 two groups ahead (double buffer)");
    }
    return 0;
}
