// dispatch_probe.hip — how long does the chip need just to START the workgroups of a launch?  k_node (625 workgroups of
// 256 threads at C2, each alive ~5 us) takes 28 us: is that the dispatcher?  Empty kernels with k_node's resources
// (~134 VGPRs, 9 KiB LDS), timed with HIP events over 50 back-to-back launches.  GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int THREADS, int NV>
__global__ void __launch_bounds__(THREADS) k_empty(float* out, int spin) {
    __shared__ float lds[2304];
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
    lds[threadIdx.x] = v[0];
    __syncthreads();
    for (int s = 0; s < spin; ++s) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 12345.678f) out[blockIdx.x] = s;
}
template <int THREADS, int NV>
void run(int blocks, int spin, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_empty<THREADS, NV>), dim3(blocks), dim3(THREADS), 0, 0, out, spin);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_empty<THREADS, NV>), dim3(blocks), dim3(THREADS), 0, 0, out, spin);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("threads %4d  regs ~%3d  blocks %5d  spin %4d : %7.2f us per launch (back to back)\n", THREADS, NV, blocks, spin, ms * 1e3 / reps);
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<256, 8>(17, 0, out); run<256, 8>(261, 0, out); run<256, 8>(625, 0, out); run<256, 8>(2500, 0, out);
        run<256, 120>(17, 0, out); run<256, 120>(261, 0, out); run<256, 120>(625, 0, out); run<256, 120>(2500, 0, out);
        run<256, 120>(625, 40, out); run<256, 120>(256, 100, out);
        run<512, 120>(313, 0, out); run<1024, 56>(157, 0, out); run<64, 120>(2500, 0, out);
    }
    return 0;
}
