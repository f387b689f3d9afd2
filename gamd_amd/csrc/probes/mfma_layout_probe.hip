// Probe: verify the chain-layout / packed-fragment contract of gamd_common.h on real hardware.
// Asymmetric random W and X; checks F1 (chain->chain), F2 (chain->row) and a 2-GEMM chain.
#include "../gamd_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__global__ void probe(const f32x4* __restrict__ Wp, const float* __restrict__ Xrows, float* __restrict__ Y1,
                      float* __restrict__ Y2, float* __restrict__ Y3) {
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    f32x16 X[4], acc[4];
    load_row_chain(Xrows + (size_t)slot * 128, half, X);
    // F1
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm128<false>(Wp, lane, X, acc);
    store_row_chain(Y1 + (size_t)slot * 128, half, acc);
    // chain: Y3 = W * (W * X)   (second GEMM consumes acc directly)
    f32x16 acc2[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    gemm128<false>(Wp, lane, acc, acc2);
    store_row_chain(Y3 + (size_t)slot * 128, half, acc2);
    // F2: row layout out
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm128<true>(Wp, lane, X, acc);
    for (int tp = 0; tp < 4; ++tp)
        for (int r = 0; r < 16; ++r) {
            const int s = (r & 3) + 8 * (r >> 2) + 4 * half;
            Y2[(size_t)s * 128 + 32 * tp + slot] = acc[tp][r];
        }
}

int main() {
    std::vector<float> W(128 * 128), X(32 * 128), Wp(128 * 128);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : X) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (int tp = 0; tp < 4; ++tp) for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
            int n = 32 * tp + (lane & 31), k = 32 * t + 8 * q + 4 * (lane >> 5) + j;
            Wp[((((tp * 4 + t) * 4 + q) * 64 + lane) * 4) + j] = W[n * 128 + k];
        }
    float *dW, *dX, *dY1, *dY2, *dY3;
    hipMalloc(&dW, 65536); hipMalloc(&dX, 16384); hipMalloc(&dY1, 16384); hipMalloc(&dY2, 16384); hipMalloc(&dY3, 16384);
    hipMemcpy(dW, Wp.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), 16384, hipMemcpyHostToDevice);
    probe<<<1, 64>>>((const f32x4*)dW, dX, dY1, dY2, dY3);
    std::vector<float> Y1(32 * 128), Y2(32 * 128), Y3(32 * 128);
    hipMemcpy(Y1.data(), dY1, 16384, hipMemcpyDeviceToHost);
    hipMemcpy(Y2.data(), dY2, 16384, hipMemcpyDeviceToHost);
    hipMemcpy(Y3.data(), dY3, 16384, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0;
    std::vector<double> R(32 * 128);
    for (int s = 0; s < 32; ++s) for (int n = 0; n < 128; ++n) {
        double a = 0; for (int k = 0; k < 128; ++k) a += (double)W[n * 128 + k] * X[s * 128 + k];
        R[s * 128 + n] = a;
        e1 = fmax(e1, fabs(a - Y1[s * 128 + n])); e2 = fmax(e2, fabs(a - Y2[s * 128 + n]));
    }
    for (int s = 0; s < 32; ++s) for (int n = 0; n < 128; ++n) {
        double a = 0; for (int k = 0; k < 128; ++k) a += (double)W[n * 128 + k] * R[s * 128 + k];
        e3 = fmax(e3, fabs(a - Y3[s * 128 + n]));
    }
    printf("mfma_layout_probe: errF1=%.3e errF2=%.3e errChain=%.3e  %s\n", e1, e2, e3,
           (e1 < 1e-4 && e2 < 1e-4 && e3 < 1e-3) ? "OK" : "FAIL");
    return (e1 < 1e-4 && e2 < 1e-4 && e3 < 1e-3) ? 0 : 1;
}
