// Probe: chain layout with v_mfma_f32_32x32x16_bf16 (K = 16 per instruction, 8 bf16 per lane per operand).
// Activation stays in fp32 accumulator registers; per 32-feature tile t a lane owns 16 features
//   feat(t, r, half) = 32t + (r&3) + 8(r>>2) + 4half,  r = 0..15
// and feeds them to the next GEMM as two K-steps (u = 0,1) of 8 packed bf16 (r = 8u..8u+7).
// Weight fragment for (tp, t, u): lane (n = lane&31, half) holds W[32tp+n][feat(t, 8u+j, half)], j = 0..7.
#include "../gamd_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float x) {      // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static unsigned short h_f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int u) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(v[8 * u + j]);
    return o;
}

template <bool F2>
__device__ __forceinline__ void gemm_bf16(const bf16x8* W, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 w = W[((tp * 4 + t) * 2 + u) * 64 + lane];
                const bf16x8 x = pack8(X[t], u);
                acc[tp] = F2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, acc[tp], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc[tp], 0, 0, 0);
            }
}

__global__ void probe(const bf16x8* __restrict__ Wp, const float* __restrict__ Xrows, float* __restrict__ Y1,
                      float* __restrict__ Y2, float* __restrict__ Y3) {
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    f32x16 X[4], acc[4], acc2[4];
    load_row_chain(Xrows + (size_t)slot * 128, half, X);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc2[t][r] = 0.f; }
    gemm_bf16<false>(Wp, lane, X, acc);
    store_row_chain(Y1 + (size_t)slot * 128, half, acc);
    gemm_bf16<false>(Wp, lane, acc, acc2);                       // chain: consumes acc directly
    store_row_chain(Y3 + (size_t)slot * 128, half, acc2);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm_bf16<true>(Wp, lane, X, acc);
    for (int tp = 0; tp < 4; ++tp)
        for (int r = 0; r < 16; ++r) {
            const int s = (r & 3) + 8 * (r >> 2) + 4 * half;
            Y2[(size_t)s * 128 + 32 * tp + slot] = acc[tp][r];
        }
}

int main() {
    std::vector<float> W(128 * 128), X(32 * 128);
    std::vector<unsigned short> Wp(128 * 128);
    srand(2);
    for (auto& v : W) v = h_bf2f(h_f2bf((rand() / (float)RAND_MAX) - 0.5f));     // bf16-exact test data
    for (auto& v : X) v = h_bf2f(h_f2bf((rand() / (float)RAND_MAX) - 0.5f));
    for (int tp = 0; tp < 4; ++tp) for (int t = 0; t < 4; ++t) for (int u = 0; u < 2; ++u)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int r = 8 * u + j, half = lane >> 5;
            const int n = 32 * tp + (lane & 31), k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            Wp[((((tp * 4 + t) * 2 + u) * 64 + lane) * 8) + j] = h_f2bf(W[n * 128 + k]);
        }
    void *dW; float *dX, *dY1, *dY2, *dY3;
    hipMalloc(&dW, 32768); hipMalloc(&dX, 16384); hipMalloc(&dY1, 16384); hipMalloc(&dY2, 16384); hipMalloc(&dY3, 16384);
    hipMemcpy(dW, Wp.data(), 32768, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), 16384, hipMemcpyHostToDevice);
    probe<<<1, 64>>>((const bf16x8*)dW, dX, dY1, dY2, dY3);
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
        double a = 0; for (int k = 0; k < 128; ++k) a += (double)W[n * 128 + k] * h_bf2f(h_f2bf((float)R[s * 128 + k]));
        e3 = fmax(e3, fabs(a - Y3[s * 128 + n]));
    }
    const bool ok = e1 < 1e-4 && e2 < 1e-4 && e3 < 2e-2;
    printf("mfma_bf16_layout_probe: errF1=%.3e errF2=%.3e errChain=%.3e  %s\n", e1, e2, e3, ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
