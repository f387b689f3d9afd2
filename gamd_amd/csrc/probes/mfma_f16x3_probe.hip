// Probe: fp32-grade GEMM on the fp16 matrix pipe by operand splitting (x = hi + lo, both fp16;
// y = W_hi x_hi + (W_hi x_lo + W_lo x_hi), fp32 accumulate) in the chain layout of gamd_common.h, with
// v_mfma_f32_32x32x16_f16.  Checks (1) that fp16 subnormal operands are honoured by the MFMA (the lo parts of small
// values are subnormal), (2) the layout (same as the bf16 variant), (3) the error against an f64 reference for
// fp32 data, next to the plain fp32 MFMA of the shipped kernels.
#include "../gamd_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8(const f32x16& v, int u, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[8 * u + j];
        const _Float16 h = (_Float16)x;
        hi[j] = h;
        lo[j] = (_Float16)(x - (float)h);
    }
}

template <bool F2>
__device__ __forceinline__ void gemm_f16x3(const f16x8* Wh, const f16x8* Wl, int lane, const f32x16 (&X)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 xh, xl;
            split8(X[t], u, xh, xl);
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const f16x8 wh = Wh[((tp * 4 + t) * 2 + u) * 64 + lane];
                const f16x8 wl = Wl[((tp * 4 + t) * 2 + u) * 64 + lane];
                if (F2) {
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc[tp], 0, 0, 0);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc[tp], 0, 0, 0);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc[tp], 0, 0, 0);
                } else {
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[tp], 0, 0, 0);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[tp], 0, 0, 0);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[tp], 0, 0, 0);
                }
            }
        }
}

__global__ void probe(const f16x8* __restrict__ Wh, const f16x8* __restrict__ Wl, const float* __restrict__ Wp32,
                      const float* __restrict__ Xrows, float* __restrict__ Y1, float* __restrict__ Y2,
                      float* __restrict__ Y32, float* __restrict__ den) {
    const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
    f32x16 X[4], acc[4];
    load_row_chain(Xrows + (size_t)slot * 128, half, X);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm_f16x3<false>(Wh, Wl, lane, X, acc);
    store_row_chain(Y1 + (size_t)slot * 128, half, acc);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm_f16x3<true>(Wh, Wl, lane, X, acc);
    for (int tp = 0; tp < 4; ++tp)
        for (int r = 0; r < 16; ++r) {
            const int s = (r & 3) + 8 * (r >> 2) + 4 * half;
            Y2[(size_t)s * 128 + 32 * tp + slot] = acc[tp][r];
        }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm128<false>((const f32x4*)Wp32, lane, X, acc);
    store_row_chain(Y32 + (size_t)slot * 128, half, acc);
    // subnormal test: a = 2^-20 (fp16 subnormal), b = 1 -> sum over K=16 is 2^-16 if honoured, 0 if flushed
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) { den[0] = c[0]; den[1] = (float)a[0]; }
    // subnormal RESULT of the conversion: (half)(3e-6f) must not be flushed either
    if (lane == 1) { volatile float tiny = 3.0e-6f; den[2] = (float)(_Float16)tiny; }
}

static unsigned short h_f2h(float f) {            // fp32 -> fp16 RNE incl. subnormals
    _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u;
}
static float h_h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main() {
    std::vector<float> W(128 * 128), X(32 * 128), Wp32(128 * 128);
    std::vector<unsigned short> Wh(128 * 128), Wl(128 * 128);
    srand(2);
    for (auto& v : W) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.18f;           // ~U(-1/sqrt(128), 1/sqrt(128))
    for (auto& v : X) v = ((rand() / (float)RAND_MAX) - 0.5f) * 4.0f * ((rand() & 7) ? 1.f : 1e-3f);   // some tiny entries
    for (int tp = 0; tp < 4; ++tp) for (int t = 0; t < 4; ++t) for (int u = 0; u < 2; ++u)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int r = 8 * u + j, half = lane >> 5;
            const int n = 32 * tp + (lane & 31), k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float w = W[n * 128 + k];
            const unsigned short hi = h_f2h(w);
            const size_t at = ((((tp * 4 + t) * 2 + u) * 64 + lane) * 8) + j;
            Wh[at] = hi; Wl[at] = h_f2h(w - h_h2f(hi));
        }
    for (int tp = 0; tp < 4; ++tp) for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j)
            Wp32[((((tp * 4 + t) * 4 + q) * 64 + lane) * 4) + j] = W[(32 * tp + (lane & 31)) * 128 + 32 * t + 8 * q + 4 * (lane >> 5) + j];
    void *dWh, *dWl; float *dW32, *dX, *dY1, *dY2, *dY32, *dden;
    hipMalloc(&dWh, 32768); hipMalloc(&dWl, 32768); hipMalloc(&dW32, 65536); hipMalloc(&dX, 16384);
    hipMalloc(&dY1, 16384); hipMalloc(&dY2, 16384); hipMalloc(&dY32, 16384); hipMalloc(&dden, 64);
    hipMemcpy(dWh, Wh.data(), 32768, hipMemcpyHostToDevice); hipMemcpy(dWl, Wl.data(), 32768, hipMemcpyHostToDevice);
    hipMemcpy(dW32, Wp32.data(), 65536, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), 16384, hipMemcpyHostToDevice);
    probe<<<1, 64>>>((const f16x8*)dWh, (const f16x8*)dWl, dW32, dX, dY1, dY2, dY32, dden);
    std::vector<float> Y1(32 * 128), Y2(32 * 128), Y32(32 * 128); float den[4];
    hipMemcpy(Y1.data(), dY1, 16384, hipMemcpyDeviceToHost); hipMemcpy(Y2.data(), dY2, 16384, hipMemcpyDeviceToHost);
    hipMemcpy(Y32.data(), dY32, 16384, hipMemcpyDeviceToHost); hipMemcpy(den, dden, 16, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e32 = 0, ymax = 0;
    for (int s = 0; s < 32; ++s) for (int n = 0; n < 128; ++n) {
        double ref = 0; for (int k = 0; k < 128; ++k) ref += (double)W[n * 128 + k] * (double)X[s * 128 + k];
        ymax = fmax(ymax, fabs(ref));
        e1 = fmax(e1, fabs(Y1[s * 128 + n] - ref)); e2 = fmax(e2, fabs(Y2[s * 128 + n] - ref)); e32 = fmax(e32, fabs(Y32[s * 128 + n] - ref));
    }
    printf("subnormal operand: mfma sum = %g (expected %g), a = %g; cvt of 3e-6 -> %g\n", den[0], 16 * 9.5367431640625e-07, den[1], den[2]);
    printf("max|y| = %g; max abs err vs f64: f16x3 F1 %.3e  F2 %.3e   fp32 MFMA %.3e\n", ymax, e1, e2, e32);
    printf("relative (max/max): f16x3 %.3e  fp32 %.3e\n", fmax(e1, e2) / ymax, e32 / ymax);
    const bool ok = den[0] > 0 && fmax(e1, e2) / ymax < 2e-6;
    printf("%s\n", ok ? "PROBE OK" : "PROBE FAILED");
    return ok ? 0 : 1;
}
