#include "../gamd_common.h"
#include <cstdio>
#include <vector>
#include <cmath>
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
struct GeluCoef { gelu_f2 q[7]; };
__device__ __forceinline__ GeluCoef gelu_coef() {
    GeluCoef k;
    const float c[7] = {GAMD_GELU_Q0, GAMD_GELU_Q1, GAMD_GELU_Q2, GAMD_GELU_Q3, GAMD_GELU_Q4, GAMD_GELU_Q5, GAMD_GELU_Q6};
#pragma unroll
    for (int i = 0; i < 7; ++i) { k.q[i] = gelu_f2{c[i], c[i]}; asm volatile("" : "+v"(k.q[i])); }
    return k;
}
__device__ __forceinline__ gelu_f2 pk_fma(gelu_f2 a, gelu_f2 b, gelu_f2 c) {
    gelu_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ gelu_f2 gelu_pair(gelu_f2 x, const GeluCoef& k) {
    const gelu_f2 a = {__builtin_amdgcn_fmed3f(fabsf(x[0]), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(x[1]), 0.0f, 6.0f)};
    gelu_f2 q = pk_fma(k.q[6], a, k.q[5]);
    q = pk_fma(q, a, k.q[4]);
    q = pk_fma(q, a, k.q[3]);
    q = pk_fma(q, a, k.q[2]);
    q = pk_fma(q, a, k.q[1]);
    q = pk_fma(q, a, k.q[0]);
    const gelu_f2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const gelu_f2 relu = {x[0] - __builtin_amdgcn_fmed3f(x[0], -3.0e38f, 0.0f), x[1] - __builtin_amdgcn_fmed3f(x[1], -3.0e38f, 0.0f)};
    gelu_f2 r;
    asm("s_nop 1\n\tv_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(e), "v"(relu));
    return r;
}
__global__ void k(const float* x, float* y0, float* y1, int n) {
    const GeluCoef gk = gelu_coef();
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        gelu_f2 r = gelu_pair(gelu_f2{x[i], x[i + 1]}, gk);
        y0[i] = r[0]; y0[i + 1] = r[1];
        y1[i] = gamd_gelu_hw(x[i]); y1[i + 1] = gamd_gelu_hw(x[i + 1]);
    }
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n), a(n), b(n);
    for (int i = 0; i < n; ++i) h[i] = -9.0f + 18.0f * (float)((i * 2654435761u) % 100003) / 100003.0f;
    float *dx, *d0, *d1; hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, d0, d1, n);
    hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; double md = 0;
    for (int i = 0; i < n; ++i) { if (a[i] != b[i]) { if (bad < 5) printf("x=%g pk=%g scalar=%g\n", h[i], a[i], b[i]); ++bad; } md = fmax(md, fabs(a[i] - b[i])); }
    printf("mismatches %d of %d, max diff %g\n", bad, n, md);
    return 0;
}
