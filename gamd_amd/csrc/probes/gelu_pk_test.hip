#include "../gamd_common.h"
#include <cstdio>
#include <vector>
#include <cmath>
// gelu_pair / gelu_coef: the production forms in gamd_common.h
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
