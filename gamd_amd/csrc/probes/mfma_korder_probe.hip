// mfma_korder_probe.hip — is one v_mfma_f32_16x16x4_f32 the same fused multiply-add chain, bit for bit, as two
// v_mfma_f32_32x32x2_f32 over the same four K values?  (A 16-edge half-tile conv kernel can only be bit-identical to the 32-edge
// one if both shapes accumulate k = 0, 1, 2, 3 as ((((c + a0 b0) + a1 b1) + a2 b2) + a3 b3) with one rounding per step.)
// Random operands incl. large cancellations; compared with the fmaf chain computed by plain VALU code.  GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: [32][4] row-major, B: [4][32], C: [32][32]; out32 = two 32x32x2 MFMAs (k 0,1 then 2,3), out16 = 16x16x4 on the 4 quadrants,
// ref = fmaf chain
__global__ void k(const float* A, const float* B, const float* C, float* out32, float* out16, float* ref) {
    const int l = threadIdx.x;
    {   // 32x32x2: lane -> (row / col = l & 31, k = l >> 5)
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
        for (int s = 0; s < 2; ++s) {
            const float a = A[(l & 31) * 4 + 2 * s + (l >> 5)], b = B[(2 * s + (l >> 5)) * 32 + (l & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
    }
    for (int qi = 0; qi < 2; ++qi)
        for (int qj = 0; qj < 2; ++qj) {   // 16x16x4: lane -> (row / col = l & 15, k = l >> 4); D regs r -> row 4 (l >> 4) + r
            f32x4 acc;
            for (int r = 0; r < 4; ++r) acc[r] = C[(16 * qi + 4 * (l >> 4) + r) * 32 + 16 * qj + (l & 15)];
            const float a = A[(16 * qi + (l & 15)) * 4 + (l >> 4)], b = B[(l >> 4) * 32 + 16 * qj + (l & 15)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) out16[(16 * qi + 4 * (l >> 4) + r) * 32 + 16 * qj + (l & 15)] = acc[r];
        }
    for (int e = l; e < 1024; e += 64) {
        const int i = e >> 5, j = e & 31;
        float c = C[e];
        for (int kk = 0; kk < 4; ++kk) c = __builtin_fmaf(A[i * 4 + kk], B[kk * 32 + j], c);
        ref[e] = c;
    }
}

int main() {
    std::vector<float> A(128), B(128), C(1024), o32(1024), o16(1024), rf(1024);
    float *dA, *dB, *dC, *d32, *d16, *dR;
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dC, 4096);
    (void)hipMalloc(&d32, 4096); (void)hipMalloc(&d16, 4096); (void)hipMalloc(&dR, 4096);
    long bad32 = 0, bad16 = 0, bad_x = 0, n = 0;
    srand(7);
    for (int trial = 0; trial < 2000; ++trial) {
        const float scale = (trial & 1) ? 1.0f : 1e3f;
        for (auto& v : A) v = scale * (rand() / (float)RAND_MAX - 0.5f);
        for (auto& v : B) v = rand() / (float)RAND_MAX - 0.5f;
        for (auto& v : C) v = (trial & 2) ? -scale * 0.25f * (rand() / (float)RAND_MAX) : (rand() / (float)RAND_MAX - 0.5f);   // cancellations
        (void)hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, d32, d16, dR);
        (void)hipMemcpy(o32.data(), d32, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o16.data(), d16, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(rf.data(), dR, 4096, hipMemcpyDeviceToHost);
        for (int e = 0; e < 1024; ++e) {
            ++n;
            if (memcmp(&o32[e], &rf[e], 4)) ++bad32;
            if (memcmp(&o16[e], &rf[e], 4)) ++bad16;
            if (memcmp(&o16[e], &o32[e], 4)) ++bad_x;
        }
    }
    printf("%ld elements: 2 x 32x32x2 != fmaf chain: %ld;  16x16x4 != fmaf chain: %ld;  16x16x4 != 2 x 32x32x2: %ld\n", n, bad32, bad16, bad_x);
    return 0;
}
