// integrate.hip — the split BAOAB Langevin scheme of the reference's hacked OpenMM integrators,
// as two elementwise kernels so the MD loop never leaves the device.
//
//   first half  (HackLangevinIntegrator, code/hack_integrator.py:141-165, no constraints):
//       B: v += (dt/2) f_last/m ;  A: x += (dt/2) v ;  O: v = a v + b sigma xi ;  A: x += (dt/2) v
//   second half (HackHalfVelocityIntegrator, code/hack_integrator.py:175-178):
//       B: v += (dt/2) f_gnn/m
// Units: x Angstrom, v Angstrom/ps, f kJ/mol/nm, m amu  ->  a[Angstrom/ps^2] = 10 f/m.
// Noise: Philox4x32-10 counter RNG keyed by (seed, step, atom) + Box-Muller, so a trajectory is
// reproducible and independent of launch geometry.  Positions are re-wrapped into the box each step
// (the reference reads them back with enforcePeriodicBox=True, test_langevin.py:102-105).
#include "gamd_common.h"
#include "gamd_internal.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void k_baoab_first(MdArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(a.step & 0xffffffffu), (uint32_t)(a.step >> 32), 0x47414D44u};
    philox4x32_10(c, (uint32_t)(a.seed & 0xffffffffu), (uint32_t)(a.seed >> 32));
    const float r0 = sqrtf(-2.0f * logf(u01(c[0]))), t0 = 6.28318530717958647692f * u01(c[1]);
    const float r1 = sqrtf(-2.0f * logf(u01(c[2]))), t1 = 6.28318530717958647692f * u01(c[3]);
    const float xi[3] = {r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1)};
    const float hdt = 0.5f * a.dt, kick = hdt * 10.0f * a.inv_mass;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d], x = a.x[3 * i + d];
        v += kick * a.f[3 * i + d];          // B
        x += hdt * v;                        // A
        v = a.a * v + a.b_sigma * xi[d];     // O
        x += hdt * v;                        // A
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(x, a.box[d]);
    }
}

__global__ void k_baoab_second(MdArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * a.n) return;
    a.v[i] += 0.5f * a.dt * 10.0f * a.inv_mass * a.f[i];
}

}  // namespace

int launch_baoab_first(const MdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_baoab_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_second(const MdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_baoab_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
