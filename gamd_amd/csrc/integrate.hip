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

// ---- Nose-Hoover chain -------------------------------------------------------------------------
// sum of m v^2 (kJ/mol: v converted to nm/ps) per block, optionally after the half kick of the second half
template <bool KICK>
__global__ void __launch_bounds__(256) k_nhc_ke2(NhcArgs a) {
    __shared__ double red[4];
    double s = 0.0;
    const float kick = 0.5f * a.dt * 10.0f / a.mass;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * a.n; i += gridDim.x * blockDim.x) {
        float v = a.v[i];
        if (KICK) { v += kick * a.f[i]; a.v[i] = v; }          // hack_integrator.py:427 v+0.5*dt*gnn_force/m
        const double vn = 0.1 * (double)v;
        s += (double)a.mass * vn * vn;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// propagateNHC (hack_integrator.py:289-316), in double like OpenMM's global variables
__global__ void k_nhc_chain(NhcArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double KE2 = 0.0;
    for (int b = 0; b < a.n_blocks; ++b) KE2 += a.partial[b];
    const int M = a.M;
    double* xi = a.state;
    double* vxi = a.state + M;
    double* G = a.state + 2 * M;
    const double Q = a.kT / (a.freq * a.freq), Q0 = a.ndf * Q;
    double scale = 1.0;
    G[0] = (KE2 - a.ndf * a.kT) / Q0;
    for (int nc = 0; nc < a.n_c; ++nc)
        for (int ys = 0; ys < a.n_ys; ++ys) {
            const double wdt = a.w[ys] * (double)a.dt / a.n_c;
            vxi[M - 1] += 0.25 * wdt * G[M - 1];
            for (int j = M - 2; j >= 0; --j) {
                const double aa = exp(-0.125 * wdt * vxi[j + 1]);
                vxi[j] = aa * (aa * vxi[j] + 0.25 * wdt * G[j]);
            }
            scale *= exp(-0.5 * wdt * vxi[0]);
            for (int j = 0; j < M; ++j) xi[j] += 0.5 * wdt * vxi[j];
            G[0] = (scale * scale * KE2 - a.ndf * a.kT) / Q0;
            for (int j = 0; j < M - 1; ++j) {
                const double aa = exp(-0.125 * wdt * vxi[j + 1]);
                vxi[j] = aa * (aa * vxi[j] + 0.25 * wdt * G[j]);
                const double Qj = j == 0 ? Q0 : Q;
                G[j + 1] = (Qj * vxi[j] * vxi[j] - a.kT) / Q;
            }
            vxi[M - 1] += 0.25 * wdt * G[M - 1];
        }
    a.state[3 * M] = scale;
    a.state[3 * M + 1] = KE2;
}

// first half tail: v = scale*v; v += dt/2 f_last/m; x += dt v   (hack_integrator.py:274-280)
__global__ void k_nhc_apply_first(NhcArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float scale = (float)a.state[3 * a.M];
    const float kick = 0.5f * a.dt * 10.0f / a.mass;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d] * scale;
        v += kick * a.f[3 * i + d];
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(a.x[3 * i + d] + a.dt * v, a.box[d]);
    }
}

__global__ void k_nhc_apply_second(NhcArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * a.n) return;
    a.v[i] *= (float)a.state[3 * a.M];
}

}  // namespace

int launch_nhc_first(const NhcArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_nhc_ke2<false>, dim3(a.n_blocks), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3(1), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_apply_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_nhc_second(const NhcArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_nhc_ke2<true>, dim3(a.n_blocks), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3(1), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_apply_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    return 0;
}

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
