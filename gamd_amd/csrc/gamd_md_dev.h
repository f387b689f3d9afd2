// gamd_md_dev.h — per-atom device code of the split BAOAB step (hack_integrator.py:141-165,175-178), shared by the
// integrator kernels (integrate.hip) and the fused small-system step kernel (neighbor.hip).
#pragma once
#include "gamd_common.h"
#include "gamd_internal.h"

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

// three standard normals for atom i at step a.step (same stream whether the atom is integrated alone or as
// part of a rigid molecule)
__device__ __forceinline__ void atom_noise(unsigned long long seed, unsigned long long step, int i, float (&xi)[3]) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(step & 0xffffffffu), (uint32_t)(step >> 32), 0x47414D44u};
    philox4x32_10(c, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32));
    const float r0 = sqrtf(-2.0f * logf(u01(c[0]))), t0 = 6.28318530717958647692f * u01(c[1]);
    const float r1 = sqrtf(-2.0f * logf(u01(c[2]))), t1 = 6.28318530717958647692f * u01(c[3]);
    xi[0] = r0 * cosf(t0); xi[1] = r0 * sinf(t0); xi[2] = r1 * cosf(t1);
}

// 1/m of atom i: species-0 atoms (H) take the second mass when one is given
__device__ __forceinline__ float atom_inv_mass(const uint8_t* species, float inv_mass, float inv_mass_h, int i) {
    return (species && inv_mass_h > 0.f && species[i] == 0) ? inv_mass_h : inv_mass;
}

// first half of a step for atom i: B A O A, positions re-wrapped (hack_integrator.py:141-165)
__device__ __forceinline__ void d_baoab_first_atom(const MdArgs& a, int i) {
    float xi[3];
    atom_noise(a.seed, a.step, i, xi);
    const float w = atom_inv_mass(a.species, a.inv_mass, a.inv_mass_h, i);
    const float hdt = 0.5f * a.dt, kick = hdt * a.len * w, bs = a.b_len_kT * sqrtf(w);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d], x = a.x[3 * i + d];
        v += kick * a.f[3 * i + d];          // B
        x += hdt * v;                        // A
        v = a.a * v + bs * xi[d];            // O
        x += hdt * v;                        // A
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(x, a.box[d]);
    }
}

// second half for degree of freedom i in [0, 3n): B with the new forces (hack_integrator.py:175-178)
__device__ __forceinline__ void d_baoab_second_dof(const MdArgs& a, int i) {
    a.v[i] += 0.5f * a.dt * a.len * atom_inv_mass(a.species, a.inv_mass, a.inv_mass_h, i / 3) * a.f[i];
}
