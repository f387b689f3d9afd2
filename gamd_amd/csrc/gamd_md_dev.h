// gamd_md_dev.h — per-atom / per-rigid-molecule device code of the split BAOAB step (hack_integrator.py:141-165,175-178),
// shared by the integrator kernels (integrate.hip) and the neighbour-stage kernels that carry the two halves in skin mode
// (neighbor.hip: k_step_small, k_skin_check).
#pragma once
#include "gamd_common.h"
#include "gamd_internal.h"

// No floating-point contraction in the integrator code: the same per-atom / per-molecule function is inlined into several
// kernels (k_baoab_*, k_baoab_second_com, k_skin_check, k_step_small), which multiply-add pairs hipcc fuses depends on the
// kernel around it, and a trajectory must not depend on which of them carried a half step (a run enqueued in several
// gamd_md_run calls equals the same run in one; fused == stand-alone launches).  Restored at the end of this header.
#pragma clang fp contract(off)

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

// Several boxes in one launch (MdArgs::bx.n_boxes > 1): atom i of the batch is atom i - b n_per_box of box b and draws the
// noise a single-box run with seed + b draws for it, so a batch reproduces its boxes run one by one.
struct BoxAtom { int box, local; };
__device__ __forceinline__ BoxAtom md_box_atom(const BoxRef& r, int i) {
    if (r.n_boxes <= 1) return BoxAtom{0, i};
    const int b = gamd_box_of(r, i);
    return BoxAtom{b, i - b * r.n_per_box};
}
__device__ __forceinline__ void md_box(const BoxRef& r, const float (&box)[3], int b, float (&out)[3]) {
    if (r.n_boxes <= 1) { out[0] = box[0]; out[1] = box[1]; out[2] = box[2]; return; }
    const float4 v = r.boxes[3 * b];
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

// centre-of-mass velocity of box `box` from k_com_partial's per-block sums, added up in block order (every thread of the box
// gets the same bits); OpenMM's CMMotionRemover accumulates in double as well
__device__ __forceinline__ void md_com_velocity(const MdCom& c, int box, float (&out)[3]) {
    double px = 0.0, py = 0.0, pz = 0.0, m = 0.0;
    const double* p = c.partial + (size_t)box * c.blocks * 4;
    for (int k = 0; k < c.blocks; ++k) { px += p[4 * k]; py += p[4 * k + 1]; pz += p[4 * k + 2]; m += p[4 * k + 3]; }
    const double inv = 1.0 / m;
    out[0] = (float)(px * inv); out[1] = (float)(py * inv); out[2] = (float)(pz * inv);
}

// first half of a step for atom i: [COM motion removal] B A O A, positions re-wrapped (hack_integrator.py:141-165)
__device__ __forceinline__ void d_baoab_first_atom(const MdArgs& a, int i) {
    float xi[3], box[3], com[3] = {0.f, 0.f, 0.f};
    const BoxAtom ba = md_box_atom(a.bx, i);
    md_box(a.bx, a.box, ba.box, box);
    if (a.com.enabled) md_com_velocity(a.com, ba.box, com);          // updateContextState :142
    atom_noise(a.seed + (unsigned long long)ba.box, a.step, ba.local, xi);
    const float w = atom_inv_mass(a.species, a.inv_mass, a.inv_mass_h, i);
    const float hdt = 0.5f * a.dt, kick = hdt * a.len * w, bs = a.b_len_kT * sqrtf(w);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d] - com[d], x = a.x[3 * i + d];
        v += kick * a.f[3 * i + d];          // B
        x += hdt * v;                        // A
        v = a.a * v + bs * xi[d];            // O
        x += hdt * v;                        // A
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(x, box[d]);
    }
}

// second half for degree of freedom i in [0, 3n): B with the new forces (hack_integrator.py:175-178)
__device__ __forceinline__ void d_baoab_second_dof(const MdArgs& a, int i) {
    a.v[i] += 0.5f * a.dt * a.len * atom_inv_mass(a.species, a.inv_mass, a.inv_mass_h, i / 3) * a.f[i];
}

namespace gamd_md {

// ---- rigid 3-site water ------------------------------------------------------------------------
// The water drivers integrate rigid molecules: OpenMM applies the constraints where the hacked integrators
// say addConstrainPositions / addConstrainVelocities (hack_integrator.py:145-164,178,277-280,427-428).  For a
// 3-site molecule both have closed forms: SETTLE (Miyamoto & Kollman, J. Comput. Chem. 13, 952 (1992)) for the
// positions and a 3x3 linear solve for the velocities.  One thread owns one molecule (atoms O,H,H).
struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Vec3 unit(Vec3 a) { return (1.0f / sqrtf(dot(a, a))) * a; }

// x0: constrained reference geometry; x1: unconstrained new positions -> constrained new positions (in place).
// Works in coordinates relative to the old oxygen so that fp32 carries the bond lengths, not the box size.
__device__ __forceinline__ void settle_positions(const Vec3 (&x0)[3], Vec3 (&x1)[3], const RigidWater& g) {
    const Vec3 b0 = x0[1] - x0[0], c0 = x0[2] - x0[0];
    const Vec3 A1 = x1[0] - x0[0], B1 = x1[1] - x0[0], C1 = x1[2] - x0[0];
    const float inv_m = 1.0f / (g.m_o + 2.0f * g.m_h);
    const Vec3 d0 = inv_m * ((g.m_o * A1) + (g.m_h * (B1 + C1)));
    const Vec3 a1 = A1 - d0, b1 = B1 - d0, c1 = C1 - d0;
    Vec3 n0 = cross(b0, c0), n1 = cross(a1, n0), n2 = cross(n0, n1);
    n0 = unit(n0); n1 = unit(n1); n2 = unit(n2);
    const float b0x = dot(b0, n1), b0y = dot(b0, n2), c0x = dot(c0, n1), c0y = dot(c0, n2);
    const float a1z = dot(a1, n0);
    const float b1x = dot(b1, n1), b1y = dot(b1, n2), b1z = dot(b1, n0);
    const float c1x = dot(c1, n1), c1y = dot(c1, n2), c1z = dot(c1, n0);
    const float sinphi = a1z / g.ra, cosphi = sqrtf(1.0f - sinphi * sinphi);
    const float sinpsi = (b1z - c1z) / (2.0f * g.rc * cosphi), cospsi = sqrtf(1.0f - sinpsi * sinpsi);
    const float a2y = g.ra * cosphi, a2z = g.ra * sinphi;
    const float b2x = -g.rc * cospsi, b2y = -g.rb * cosphi - g.rc * sinpsi * sinphi, b2z = -g.rb * sinphi + g.rc * sinpsi * cosphi;
    const float c2x = g.rc * cospsi, c2y = -g.rb * cosphi + g.rc * sinpsi * sinphi, c2z = -g.rb * sinphi - g.rc * sinpsi * cosphi;
    const float alpha = b2x * (b0x - c0x) + b0y * b2y + c0y * c2y;
    const float beta = b2x * (c0y - b0y) + b0x * b2y + c0x * c2y;
    const float gamma = (b0x * b1y - b1x * b0y) + (c0x * c1y - c1x * c0y);
    const float a2b2 = alpha * alpha + beta * beta;
    const float sint = (alpha * gamma - beta * sqrtf(a2b2 - gamma * gamma)) / a2b2, cost = sqrtf(1.0f - sint * sint);
    const float a3x = -a2y * sint, a3y = a2y * cost;
    const float b3x = b2x * cost - b2y * sint, b3y = b2x * sint + b2y * cost;
    const float c3x = c2x * cost - c2y * sint, c3y = c2x * sint + c2y * cost;
    const Vec3 base = x0[0] + d0;
    x1[0] = base + ((a3x * n1) + (a3y * n2)) + (a2z * n0);
    x1[1] = base + ((b3x * n1) + (b3y * n2)) + (b2z * n0);
    x1[2] = base + ((c3x * n1) + (c3y * n2)) + (c2z * n0);
}

// remove the relative velocity along the three bonds: v_i += w_i sum_k (+-) g_k r_k with A g = -r_k.u_k
__device__ __forceinline__ void settle_velocities(const Vec3 (&x)[3], Vec3 (&v)[3], const RigidWater& g) {
    const float wo = 1.0f / g.m_o, wh = 1.0f / g.m_h;
    const Vec3 r0 = x[0] - x[1], r1 = x[0] - x[2], r2 = x[1] - x[2];
    const float y0 = -dot(r0, v[0] - v[1]), y1 = -dot(r1, v[0] - v[2]), y2 = -dot(r2, v[1] - v[2]);
    const float a00 = (wo + wh) * dot(r0, r0), a01 = wo * dot(r0, r1), a02 = -wh * dot(r0, r2);
    const float a11 = (wo + wh) * dot(r1, r1), a12 = wh * dot(r1, r2), a22 = 2.0f * wh * dot(r2, r2);
    // symmetric 3x3 solve by cofactors
    const float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const float inv_det = 1.0f / ((a00 * c00 + a01 * c01) + a02 * c02);
    const float g0 = ((c00 * y0 + c01 * y1) + c02 * y2) * inv_det;
    const float g1 = ((c01 * y0 + c11 * y1) + c12 * y2) * inv_det;
    const float g2 = ((c02 * y0 + c12 * y1) + c22 * y2) * inv_det;
    v[0] = v[0] + (wo * ((g0 * r0) + (g1 * r1)));
    v[1] = v[1] + (wh * ((g2 * r2) - (g0 * r0)));
    v[2] = v[2] - (wh * ((g1 * r1) + (g2 * r2)));
}

__device__ __forceinline__ void load_mol(const float* p, int m, Vec3 (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = {p[9 * m + 3 * k], p[9 * m + 3 * k + 1], p[9 * m + 3 * k + 2]};
}
__device__ __forceinline__ void store_mol(float* p, int m, const Vec3 (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { p[9 * m + 3 * k] = o[k].x; p[9 * m + 3 * k + 1] = o[k].y; p[9 * m + 3 * k + 2] = o[k].z; }
}
// keep the molecule whole: translate all three atoms by the lattice vector that brings the oxygen into the box
__device__ __forceinline__ void wrap_mol(Vec3 (&x)[3], const float (&box)[3]) {
    const Vec3 s = {floorf(x[0].x / box[0]) * box[0], floorf(x[0].y / box[1]) * box[1], floorf(x[0].z / box[2]) * box[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = x[k] - s;
}

// first half of a step for rigid molecule m (atoms 3m, 3m+1, 3m+2 = O, H, H): B A O A with the constraints applied where
// the hacked integrator has addConstrainVelocities / addConstrainPositions (hack_integrator.py:145-164)
__device__ __forceinline__ void d_baoab_first_mol(const MdArgs& a, int m) {
    Vec3 x[3], v[3], f[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
    const float hdt = 0.5f * a.dt;
    const BoxAtom ba = md_box_atom(a.bx, 3 * m);            // n_per_box is a multiple of 3: a molecule lies in one box
    float box[3];
    md_box(a.bx, a.box, ba.box, box);
    if (a.com.enabled) {                                                                   // updateContextState :142
        float com[3];
        md_com_velocity(a.com, ba.box, com);
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] - Vec3{com[0], com[1], com[2]};
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = v[k] + ((hdt * a.len * w[k]) * f[k]);          // B  :145
    settle_velocities(x, v, a.rigid);                                                     //    :146
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        Vec3 x1[3], xc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { x1[k] = x[k] + (hdt * v[k]); xc[k] = x1[k]; }      // A  :149 / :160
        settle_positions(x, xc, a.rigid);                                                 //    :151 / :162
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = v[k] + ((1.0f / hdt) * (xc[k] - x1[k])); x[k] = xc[k]; }   // :152 / :163
        settle_velocities(x, v, a.rigid);                                                 //    :153 / :164
        if (stage == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {                                                 // O  :157
                float xi[3];
                atom_noise(a.seed + (unsigned long long)ba.box, a.step, ba.local + k, xi);
                const float bs = a.b_len_kT * sqrtf(w[k]);
                v[k] = (a.a * v[k]) + Vec3{bs * xi[0], bs * xi[1], bs * xi[2]};
            }
            settle_velocities(x, v, a.rigid);                                             //    :158
        }
    }
    wrap_mol(x, box);
    store_mol(a.x, m, x); store_mol(a.v, m, v);
}

// second half for rigid molecule m: B + velocity constraints (hack_integrator.py:177-178)
__device__ __forceinline__ void d_baoab_second_mol(const MdArgs& a, int m) {
    Vec3 x[3], v[3], f[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = v[k] + ((0.5f * a.dt * a.len * w[k]) * f[k]);
    settle_velocities(x, v, a.rigid);
    store_mol(a.v, m, v);
}

}  // namespace gamd_md

#pragma clang fp contract(fast)
