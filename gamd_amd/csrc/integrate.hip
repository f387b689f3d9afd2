// integrate.hip — the split BAOAB Langevin scheme of the reference's hacked OpenMM integrators,
// as two elementwise kernels so the MD loop never leaves the device.
//
//   first half  (HackLangevinIntegrator, code/hack_integrator.py:141-165, no constraints):
//       B: v += (dt/2) f_last/m ;  A: x += (dt/2) v ;  O: v = a v + b sigma xi ;  A: x += (dt/2) v
//   second half (HackHalfVelocityIntegrator, code/hack_integrator.py:175-178):
//       B: v += (dt/2) f_gnn/m
// Units: x in a length unit L (Angstrom for LJ/TIP, bohr for the DFT model), v L/ps, f kJ/mol/nm, m amu
//   ->  a[L/ps^2] = len f/m with len = L per nm (10 for Angstrom).  Species-0 atoms (H) may have their own mass.
// Noise: Philox4x32-10 counter RNG keyed by (seed, step, atom) + Box-Muller, so a trajectory is
// reproducible and independent of launch geometry.  Positions are re-wrapped into the box each step
// (the reference reads them back with enforcePeriodicBox=True, test_langevin.py:102-105).
#include "gamd_common.h"
#include "gamd_internal.h"
#include "gamd_md_dev.h"

namespace {

// A neighbour buffer overflowed in an earlier force evaluation of this enqueued run: the forces are stale, so every
// integrator kernel returns without touching x, v or the thermostat chain; the first one to notice records where the
// run stopped (2 * step index + half) for the host to resume from after regrowing (gamd_sync_status).
#define GAMD_MD_GATE(HALF)                                                                        \
    do {                                                                                          \
        if (a.devflags[DEVFLAG_FROZEN]) {                                                         \
            if (blockIdx.x == 0 && threadIdx.x == 0 && a.devflags[DEVFLAG_FROZEN_AT] < 0)         \
                a.devflags[DEVFLAG_FROZEN_AT] = 2 * a.step_index + (HALF);                        \
            return;                                                                               \
        }                                                                                         \
    } while (0)

__global__ void k_baoab_first(MdArgs a) {
    GAMD_MD_GATE(0);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) d_baoab_first_atom(a, i);
}

__global__ void k_baoab_second(MdArgs a) {
    GAMD_MD_GATE(1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * a.n) d_baoab_second_dof(a, i);
}

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

// HackLangevinIntegrator with constraints, hack_integrator.py:141-165, one molecule per thread
__global__ void k_baoab_first_rigid(MdArgs a) {
    GAMD_MD_GATE(0);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m >= a.n) return;
    Vec3 x[3], v[3], f[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
    const float hdt = 0.5f * a.dt;
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
                atom_noise(a.seed, a.step, 3 * m + k, xi);
                const float bs = a.b_len_kT * sqrtf(w[k]);
                v[k] = (a.a * v[k]) + Vec3{bs * xi[0], bs * xi[1], bs * xi[2]};
            }
            settle_velocities(x, v, a.rigid);                                             //    :158
        }
    }
    wrap_mol(x, a.box);
    store_mol(a.x, m, x); store_mol(a.v, m, v);
}

// HackHalfVelocityIntegrator with constraints, hack_integrator.py:177-178
__global__ void k_baoab_second_rigid(MdArgs a) {
    GAMD_MD_GATE(1);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m >= a.n) return;
    Vec3 x[3], v[3], f[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = v[k] + ((0.5f * a.dt * a.len * w[k]) * f[k]);
    settle_velocities(x, v, a.rigid);
    store_mol(a.v, m, v);
}

// ---- Nose-Hoover chain -------------------------------------------------------------------------
// sum of m v^2 (kJ/mol: v converted to nm/ps) per block, optionally after the half kick of the second half
// (and, for rigid water, the velocity constraint that follows it: hack_integrator.py:427-428)
template <bool KICK>
__global__ void __launch_bounds__(256) k_nhc_ke2(NhcArgs a) {
    GAMD_MD_GATE(KICK ? 1 : 0);
    __shared__ double red[4];
    double s = 0.0;
    const double inv_len = 1.0 / (double)a.len;
    if (a.use_rigid) {
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; 3 * m < a.n; m += gridDim.x * blockDim.x) {
            Vec3 v[3];
            load_mol(a.v, m, v);
            const float ms[3] = {a.rigid.m_o, a.rigid.m_h, a.rigid.m_h};
            if (KICK) {
                Vec3 x[3], f[3];
                load_mol(a.x, m, x); load_mol(a.f, m, f);
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = v[k] + ((0.5f * a.dt * a.len / ms[k]) * f[k]);
                settle_velocities(x, v, a.rigid);
                store_mol(a.v, m, v);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vx = inv_len * v[k].x, vy = inv_len * v[k].y, vz = inv_len * v[k].z;
                s += (double)ms[k] * ((vx * vx + vy * vy) + vz * vz);
            }
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * a.n; i += gridDim.x * blockDim.x) {
            const float w = atom_inv_mass(a.species, 1.0f / a.mass, a.mass_h > 0.f ? 1.0f / a.mass_h : 0.f, i / 3);
            float v = a.v[i];
            if (KICK) { v += 0.5f * a.dt * a.len * w * a.f[i]; a.v[i] = v; }   // hack_integrator.py:427 v+0.5*dt*gnn_force/m
            const double vn = inv_len * (double)v;
            s += vn * vn / (double)w;
        }
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
    if (a.devflags[DEVFLAG_FROZEN]) return;                 // k_nhc_ke2 in front of it has recorded the position
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
    GAMD_MD_GATE(0);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float scale = (float)a.state[3 * a.M];
    const float kick = 0.5f * a.dt * a.len * atom_inv_mass(a.species, 1.0f / a.mass, a.mass_h > 0.f ? 1.0f / a.mass_h : 0.f, i);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d] * scale;
        v += kick * a.f[3 * i + d];
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(a.x[3 * i + d] + a.dt * v, a.box[d]);
    }
}

// the same with constraints: x1 = x + dt v; ConstrainPositions; v += (x - x1)/dt   (:277-280)
__global__ void k_nhc_apply_first_rigid(NhcArgs a) {
    GAMD_MD_GATE(0);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m >= a.n) return;
    Vec3 x[3], v[3], f[3], x1[3], xc[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const float scale = (float)a.state[3 * a.M];
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = (scale * v[k]) + ((0.5f * a.dt * a.len * w[k]) * f[k]);
        x1[k] = x[k] + (a.dt * v[k]);
        xc[k] = x1[k];
    }
    settle_positions(x, xc, a.rigid);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = v[k] + ((1.0f / a.dt) * (xc[k] - x1[k]));
    wrap_mol(xc, a.box);
    store_mol(a.x, m, xc); store_mol(a.v, m, v);
}

__global__ void k_nhc_apply_second(NhcArgs a) {
    GAMD_MD_GATE(1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * a.n) return;
    a.v[i] *= (float)a.state[3 * a.M];
}

}  // namespace

int launch_nhc_first(const NhcArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_nhc_ke2<false>, dim3(a.n_blocks), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3(1), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    if (a.use_rigid) hipLaunchKernelGGL(k_nhc_apply_first_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_nhc_apply_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_nhc_second(const NhcArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_nhc_ke2<true>, dim3(a.n_blocks), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3(1), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_apply_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_first(const MdArgs& a, hipStream_t st) {
    if (a.use_rigid) hipLaunchKernelGGL(k_baoab_first_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_baoab_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_second(const MdArgs& a, hipStream_t st) {
    if (a.use_rigid) hipLaunchKernelGGL(k_baoab_second_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_baoab_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
