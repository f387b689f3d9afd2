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

using namespace gamd_md;


// HackLangevinIntegrator with constraints, hack_integrator.py:141-165, one molecule per thread
__global__ void k_baoab_first_rigid(MdArgs a) {
    GAMD_MD_GATE(0);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m < a.n) d_baoab_first_mol(a, m);
}

// HackHalfVelocityIntegrator with constraints, hack_integrator.py:177-178
__global__ void k_baoab_second_rigid(MdArgs a) {
    GAMD_MD_GATE(1);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m < a.n) d_baoab_second_mol(a, m);
}

// ---- centre-of-mass motion removal (MdCom) ------------------------------------------------------
// per-block sums of m v and m over this block's share of box blockIdx.y's atoms; double accumulation, fixed reduction tree
// by_molecule (rigid water): a thread adds up whole O,H,H molecules q = t, t + T, ... instead of atoms i = t, t + T, ...: the
// assignment k_baoab_second_com needs (its threads own molecules), so that the sums have the same bits whichever of the two
// kernels produced them (a run enqueued in several gamd_md_run calls equals the same run in one)
__device__ __forceinline__ void com_add_atom(double (&s)[4], const float* __restrict__ v, const uint8_t* __restrict__ species,
                                             float inv_mass, float inv_mass_h, int i) {
#pragma clang fp contract(off)                              // the same bits in k_com_partial and k_baoab_second_com
    const double m = 1.0 / (double)atom_inv_mass(species, inv_mass, inv_mass_h, i);
    s[0] += m * (double)v[3 * i]; s[1] += m * (double)v[3 * i + 1]; s[2] += m * (double)v[3 * i + 2]; s[3] += m;
}
__global__ void __launch_bounds__(256) k_com_partial(MdCom c, const float* __restrict__ v, const uint8_t* __restrict__ species,
                                                     float inv_mass, float inv_mass_h, int n, BoxRef bx, const int* devflags,
                                                     int by_molecule) {
    if (devflags[DEVFLAG_FROZEN]) return;                   // the first-half kernel behind this one records the position
    __shared__ double red[4][4];
    const int npb = bx.n_boxes > 1 ? bx.n_per_box : n, a0 = blockIdx.y * npb, a1 = a0 + npb;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (by_molecule) {
        for (int q = a0 / 3 + blockIdx.x * blockDim.x + threadIdx.x; 3 * q < a1; q += gridDim.x * blockDim.x)
            for (int k = 0; k < 3; ++k) com_add_atom(s, v, species, inv_mass, inv_mass_h, 3 * q + k);
    } else {
        for (int i = a0 + blockIdx.x * blockDim.x + threadIdx.x; i < a1; i += gridDim.x * blockDim.x)
            com_add_atom(s, v, species, inv_mass, inv_mass_h, i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_down(s[k], d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4)
        c.partial[((size_t)blockIdx.y * c.blocks + blockIdx.x) * 4 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// B of step s - 1 and the momentum sums of step s's CMMotionRemover in ONE launch (skin mode with remove_cm_motion: the B has
// to be complete for every atom before the sums, and the sums before the first half that rides in the neighbour kernel — but
// B and the sums themselves are per atom / per molecule, so one grid-stride pass does both).  Same grid, same partial layout
// and same summation tree as k_com_partial: the sums have the bits the two launches gave.
__global__ void __launch_bounds__(256) k_baoab_second_com(MdArgs a) {
    GAMD_MD_GATE(1);                                        // a.step_index is the step whose second half this is
    __shared__ double red[4][4];
    const MdCom& c = a.com;
    const int n = a.n, npb = a.bx.n_boxes > 1 ? a.bx.n_per_box : n, a0 = blockIdx.y * npb, a1 = a0 + npb;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (a.use_rigid) {
        // a thread of this kernel owns whole molecules (q = t, t + T, ...): k_com_partial's by_molecule assignment
        for (int q = a0 / 3 + blockIdx.x * blockDim.x + threadIdx.x; 3 * q < a1; q += gridDim.x * blockDim.x) {
            d_baoab_second_mol(a, q);
            for (int k = 0; k < 3; ++k) com_add_atom(s, a.v, a.species, a.inv_mass, a.inv_mass_h, 3 * q + k);
        }
    } else {
        for (int i = a0 + blockIdx.x * blockDim.x + threadIdx.x; i < a1; i += gridDim.x * blockDim.x) {
#pragma unroll
            for (int d = 0; d < 3; ++d) d_baoab_second_dof(a, 3 * i + d);
            com_add_atom(s, a.v, a.species, a.inv_mass, a.inv_mass_h, i);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_down(s[k], d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4)
        c.partial[((size_t)blockIdx.y * c.blocks + blockIdx.x) * 4 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- Nose-Hoover chain -------------------------------------------------------------------------
// sum of m v^2 (kJ/mol: v converted to nm/ps) per block, optionally after the half kick of the second half
// (and, for rigid water, the velocity constraint that follows it: hack_integrator.py:427-428)
// Several boxes (a.bx.n_boxes > 1): blockIdx.y is the box; every box has its own chain, kinetic energy and velocity scale.
__device__ __forceinline__ int nhc_npb(const NhcArgs& a) { return a.bx.n_boxes > 1 ? a.bx.n_per_box : a.n; }
__device__ __forceinline__ int nhc_state_stride(const NhcArgs& a) { return 3 * a.M + 2; }

template <bool KICK>
__global__ void __launch_bounds__(256) k_nhc_ke2(NhcArgs a) {
    GAMD_MD_GATE(KICK ? 1 : 0);
    __shared__ double red[4];
    double s = 0.0;
    const double inv_len = 1.0 / (double)a.len;
    const int npb = nhc_npb(a), a0 = blockIdx.y * npb, a1 = a0 + npb;        // this box's atoms [a0, a1)
    // (first half: propagateNHC takes KE2 from the velocities as they are, hack_integrator.py:271; the CMMotionRemover of
    // addUpdateContextState() :272 acts AFTER the chain has scaled them -> k_nhc_apply_first)
    if (a.use_rigid) {
        for (int m = a0 / 3 + blockIdx.x * blockDim.x + threadIdx.x; 3 * m < a1; m += gridDim.x * blockDim.x) {
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
        for (int i = 3 * a0 + blockIdx.x * blockDim.x + threadIdx.x; i < 3 * a1; i += gridDim.x * blockDim.x) {
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
    if (threadIdx.x == 0) a.partial[blockIdx.y * a.n_blocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// propagateNHC (hack_integrator.py:289-316), in double like OpenMM's global variables
__global__ void k_nhc_chain(NhcArgs a) {
    const int box = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per box (one box: thread 0)
    if (box >= (a.bx.n_boxes > 1 ? a.bx.n_boxes : 1)) return;
    if (a.devflags[DEVFLAG_FROZEN]) return;                 // k_nhc_ke2 in front of it has recorded the position
    double KE2 = 0.0;
    for (int b = 0; b < a.n_blocks; ++b) KE2 += a.partial[box * a.n_blocks + b];
    const int M = a.M;
    double* st = a.state + (size_t)box * nhc_state_stride(a);
    double* xi = st;
    double* vxi = st + M;
    double* G = st + 2 * M;
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
    st[3 * M] = scale;
    st[3 * M + 1] = KE2;
}

// first half tail: v = scale*v (end of propagateNHC, :316); updateContextState (:272) = the CMMotionRemover on the scaled
// velocities, sum m (scale v) / sum m = scale * vcom with vcom from k_com_partial's sums over the unscaled ones;
// v += dt/2 f_last/m; x += dt v   (hack_integrator.py:273-280)
__global__ void k_nhc_apply_first(NhcArgs a) {
    GAMD_MD_GATE(0);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const BoxAtom ba = md_box_atom(a.bx, i);
    float box[3];
    md_box(a.bx, a.box, ba.box, box);
    const float scale = (float)a.state[(size_t)ba.box * nhc_state_stride(a) + 3 * a.M];
    const float kick = 0.5f * a.dt * a.len * atom_inv_mass(a.species, 1.0f / a.mass, a.mass_h > 0.f ? 1.0f / a.mass_h : 0.f, i);
    float com[3] = {0.f, 0.f, 0.f};
    if (a.com.enabled) md_com_velocity(a.com, ba.box, com);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = a.v[3 * i + d] * scale;
        v -= scale * com[d];
        v += kick * a.f[3 * i + d];
        a.v[3 * i + d] = v;
        a.x[3 * i + d] = gamd_remainder(a.x[3 * i + d] + a.dt * v, box[d]);
    }
}

// the same with constraints: x1 = x + dt v; ConstrainPositions; v += (x - x1)/dt   (:277-280)
__global__ void k_nhc_apply_first_rigid(NhcArgs a) {
    GAMD_MD_GATE(0);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (3 * m >= a.n) return;
    Vec3 x[3], v[3], f[3], x1[3], xc[3];
    load_mol(a.x, m, x); load_mol(a.v, m, v); load_mol(a.f, m, f);
    const BoxAtom ba = md_box_atom(a.bx, 3 * m);
    float box[3];
    md_box(a.bx, a.box, ba.box, box);
    const float scale = (float)a.state[(size_t)ba.box * nhc_state_stride(a) + 3 * a.M];
    const float w[3] = {1.0f / a.rigid.m_o, 1.0f / a.rigid.m_h, 1.0f / a.rigid.m_h};
    float com[3] = {0.f, 0.f, 0.f};
    if (a.com.enabled) md_com_velocity(a.com, ba.box, com);
    const Vec3 vcom{scale * com[0], scale * com[1], scale * com[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = ((scale * v[k]) - vcom) + ((0.5f * a.dt * a.len * w[k]) * f[k]);
        x1[k] = x[k] + (a.dt * v[k]);
        xc[k] = x1[k];
    }
    settle_positions(x, xc, a.rigid);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = v[k] + ((1.0f / a.dt) * (xc[k] - x1[k]));
    wrap_mol(xc, box);
    store_mol(a.x, m, xc); store_mol(a.v, m, v);
}

__global__ void k_nhc_apply_second(NhcArgs a) {
    GAMD_MD_GATE(1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * a.n) return;
    a.v[i] *= (float)a.state[(size_t)md_box_atom(a.bx, i / 3).box * nhc_state_stride(a) + 3 * a.M];
}

}  // namespace

int launch_com_partial(const MdCom& com, const float* v, const uint8_t* species, float inv_mass, float inv_mass_h, int n,
                       const BoxRef& bx, const int* devflags, int by_molecule, hipStream_t st) {
    const int nb = bx.n_boxes > 1 ? bx.n_boxes : 1;
    hipLaunchKernelGGL(k_com_partial, dim3(com.blocks, nb), dim3(256), 0, st, com, v, species, inv_mass, inv_mass_h, n, bx, devflags,
                       by_molecule);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_nhc_first(const NhcArgs& a, hipStream_t st) {
    const int nb = a.bx.n_boxes > 1 ? a.bx.n_boxes : 1;
    if (a.com.enabled) {
        int r = launch_com_partial(a.com, a.v, a.species, 1.0f / a.mass, a.mass_h > 0.f ? 1.0f / a.mass_h : 0.f, a.n, a.bx,
                                   a.devflags, a.use_rigid, st);
        if (r) return r;
    }
    hipLaunchKernelGGL(k_nhc_ke2<false>, dim3(a.n_blocks, nb), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3((nb + 63) / 64), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    if (a.use_rigid) hipLaunchKernelGGL(k_nhc_apply_first_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_nhc_apply_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_nhc_second(const NhcArgs& a, hipStream_t st) {
    const int nb = a.bx.n_boxes > 1 ? a.bx.n_boxes : 1;
    hipLaunchKernelGGL(k_nhc_ke2<true>, dim3(a.n_blocks, nb), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_chain, dim3((nb + 63) / 64), dim3(64), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_nhc_apply_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_first(const MdArgs& a, hipStream_t st) {
    if (a.com.enabled) {
        int r = launch_com_partial(a.com, a.v, a.species, a.inv_mass, a.inv_mass_h, a.n, a.bx, a.devflags, a.use_rigid, st);
        if (r) return r;
    }
    if (a.use_rigid) hipLaunchKernelGGL(k_baoab_first_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_baoab_first, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_second_com(const MdArgs& a, hipStream_t st) {
    const int nb = a.bx.n_boxes > 1 ? a.bx.n_boxes : 1;
    hipLaunchKernelGGL(k_baoab_second_com, dim3(a.com.blocks, nb), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_baoab_second(const MdArgs& a, hipStream_t st) {
    if (a.use_rigid) hipLaunchKernelGGL(k_baoab_second_rigid, dim3((a.n / 3 + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_baoab_second, dim3((3 * a.n + 255) / 256), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
