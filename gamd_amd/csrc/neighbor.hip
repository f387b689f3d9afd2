// neighbor.hip — periodic cell-list radius search -> destination-sorted CSR, on device, no host sync.
//
// Replaces (reference, code/):
//   graph_utils.py:21-26,29-44   jax-md partition.neighbor_list (cell list, padded dense idx)
//   graph_utils.py:51-61         exact-cutoff mask  dr^2 < rc^2  (strict), self pair kept
//   LJ/train_network_lj.py:166-185  dense -> COO compaction
//   md_module.py:63-78,93-126    O(N^2) search of the dynamic-box model (norm <= rc, self excluded)
//
// Pipeline (all stream-ordered, sizes read from device memory by later kernels):
//   bin -> scan(cells) -> fill cells -> sort each cell by atom id (determinism) -> gather sorted
//   positions -> count neighbours (27-cell sweep) -> scan(deg) -> fill CSR -> chunk metadata.
// Atoms are renumbered in cell order ("sorted order"); every later kernel works in that order, which
// makes the h[src] gathers of a destination tile land in a compact slice of the node tables.
#include "gamd_common.h"
#include "gamd_internal.h"
#include "gamd_md_dev.h"

namespace {

__device__ __forceinline__ int cell_coord(float p, float box, int nc) {
    int c = (int)floorf(p * ((float)nc / box));
    return c < 0 ? 0 : (c >= nc ? nc - 1 : c);     // remainder() may round up to exactly `box`
}

#define GAMD_GATE() do { if (a.gate && *a.gate == 0) return; } while (0)

// node feature that rides along in pos_s.w: the caller's float feature (nn_module.py:554 feeds it to node_encoder), or
// the species flag (O=1, H=0: water/test_script/test_nosehoover.py:82-89)
__device__ __forceinline__ float node_feature(const NbrArgs& a, int v) {
    return a.feat ? a.feat[v] : (a.species ? (float)a.species[v] : 0.f);
}

// box of atom i (caller's order or sorted order: box b owns [b n_per_box, (b + 1) n_per_box) in both) and its dimensions
__device__ __forceinline__ int box_id(const NbrArgs& a, int i) { return a.bx.n_boxes > 1 ? gamd_box_of(a.bx, i) : 0; }
// (The single-box values pass through an empty asm: without it hipcc folds the two branches into loads through
// select(&a.box, &boxes[3 b]) — the address of a member of the argument block escapes, scalar replacement gives up, and a kernel
// that works on a modified copy of the block (k_step_small's candidate pass) keeps all 376 bytes of it in scratch memory and reads
// every field back from there inside its sweep loops: a 258-atom rebuild took 0.5 ms instead of 0.1.)
__device__ __forceinline__ BoxDims box_dims(const NbrArgs& a, int b) {
    if (a.bx.n_boxes <= 1) {
        float bx = a.box[0], by = a.box[1], bz = a.box[2], hx = a.half[0], hy = a.half[1], hz = a.half[2];
        asm volatile("" : "+r"(bx), "+r"(by), "+r"(bz), "+r"(hx), "+r"(hy), "+r"(hz));
        return BoxDims{bx, by, bz, hx, hy, hz};
    }
    const float4 B = a.bx.boxes[3 * b], H = a.bx.boxes[3 * b + 1];
    return BoxDims{B.x, B.y, B.z, H.x, H.y, H.z};
}
// cell grid of box b: cells along x, y, z and the index of its first cell
struct BoxCells { int nx, ny, nz, base; };
__device__ __forceinline__ BoxCells box_cells(const NbrArgs& a, int b) {
    if (a.bx.n_boxes <= 1) {
        int nx = a.nc[0], ny = a.nc[1], nz = a.nc[2];
        asm volatile("" : "+r"(nx), "+r"(ny), "+r"(nz));
        return BoxCells{nx, ny, nz, 0};
    }
    const int4 c = reinterpret_cast<const int4*>(a.bx.boxes)[3 * b + 2];
    return BoxCells{c.x, c.y, c.z, c.w};
}

__device__ __forceinline__ void d_bin(const NbrArgs& a, int i) {
    const int bi = box_id(a, i);
    const BoxDims B = box_dims(a, bi);
    float4 p;
    p.x = gamd_remainder(a.pos[3 * i + 0], B.bx);    // graph_utils.py:31 jnp.mod(pos, box)
    p.y = gamd_remainder(a.pos[3 * i + 1], B.by);
    p.z = gamd_remainder(a.pos[3 * i + 2], B.bz);
    p.w = 0.f;
    a.pos_w[i] = p;
    if (a.ref_pos) a.ref_pos[i] = p;
    // cells are numbered box-major: atoms of different boxes never share a cell, so they are never neighbours
    const BoxCells G = box_cells(a, bi);
    const int c = G.base + (cell_coord(p.x, B.bx, G.nx) * G.ny + cell_coord(p.y, B.by, G.ny)) * G.nz + cell_coord(p.z, B.bz, G.nz);
    (void)GAMD_CHK_RANGE(a.sticky, c, 0, a.ncell - 1, GAMD_CHK_CELL);
    a.cell_of[i] = c;
    atomicAdd(&a.cell_cnt[c], 1);
}

__global__ void k_bin(NbrArgs a) {
    GAMD_GATE();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) d_bin(a, i);
}

// single-block exclusive scan, any length; out has n+1 entries (out[n] = total).  4 consecutive items per thread per
// pass (4 096 per pass), wave shuffles + one LDS hop.
template <typename F>
__device__ __forceinline__ void block_exclusive_scan(int n, F load, int* __restrict__ out) {
    constexpr int IPT = 4;
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * IPT) {
        const int i0 = base + tid * IPT;
        int v[IPT];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) { v[k] = (i0 + k < n) ? load(i0 + k) : 0; mine += v[k]; }
        int x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += wave_tot[w];
        const int carry = carry_s;
        int run = carry + woff + x - mine;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry_s;
}

__global__ void __launch_bounds__(1024) k_scan_cells(NbrArgs a) {
    GAMD_GATE();
    block_exclusive_scan(a.ncell, [&](int i) { return a.cell_cnt[i]; }, a.cell_start);
}

__device__ __forceinline__ void d_fill_cells(const NbrArgs& a, int i) {
    const int c = GAMD_CHK_RANGE(a.sticky, a.cell_of[i], 0, a.ncell - 1, GAMD_CHK_CELL);
    const int s = atomicAdd(&a.cell_fill[c], 1);
    a.perm[GAMD_CHK_RANGE(a.sticky, a.cell_start[c] + s, 0, a.n - 1, GAMD_CHK_PERM)] = i;
}

__global__ void k_fill_cells(NbrArgs a) {
    GAMD_GATE();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) d_fill_cells(a, i);
}

// atomics above give an arbitrary order inside a cell; sort by original atom id so that the CSR
// (and with it every floating-point summation order downstream) is bit-reproducible run to run.
// One wave per cell: rank by counting (cells hold ~14 atoms), then gather the sorted positions.
__device__ __forceinline__ void d_sort_gather(const NbrArgs& a, int c, int lane) {
    const int s = a.cell_start[c], e = a.cell_start[c + 1], cnt = e - s;
    if (cnt <= 64) {
        const int v = lane < cnt ? a.perm[s + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += (__shfl(v, j, 64) < v) ? 1 : 0;
        if (lane < cnt) {
            a.perm[s + rank] = v;                 // ids are distinct -> ranks are a permutation
            float4 p = a.pos_w[v];
            p.w = node_feature(a, v);
            a.pos_s[s + rank] = p;
            a.inv_perm[v] = s + rank;
        }
    } else {
        if (lane == 0) {                          // over-full cell (cutoff >> spacing): serial fallback
            for (int i = s + 1; i < e; ++i) {
                const int v = a.perm[i];
                int j = i - 1;
                while (j >= s && a.perm[j] > v) { a.perm[j + 1] = a.perm[j]; --j; }
                a.perm[j + 1] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int i = s + lane; i < e; i += 64) {
            const int v = ((volatile int*)a.perm)[i];
            float4 p = a.pos_w[v];
            p.w = node_feature(a, v);
            a.pos_s[i] = p;
            a.inv_perm[v] = i;
        }
    }
}

__global__ void __launch_bounds__(256) k_sort_gather(NbrArgs a) {
    GAMD_GATE();
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c < a.ncell) d_sort_gather(a, c, threadIdx.x & 63);
}

// Squared minimum-image distance of the neighbour test, op for op what the reference's mask evaluates in fp32:
//   jax-md flavour  dR = displacement(R_centre, R_neigh) = periodic(R_centre - R_neigh)  (graph_utils.py:53-56 map_neighbor
//                   over space.periodic's displacement: mod(dR + side/2, side) - side/2), then sum(dR ** 2) (:59);
//   torch flavour   dist_mat[a, b] = pos[b] - pos[a] with centre = b (md_module.py:65-66, :121), i.e. centre - neighbour too.
// Centre MINUS neighbour (the rounding of mod(d + L/2, L) - L/2 is not symmetric in d), every product rounded before the
// adds, (x^2 + y^2) + z^2 in that order (what torch's / XLA's reduction over the last axis of three gives): a pair that sits
// on the cutoff to fp32 rounding falls on the same side here as in the oracle's restatement, so the edge SETS are equal,
// not just equal up to near-cutoff pairs.
__device__ __forceinline__ float gamd_mask_d2(const float4& pc, const float4& pb, const BoxDims& B) {
#pragma clang fp contract(off)
    const float rx = gamd_min_image_wrapped(pc.x - pb.x, B.bx, B.hx);
    const float ry = gamd_min_image_wrapped(pc.y - pb.y, B.by, B.hy);
    const float rz = gamd_min_image_wrapped(pc.z - pb.z, B.bz, B.hz);
    const float xx = rx * rx, yy = ry * ry, zz = rz * rz;
    return (xx + yy) + zz;
}

// 27-cell sweep shared by the count and the fill pass: one half-wave (32 lanes) per centre atom, lanes
// test the atoms of a cell in parallel.  visit(ok, b) is called by every lane for every pass; the
// accepted neighbours of a pass are compacted in lane order with a ballot, so the CSR order is fixed:
// cells in (dx,dy,dz) order, atoms by ascending original id inside a cell.
// position tables of the sweeps: the argument block's global table, or k_step_small's LDS copy through a pointer typed with the
// LDS address space (ds_read_b128 instead of a flat load; float4 is a class type without address-space-qualified copies, so the
// LDS side is read as a plain 4-vector)
typedef __attribute__((address_space(3))) const f32x4* LdsPos;
typedef __attribute__((address_space(3))) const int* LdsCells;
__device__ __forceinline__ float4 pos_at(const float4* p, int i) { return p[i]; }
__device__ __forceinline__ float4 pos_at(LdsPos p, int i) { const f32x4 v = p[i]; return make_float4(v[0], v[1], v[2], v[3]); }

// visit(valid, d2, b): lane's candidate b of this pass (valid: the pass has an atom for this lane) and its squared distance
template <typename V, typename PosPtr, typename CellPtr>
__device__ __forceinline__ void sweep_d2(const NbrArgs& a, int ctr, int l, V visit, PosPtr pos_s, CellPtr cell_start) {
    // pos_s / cell_start: the argument block's tables, or the LDS copies k_step_small sweeps — as pointers of their own, typed
    // with the LDS address space there (ds_read instead of flat loads: three dependent loads per visited cell)
    const float4 pc = pos_at(pos_s, ctr);
    const int bi = box_id(a, ctr);
    const BoxDims B = box_dims(a, bi);
    const BoxCells G = box_cells(a, bi);
    const int cx = cell_coord(pc.x, B.bx, G.nx);
    const int cy = cell_coord(pc.y, B.by, G.ny);
    const int cz = cell_coord(pc.z, B.bz, G.nz);
    // axes with fewer than 3 cells: visit every cell of that axis exactly once
    const int lx = G.nx >= 3 ? -1 : -cx, hx = G.nx >= 3 ? 1 : G.nx - 1 - cx;
    const int ly = G.ny >= 3 ? -1 : -cy, hy = G.ny >= 3 ? 1 : G.ny - 1 - cy;
    const int lz = G.nz >= 3 ? -1 : -cz, hz = G.nz >= 3 ? 1 : G.nz - 1 - cz;
    for (int dx = lx; dx <= hx; ++dx) {
        int x = cx + dx; x += x < 0 ? G.nx : 0; x -= x >= G.nx ? G.nx : 0;
        for (int dy = ly; dy <= hy; ++dy) {
            int y = cy + dy; y += y < 0 ? G.ny : 0; y -= y >= G.ny ? G.ny : 0;
            for (int dz = lz; dz <= hz; ++dz) {
                int z = cz + dz; z += z < 0 ? G.nz : 0; z -= z >= G.nz ? G.nz : 0;
                const int c = G.base + (x * G.ny + y) * G.nz + z;
                const int s = cell_start[c], e = cell_start[c + 1];
                for (int b0 = s; b0 < e; b0 += 32) {
                    const int b = b0 + l;
                    float d2 = 0.f;
                    if (b < e) { const float4 pb = pos_at(pos_s, b); d2 = gamd_mask_d2(pc, pb, B); }
                    visit(b < e, d2, b);
                }
            }
        }
    }
}
template <typename V>
__device__ __forceinline__ void sweep_d2(const NbrArgs& a, int ctr, int l, V visit) { sweep_d2(a, ctr, l, visit, a.pos_s, a.cell_start); }
// the cutoff test of the two flavours on a squared distance
__device__ __forceinline__ bool in_range(int flavour, float d2, float rc, float rc2, bool is_self) {
    return flavour == 0 ? d2 < rc2                                   // graph_utils.py:59 (strict, self pair kept)
                        : (sqrtf(d2) <= rc) && !is_self;             // md_module.py:111
}
template <typename V, typename PosPtr, typename CellPtr>
__device__ __forceinline__ void sweep(const NbrArgs& a, int ctr, int l, V visit, PosPtr pos_s, CellPtr cell_start) {
    sweep_d2(a, ctr, l, [&](bool valid, float d2, int b) { visit(valid && in_range(a.flavour, d2, a.rc, a.rc2, b == ctr), b); },
             pos_s, cell_start);
}
template <typename V>
__device__ __forceinline__ void sweep(const NbrArgs& a, int ctr, int l, V visit) { sweep(a, ctr, l, visit, a.pos_s, a.cell_start); }

// my half-wave's 32-bit slice of a 64-lane ballot
__device__ __forceinline__ unsigned half_ballot(bool p) {
    const unsigned long long m = __ballot(p);
    return (unsigned)(m >> (threadIdx.x & 32));
}

// one half-wave per centre atom `ctr` (both halves of a wave must call this together, live or not)
template <typename PosPtr, typename CellPtr>
__device__ __forceinline__ void d_count(const NbrArgs& a, int ctr, int l, PosPtr pos_s, CellPtr cell_start) {
    const bool live = ctr < a.n;
    int cnt = 0;
    // both halves of a wave must run the same number of ballots: sweep a clamped atom, discard below
    sweep(a, live ? ctr : a.n - 1, l, [&](bool ok, int) { cnt += __popc(half_ballot(ok)); }, pos_s, cell_start);
    if (live && l == 0) a.deg[ctr] = cnt + (a.self_loop ? 1 : 0);
}
__device__ __forceinline__ void d_count(const NbrArgs& a, int ctr, int l) { d_count(a, ctr, l, a.pos_s, a.cell_start); }

__global__ void __launch_bounds__(256) k_count(NbrArgs a) {
    GAMD_GATE();
    d_count(a, (blockIdx.x * blockDim.x + threadIdx.x) >> 5, threadIdx.x & 31);
}

// Several boxes (gamd_config.n_boxes > 1): on entry row_ptr = exscan(deg) over all atoms.  Every box's first row is moved to
// the next 16-edge chunk boundary by giving the LAST atom of the box in front of it pad_b = (-E_b) mod 16 extra slots (deg is
// updated; the fill pass points them at the all-zero row).  The partial-sum pieces of an atom are cut at chunk boundaries, so
// with aligned starts box b's rows are cut exactly where a single-box evaluation cuts them: the forces of a batch are
// bit-identical to the boxes evaluated one by one.  Whole 1024-thread workgroup; at most 15 slots per box.
__device__ __forceinline__ void d_box_align(const NbrArgs& a) {
    const int nb = a.bx.n_boxes, npb = a.bx.n_per_box;
    block_exclusive_scan(nb, [&](int b) {
        const int T = a.row_ptr[(b + 1) * npb] - a.row_ptr[b * npb];
        return b + 1 < nb ? ((-T) & (GAMD_CHUNK - 1)) : 0;
    }, a.box_shift);
    __syncthreads();
    for (int i = threadIdx.x; i < a.n; i += 1024) {
        const int b = gamd_box_of(a.bx, i);
        const int sh = a.box_shift[b];
        if (sh) a.row_ptr[i] += sh;
        if (i == (b + 1) * npb - 1) a.deg[i] += a.box_shift[b + 1] - sh;
    }
    if (threadIdx.x == 0) a.row_ptr[a.n] += a.box_shift[nb];
    __syncthreads();
}

// row_ptr = exscan(deg); NA = inclusive count of non-empty segments that start off a chunk boundary;
// publishes E, the piece count and the overflow flag.
// The same two scans for n <= 16 384 in ONE pass over registers: 16 consecutive rows per thread (four 16-byte loads), row_ptr
// and the off-boundary flags derived from it never leave the thread between the two scans — the multi-pass form above
// re-reads row_ptr through L2 and needs 6 passes with 4 workgroup barriers each at n = 10 000 (17 us -> see DESIGN).
__device__ __forceinline__ void d_scan_deg_fast(const NbrArgs& a) {
    constexpr int IPT = 16;
    __shared__ int s_tot[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = tid * IPT;
    int v[IPT];
    if (i0 + IPT <= a.n) {
#pragma unroll
        for (int q = 0; q < IPT / 4; ++q) {
            const int4 t = *reinterpret_cast<const int4*>(a.deg + i0 + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k) v[k] = (i0 + k < a.n) ? a.deg[i0 + k] : 0;
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) mine += v[k];
    int x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_tot[0][wv] = x;
    __syncthreads();
    int run = x - mine, E_all = 0;
    for (int w = 0; w < 16; ++w) { const int t = s_tot[0][w]; if (w < wv) run += t; E_all += t; }
    int f[IPT], rp[IPT], mine2 = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { rp[k] = run; run += v[k]; }
    if (a.bx.n_boxes > 1 && a.bx.n_boxes <= 1024) {
        // box alignment (see d_box_align) without leaving the workgroup: box starts -> LDS, thread b owns box b's padding,
        // one more block scan over the boxes, the shifts applied to the rows still in registers
        __shared__ int s_bstart[1025], s_bshift[1025];
        __shared__ int s_ptot[16];
        const int nb = a.bx.n_boxes, npb = a.bx.n_per_box;
        int bk = i0 < a.n ? gamd_box_of(a.bx, i0) : nb;          // box of row i0; rows are consecutive, boxes follow
        int nextb = (bk + 1) * npb;
        {
            int b = bk, nx = nextb;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const int i = i0 + k;
                if (i < a.n) { if (i == nx) { ++b; nx += npb; } if (i == b * npb) s_bstart[b] = rp[k]; }
            }
        }
        if (tid == 0) s_bstart[nb] = E_all;
        __syncthreads();
        const int pad = (tid + 1 < nb) ? ((-(s_bstart[tid + 1] - s_bstart[tid])) & (GAMD_CHUNK - 1)) : 0;
        int px = pad;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(px, d, 64); if (lane >= d) px += y; }
        if (lane == 63) s_ptot[wv] = px;
        __syncthreads();
        int pbase = px - pad, ptotal = 0;
        for (int w = 0; w < 16; ++w) { const int t = s_ptot[w]; if (w < wv) pbase += t; ptotal += t; }
        if (tid < nb) s_bshift[tid] = pbase;
        if (tid == 0) s_bshift[nb] = ptotal;
        __syncthreads();
        {
            int b = bk, nx = nextb;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const int i = i0 + k;
                if (i < a.n) {
                    if (i == nx) { ++b; nx += npb; }
                    rp[k] += s_bshift[b];
                    if (i == nx - 1) { v[k] += s_bshift[b + 1] - s_bshift[b]; a.deg[i] = v[k]; }
                }
            }
        }
        E_all += ptotal;
    } else if (a.bx.n_boxes > 1) {
        // (more than 1 024 boxes) publish the plain offsets, align the boxes' first rows in global memory, take the result back
#pragma unroll
        for (int k = 0; k < IPT; ++k) if (i0 + k < a.n) a.row_ptr[i0 + k] = rp[k];
        if (tid == 0) a.row_ptr[a.n] = E_all;
        __syncthreads();
        d_box_align(a);
#pragma unroll
        for (int k = 0; k < IPT; ++k) if (i0 + k < a.n) { rp[k] = a.row_ptr[i0 + k]; v[k] = a.deg[i0 + k]; }
        E_all = a.row_ptr[a.n];
    }
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        f[k] = (i0 + k < a.n && v[k] > 0 && (rp[k] % GAMD_CHUNK) != 0) ? 1 : 0;
        mine2 += f[k];
    }
    if (i0 + IPT <= a.n) {
#pragma unroll
        for (int q = 0; q < IPT / 4; ++q)
            *reinterpret_cast<int4*>(a.row_ptr + i0 + 4 * q) = make_int4(rp[4 * q], rp[4 * q + 1], rp[4 * q + 2], rp[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k) if (i0 + k < a.n) a.row_ptr[i0 + k] = rp[k];
    }
    if (tid == 0) a.row_ptr[a.n] = E_all;
    int y2 = mine2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(y2, d, 64); if (lane >= d) y2 += y; }
    if (lane == 63) s_tot[1][wv] = y2;
    __syncthreads();
    int run2 = y2 - mine2, NA_all = 0;
    for (int w = 0; w < 16; ++w) { const int t = s_tot[1][w]; if (w < wv) run2 += t; NA_all += t; }
    int na[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { na[k] = run2; run2 += f[k]; }
    if (i0 + IPT <= a.n) {
#pragma unroll
        for (int q = 0; q < IPT / 4; ++q)
            *reinterpret_cast<int4*>(a.na_excl + i0 + 4 * q) = make_int4(na[4 * q], na[4 * q + 1], na[4 * q + 2], na[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k) if (i0 + k < a.n) a.na_excl[i0 + k] = na[k];
    }
    if (tid == 0) {
        a.na_excl[a.n] = NA_all;
        a.counters[CNT_E] = E_all;
        a.counters[CNT_PIECES] = (E_all + GAMD_CHUNK - 1) / GAMD_CHUNK + NA_all;
        if ((long long)E_all > a.e_cap) {
            a.counters[CNT_OVERFLOW] = 1; a.sticky[STICKY_EDGE_OVERFLOW] = 1; a.devflags[DEVFLAG_FROZEN] = 1;
        }
        a.counters[CNT_TILES] = (E_all + GAMD_TILE - 1) / GAMD_TILE;
    }
}

__device__ __forceinline__ void d_scan_deg(const NbrArgs& a) {
    if (a.n <= 16 * 1024) { d_scan_deg_fast(a); return; }
    block_exclusive_scan(a.n, [&](int i) { return a.deg[i]; }, a.row_ptr);
    __syncthreads();
    if (a.bx.n_boxes > 1) d_box_align(a);
    block_exclusive_scan(a.n, [&](int i) { return (a.deg[i] > 0 && (a.row_ptr[i] % GAMD_CHUNK) != 0) ? 1 : 0; },
                         a.na_excl);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int E = a.row_ptr[a.n];
        a.counters[CNT_E] = E;
        a.counters[CNT_PIECES] = (E + GAMD_CHUNK - 1) / GAMD_CHUNK + a.na_excl[a.n];
        if ((long long)E > a.e_cap) {
            a.counters[CNT_OVERFLOW] = 1; a.sticky[STICKY_EDGE_OVERFLOW] = 1; a.devflags[DEVFLAG_FROZEN] = 1;
        }
        a.counters[CNT_TILES] = (E + GAMD_TILE - 1) / GAMD_TILE;
    }
}

// The fixed-stride candidate rows were rebuilt in this call (k_filter_count): publish their size.  A row longer than the stride is an
// overflow (the list is truncated: freeze, the host regrows to 1.25 x longest row x n and resumes).  Total and longest row are
// reduced HERE from the row lengths (one coalesced pass of this workgroup over cand_deg): as a pair of global atomics per
// workgroup of the count pass they were 1 500 operations on two addresses at 6 000 atoms, which the L2 serialises — 13 of the
// 36 us of that kernel on a rebuild step.  Whole workgroup of NT threads.
template <int NT>
__device__ __forceinline__ void d_publish_candidates(const NbrArgs& a) {
    __shared__ int s_ctot[NT / 64], s_cmax[NT / 64];
    int tot = 0, longest = 0;
    for (int i = threadIdx.x; i < a.n; i += NT) {
        const int w = a.cand_deg[i];
        tot += w < a.cand_stride ? w : a.cand_stride;
        longest = w > longest ? w : longest;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        tot += __shfl_xor(tot, d, 64);
        const int o = __shfl_xor(longest, d, 64);
        longest = o > longest ? o : longest;
    }
    if ((threadIdx.x & 63) == 0) { s_ctot[threadIdx.x >> 6] = tot; s_cmax[threadIdx.x >> 6] = longest; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tot = 0; longest = 0;
        for (int w = 0; w < NT / 64; ++w) { tot += s_ctot[w]; longest = s_cmax[w] > longest ? s_cmax[w] : longest; }
        a.counters[CNT_NCAND] = tot;
        a.counters[CNT_CAND_MAX] = longest;
        a.sticky[STICKY_NCAND] = tot;
        a.sticky[STICKY_REBUILDS] = ++a.devflags[DEVFLAG_REBUILDS];
        if (longest > a.cand_stride) {
            const long long need = (long long)longest * a.n;
            a.sticky[STICKY_NCAND] = need > 0x7fffffffll ? 0x7fffffff : (int)need;
            a.sticky[STICKY_CAND_OVERFLOW] = 1;
            a.devflags[DEVFLAG_FROZEN] = 1;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) k_scan_deg(NbrArgs a) {
    GAMD_GATE();
    if (a.cand_stride > 0 && a.counters[CNT_REBUILD]) d_publish_candidates<1024>(a);
    d_scan_deg(a);
}

// Several boxes: the scan (d_box_align) has made every box's first CSR row start on a 16-edge chunk boundary by lengthening
// the row of the box's LAST atom; the half-wave that fills that row writes the extra slots behind its real edges: source =
// the all-zero row n of the node tables (message exactly 0), destination = the atom itself.
__device__ __forceinline__ void d_fill_padding(const NbrArgs& a, int c, int l, long long w_end) {
    if (a.bx.n_boxes <= 1) return;
    const int b = gamd_box_of(a.bx, c);
    if (c != (b + 1) * a.bx.n_per_box - 1 || b + 1 >= a.bx.n_boxes) return;
    long long end = a.row_ptr[c + 1];
    if (end > a.e_cap) end = a.e_cap;
    for (long long x = w_end + l; x < end; x += 32) { a.col[x] = a.n; if (a.erow) a.erow[x] = c; }
}

template <typename PosPtr, typename CellPtr>
__device__ __forceinline__ void d_fill(const NbrArgs& a, int ctr, int l, PosPtr pos_s, CellPtr cell_start) {
    const bool live = ctr < a.n;
    const int c = live ? ctr : a.n - 1;
    long long w = a.row_ptr[c];
    sweep(a, c, l, [&](bool ok, int b) {
        const unsigned m = half_ballot(ok);
        if (ok && live) {
            const long long at = w + __popc(m & ((1u << l) - 1u));
            if (at < a.e_cap) { a.col[at] = b; if (a.erow) a.erow[at] = c; }
        }
        w += __popc(m);
    }, pos_s, cell_start);
    // self_loop_mode 1: the loop DGL's in-place add_self_loop would append (nn_module.py:650-652), last in the row
    if (a.self_loop && live && l == 0 && w < a.e_cap) { a.col[w] = c; if (a.erow) a.erow[w] = c; }
    if (live) d_fill_padding(a, c, l, w + (a.self_loop ? 1 : 0));
}
__device__ __forceinline__ void d_fill(const NbrArgs& a, int ctr, int l) { d_fill(a, ctr, l, a.pos_s, a.cell_start); }

__global__ void __launch_bounds__(256) k_fill(NbrArgs a) {
    GAMD_GATE();
    d_fill(a, (blockIdx.x * blockDim.x + threadIdx.x) >> 5, threadIdx.x & 31);
    // last kernel of a cell-list build: the cell counters are not read any more (the sweep uses cell_start) and are left
    // ZERO for the next build, so the skin path needs no memset node per step
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.ncell; k += gridDim.x * blockDim.x) { a.cell_cnt[k] = 0; a.cell_fill[k] = 0; }
}

// per 16-edge chunk: first piece id and the bit mask of edges that close a destination segment
__global__ void k_chunk_meta(NbrArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    long long E = a.counters[CNT_E];
    if (E > a.e_cap) E = a.e_cap;
    const long long x0 = (long long)c * GAMD_CHUNK;
    if (x0 >= E) {
        if (x0 < a.e_cap + GAMD_TILE) { a.chunk_piece[c] = 0; a.chunk_mask[c] = 0; }
        return;
    }
    const int first_atom = a.erow[x0];
    // pieces before this chunk: one per earlier chunk + one per off-boundary segment start <= x0
    const int na_incl = a.na_excl[first_atom] +
                        ((a.deg[first_atom] > 0 && (a.row_ptr[first_atom] % GAMD_CHUNK) != 0) ? 1 : 0);
    a.chunk_piece[c] = c + na_incl;
    unsigned mask = 0;
    int prev = first_atom;
    for (int r = 0; r < GAMD_CHUNK; ++r) {
        const long long x = x0 + r;
        if (x >= E) break;
        const int nxt = (x + 1 < E) ? a.erow[x + 1] : -1;
        if (nxt != prev) mask |= 1u << r;
        prev = nxt;
    }
    a.chunk_mask[c] = mask;
}

// ---- Verlet-skin reuse ----------------------------------------------------------------------------
// argument block of the candidate pass (rc + skin, candidate arrays, no self loops: the exact filter appends them)
__device__ __forceinline__ NbrArgs cand_args(const NbrArgs& a) {
    NbrArgs c = a;
    c.rc = a.rc_build; c.rc2 = a.rc2_build;
    c.deg = a.cand_deg; c.row_ptr = a.cand_ptr; c.col = a.cand_col; c.erow = nullptr; c.e_cap = a.cand_cap;
    c.self_loop = 0;
    c.gate = nullptr;
    return c;
}

// bin | scan | fill | sort + gather of a candidate rebuild by ONE 1024-thread workgroup (the phases of k_step_small's rebuild
// with loops over the atoms / cells); the cell counters are zero on entry (the previous rebuild's k_filter_count left them so)
//
// One workgroup means every phase is a loop of dependent memory round trips per thread (6 atoms per thread at 6 000 atoms, 32
// cells per wave at 512 cells: ~140 us per rebuild, ~100 of them in the per-cell sort, at a rebuild every 11 steps 4 % of the C5
// step).  The loops are therefore BATCHED: the loads (and the returning atomics) of up to 8 iterations are issued together and
// waited for once, so a phase costs two or three round trips per batch instead of per atom / per cell.  Same stores, same
// values: the sorted order, perm / inv_perm and the cell bounds are what the one-by-one form writes.
constexpr int CELL_BATCH = 8;

__device__ void d_bin_batch(const NbrArgs& a, int tid) {
    for (int base = tid; base < a.n; base += 1024 * CELL_BATCH) {
        float px[CELL_BATCH], py[CELL_BATCH], pz[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) { px[b] = a.pos[3 * i + 0]; py[b] = a.pos[3 * i + 1]; pz[b] = a.pos[3 * i + 2]; }
        }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) {
                const int bi = box_id(a, i);
                const BoxDims B = box_dims(a, bi);
                float4 p;
                p.x = gamd_remainder(px[b], B.bx);           // graph_utils.py:31 jnp.mod(pos, box)
                p.y = gamd_remainder(py[b], B.by);
                p.z = gamd_remainder(pz[b], B.bz);
                p.w = 0.f;
                a.pos_w[i] = p;
                if (a.ref_pos) a.ref_pos[i] = p;
                const BoxCells G = box_cells(a, bi);
                const int c = G.base + (cell_coord(p.x, B.bx, G.nx) * G.ny + cell_coord(p.y, B.by, G.ny)) * G.nz + cell_coord(p.z, B.bz, G.nz);
                a.cell_of[i] = c;
                atomicAdd(&a.cell_cnt[c], 1);
            }
        }
    }
}

__device__ void d_fill_cells_batch(const NbrArgs& a, int tid) {
    for (int base = tid; base < a.n; base += 1024 * CELL_BATCH) {
        int c[CELL_BATCH], s[CELL_BATCH], st[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) { const int i = base + 1024 * b; c[b] = i < a.n ? a.cell_of[i] : -1; }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b)
            if (c[b] >= 0) { s[b] = atomicAdd(&a.cell_fill[c[b]], 1); st[b] = a.cell_start[c[b]]; }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b)
            if (c[b] >= 0) a.perm[st[b] + s[b]] = base + 1024 * b;
    }
}

// d_sort_gather for the cells c_begin + wave, + 16, + 32, ... < c_end of this workgroup, CELL_BATCH cells at a time
__device__ void d_sort_gather_batch(const NbrArgs& a, int wave, int lane, int c_begin, int c_end) {
    for (int k0 = c_begin + wave; k0 < c_end; k0 += 16 * CELL_BATCH) {
        int s[CELL_BATCH], cnt[CELL_BATCH], v[CELL_BATCH], rank[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int k = k0 + 16 * b;
            s[b] = k < c_end ? a.cell_start[k] : 0;
            cnt[b] = k < c_end ? a.cell_start[k + 1] - s[b] : 0;
        }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) v[b] = (cnt[b] <= 64 && lane < cnt[b]) ? a.perm[s[b] + lane] : 0x7fffffff;
        // rank by counting, all cells of the batch in step (lanes beyond a cell's count hold INT_MAX: they never count)
        int maxc = 0;
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) { rank[b] = 0; if (cnt[b] <= 64 && cnt[b] > maxc) maxc = cnt[b]; }
        for (int j = 0; j < maxc; ++j) {
#pragma unroll
            for (int b = 0; b < CELL_BATCH; ++b) rank[b] += (__shfl(v[b], j, 64) < v[b]) ? 1 : 0;
        }
        float4 p[CELL_BATCH];
        float f[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b)
            if (cnt[b] <= 64 && lane < cnt[b]) { p[b] = a.pos_w[v[b]]; f[b] = node_feature(a, v[b]); }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b)
            if (cnt[b] <= 64 && lane < cnt[b]) {
                a.perm[s[b] + rank[b]] = v[b];            // ids are distinct -> ranks are a permutation
                a.pos_s[s[b] + rank[b]] = make_float4(p[b].x, p[b].y, p[b].z, f[b]);
                a.inv_perm[v[b]] = s[b] + rank[b];
            }
        unsigned over = 0;                                             // over-full cells (cutoff >> spacing): the serial form
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) over |= cnt[b] > 64 ? 1u << b : 0u;
        for (int b = 0; over >> b; ++b)
            if ((over >> b) & 1u) d_sort_gather(a, k0 + 16 * b, lane);
    }
}

__device__ void d_cells_one_wg(const NbrArgs& c) {
    const int tid = threadIdx.x;
    d_bin_batch(c, tid);                                                     // also stores ref_pos
    __syncthreads();
    block_exclusive_scan(c.ncell, [&](int i) { return c.cell_cnt[i]; }, c.cell_start);
    __syncthreads();
    d_fill_cells_batch(c, tid);
    __syncthreads();
    d_sort_gather_batch(c, tid >> 6, tid & 63, 0, c.ncell);
}

// Every call: wrap the positions and raise the rebuild flag if any atom has moved more than skin/2 since the
// candidate list was built (or the host forces it).  jax-md does the same test in update_neighbor_lst
// (graph_utils.py:36-44) with dr_threshold = cutoff/6.
// has the atom at p moved more than skin / 2 from r?  (NaN positions count as moved.)  Without fp contraction: the one-atom and the
// molecule form of the check then decide every borderline case alike, whatever kernel they are inlined into.
__device__ __forceinline__ bool d_moved_beyond(const float4& p, const float4& r, const BoxDims& B, float skin_half2) {
#pragma clang fp contract(off)
    const float dx = gamd_min_image_wrapped(p.x - r.x, B.bx, B.hx);
    const float dy = gamd_min_image_wrapped(p.y - r.y, B.by, B.hy);
    const float dz = gamd_min_image_wrapped(p.z - r.z, B.bz, B.hz);
    return !(((dx * dx + dy * dy) + dz * dz) <= skin_half2);
}

__device__ __forceinline__ bool d_skin_check(const NbrArgs& a, int i) {
    const BoxDims B = box_dims(a, box_id(a, i));
    float4 p;
    p.x = gamd_remainder(a.pos[3 * i + 0], B.bx);
    p.y = gamd_remainder(a.pos[3 * i + 1], B.by);
    p.z = gamd_remainder(a.pos[3 * i + 2], B.bz);
    p.w = 0.f;
    a.pos_w[i] = p;
    bool moved = a.force_rebuild != 0;
    if (!moved) {
        moved = d_moved_beyond(p, a.ref_pos[i], B, a.skin_half2);       // NaN positions force a rebuild too
    }
    // current position in the (so far frozen) sorted order; a rebuild later in this call overwrites pos_s and the order
    p.w = node_feature(a, i);
    a.pos_s[a.inv_perm[i]] = p;
    return moved;
}

// The same check for the (up to) three atoms i0 .. i0 + 2 of a rigid molecule with every load issued before the first store: the
// one-atom form called three times is three dependent chains (position -> stores -> next position ...) behind the integrator,
// because the compiler may not move a load across the stores of the previous atom.  Same operations per atom, same bits.
__device__ __forceinline__ bool d_skin_check_mol(const NbrArgs& a, int i0) {
    float px[3][3];
    float4 r[3];
    int ip[3];
    float w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = i0 + k < a.n ? i0 + k : a.n - 1;
        px[k][0] = a.pos[3 * i + 0]; px[k][1] = a.pos[3 * i + 1]; px[k][2] = a.pos[3 * i + 2];
        r[k] = a.ref_pos[i];
        ip[k] = a.inv_perm[i];
        w[k] = node_feature(a, i);
    }
    bool moved = a.force_rebuild != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = i0 + k;
        if (i < a.n) {
            const BoxDims B = box_dims(a, box_id(a, i));
            float4 p;
            p.x = gamd_remainder(px[k][0], B.bx);
            p.y = gamd_remainder(px[k][1], B.by);
            p.z = gamd_remainder(px[k][2], B.bz);
            p.w = 0.f;
            a.pos_w[i] = p;
            if (!a.force_rebuild) moved |= d_moved_beyond(p, r[k], B, a.skin_half2);
            p.w = w[k];
            a.pos_s[ip[k]] = p;
        }
    }
    return moved;
}

// do_second / do_first: the B of the previous MD step and the B A O A of this one for atom i first (plain BAOAB, MdFuse) —
// per-atom work in front of a per-atom check: two launches less per step
__global__ void k_skin_check(NbrArgs a, MdArgs md, int do_second, int do_first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.counters_next && i < CNT_COUNT) a.counters_next[i] = 0;      // ping-pong counter blocks: no memset node
    if (do_second | do_first) {
        if (md.devflags[DEVFLAG_FROZEN]) {                 // GAMD_MD_GATE of integrate.hip for the fused halves
            if (i == 0 && md.devflags[DEVFLAG_FROZEN_AT] < 0)
                md.devflags[DEVFLAG_FROZEN_AT] = do_second ? 2 * (md.step_index - 1) + 1 : 2 * md.step_index;
        } else if (md.use_rigid) {                         // thread i owns molecule i (atoms 3 i .. 3 i + 2 = O, H, H)
            if (3 * i < a.n) {
                if (do_second) gamd_md::d_baoab_second_mol(md, i);
                if (do_first) gamd_md::d_baoab_first_mol(md, i);
            }
        } else if (i < a.n) {
            if (do_second) {
#pragma unroll
                for (int d = 0; d < 3; ++d) d_baoab_second_dof(md, 3 * i + d);
            }
            if (do_first) d_baoab_first_atom(md, i);
        }
    }
    bool moved = false;
    if ((do_second | do_first) && md.use_rigid) {          // the thread that moved the molecule checks its three atoms
        if (3 * i < a.n) moved = d_skin_check_mol(a, 3 * i);
    } else if (i < a.n) {
        moved = d_skin_check(a, i);
    }
    if (moved) a.counters[CNT_REBUILD] = 1;
}

// The four cell-list phases of a candidate rebuild in ONE gated single-workgroup launch (up to 16 384 atoms; the four
// grid-wide kernels above that or when rebuilds are frequent): it runs once in 10-100 steps, what counts is that a reuse step
// pays for one launch that returns at once instead of four.
__global__ void __launch_bounds__(1024) k_cells_one_wg(NbrArgs a) {
    if (a.counters[CNT_REBUILD] == 0) return;
    d_cells_one_wg(cand_args(a));
}

// The same rebuild by CELLS_SL_WGS workgroups that never wait for each other.  One workgroup is bound by what ONE compute unit's
// vector-memory path can do with ~10 scattered passes over the atoms (74-80 us at 6 000 atoms, a rebuild every ~10 steps at C5:
// profiles/r05_experiments.md section 7); several workgroups normally need device-wide barriers between bin | scan | fill | sort.
// They do not if every workgroup repeats the cheap part: each one histograms ALL atoms into its own LDS copy of the cell
// counters (a coalesced stream of 12 B per atom, LDS atomics) and scans it, so each knows every cell's first slot; then it takes
// the cells whose first slot lies in its 1 / CELLS_SL_WGS share of the sorted order, streams the atoms a second time to collect
// the members of those cells (LDS fill counters, perm in its own slots of global memory) and sorts / gathers them as above.
// Everything a workgroup reads back from global memory it has written itself.  Outputs (pos_w, ref_pos, cell_of, cell_start,
// perm, pos_s, inv_perm) are the values the one-workgroup form writes: the order inside a cell is by atom id in both.
constexpr int CELLS_SL_MAXCELL = 4096;
constexpr int CELLS_SL_WGS = 32;

struct BinOut { float4 p; int c; };
__device__ __forceinline__ BinOut d_wrap_cell(const NbrArgs& a, int i, float px, float py, float pz) {
    const int bi = box_id(a, i);
    const BoxDims B = box_dims(a, bi);
    BinOut o;
    o.p.x = gamd_remainder(px, B.bx);                 // graph_utils.py:31 jnp.mod(pos, box)
    o.p.y = gamd_remainder(py, B.by);
    o.p.z = gamd_remainder(pz, B.bz);
    o.p.w = 0.f;
    const BoxCells G = box_cells(a, bi);
    o.c = G.base + (cell_coord(o.p.x, B.bx, G.nx) * G.ny + cell_coord(o.p.y, B.by, G.ny)) * G.nz + cell_coord(o.p.z, B.bz, G.nz);
    return o;
}

__global__ void __launch_bounds__(1024) k_cells_sliced(NbrArgs a0) {
    if (a0.counters[CNT_REBUILD] == 0) return;
    __shared__ int s_cnt[CELLS_SL_MAXCELL];           // histogram, later the fill counters of this workgroup's cells
    __shared__ int s_start[CELLS_SL_MAXCELL + 1];
    __shared__ int s_range[2];
    const NbrArgs a = cand_args(a0);
    const int tid = threadIdx.x, w = blockIdx.x, nwg = gridDim.x;
    for (int k = tid; k < a.ncell; k += 1024) s_cnt[k] = 0;
    __syncthreads();
    // pass 1: histogram of all atoms
    for (int base = tid; base < a.n; base += 1024 * CELL_BATCH) {
        float px[CELL_BATCH], py[CELL_BATCH], pz[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) { px[b] = a.pos[3 * i + 0]; py[b] = a.pos[3 * i + 1]; pz[b] = a.pos[3 * i + 2]; }
        }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) atomicAdd(&s_cnt[d_wrap_cell(a, i, px[b], py[b], pz[b]).c], 1);
        }
    }
    __syncthreads();
    block_exclusive_scan(a.ncell, [&](int i) { return s_cnt[i]; }, s_start);
    __syncthreads();
    // this workgroup's cells: first slot in [w, w + 1) * chunk of the sorted order (trailing empty cells go to the last one)
    if (tid < 2) {
        const int which = w + tid;
        int c;
        if (which == 0) c = 0;
        else if (which >= nwg) c = a.ncell;
        else {
            const int chunk = (a.n + nwg - 1) / nwg;
            const int target = which * chunk < a.n ? which * chunk : a.n;
            int lo = 0, hi = a.ncell;                 // first c in [0, ncell] with s_start[c] >= target (s_start[ncell] = n)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_start[mid] >= target) hi = mid; else lo = mid + 1; }
            c = lo;
        }
        s_range[tid] = c;
    }
    for (int k = tid; k < a.ncell; k += 1024) s_cnt[k] = 0;
    __syncthreads();
    const int c0 = s_range[0], c1 = s_range[1];
    // (cell_start[c1] is written by this workgroup AND by the owner of cell c1, with the same value: the sort below reads it)
    for (int k = c0 + tid; k <= c1; k += 1024) a.cell_start[k] = s_start[k];
    // pass 2: the members of my cells -> my slots of perm; their wrapped positions / cell ids
    for (int base = tid; base < a.n; base += 1024 * CELL_BATCH) {
        float px[CELL_BATCH], py[CELL_BATCH], pz[CELL_BATCH];
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) { px[b] = a.pos[3 * i + 0]; py[b] = a.pos[3 * i + 1]; pz[b] = a.pos[3 * i + 2]; }
        }
#pragma unroll
        for (int b = 0; b < CELL_BATCH; ++b) {
            const int i = base + 1024 * b;
            if (i < a.n) {
                const BinOut o = d_wrap_cell(a, i, px[b], py[b], pz[b]);
                if (o.c >= c0 && o.c < c1) {
                    a.pos_w[i] = o.p;
                    if (a.ref_pos) a.ref_pos[i] = o.p;
                    a.cell_of[i] = o.c;
                    a.perm[s_start[o.c] + atomicAdd(&s_cnt[o.c], 1)] = i;
                }
            }
        }
    }
    __syncthreads();
    d_sort_gather_batch(a, tid >> 6, tid & 63, c0, c1);
}

// ---- small systems (n <= 1024), Verlet-skin mode: everything in front of the exact filter in ONE workgroup ------------
// The reference's own drivers run 258 / 774 atoms, where a step is bound by the ~4-5 us every kernel costs in the stream,
// not by work.  One 1024-thread workgroup (one thread per atom) therefore does, in this order:
//   [B of the previous MD step] [B A O A of this step]     (plain BAOAB only: MdFuse; skipped when the run is frozen)
//   wrap + displacement check, positions into the sorted order, clear of the other counter block
//   the cell build of a candidate rebuild (bin | scan | fill | sort), behind a workgroup-wide OR of the check (once in 50-100
//   steps), with workgroup barriers where the large-system path has kernel boundaries.
// The candidate rows are then written by the count pass of the exact filter (k_filter_count, as above 1 024 atoms): 3 launches.
__global__ void __launch_bounds__(1024) k_step_small(NbrArgs a, MdArgs md, int do_second, int do_first) {
    const int tid = threadIdx.x;
    if (do_second | do_first) {
        if (md.devflags[DEVFLAG_FROZEN]) {                 // GAMD_MD_GATE of integrate.hip for the fused halves
            if (tid == 0 && md.devflags[DEVFLAG_FROZEN_AT] < 0)
                md.devflags[DEVFLAG_FROZEN_AT] = do_second ? 2 * (md.step_index - 1) + 1 : 2 * md.step_index;
        } else if (md.use_rigid) {                         // thread t owns molecule t (atoms 3 t .. 3 t + 2 = O, H, H)
            if (3 * tid < a.n) {
                if (do_second) gamd_md::d_baoab_second_mol(md, tid);
                if (do_first) gamd_md::d_baoab_first_mol(md, tid);
            }
        } else if (tid < a.n) {
            if (do_second) {
#pragma unroll
                for (int d = 0; d < 3; ++d) d_baoab_second_dof(md, 3 * tid + d);
            }
            if (do_first) d_baoab_first_atom(md, tid);
        }
    }
    bool moved = false;
    if ((do_second | do_first) && md.use_rigid) {          // the thread that moved the molecule checks its three atoms
        if (3 * tid < a.n) moved = d_skin_check_mol(a, 3 * tid);
    } else if (tid < a.n) {
        moved = d_skin_check(a, tid);
    }
    if (tid < CNT_COUNT) a.counters_next[tid] = 0;
    if (!__syncthreads_or(moved ? 1 : 0)) return;
    if (tid == 0) a.counters[CNT_REBUILD] = 1;
    // candidate pass: rc + skin, candidate arrays, no self loops (the exact filter appends them)
    NbrArgs c = cand_args(a);
    for (int k = tid; k < c.ncell; k += 1024) { c.cell_cnt[k] = 0; c.cell_fill[k] = 0; }
    __syncthreads();
    if (tid < c.n) d_bin(c, tid);                           // also stores ref_pos
    __syncthreads();
    block_exclusive_scan(c.ncell, [&](int i) { return c.cell_cnt[i]; }, c.cell_start);
    __syncthreads();
    if (tid < c.n) d_fill_cells(c, tid);
    __syncthreads();
    for (int k = tid >> 6; k < c.ncell; k += 16) d_sort_gather(c, k, tid & 63);
    __syncthreads();
    // (the candidate rows themselves are written by the count pass behind this kernel, k_filter_count, spread over the chip: as two
    //  sweeps of this ONE workgroup they were 0.26 of the 0.27 ms a rebuild step cost at 258 atoms)
}

// exact cutoff on the candidate rows: one half-wave per centre atom, candidates keep their order
template <bool FILL, typename RowPtr>
__device__ __forceinline__ void d_filter(const NbrArgs& a, int ctr, int l, RowPtr row_ptr) {
    const bool live = ctr < a.n;
    const int c = live ? ctr : a.n - 1;
    const float4 pc = a.pos_s[c];
    const BoxDims B = box_dims(a, box_id(a, c));
    long long s, e;
    if (a.cand_stride > 0) {
        const int dgc = a.cand_deg[c];
        s = (long long)c * a.cand_stride;
        e = s + (dgc < a.cand_stride ? dgc : a.cand_stride);
    } else {
        s = a.cand_ptr[c]; e = a.cand_ptr[c + 1];
        if (e > a.cand_cap) e = a.cand_cap;
        if (s > e) s = e;
    }
    long long w = FILL ? (long long)row_ptr[c] : 0;
    int cnt = 0;
    for (long long b0 = s; b0 < e; b0 += 32) {
        bool ok = false;
        int b = 0;
        if (b0 + l < e) {
            b = GAMD_CHK_RANGE(a.sticky, a.cand_col[b0 + l], 0, a.n, GAMD_CHK_CAND);
            const float4 pb = a.pos_s[b];
            const float d2 = gamd_mask_d2(pc, pb, B);
            if (a.flavour == 0) ok = d2 < a.rc2;
            else ok = (sqrtf(d2) <= a.rc) && (b != c);
        }
        const unsigned m = half_ballot(ok);
        if (FILL) {
            if (ok && live) {
                const long long at = w + __popc(m & ((1u << l) - 1u));
                if (at < a.e_cap) { a.col[at] = b; a.erow[at] = c; }
            }
            w += __popc(m);
        } else {
            cnt += __popc(m);
        }
    }
    if (FILL && a.self_loop && live && l == 0 && w < a.e_cap) { a.col[w] = c; a.erow[w] = c; }
    if (FILL && live) d_fill_padding(a, c, l, w + (a.self_loop ? 1 : 0));
    if (!FILL && live && l == 0) a.deg[ctr] = cnt + (a.self_loop ? 1 : 0);
}

template <bool FILL>
__global__ void __launch_bounds__(256) k_filter(NbrArgs a) {
    d_filter<FILL>(a, (blockIdx.x * blockDim.x + threadIdx.x) >> 5, threadIdx.x & 31, a.row_ptr);
    if (FILL && a.cand_stride > 0) {
        // Chunk metadata (first piece id, mask of the edges that close a destination segment) of the 16-edge chunks that START
        // in this half-wave's row, one lane per chunk, from the row pointers alone — an edge closes a segment iff it is the
        // last of its row (k_filter_fill_small's rule; row_ptr / na_excl are complete, the scan ran before): no k_chunk_meta
        // launch, and no dependence on other workgroups' erow stores.
        const int ctr = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, l = threadIdx.x & 31;
        long long E = a.counters[CNT_E];
        if (E > a.e_cap) E = a.e_cap;
        if (ctr < a.n) {
            const long long rp = a.row_ptr[ctr], re = a.row_ptr[ctr + 1];          // this row: [rp, re)
            const int na_incl = a.na_excl[ctr] + ((re > rp && (rp % GAMD_CHUNK) != 0) ? 1 : 0);
            for (long long c = (rp + GAMD_CHUNK - 1) / GAMD_CHUNK + l; c * GAMD_CHUNK < re && c * GAMD_CHUNK < E; c += 32) {
                const long long x0 = c * GAMD_CHUNK;
                a.chunk_piece[c] = (int)c + na_incl;
                unsigned mask = 0;
                int at = ctr;
                long long end = re;                                                  // end of row `at`
                for (int r = 0; r < GAMD_CHUNK; ++r) {
                    const long long xe = x0 + r;
                    if (xe >= E) break;
                    while (end <= xe) { ++at; end = a.row_ptr[at + 1]; }
                    if (xe == end - 1 || xe == E - 1) mask |= 1u << r;
                }
                a.chunk_mask[c] = mask;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {          // the second chunk of a last tile that is at most half full
            const long long n_chunks = 2 * ((E + GAMD_TILE - 1) / GAMD_TILE);
            if (n_chunks > 0 && (n_chunks - 1) * GAMD_CHUNK >= E) { a.chunk_piece[n_chunks - 1] = 0; a.chunk_mask[n_chunks - 1] = 0; }
        }
    }
}

// ---- fill pass of the exact filter that scans the degrees itself (one box, 1 024 < n <= 16 384) ---------------------------
// k_filter_count | k_scan_deg | k_filter<true> is count | one workgroup | fill: the single-workgroup scan costs ~6 us plus a
// kernel boundary on every step.  Here every fill workgroup (8 rows) computes the exclusive prefixes it needs itself — the sum of
// deg[] and the number of off-boundary segment starts over the rows in front of it, 4 096 rows per pass through registers (the
// arithmetic of d_scan_deg_fast) — publishes row_ptr / na_excl of its own rows, fills them, and writes the chunk metadata of the
// chunks that start in them from deg[] alone.  The workgroup that holds the last row publishes the totals (and, on a rebuild
// step, the size of the new candidate rows: what k_scan_deg did).  Same row_ptr / na_excl / col / erow / chunk arrays.
struct ShiftedRows {
    const int* p; int r0;
    __device__ __forceinline__ int operator[](int i) const { return p[i - r0]; }
};

__global__ void __launch_bounds__(256) k_filter_fill_scan(NbrArgs a) {
    constexpr int IPT = 16, SPAN = 256 * IPT, ROWS = 8;
    __shared__ int s_w[2][4];
    __shared__ int s_rp[ROWS + 1], s_na[ROWS + 1];               // exclusive prefixes at rows r0 .. r0 + 8
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * ROWS;
    const int r_end = r0 + ROWS < a.n ? r0 + ROWS : a.n;         // rows [0, r_end) are summed; prefixes wanted at r0 .. r_end
    const bool last = r_end == a.n;
    int carry_rp = 0, carry_na = 0;
    for (int base = 0; base < r_end; base += SPAN) {
        const int i0 = base + tid * IPT;
        int v[IPT];
        if (i0 + IPT <= r_end) {
#pragma unroll
            for (int q = 0; q < IPT / 4; ++q) {
                const int4 t = *reinterpret_cast<const int4*>(a.deg + i0 + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPT; ++k) v[k] = (i0 + k < r_end) ? a.deg[i0 + k] : 0;
        }
        int mine = 0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) mine += v[k];
        int x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_w[0][wv] = x;
        __syncthreads();
        int run = carry_rp + x - mine, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int t = s_w[0][w]; if (w < wv) run += t; tot += t; }
        int f[IPT], mine2 = 0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int i = i0 + k;
            if (i >= r0 && i < r_end) s_rp[i - r0] = run;
            f[k] = (i < r_end && v[k] > 0 && (run % GAMD_CHUNK) != 0) ? 1 : 0;
            mine2 += f[k];
            run += v[k];
        }
        int y2 = mine2;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(y2, d, 64); if (lane >= d) y2 += y; }
        if (lane == 63) s_w[1][wv] = y2;
        __syncthreads();
        int run2 = carry_na + y2 - mine2, tot2 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int t = s_w[1][w]; if (w < wv) run2 += t; tot2 += t; }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int i = i0 + k;
            if (i >= r0 && i < r_end) s_na[i - r0] = run2;
            run2 += f[k];
        }
        carry_rp += tot; carry_na += tot2;
        __syncthreads();                                         // s_w is rewritten by the next pass
    }
    if (tid == 0) { s_rp[r_end - r0] = carry_rp; s_na[r_end - r0] = carry_na; }
    __syncthreads();
    if (tid <= r_end - r0 && (tid < r_end - r0 || last)) { a.row_ptr[r0 + tid] = s_rp[tid]; a.na_excl[r0 + tid] = s_na[tid]; }
    const int E_all = carry_rp;                                  // (only the last workgroup's value is the total)
    if (last) {
        if (a.counters[CNT_REBUILD]) d_publish_candidates<256>(a);          // what k_scan_deg does on the five-launch path
        if (tid == 0) {
            a.counters[CNT_E] = E_all;
            a.counters[CNT_PIECES] = (E_all + GAMD_CHUNK - 1) / GAMD_CHUNK + carry_na;
            if ((long long)E_all > a.e_cap) {
                a.counters[CNT_OVERFLOW] = 1; a.sticky[STICKY_EDGE_OVERFLOW] = 1; a.devflags[DEVFLAG_FROZEN] = 1;
            }
            a.counters[CNT_TILES] = (E_all + GAMD_TILE - 1) / GAMD_TILE;
            // the second chunk of a last tile that is at most half full
            const long long E = (long long)E_all > a.e_cap ? a.e_cap : (long long)E_all;
            const long long n_chunks = 2 * ((E + GAMD_TILE - 1) / GAMD_TILE);
            if (n_chunks > 0 && (n_chunks - 1) * GAMD_CHUNK >= E) { a.chunk_piece[n_chunks - 1] = 0; a.chunk_mask[n_chunks - 1] = 0; }
        }
    }
    // rows of this workgroup: one half-wave per atom
    const int ctr = (blockIdx.x * 256 + tid) >> 5, l = tid & 31;
    d_filter<true>(a, ctr, l, ShiftedRows{s_rp, r0});
    // chunk metadata of the 16-edge chunks that START in this half-wave's row, one lane per chunk: an edge closes a segment iff it
    // is the last of its row; the rows behind this one are walked through deg[] (complete since the count pass)
    if (ctr < a.n) {
        const long long rp = s_rp[ctr - r0], re = s_rp[ctr - r0 + 1];
        const int na_incl = s_na[ctr - r0] + ((re > rp && (rp % GAMD_CHUNK) != 0) ? 1 : 0);
        for (long long c = (rp + GAMD_CHUNK - 1) / GAMD_CHUNK + l; c * GAMD_CHUNK < re && c * GAMD_CHUNK < a.e_cap; c += 32) {
            const long long x0 = c * GAMD_CHUNK;
            a.chunk_piece[c] = (int)c + na_incl;
            unsigned mask = 0;
            int at = ctr;
            long long end = re;                                  // end of row `at`
            for (int r = 0; r < GAMD_CHUNK; ++r) {
                const long long xe = x0 + r;
                if (xe >= a.e_cap) break;
                while (end <= xe && at + 1 < a.n) { ++at; end += a.deg[at]; }
                if (end <= xe) break;                            // behind the last edge
                if (xe == end - 1) mask |= 1u << r;
            }
            a.chunk_mask[c] = mask;
        }
    }
}

// ---- count pass of the exact filter for n > 1024, carrying the candidate fill ---------------------------------------------
// One half-wave per centre atom.
//   reuse step    the exact cutoff on the atom's fixed-width candidate row -> deg
//   rebuild step  (the flag k_skin_check raised; the cells are rebuilt) ONE sweep of the 27 cells computes every squared
//                 distance once and uses it twice: < rc + skin -> the atom's new candidate row (what k_cand_fill did in a
//                 gated launch of its own on every step), < rc -> deg.  Row length, total and longest row go to cand_deg /
//                 counters (published by the row scan behind this kernel); the cell counters are left zero for the next
//                 rebuild.
__global__ void __launch_bounds__(256) k_filter_count(NbrArgs a) {
    const int ctr = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, l = threadIdx.x & 31;
    if (a.counters[CNT_REBUILD] == 0) {
        d_filter<false>(a, ctr, l, a.row_ptr);
        return;
    }
    const bool live = ctr < a.n;
    const int cc = live ? ctr : a.n - 1;
    const long long row0 = (long long)cc * a.cand_stride;
    int w = 0, cnt = 0;
    sweep_d2(a, cc, l, [&](bool valid, float d2, int b) {
        const bool cand = valid && in_range(a.flavour, d2, a.rc_build, a.rc2_build, b == cc);
        const bool edge = valid && in_range(a.flavour, d2, a.rc, a.rc2, b == cc);
        const unsigned m = half_ballot(cand);
        if (cand && live) {
            const int at = w + __popc(m & ((1u << l) - 1u));
            if (at < a.cand_stride) a.cand_col[row0 + at] = b;
        }
        w += __popc(m);
        cnt += __popc(half_ballot(edge));
    });
    // (total and longest row: k_scan_deg reduces them from cand_deg)
    if (live && l == 0) {
        a.cand_deg[ctr] = w;
        a.deg[ctr] = cnt + (a.self_loop ? 1 : 0);
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.ncell; k += gridDim.x * blockDim.x) { a.cell_cnt[k] = 0; a.cell_fill[k] = 0; }
}

// Small systems (n <= 1024): exact-filter fill, the scan of the degrees and the chunk metadata in ONE launch.  Every
// workgroup scans deg[] (and the off-boundary segment starts) itself into LDS, 4 atoms per thread, instead of waiting for
// a separate single-workgroup scan kernel; it then fills its rows, and the chunk metadata is derived from the row
// pointers alone (an edge closes a segment iff it is the last of its row), so it does not have to wait for other
// workgroups' erow stores either.  Workgroup 0 publishes row_ptr / na_excl / the counters for the later kernels.
__global__ void __launch_bounds__(256) k_filter_fill_small(NbrArgs a) {
    __shared__ int s_rp[1025];
    __shared__ int s_na[1025];
    __shared__ int s_wave[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int v[4], f[4];
    const int i0 = tid * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < a.n) ? a.deg[i0 + k] : 0;
    int x = (v[0] + v[1]) + (v[2] + v[3]);
    const int mine = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_wave[0][wv] = x;
    __syncthreads();
    int run = x - mine;
    for (int w = 0; w < wv; ++w) run += s_wave[0][w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k <= a.n) s_rp[i0 + k] = run;
        f[k] = (i0 + k < a.n && v[k] > 0 && (run % GAMD_CHUNK) != 0) ? 1 : 0;
        run += v[k];
    }
    if (tid == 255 && a.n == 1024) s_rp[1024] = run;
    int y2 = (f[0] + f[1]) + (f[2] + f[3]);
    const int mine2 = y2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(y2, d, 64); if (lane >= d) y2 += y; }
    if (lane == 63) s_wave[1][wv] = y2;
    __syncthreads();
    int run2 = y2 - mine2;
    for (int w = 0; w < wv; ++w) run2 += s_wave[1][w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k <= a.n) s_na[i0 + k] = run2;
        run2 += f[k];
    }
    if (tid == 255 && a.n == 1024) s_na[1024] = run2;
    __syncthreads();
    const int E_all = s_rp[a.n];
    long long E = E_all;
    if (E > a.e_cap) E = a.e_cap;
    if (blockIdx.x == 0) {
        if (a.cand_stride > 0 && a.counters[CNT_REBUILD]) d_publish_candidates<256>(a);
        for (int i = tid; i <= a.n; i += 256) { a.row_ptr[i] = s_rp[i]; a.na_excl[i] = s_na[i]; }
        if (tid == 0) {
            a.counters[CNT_E] = E_all;
            a.counters[CNT_PIECES] = (E_all + GAMD_CHUNK - 1) / GAMD_CHUNK + s_na[a.n];
            if ((long long)E_all > a.e_cap) {
                a.counters[CNT_OVERFLOW] = 1; a.sticky[STICKY_EDGE_OVERFLOW] = 1; a.devflags[DEVFLAG_FROZEN] = 1;
            }
            a.counters[CNT_TILES] = (E_all + GAMD_TILE - 1) / GAMD_TILE;
        }
    }
    // rows of this workgroup: one half-wave per atom
    d_filter<true>(a, (blockIdx.x * 256 + tid) >> 5, tid & 31, s_rp);
    // chunk metadata, chunks strided over the whole grid; every chunk of every (partly) valid 32-edge tile is written
    const int n_chunks = 2 * (int)((E + GAMD_TILE - 1) / GAMD_TILE);
    for (int c = blockIdx.x * 256 + tid; c < n_chunks; c += gridDim.x * 256) {
        const long long x0 = (long long)c * GAMD_CHUNK;
        if (x0 >= E) { a.chunk_piece[c] = 0; a.chunk_mask[c] = 0; continue; }
        int lo = 0, hi = a.n - 1;                         // last row with row_ptr <= x0 that is not empty up to x0: owner of edge x0
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_rp[mid] <= x0) lo = mid; else hi = mid - 1; }
        int at = lo;
        while (s_rp[at + 1] <= x0) ++at;                  // skip empty rows that share the offset
        const int na_incl = s_na[at] + ((s_rp[at + 1] > s_rp[at] && (s_rp[at] % GAMD_CHUNK) != 0) ? 1 : 0);
        a.chunk_piece[c] = c + na_incl;
        unsigned mask = 0;
        for (int r = 0; r < GAMD_CHUNK; ++r) {
            const long long xe = x0 + r;
            if (xe >= E) break;
            while (s_rp[at + 1] <= xe) ++at;
            if (xe == (long long)s_rp[at + 1] - 1 || xe == E - 1) mask |= 1u << r;
        }
        a.chunk_mask[c] = mask;
    }
}

// ---- CSR from an explicit edge list (model-level forward([pos],[edge_idx]), nn_module.py:636-653) ----
__global__ void k_identity_sort(NbrArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const BoxDims B = box_dims(a, box_id(a, i));
    float4 p;
    p.x = gamd_remainder(a.pos[3 * i + 0], B.bx);
    p.y = gamd_remainder(a.pos[3 * i + 1], B.by);
    p.z = gamd_remainder(a.pos[3 * i + 2], B.bz);
    p.w = node_feature(a, i);
    a.pos_w[i] = p;
    a.pos_s[i] = p;
    a.perm[i] = i;
    a.inv_perm[i] = i;
    a.deg[i] = a.self_loop ? 1 : 0;                        // the appended loop occupies the last slot of the row
}

__global__ void k_edges_count(NbrArgs a, const int* __restrict__ centre, const int* __restrict__ neigh, long long ne) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int c = centre[e], j = neigh[e];
    if (c >= 0 && c < a.n && j >= 0 && j < a.n) atomicAdd(&a.deg[c], 1);
    else a.counters[CNT_OVERFLOW] = 2;                     // index out of range: edge dropped, call fails
}

__global__ void k_edges_fill(NbrArgs a, const int* __restrict__ centre, const int* __restrict__ neigh, long long ne,
                             int* __restrict__ tmp_eid) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int c = centre[e], j = neigh[e];
    if (c < 0 || c >= a.n || j < 0 || j >= a.n) return;
    const int s = atomicAdd(&a.cell_of[c], 1);             // cell_of doubles as the per-row cursor here
    const long long at = (long long)a.row_ptr[c] + s;
    if (at < a.e_cap) tmp_eid[at] = (int)e;
}

// one wave per destination row: order the row's edges by their position in the caller's list
// (atomics above scatter them), then emit col / erow
__global__ void __launch_bounds__(256) k_edges_sort_rows(NbrArgs a, const int* __restrict__ neigh,
                                                         const int* __restrict__ tmp_eid) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= a.n) return;
    const long long s = a.row_ptr[row];
    long long e = a.row_ptr[row + 1];
    if (e > a.e_cap) e = a.e_cap;
    const int total = (int)(e - s);
    // row layout: [the caller's edges][the appended loop (self_loop_mode 1)][padding slots (last atom of a box, n_boxes > 1)]
    int cnt = a.cell_of[row];                                 // edges k_edges_fill placed in this row
    if (cnt > total) cnt = total;
    int behind = cnt;
    if (a.self_loop && behind < total) {
        if (lane == 0) { a.col[s + behind] = row; a.erow[s + behind] = row; }
        ++behind;
    }
    for (int x = behind + lane; x < total; x += 64) { a.col[s + x] = a.n; a.erow[s + x] = row; }
    if (cnt <= 0) return;
    if (cnt <= 64) {
        const int v = lane < cnt ? tmp_eid[s + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += (__shfl(v, j, 64) < v) ? 1 : 0;
        if (lane < cnt) { a.col[s + rank] = neigh[v]; a.erow[s + rank] = row; }
    } else {
        for (int i = lane; i < cnt; i += 64) {                // rank by counting, O(cnt^2 / 64)
            const int v = tmp_eid[s + i];
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += (tmp_eid[s + j] < v) ? 1 : 0;
            a.col[s + rank] = neigh[v];
            a.erow[s + rank] = row;
        }
    }
}

}  // namespace

int launch_csr_from_edges(const NbrArgs& a, const int* centre, const int* neigh, long long n_edges, int* tmp_eid,
                          hipStream_t st) {
    hipError_t e;
    e = hipMemsetAsync(a.counters, 0, sizeof(int) * CNT_COUNT, st); if (e) return (int)e;
    e = hipMemsetAsync(a.cell_of, 0, sizeof(int) * (size_t)a.n, st); if (e) return (int)e;
    const int tb = 256, gb = (a.n + tb - 1) / tb;
    hipLaunchKernelGGL(k_identity_sort, dim3(gb), dim3(tb), 0, st, a); GAMD_CHECK_LAUNCH();
    if (n_edges > 0) {
        const unsigned ge = (unsigned)((n_edges + tb - 1) / tb);
        hipLaunchKernelGGL(k_edges_count, dim3(ge), dim3(tb), 0, st, a, centre, neigh, n_edges); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_scan_deg, dim3(1), dim3(1024), 0, st, a); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_edges_fill, dim3(ge), dim3(tb), 0, st, a, centre, neigh, n_edges, tmp_eid); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_edges_sort_rows, dim3((a.n + 3) / 4), dim3(256), 0, st, a, neigh, tmp_eid); GAMD_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(k_scan_deg, dim3(1), dim3(1024), 0, st, a); GAMD_CHECK_LAUNCH();
    }
    const long long nchunk_cap = (a.e_cap + GAMD_TILE) / GAMD_CHUNK;
    hipLaunchKernelGGL(k_chunk_meta, dim3((unsigned)((nchunk_cap + 255) / 256)), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_neighbor_skin(const NbrArgs& a, hipStream_t st, const MdFuse* fuse) {
    hipError_t e;
    const int tb = 256, gb = (a.n + tb - 1) / tb, ga = (a.n + 7) / 8;
    if (a.counters_next && a.use_small) {                  // the host's choice (gamd_api.hip: small_path)
        if (a.n > 1024) return -22;                        // k_step_small keeps one atom per thread of ONE workgroup
        // small system (n <= 1024): 3 launches and no memset node instead of 13 + 1 (+ 2 integrator launches): counters
        // ping-pong, cell arrays cleared inside the rebuild, integrator halves folded in
        MdArgs md{};
        if (fuse) md = *fuse->md;
        hipLaunchKernelGGL(k_step_small, dim3(1), dim3(1024), 0, st, a, md, fuse ? fuse->do_second : 0, fuse ? fuse->do_first : 0);
        GAMD_CHECK_LAUNCH();
        NbrArgs x = a;
        x.ref_pos = nullptr;
        hipLaunchKernelGGL(k_filter_count, dim3(ga), dim3(256), 0, st, x); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_filter_fill_small, dim3(ga), dim3(256), 0, st, x); GAMD_CHECK_LAUNCH();
        return 0;
    }
    // counters: ping-pong blocks (k_skin_check clears the other one) or one memset; the cell counters are left zero by the
    // rebuild's last kernel.  Four launches per reuse step: check (+ integrator halves) | gated cell build | count (+ candidate
    // fill on a rebuild step) | fill with its own row scan (+ chunk metadata); batches of boxes and n > 16 384: row scan | fill.
    if (!a.counters_next) { e = hipMemsetAsync(a.counters, 0, sizeof(int) * CNT_COUNT, st); if (e) return (int)e; }
    {
        MdArgs md{};
        if (fuse) md = *fuse->md;
        hipLaunchKernelGGL(k_skin_check, dim3(gb), dim3(tb), 0, st, a, md, fuse ? fuse->do_second : 0, fuse ? fuse->do_first : 0);
        GAMD_CHECK_LAUNCH();
    }
    if (a.cells_one_wg && a.n <= 16 * 1024) {
        if (a.ncell <= CELLS_SL_MAXCELL) hipLaunchKernelGGL(k_cells_sliced, dim3(CELLS_SL_WGS), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL(k_cells_one_wg, dim3(1), dim3(1024), 0, st, a);
        GAMD_CHECK_LAUNCH();
    } else {
        // frequent rebuilds, or more atoms than one workgroup should bin: the four cell-list phases as grid-wide kernels, every
        // one gated on the flag k_skin_check has just written
        NbrArgs c = a;
        c.gate = a.counters + CNT_REBUILD;
        c.rc = a.rc_build; c.rc2 = a.rc2_build;
        c.deg = a.cand_deg; c.row_ptr = a.cand_ptr; c.col = a.cand_col; c.erow = nullptr; c.e_cap = a.cand_cap;
        c.self_loop = 0;
        hipLaunchKernelGGL(k_bin, dim3(gb), dim3(tb), 0, st, c); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, c); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_fill_cells, dim3(gb), dim3(tb), 0, st, c); GAMD_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sort_gather, dim3((a.ncell + 3) / 4), dim3(256), 0, st, c); GAMD_CHECK_LAUNCH();
    }
    // exact list of this step (on a rebuild step the count pass also writes the new candidate rows)
    NbrArgs x = a;
    x.ref_pos = nullptr;
    hipLaunchKernelGGL(k_filter_count, dim3(ga), dim3(256), 0, st, x); GAMD_CHECK_LAUNCH();
    if (a.bx.n_boxes <= 1 && a.n <= 16 * 1024 && a.cand_stride > 0) {
        // one box: the fill pass scans the degrees itself (4 launches per reuse step)
        hipLaunchKernelGGL(k_filter_fill_scan, dim3(ga), dim3(256), 0, st, x); GAMD_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_scan_deg, dim3(1), dim3(1024), 0, st, x); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_filter<true>, dim3(ga), dim3(256), 0, st, x); GAMD_CHECK_LAUNCH();
    return 0;
}

int launch_neighbor_build(const NbrArgs& a, hipStream_t st) {
    hipError_t e;
    // counters | cell_cnt | cell_fill live in one allocation (gamd_api.hip): a single memset node
    e = hipMemsetAsync(a.counters, 0, sizeof(int) * (CNT_COUNT + 2 * (size_t)a.ncell_cap), st); if (e) return (int)e;
    const int tb = 256, gb = (a.n + tb - 1) / tb;
    hipLaunchKernelGGL(k_bin, dim3(gb), dim3(tb), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_fill_cells, dim3(gb), dim3(tb), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sort_gather, dim3((a.ncell + 3) / 4), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    const int ga = (a.n + 7) / 8;                      // 8 half-waves (atoms) per 256-thread block
    hipLaunchKernelGGL(k_count, dim3(ga), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scan_deg, dim3(1), dim3(1024), 0, st, a); GAMD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_fill, dim3(ga), dim3(256), 0, st, a); GAMD_CHECK_LAUNCH();
    const long long nchunk_cap = (a.e_cap + GAMD_TILE) / GAMD_CHUNK;
    hipLaunchKernelGGL(k_chunk_meta, dim3((unsigned)((nchunk_cap + 255) / 256)), dim3(256), 0, st, a);
    GAMD_CHECK_LAUNCH();
    return 0;
}
