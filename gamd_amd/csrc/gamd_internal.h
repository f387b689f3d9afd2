// gamd_internal.h — kernel argument blocks and launcher prototypes shared by the .hip translation
// units of libgamd_hip.so.  Not part of the public C ABI (that is include/gamd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "gamd_common.h"

enum { CNT_E = 0, CNT_PIECES = 1, CNT_OVERFLOW = 2, CNT_TILES = 3, CNT_REBUILD = 4, CNT_NCAND = 5,
       CNT_CAND_MAX = 6,      // fixed-stride candidate rows: the longest row of the rebuild of this call
       CNT_COUNT = 8 };
// host-mapped, never cleared by the per-call memset: overflow must survive later steps of an enqueued MD run
enum { STICKY_EDGE_OVERFLOW = 0, STICKY_CAND_OVERFLOW = 1, STICKY_REBUILDS = 2, STICKY_NCAND = 3,
       STICKY_NONFINITE = 4,   // the decoder produced a non-finite force component (NaN / inf positions, or an operand
                               // beyond the fp16 range in the split-fp16 edge MLP, which turns into inf / NaN downstream)
       // checked build (-DGAMD_CHECKED, libgamd_hip_chk.so): the first device-side range check that failed — its code (the
       // GAMD_CHK_* ids below), the offending value and the source line.  Never written by the release build.
       STICKY_CHECK_CODE = 5, STICKY_CHECK_VALUE = 6, STICKY_CHECK_LINE = 7,
       STICKY_COUNT = 8 };
// ---- checked build ----------------------------------------------------------------------------------------------------
// SURVEY.md section 5 ("add a debug build with bounds checks"; the reference's only failure handling is the overflow test of
// graph_utils.py:41-42).  `make checked` compiles every kernel with -DGAMD_CHECKED: each index that a kernel READS FROM MEMORY and
// then uses as an address (CSR source / destination atoms, candidate rows, cell numbers, permutations, piece numbers, CSR write
// positions) passes through GAMD_CHK_RANGE, which records the first violation in the host-mapped sticky block and returns a safe
// value instead (no wild access, no trap: the process survives and gamd_sync_status / the synchronous entry points return -35
// with code, value and line).  In the release build the macro is the identity: same instructions as without it.
enum { GAMD_CHK_CELL = 101, GAMD_CHK_CAND = 102, GAMD_CHK_EDGE_POS = 103, GAMD_CHK_PERM = 104, GAMD_CHK_ENC_SRC = 111, GAMD_CHK_ENC_DST = 112,
       GAMD_CHK_CONV_SRC = 121, GAMD_CHK_CONV_DST = 122, GAMD_CHK_PIECE = 123, GAMD_CHK_NODE_PIECES = 131, GAMD_CHK_INJECTED = 199 };
#ifdef GAMD_CHECKED
template <typename T>
__device__ __forceinline__ T gamd_chk_range(int* sticky, T v, long long lo, long long hi, int code, int line) {
    if ((long long)v >= lo && (long long)v <= hi) return v;
    if (sticky && sticky[STICKY_CHECK_CODE] == 0) {          // first failure wins (a benign race between failing lanes)
        sticky[STICKY_CHECK_VALUE] = (int)v;
        sticky[STICKY_CHECK_LINE] = line;
        __threadfence_system();
        sticky[STICKY_CHECK_CODE] = code;
    }
    return (T)lo;
}
#define GAMD_CHK_RANGE(sticky, v, lo, hi, code) gamd_chk_range((sticky), (v), (long long)(lo), (long long)(hi), (code), __LINE__)
#else
#define GAMD_CHK_RANGE(sticky, v, lo, hi, code) (v)
#endif
// device-resident flags, cleared by the host only (gamd_create, after a regrow):
//   DEVFLAG_FROZEN     a neighbour buffer (edges or candidates) overflowed: the CSR of that call is truncated.  Node kernels
//                      and every integrator kernel return without touching their outputs while it is set, so an enqueued MD
//                      run stops at the last consistent state instead of integrating stale forces
//   DEVFLAG_FROZEN_AT  2 * (index of the step inside the md_run call) + (0: before its first half, 1: before its second
//                      half) of the first integrator kernel that found the flag set, -1 if none did: where to resume
//   DEVFLAG_REBUILDS   candidate rebuilds so far, counted HERE (device memory) and published to the host-mapped sticky block by a
//                      plain store: a read-modify-write on the host-mapped word is a PCIe read in the middle of a single-
//                      workgroup kernel — measured ~200 us per rebuild step (round 5).  Never cleared.
enum { DEVFLAG_FROZEN = 0, DEVFLAG_FROZEN_AT = 1, DEVFLAG_REBUILDS = 2, DEVFLAG_COUNT = 4 };

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, DEVICE): a process that holds handles on several
// devices (gamd_config.device; SURVEY.md 8e "one stream per device from one process") must raise the limit on each of them, or
// the second device launches > 64 KiB of dynamic LDS without it.  One flag per device and kernel instantiation; launchers run
// under the handle's DeviceGuard, so the current device is the one the launch goes to.
// Different handles may be driven from different threads (include/gamd_hip.h): two threads can make their first launch of a
// kernel on one device at the same time, so the flags are atomics (relaxed: hipFuncSetAttribute is idempotent, a lost race
// only repeats it).
constexpr int GAMD_MAX_DEVICES = 64;
struct PerDeviceOnce { std::atomic<bool> done[GAMD_MAX_DEVICES] = {}; };
template <typename... Fn>
inline int gamd_allow_dynamic_lds(PerDeviceOnce& once, int bytes, Fn... fns) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < GAMD_MAX_DEVICES && once.done[dev].load(std::memory_order_relaxed)) return 0;
    const void* list[] = {(const void*)fns...};
    for (const void* fn : list) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
    }
    if (dev >= 0 && dev < GAMD_MAX_DEVICES) once.done[dev].store(true, std::memory_order_relaxed);
    return 0;
}

// ---- several independent boxes in one set of launches (gamd_config.n_boxes > 1) -----------------------------------------
// Box b owns atoms [b * n_per_box, (b + 1) * n_per_box) in the caller's order AND in the sorted order (cells are numbered
// box-major: cell = first cell of box b + local cell, and every box holds exactly n_per_box atoms).  boxes[3 b] = (Lx, Ly, Lz, 0),
// boxes[3 b + 1] = (Lx / 2, Ly / 2, Lz / 2, 0), boxes[3 b + 2] = the bits of int4 (cells along x, y, z, first cell of box b):
// every box has the cell grid a single-box handle would give it, so its CSR rows come out in the same order.
// n_boxes <= 1: the by-value box of the argument block is used and nothing below is read.  The reference evaluates several graphs per forward through build_graph_batches + dgl.batch
// (nn_module.py:655-661, :520-527); here they share every launch.
struct BoxRef {
    int n_boxes, n_per_box;
    float inv_npb;             // 1 / n_per_box
    const float4* boxes;       // device, [n_boxes][3]
};
__device__ __forceinline__ int gamd_box_of(const BoxRef& r, int i) {
    int q = (int)(((float)i + 0.5f) * r.inv_npb);          // off by at most one for i < 2^23 (gamd_create's limit)
    const int rem = i - q * r.n_per_box;
    q += rem >= r.n_per_box ? 1 : (rem < 0 ? -1 : 0);
    return q;
}
struct BoxDims { float bx, by, bz, hx, hy, hz; };
__device__ __forceinline__ BoxDims gamd_box_dims(const BoxRef& r, const float (&box)[3], const float (&half)[3], int box_id) {
    if (r.n_boxes <= 1) return BoxDims{box[0], box[1], box[2], half[0], half[1], half[2]};
    const float4 b = r.boxes[3 * box_id], h = r.boxes[3 * box_id + 1];
    return BoxDims{b.x, b.y, b.z, h.x, h.y, h.z};
}

// ---- neighbour build --------------------------------------------------------------------------
struct NbrArgs {
    int n;                 // atoms (all boxes together)
    BoxRef bx;             // n_boxes > 1: per-box dimensions and cell grids (nc[] below is then unused)
    int* box_shift;        // [n_boxes + 1] scratch of the row scan: padding in front of each box's first CSR row (see d_scan_deg)
    int flavour;           // 0: jax-md path (dr^2 < rc^2, self kept); 1: torch path (|dr| <= rc, no self)
    float box[3], half[3]; // box and 0.5*box in fp32 (nn_module.py:617-621)
    float rc, rc2;
    int nc[3], ncell;
    int ncell_cap;         // counters | cell_cnt[ncell_cap] | cell_fill[ncell_cap] are one allocation
    long long e_cap;       // edge capacity of col/erow/e_frag
    const float* pos;      // [n][3] caller positions (any image)
    const uint8_t* species;// [n] or null
    const float* feat;     // [n] float node feature (nn_module.py:554 node_encoder input) or null: then (float)species
    int self_loop;         // 1: append one self edge per atom at the end of its row (gamd_config.self_loop_mode)
    int* devflags;         // [DEVFLAG_COUNT]
    float4* pos_w;         // [n] wrapped, original order
    float4* pos_s;         // [n + 1] wrapped, sorted order; .w = species; row n stays zero (source of padding edges)
    int* cell_of;          // [n]
    int* cell_cnt;         // [ncell]
    int* cell_fill;        // [ncell]
    int* cell_start;       // [ncell+1]
    int* perm;             // [n + 1] sorted -> original; perm[n] = -2 (the zero row that padding edges point at)
    int* inv_perm;         // [n] original -> sorted
    int* deg;              // [n]
    int* row_ptr;          // [n+1]
    int* na_excl;          // [n+1]
    int* col;              // [e_cap] source (neighbour) atom, sorted index
    int* erow;             // [e_cap] destination (centre) atom, sorted index
    int* chunk_piece;      // [(e_cap+32)/16]
    unsigned* chunk_mask;  // [(e_cap+32)/16]
    int* counters;         // [CNT_COUNT]
    int* counters_next;    // skin path: the OTHER counter block of the ping-pong pair, cleared by k_skin_check / k_step_small
                           // for the next call (no per-call memset node in the stream); null otherwise
    int* sticky;           // [STICKY_COUNT] host-mapped
    // Verlet-skin reuse (graph_utils.py:21-25,36-44: build with cutoff + dr_threshold, rebuild when an atom has moved
    // dr_threshold / 2): candidate CSR built with rc + skin on the steps that need it, exact filter every step
    const int* gate;       // non-null: the kernel returns unless *gate != 0 (rebuild kernels of the skin mode)
    float skin_half2;      // (skin / 2)^2
    int force_rebuild;     // first call, box change, regrown buffers
    float4* ref_pos;       // [n] wrapped positions at the last candidate build (original order)
    int* cand_deg;         // [n]
    int* cand_ptr;         // [n+1]
    int* cand_col;         // [cand_cap] sorted index of the candidate neighbour
    long long cand_cap;
    int cand_stride;       // > 0 (n > 1024): candidate row c is cand_col[c * cand_stride ..][cand_deg[c]] — fixed-width rows as in
                           // jax-md's idx[N, max_occupancy] (graph_utils.py:21-25): one kernel fills them, no count / scan / fill
                           // passes.  0 (n <= 1024, k_step_small): CSR rows cand_col[cand_ptr[c] ..][cand_deg[c]]
    int use_small;         // the host's choice (gamd_api.hip: use_small, n <= 1024): k_step_small + k_filter_fill_small, CSR
                           // candidate rows.  Explicit, not inferred from cand_stride == 0 (an undersized candidate buffer must
                           // surface as an overflow on the grid-wide path, never send n > 1024 atoms into the one-workgroup kernel)
    int cells_one_wg;      // candidate rebuild: the four cell-list phases in one single-workgroup launch (rebuilds are rare: one
                           // gated launch per reuse step instead of four) or as four grid-wide kernels (rebuilds are frequent)
    float rc_build, rc2_build;   // rc + skin
};
struct MdArgs;
// Integrator work that the small-system skin path (n <= 1024, NbrArgs::counters_next set) folds into its first kernel:
// the second half (B) of the previous step and / or the first half (B A O A) of this one (plain BAOAB, no constraints).
struct MdFuse { const MdArgs* md; int do_second, do_first; };
int launch_neighbor_build(const NbrArgs& a, hipStream_t st);
// skin mode: check, gated candidate rebuild, exact filter.  fuse != null (small-system path only): see MdFuse
int launch_neighbor_skin(const NbrArgs& a, hipStream_t st, const MdFuse* fuse = nullptr);
// CSR from a caller-supplied directed edge list (centre[e], neigh[e]); atoms keep the caller's order.
// tmp_eid: [n_edges] scratch.  Rows keep the caller's edge order (deterministic).
int launch_csr_from_edges(const NbrArgs& a, const int* centre, const int* neigh, long long n_edges, int* tmp_eid,
                          hipStream_t st);

// ---- edge encoder -----------------------------------------------------------------------------
struct EncArgs {
    const int* counters;
    const int* devflags;       // DEVFLAG_FROZEN set: return at once (see ConvEdgeArgs)
    const float4* pos_s;
    const int* col;
    const int* erow;
    const int* bond_nbr;       // [n][4] original-index bonded partners (-1 pad) or null
    const int* perm;           // sorted -> original (bond lookup)
    const int* row_ptr;        // CSR rows (self_loop: the last edge of a row is the appended zero-feature loop)
    int self_loop;             // gamd_config.self_loop_mode
    int zero_row;              // = n (all boxes): the source index of the padding slots behind a box's last row
    float box[3], half[3];
    BoxRef bx;                 // n_boxes > 1: the box of an edge is the box of its destination atom
    float length_mean, length_std, gamma;
    // edge_layer_norm over the TRUE edge-embedding width: widths below the 128-blocks the kernels work in are zero-padded by
    // gamd_finalize_weights (padded outputs are exact zeros); 1 / width and the number of padded features
    float ln_inv_width, ln_n_pad;
    int n_feat;                // 44 or 45
    int n_ksteps;              // ceil(n_feat/2)
    const float* centers;      // [40]
    RbfGrid rbf;               // uniform centres: recurrence along the grid instead of one exponential per centre
    const float* w1p;          // packed [4][6][64][4]  (K padded to 48)
    const float* w2p;          // packed 128x128
    const float* w3p;          // packed 128x128
    const float* b1; const float* b2; const float* b3;   // [128]
    const float* ln_g; const float* ln_b;                // [128]
    float* e_frag;             // [tiles][4][4][64][4]
    long long e_cap;
    float* feat_dbg;           // optional [e_cap][48] raw features (debug/parity), or null
    int* sticky;               // host-mapped flags (checked build: GAMD_CHK_RANGE reports here)
    int e_format;              // generic-width encoder (wide.hip): 0 = fp32 fragments, 1 = bf16 fragments (wide_lp.hip), 2 = (hi, lo)
                               // fp16 operand images for the split-fp16 conv kernels (the layout edge_encode_f16x3.hip writes)
};
// self_loop_mode 1: is CSR slot x (source src, destination dst) the loop that was appended behind the row's real edges?  It is the
// last slot of its row that is not a padding slot (padding follows only in the row of a box's last atom, n_boxes > 1).
__device__ __forceinline__ bool gamd_is_appended_loop(const EncArgs& a, long long x, int src, int dst) {
    if (!a.self_loop || src == a.zero_row) return false;
    const long long row_end = a.row_ptr[dst + 1];
    return x == row_end - 1 || (x + 1 < row_end && a.col[x + 1] == a.zero_row);
}

// dimensions of the box an edge lives in (the box of its destination atom, sorted index)
__device__ __forceinline__ BoxDims gamd_edge_box(const EncArgs& a, int dst) {
    return gamd_box_dims(a.bx, a.box, a.half, a.bx.n_boxes > 1 ? gamd_box_of(a.bx, dst) : 0);
}
int launch_edge_encode(const EncArgs& a, int n_blocks, hipStream_t st);
// small edge counts: one tile per 4-wave workgroup, weights straight from L2, bit-identical to launch_edge_encode
int launch_edge_encode_small(const EncArgs& a, int n_blocks, hipStream_t st);
int launch_edge_encode_bf16(const EncArgs& a, int n_blocks, hipStream_t st);   // w*p = bf16 packed fragments, e_frag bf16
int launch_edge_encode_f16x3(const EncArgs& a, int n_blocks, hipStream_t st);  // w*p = [hi | lo] fp16 fragments, e_frag pre-split
// generic widths (wide.hip): n_feat in {4, 5, 44, 45}; w3p = eht packed blocks W3[128 ob : 128 ob + 128, :],
// b3 / ln_g / ln_b are [128 eht]; e_frag is [tiles][eht][4][4][64][4]
int launch_edge_encode_wide(const EncArgs& a, int eht, int n_blocks, hipStream_t st);
// hidden_dim above 128 (wide_d.hip, fp32): D = 128 dt, dt = 2.  w1p points at 1 + dt^2 + eht dt contiguous 64 KiB blocks
// [W1: dt images of 24 KiB] | W2[ob][kb] | W3[ob][kb]; b1, b2 are [128 dt]
int launch_edge_encode_wide_d(const EncArgs& a, int eht, int dt, int n_blocks, hipStream_t st);

// ---- conv layer, edge side --------------------------------------------------------------------
struct ConvEdgeArgs {
    const int* counters;
    const int* devflags;       // DEVFLAG_FROZEN set (a neighbour buffer overflowed earlier in this enqueued run): return at once —
                               // the steps enqueued behind a frozen one cost their launches, not their GEMMs
    const int* col;
    const int* erow;
    const int* chunk_piece;
    const unsigned* chunk_mask;
    const float* e_frag;
    const float* hn;           // [n][128] LayerNorm'd node features
    const float* S;            // [n][128] src_affine(hn) + b_src + b_dst + b_edge_affine2
    const float* D;            // [n][128] dst_affine(hn) (no bias)
                               // (bf16 edge MLP: the three tables are fp16 rows of 256 B, NodeArgs::tab16, S and D pre-multiplied
                               //  by log2 e)
    const float* w1p; const float* w2p; const float* w3p; const float* w4p;   // packed 128x128
    const float* w16p;         // generic-width fp32 path: the same EHT + 2 + HT blocks packed for the 16-edge kernel (wide16.hip), or null
    const float* b1; const float* b3; const float* b4;                           // [128]
    float* partial;            // [pieces][128]
    long long e_cap;
    int zero_row;              // = n: hn / S / D have one extra all-zero row for the padding slots of the last tile
    long long* tdbg;           // profiling builds only: [blocks][8 waves][16] cycle sums, or null
    int* sticky;               // host-mapped flags (checked build: GAMD_CHK_RANGE reports here)
    long long piece_cap;       // rows of `partial` (checked build: every piece index is tested against it)
    float* emb_out;            // update_edge_emb (generic-width kernels only): e_emb = theta_edge(...) of every edge, [E][H] rows
                               // in CSR order, for launch_edge_update; null otherwise
};
int launch_conv_edge(const ConvEdgeArgs& a, int n_blocks, hipStream_t st);
int launch_conv_edge_bf16(const ConvEdgeArgs& a, int n_blocks, hipStream_t st);   // w*p = bf16 packed fragments, e_frag bf16
// small edge counts (conv_edge_small.hip): one tile per 4-wave workgroup, bit-identical to launch_conv_edge
int launch_conv_edge_small(const ConvEdgeArgs& a, int n_blocks, hipStream_t st);
int launch_conv_edge_small_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st);   // = launch_conv_edge_wide
int launch_conv_edge_f16x3(const ConvEdgeArgs& a, int n_blocks, hipStream_t st);  // w*p = [hi | lo] fp16 fragments (64 KiB)
// generic widths (wide.hip): Eh = 128 eht, H = 128 ht.  w1p points at eht + 2 + ht contiguous packed blocks
// W1[:, kb] | W2 | W3 | W4[ob, :]; b4 is [H]; hn and partial rows are H wide, S and D stay 128 wide
int launch_conv_edge_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st);
// the same on 16-edge work units (wide16.hip: v_mfma_f32_16x16x4_f32, one wave per SIMD, a.w16p), bit-identical to
// launch_conv_edge_wide; for launches whose 32-edge tiles leave the SIMDs between 1 and 1.5 (2 and 2.5, ...) quanta of work
int launch_conv_edge_wide16(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st);
// hidden_dim above 128 (wide_d.hip, fp32): w1p points at eht + dt + dt^2 + ht dt contiguous blocks
// W1[:, kb] | W2[db, :] | W3[ob][db] | W4[ob][db]; b3 is [128 dt]; S and D rows are 128 dt wide
int launch_conv_edge_wide_d(const ConvEdgeArgs& a, int eht, int ht, int dt, int n_blocks, hipStream_t st);
// the same on the fp16 matrix pipe by operand splitting (wide_lp.hip): w1p = eht + 2 + ht contiguous [hi | lo] fp16 images,
// e_frag in the encoder's e_format 2, hn rows in their natural [n][H] layout
int launch_conv_edge_f16x3_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st);
// bf16 operands (wide_lp.hip): w1p = eht + 2 + ht contiguous 32 KiB bf16 images, e_frag in the encoder's e_format 1
int launch_conv_edge_bf16_wide(const ConvEdgeArgs& a, int eht, int ht, int n_blocks, hipStream_t st);

// update_edge_emb=True (SmoothConvLayerNew, nn_module.py:91-92, :140-146): the edge embedding the NEXT layers read is
// edge_layer_norm(e_emb) of this layer.  emb: [E][128 ht] rows written by the conv kernel (emb_out); e_frag_out: the same
// fragment-order tiles the edge encoder writes (Eh == H).
struct EdgeUpdateArgs {
    const int* counters;
    const int* devflags;
    long long e_cap;
    const float* emb;
    const float* ln_g; const float* ln_b;      // [128 ht], zero-padded
    float ln_inv_width, ln_n_pad;              // LayerNorm over the true width
    float* e_frag_out;
};
int launch_edge_update(const EdgeUpdateArgs& a, int ht, int n_blocks, hipStream_t st);

// ---- node side --------------------------------------------------------------------------------
struct NodeLayerW {            // one conv layer's node-side parameters (device pointers)
    const float* ln_g; const float* ln_b;
    const float* wsp; const float* wdp; const float* wpdp;   // packed src_affine, dst_affine, phi_dst
    const float* bS;           // b_src + b_dst + b_edge_affine.2
    const float* bP;           // b_phi_dst + b_phi_edge
    const float* wpep;         // packed phi_edge
    const float* wphip;        // packed phi.mlp_layer.1
    const float* bphi;
};
struct NodeArgs {
    const int* counters;       // CNT_OVERFLOW set: the CSR is truncated and piece indices are meaningless -> do nothing
                               // (the host regrows the buffers and re-issues the call)
    const int* devflags;       // DEVFLAG_FROZEN set: same
    int* sticky;               // host-mapped; STICKY_NONFINITE is raised by the decoder
    int n;
    int mode;                  // 0: first (embed + pre(0)); 1: post(l-1) + pre(l); 2: post(L-1) + decoder
    int l0_gate;               // mode 0 inside an enqueued MD run: return at once unless this step rebuilt the candidate list
                               // (counters[CNT_REBUILD]) — layer 0's tables depend on species + weights only and are still valid
    // inputs
    const float4* pos_s;       // .w = species feature
    const float* node_emb;     // [128] (lj) or null
    const float* enc_w; const float* enc_b;   // node_encoder Linear(1->128): weight[:,0], bias (water)
    const int* row_ptr; const int* na_excl; const int* deg;
    const float* partial;
    long long piece_cap;       // rows of `partial` (checked build)
    const float* h_in;         // [n][128] residual stream before this layer's conv (mode 1,2)
    const float* P_in;         // [n][128] phi_dst(hn)+biases from pre()
    NodeLayerW post;           // layer being finished (mode 1,2)
    NodeLayerW pre;            // layer being prepared (mode 0,1)
    // decoder (mode 2)
    const float* dec_w1p; const float* dec_b1; const float* dec_w2; const float* dec_b2;   // w2: [3][128] plain
    float scale, shift;        // sqrt(var), mean of the force scaler (fp32 copy for the device path)
    float ln_inv_width, ln_n_pad;   // graph_conv.norm_layers over the TRUE node width (zero-padded to the 128-blocks): 1 / width, #pad
    int norm_bn;                    // 1: norm_layers are eval-mode BatchNorm1d (use_layer_norm=False), folded into ln_g / ln_b
    int f16x3;                      // 1 (node.hip, reduced-precision edge modes): node-side matrices are (hi | lo) fp16 images, GEMMs in split-fp16
    const int* perm;
    // outputs
    float* h_out;              // [n][128]
    float* hn_out; float* S_out; float* D_out; float* P_out;
    int hn_perm;               // 1: hn rows are stored feature-permuted, position (f & 31) * 4 + (f >> 5), so that the
                               // conv kernel's row-layout gather is one 16-byte load per edge (conv_edge_f16x3.hip)
    int tab16;                 // 1 (bf16 edge MLP, conv_edge_bf16.hip): hn, S, D are written as fp16 rows of 256 B instead of fp32
                               // rows — hn in natural feature order, S and D in the order gamd_tab16_pos() gives (the 64 features a
                               // lane of the chain layout owns come as 8 x 16 B, lanes slot and slot + 32 side by side)
    float* forces_norm;        // [n][3] normalised network output, ORIGINAL atom order (mode 2)
    float* forces;             // [n][3] denormalised fp32 (device MD loop), original order, or null
    long long* tdbg;           // profiling builds only (GAMD_NODE_TIME=1): [workgroup][wave][16] s_memtime marks of the mode-1 launches
};
int launch_node(const NodeArgs& a, hipStream_t st);
// generic widths (wide.hip): h / hn / partial rows and node_emb, enc_w, enc_b, ln_*, bphi are H = 128 ht wide;
// wsp, wdp, wpdp, wpep, dec_w1p are ht K-blocks, wphip is ht output blocks; S, D, P stay 128 wide
int launch_node_wide(const NodeArgs& a, int ht, hipStream_t st);
// hidden_dim above 128 (wide_d.hip, fp32): S, D, P rows, bS, bP, dec_b1 are 128 dt wide, dec_w2 is [3][128 dt]; wsp, wdp, wpdp, wpep,
// dec_w1p are dt x ht blocks (output block major), wphip is ht x dt
int launch_node_wide_d(const NodeArgs& a, int ht, int dt, hipStream_t st);

// ---- integrator -------------------------------------------------------------------------------
// rigid 3-site water (atoms O,H,H): masses and the SETTLE canonical triangle (Miyamoto & Kollman 1992):
// rc = d_HH/2, ra = distance O - centre of mass, rb = distance centre of mass - HH midpoint
struct RigidWater { float m_o, m_h, ra, rb, rc; };
// OpenMM's CMMotionRemover, which the first-half integrators run through addUpdateContextState() at the top of every step
// (hack_integrator.py:142, :272) when the System carries one (the water drivers' openmmtools WaterBox does; :226-235 takes 3
// degrees of freedom off for it): v_i -= sum_j m_j v_j / sum_j m_j, per box.  k_com_partial leaves per-block sums
// (sum m vx, sum m vy, sum m vz, sum m) in `partial`; every first-half thread adds up the `blocks` rows of its box in order.
struct MdCom {
    int enabled;
    int blocks;                // partial rows per box
    double* partial;           // [n_boxes][blocks][4]
};
struct MdArgs {
    int n;
    float* x; float* v;        // [n][3] length unit L (Angstrom | bohr), L/ps
    const float* f;            // [n][3] kJ/mol/nm
    const uint8_t* species;    // [n] or null: species-0 atoms use inv_mass_h when it is > 0
    float inv_mass, inv_mass_h;// 1/amu
    float len;                 // L per nm (10 for Angstrom)
    float dt;                  // ps
    float a, b_len_kT;         // exp(-gamma dt), sqrt(1-a^2)*len*sqrt(kT): O-step sigma = b_len_kT*sqrt(1/m)
    float box[3];
    BoxRef bx;                 // n_boxes > 1: per-box dimensions; box b draws its noise as a single box with seed + b would
    int use_rigid;             // 1: O,H,H triples are rigid (one thread per molecule)
    RigidWater rigid;
    unsigned long long seed; unsigned long long step;
    int* devflags;             // [DEVFLAG_COUNT]: frozen -> return, first kernel to see it records 2 * step_index + half
    int step_index;            // index of this step inside the gamd_md_run call
    MdCom com;                 // centre-of-mass motion removal at the start of the first half (hack_integrator.py:142)
};
int launch_baoab_first(const MdArgs& a, hipStream_t st);    // B A O A  (hack_integrator.py:141-165)
int launch_baoab_second(const MdArgs& a, hipStream_t st);   // B        (hack_integrator.py:175-178)
// B of the step a.step_index AND the per-block momentum sums (a.com) the NEXT step's CMMotionRemover needs, in one launch
int launch_baoab_second_com(const MdArgs& a, hipStream_t st);

// Nose-Hoover chain of the reference drivers (hack_integrator.py:182-330 first half, :334-493 second half;
// one chain state shared by both halves, as copy_state_from_integrator does every step)
struct NhcArgs {
    int n;
    float* x; float* v;        // [n][3] Angstrom, Angstrom/ps
    const float* f;            // [n][3] kJ/mol/nm
    const uint8_t* species;    // [n] or null
    float mass, mass_h;        // amu; species-0 atoms use mass_h when it is > 0
    float len;                 // length units per nm
    float dt;                  // ps
    float box[3];
    BoxRef bx;                 // n_boxes > 1: one chain per box (state / partial are per box), ndf is per box
    int use_rigid;
    RigidWater rigid;
    double kT, freq, ndf;      // kJ/mol, 1/ps, degrees of freedom
    int M, n_c, n_ys;          // chain length, multi-time-step count, Yoshida-Suzuki order
    double w[5];
    double* state;             // [n_boxes][3*M + 2]: xi[M], vxi[M], G[M], scale, KE2
    double* partial;           // [n_boxes][n_blocks] per-block sums of m v^2
    int n_blocks;              // blocks per box
    int* devflags;             // as in MdArgs
    int step_index;
    MdCom com;                 // as in MdArgs; in the first half the remover runs behind propagateNHC (hack_integrator.py:271-272):
                               // KE2 and the chain see the velocities as they are, the scaled velocities lose their COM part
};
// per-block momentum sums of the current velocities (first kernel of a step whose integrator removes the COM motion)
int launch_com_partial(const MdCom& com, const float* v, const uint8_t* species, float inv_mass, float inv_mass_h, int n,
                       const BoxRef& bx, const int* devflags, int by_molecule, hipStream_t st);
int launch_nhc_first(const NhcArgs& a, hipStream_t st);     // propagateNHC; v *= scale; v += dt/2 f/m; x += dt v
int launch_nhc_second(const NhcArgs& a, hipStream_t st);    // v += dt/2 f/m; propagateNHC; v *= scale
