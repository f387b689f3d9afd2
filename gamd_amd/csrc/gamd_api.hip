// gamd_api.hip — handle, weight packing and the extern "C" entry points of include/gamd_hip.h.
#include "../../include/gamd_hip.h"
#include "gamd_common.h"
#include "gamd_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail(-1000 - (int)e__, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// Initialising work — zeroing a fresh buffer, uploading a small table — goes to ONE stream and is waited for on THAT stream
// before the call returns: the caller's stream inside the entry points that take one, the handle's private non-blocking
// stream everywhere else (gamd_create, gamd_finalize_weights, gamd_set_bonds).  Nothing is ordered on, or waits for, the NULL
// stream: a hipMemset / hipMemcpy there is asynchronous to the host for device memory and not ordered with a non-blocking
// stream at all (round 5: the momentum sums of a run's first step, com_partial, were wiped after k_com_partial on the caller's
// non-blocking stream had written them, once in ~300 runs), and a NULL-stream synchronise inside a library stalls every
// blocking stream of the process.  InitStream is set by every entry point (RAII, per thread: different handles may be driven
// from different threads).
thread_local hipStream_t tl_init_stream = nullptr;
struct InitStream {
    hipStream_t prev;
    explicit InitStream(hipStream_t st) : prev(tl_init_stream) { tl_init_stream = st; }
    ~InitStream() { tl_init_stream = prev; }
    InitStream(const InitStream&) = delete;
    InitStream& operator=(const InitStream&) = delete;
};
// host -> device upload of a small table from pageable memory, landed before it returns
hipError_t init_upload(void* dst, const void* src, size_t bytes) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, tl_init_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(tl_init_stream);
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t want, bool zero) {
        if (want <= bytes && p) return 0;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return (int)e; p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return (int)e;
        bytes = want;
        if (zero) {
            // on the call's stream (InitStream) and waited for there: allocations are rare, and what a call allocates and
            // initialises has landed before it returns whatever stream the next call comes on
            e = hipMemsetAsync(p, 0, want, tl_init_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(tl_init_stream);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct LayerDev {
    // edge side
    const float *w1p, *w2p, *w3p, *w4p, *b1, *b3, *b4;
    const float *w16p = nullptr;                        // generic-width fp32: the blocks again, packed for wide16.hip
    const float *e_ln_g = nullptr, *e_ln_b = nullptr;   // update_edge_emb: this layer's edge_layer_norm
    NodeLayerW node;
};

// every entry point runs on the handle's device and leaves the caller's current device as it found it
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// the gamd_md_run / gamd_md_run_nhc call whose steps are still in the stream: what gamd_sync_status needs to finish the
// run after a neighbour-buffer overflow froze it
struct MdPending {
    bool active = false;
    int kind = 0;                      // 0: split BAOAB, 1: split Nose-Hoover chain
    MdArgs m{};
    NhcArgs a{};
    unsigned long long first_step = 0;
    long long n_steps = 0;
    float* x = nullptr;
    float* f = nullptr;
    const uint8_t* species = nullptr;
    hipStream_t st = nullptr;
};

}  // namespace

static_assert(sizeof(gamd_config) == 96 && offsetof(gamd_config, n_boxes) == 88 && offsetof(gamd_config, edge_capacity) == 40,
              "gamd_config layout is part of the C ABI (gamd_amd/_lib.py mirrors it)");

struct gamd_handle {
    gamd_config cfg{};
    int dev = 0;
    int n = 0, L = 0, n_feat = 44, n_cu = 256;   // n: atoms of ALL boxes together (n_boxes * n_per_box)
    int n_boxes = 1, n_per_box = 0;              // gamd_config.n_boxes: independent boxes evaluated in one set of launches
    bool use_small = false;                      // skin mode: the single-workgroup small-system path of neighbor.hip (decided once)
    DevBuf boxes_dev, box_shift;                 // n_boxes > 1: per-box dimensions (BoxRef::boxes), scratch of the row scan
    std::vector<float> boxes_host;               // [n_boxes][3] as last set
    int H = 128, Eh = 128, HT = 1, EHT = 1;      // node width, edge-embedding width (PADDED to 128-blocks) and their block counts
    int H_true = 128, Eh_true = 128, D_true = 128;   // encoding_size, edge_embedding_dim, hidden_dim as given (<= the padded ones)
    int Dp = 128, DT = 1;                        // hidden_dim padded to 128-blocks; DT = 2: the kernels of wide_d.hip (fp32)
    int norm_bn = 0;                             // graph_conv.norm_layers are BatchNorm1d (running statistics in the state_dict)
    bool update_edge = false;                    // update_edge_emb=True: conv.<l>.edge_layer_norm keys in the state_dict
    bool node_f16 = false;                       // node.hip's GEMMs in split-fp16 (reduced-precision edge modes, 128-wide kernels)
    bool wide_enc = false, wide_conv = false;    // generic-width kernels of wide.hip
    long long small_tile_limit = 512;            // fp32 path: at most this many 32-edge tiles -> conv_edge_small.hip
    std::map<std::string, HostTensor> host_w;
    bool finalized = false;
    double scaler_mean = 0.0, scaler_var = 1.0;

    // packed weights on device
    DevBuf wblob;
    std::vector<LayerDev> layers;
    const float *enc_w1p = nullptr, *enc_w2p = nullptr, *enc_w3p = nullptr, *enc_b1 = nullptr, *enc_b2 = nullptr,
                *enc_b3 = nullptr, *enc_lng = nullptr, *enc_lnb = nullptr, *centers = nullptr;
    const float *node_emb = nullptr, *nenc_w = nullptr, *nenc_b = nullptr;
    const float *dec_w1p = nullptr, *dec_b1 = nullptr, *dec_w2 = nullptr, *dec_b2 = nullptr;
    float length_mean = 0.f, length_std = 1.f;
    RbfGrid rbf{};                               // set when edge_expand.centers is a uniform grid

    // per-atom buffers
    DevBuf pos_w, pos_s, cell_of, perm, inv_perm, deg, row_ptr, na_excl, bond_nbr;
    DevBuf hbuf, hn, S, D, P, f_norm, f_den;
    // Layer-0 node tables of their own (skin mode): h0 and pre(0)'s hn / S / D / P depend on the species and the weights only,
    // not on the positions — in sorted atom order they change when the candidate list is rebuilt (the atoms are renumbered), not
    // otherwise.  Inside an enqueued MD run the first node launch of a step therefore returns at once unless that step rebuilt
    // (NodeArgs::l0_gate): the other layers' tables are overwritten layer by layer, these are not.
    DevBuf l0_h, l0_hn, l0_S, l0_D, l0_P;
    // cells
    DevBuf cell_cnt, cell_fill, cell_start;
    int ncell_cap = 0;
    // edges
    long long e_cap = 0;
    long long piece_cap = 0;        // rows of `partial`
    DevBuf col, erow, chunk_piece, chunk_mask, e_frag, partial, feat_dbg, e_emb, e_frag2;
    DevBuf counters, tdbg, tmp_eid, ke_partial, com_partial;
    DevBuf cnt2;                    // small systems in skin mode: two counter blocks used alternately (no per-call memset)
    int cnt_parity = 0;
    long long skin_calls = 0;       // skin-mode force evaluations so far (rebuild-frequency estimate)
    int* cur_counters = nullptr;    // the counter block of the call being enqueued
    int* counters_host = nullptr;   // pinned
    // gamd_forces_host: pinned staging buffers ([n][3] floats each) and the device copy of the positions, allocated on first use
    float* host_in = nullptr;
    float* host_out = nullptr;
    DevBuf pos_in;
    int* sticky_host = nullptr;     // pinned + mapped: overflow flags and rebuild count, written by kernels directly
    int* sticky_dev = nullptr;
    hipStream_t init_stream = nullptr;   // private non-blocking stream: initialising memsets / uploads of the entry points without a stream argument
    DevBuf devflags;                // [DEVFLAG_COUNT] device-resident freeze flag + where an MD run stopped
    const float* feat_dev = nullptr;   // gamd_set_node_features
    const uint8_t* rigid_checked = nullptr;   // species pointer whose O,H,H layout has been validated
    MdPending pending;
    bool has_bonds = false;

    // Verlet-skin reuse (cfg.neighbor_skin > 0)
    float skin = 0.f;
    DevBuf ref_pos, cand_deg, cand_ptr, cand_col;
    long long cand_cap = 0;
    bool cand_valid = false;

    float box[3] = {0, 0, 0};
    int nc[3] = {1, 1, 1};
    int ncell = 1;                  // cells of all boxes together

    // live timing of the conv-edge kernel (gamd_timing_*)
    bool timing = false;
    std::vector<hipEvent_t> tev;     // pairs (start, stop)
    std::vector<int> tev_kind;       // per pair: 0 = conv-layer edge kernel(s) of layer l, 1 = edge encoder; -(l+1) coded below
    size_t tev_used = 0;
    // one event at the top of every MD step of an enqueued run (and one behind the last): gamd_timing_read_steps
    std::vector<hipEvent_t> sev;
    std::vector<uint8_t> sev_closes; // per event: 1 = recorded BEHIND the last step of an enqueue (the interval to the next event
                                     // is the host's gap between two runs, not a step)
    size_t sev_used = 0;
    static constexpr size_t EVENT_POOL_CAP = 1u << 16;   // timing left on across a long run: recording stops here (never unbounded)
};

namespace {

// batches never take the single-workgroup small-system path of neighbor.hip (k_step_small / k_filter_fill_small)
bool small_path(const gamd_handle* h) { return h->use_small; }

BoxRef box_ref(const gamd_handle* h) {
    BoxRef r{};
    r.n_boxes = h->n_boxes;
    r.n_per_box = h->n_per_box;
    r.inv_npb = 1.0f / (float)h->n_per_box;
    r.boxes = h->n_boxes > 1 ? h->boxes_dev.as<float4>() : nullptr;
    return r;
}

// centre-of-mass motion removal (MdCom): per-box, per-block momentum sums
int fill_com(gamd_handle* h, int enabled, MdCom* c) {
    c->enabled = enabled ? 1 : 0;
    c->blocks = std::max(1, std::min(16, (h->n_per_box + 1023) / 1024));
    c->partial = nullptr;
    if (!c->enabled) return 0;
    if (h->com_partial.ensure(sizeof(double) * 4 * (size_t)c->blocks * (size_t)h->n_boxes, true)) return fail(-12, "allocation failed");
    c->partial = h->com_partial.as<double>();
    return 0;
}

int alloc_candidates(gamd_handle* h, long long cap) {
    cap = std::max<long long>(cap, h->n);                         // fixed-width rows: at least one slot per atom
    if (h->cand_col.ensure(sizeof(int) * ((size_t)cap + 64), true)) return fail(-12, "candidate buffer allocation failed");
    h->cand_cap = cap;
    h->cand_valid = false;
    return 0;
}

int alloc_edges(gamd_handle* h, long long e_cap) {
    const size_t ec = (size_t)e_cap + 2 * GAMD_TILE;
    int r = 0;
    r |= h->col.ensure(sizeof(int) * ec, true);
    r |= h->erow.ensure(sizeof(int) * ec, true);
    r |= h->chunk_piece.ensure(sizeof(int) * (ec / GAMD_CHUNK + 2), true);
    r |= h->chunk_mask.ensure(sizeof(unsigned) * (ec / GAMD_CHUNK + 2), true);
    r |= h->e_frag.ensure(sizeof(float) * 4096 * (size_t)h->EHT * (ec / GAMD_TILE + 1), false);
    r |= h->partial.ensure(sizeof(float) * (size_t)h->H * (ec / GAMD_CHUNK + (size_t)h->n + 2), false);
    h->piece_cap = (long long)(ec / GAMD_CHUNK + (size_t)h->n + 2);
    if (h->update_edge) {                        // e_emb rows of one layer and the updated embedding tiles (H == Eh)
        r |= h->e_emb.ensure(sizeof(float) * (size_t)h->H * ec, false);
        r |= h->e_frag2.ensure(sizeof(float) * 4096 * (size_t)h->EHT * (ec / GAMD_TILE + 1), false);
    }
    if (h->cfg.keep_stages) r |= h->feat_dbg.ensure(sizeof(float) * 48 * ec, true);
    if (r) return fail(-12, "edge buffer allocation failed for capacity %lld", e_cap);
    h->e_cap = e_cap;
    if (h->skin > 0.f) {
        const double grow = std::pow(((double)h->cfg.cutoff + h->skin) / (double)h->cfg.cutoff, 3.0);
        // candidate rows have a fixed width (capacity / n) on the grid-wide path AND on the small-system path (round 5), so what
        // must fit is the LONGEST row, not the total: ~2.5 x the mean row (e_cap is already 1.5 x the density estimate, x 1.7)
        const long long want = (long long)((double)e_cap * grow * 1.7) + 1024;
        if (want > h->cand_cap) return alloc_candidates(h, want);
    }
    return 0;
}

// box: host [n_boxes][3].  Every box gets the cell grid a single-box handle would give it (cells at least cutoff + skin
// wide), cells numbered box after box; h->box / h->nc keep box 0 (what the single-box kernels' argument blocks carry).
int set_box(gamd_handle* h, const float* box, hipStream_t st = nullptr) {
    const int nb = h->n_boxes;
    bool changed = h->boxes_host.size() != (size_t)nb * 3;
    for (int b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
            const float v = box[3 * b + d];
            if (!(v > 0.f)) return fail(-22, "box[%d][%d] = %g is not positive", b, d, (double)v);
            if (!changed && h->boxes_host[(size_t)3 * b + d] != v) changed = true;
        }
    if (!changed) return 0;
    h->cand_valid = false;                                        // candidates were built for another box
    std::vector<int> grid((size_t)nb * 4);
    long long ncell = 0;
    for (int b = 0; b < nb; ++b) {
        long long nc_b = 1;
        for (int d = 0; d < 3; ++d) {
            const int nc = std::max(1, (int)std::floor((double)box[3 * b + d] / (((double)h->cfg.cutoff + (double)h->skin) * 1.0001)));
            grid[(size_t)4 * b + d] = nc;
            nc_b *= nc;
        }
        if (ncell + nc_b > (1ll << 30)) return fail(-22, "cell grid too large");
        grid[(size_t)4 * b + 3] = (int)ncell;
        ncell += nc_b;
    }
    for (int d = 0; d < 3; ++d) { h->box[d] = box[d]; h->nc[d] = grid[d]; }
    h->ncell = (int)ncell;
    if ((int)ncell > h->ncell_cap) {
        int r = 0;
        // counters | cell_cnt | cell_fill in one buffer so the per-call clear is a single memset
        r |= h->counters.ensure(sizeof(int) * (CNT_COUNT + 2 * (size_t)ncell), true);
        r |= h->cell_start.ensure(sizeof(int) * ((size_t)ncell + 1), true);
        if (r) return fail(-12, "cell buffer allocation failed");
        h->ncell_cap = (int)ncell;
    }
    h->boxes_host.assign(box, box + (size_t)nb * 3);
    if (nb > 1) {
        // kernels of earlier calls may still read the old dimensions: drain the stream before overwriting them (a box
        // change is rare: NPT-style drivers)
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<float> img((size_t)nb * 12, 0.f);
        for (int b = 0; b < nb; ++b) {
            for (int d = 0; d < 3; ++d) { img[(size_t)12 * b + d] = box[3 * b + d]; img[(size_t)12 * b + 4 + d] = 0.5f * box[3 * b + d]; }
            memcpy(&img[(size_t)12 * b + 8], &grid[(size_t)4 * b], 4 * sizeof(int));
        }
        HIP_TRY(init_upload(h->boxes_dev.p, img.data(), sizeof(float) * img.size()));
    }
    return 0;
}

NbrArgs nbr_args(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev) {
    NbrArgs a{};
    a.n = h->n;
    a.flavour = h->cfg.nbr_flavour;
    for (int d = 0; d < 3; ++d) {
        a.box[d] = h->box[d];
        a.half[d] = 0.5f * h->box[d];
        a.nc[d] = h->nc[d];
    }
    a.rc = h->cfg.cutoff;
    a.rc2 = (float)((double)h->cfg.cutoff * (double)h->cfg.cutoff);   // graph_utils.py:59 cutoff ** 2
    a.ncell = h->ncell;
    a.bx = box_ref(h);
    a.box_shift = h->box_shift.as<int>();
    a.e_cap = h->e_cap;
    a.pos = pos_dev;
    a.species = species_dev;
    a.feat = h->feat_dev;
    a.self_loop = h->cfg.self_loop_mode == GAMD_SELF_LOOP_APPEND_ZERO_FEATURE ? 1 : 0;
    a.devflags = h->devflags.as<int>();
    a.pos_w = h->pos_w.as<float4>();
    a.pos_s = h->pos_s.as<float4>();
    a.cell_of = h->cell_of.as<int>();
    a.ncell_cap = h->ncell_cap;
    a.cell_cnt = h->counters.as<int>() + CNT_COUNT;
    a.cell_fill = h->counters.as<int>() + CNT_COUNT + h->ncell_cap;
    a.cell_start = h->cell_start.as<int>();
    a.perm = h->perm.as<int>();
    a.inv_perm = h->inv_perm.as<int>();
    a.deg = h->deg.as<int>();
    a.row_ptr = h->row_ptr.as<int>();
    a.na_excl = h->na_excl.as<int>();
    a.col = h->col.as<int>();
    a.erow = h->erow.as<int>();
    a.chunk_piece = h->chunk_piece.as<int>();
    a.chunk_mask = h->chunk_mask.as<unsigned>();
    a.counters = h->cur_counters ? h->cur_counters : h->counters.as<int>();
    a.sticky = h->sticky_dev;
    if (h->skin > 0.f) {
        a.skin_half2 = 0.25f * h->skin * h->skin;
        a.rc_build = h->cfg.cutoff + h->skin;
        a.rc2_build = (float)((double)a.rc_build * (double)a.rc_build);
        a.cand_deg = h->cand_deg.as<int>();
        a.cand_ptr = h->cand_ptr.as<int>();
        a.cand_col = h->cand_col.as<int>();
        a.cand_cap = h->cand_cap;
        // fixed-width candidate rows (neighbor.hip): the width follows the capacity (alloc_candidates keeps cand_cap >= n, so
        // the width is at least 1: a buffer that is too small shows up as rows longer than the stride = the overflow -> regrow
        // protocol, not as a zero stride)
        a.use_small = small_path(h) ? 1 : 0;
        a.cand_stride = (int)std::max<long long>(1, std::min<long long>(h->cand_cap / h->n, 1 << 20));
    }
    return a;
}

// ---- weight packing ---------------------------------------------------------------------------
// W [128 out][128 in] block of a row-major matrix with row stride ld -> fragment order of gamd_common.h
void pack128(const float* W, float* out, int ld = 128) {
    for (int tp = 0; tp < 4; ++tp)
        for (int t = 0; t < 4; ++t)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int n = 32 * tp + (lane & 31), k = 32 * t + 8 * q + 4 * (lane >> 5) + j;
                        out[((((tp * 4 + t) * 4 + q) * 64 + lane) * 4) + j] = W[(size_t)n * ld + k];
                    }
}
// W [128 out][128 in] -> operand order of node.hip's 16x16x4 chain: block (ob, blk) = 16 output features x 16 inputs,
// lane (i = lane & 15, g = lane >> 4) holds W[16 ob + i][16 blk + 4 g + 0..3]; a wave's quarter (ob = 2 w, 2 w + 1) is contiguous
void pack16(const float* W, float* out, int ld = 128) {
    for (int ob = 0; ob < 8; ++ob)
        for (int blk = 0; blk < 8; ++blk)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    out[(((ob * 8 + blk) * 64 + lane) * 4) + r] = W[(size_t)(16 * ob + (lane & 15)) * ld + 16 * blk + 4 * (lane >> 4) + r];
}
// the same matrix as (hi | lo) fp16 fragments for node.hip's split-fp16 GEMMs on 16x16x32: fragment ((ob * 4 + m) * 2 + part),
// lane (o = lane & 15, g = lane >> 4), value j = W[16 ob + o][32 m + 16 (j >> 2) + 4 g + (j & 3)]  -- 64 KiB like the fp32 image
void split_f16(float w, uint16_t* hi, uint16_t* lo);
void pack16_f16x3(const float* W, uint16_t* out) {
    for (int ob = 0; ob < 8; ++ob)
        for (int m = 0; m < 4; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * m + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                    const size_t hi = ((((size_t)(ob * 4 + m) * 2 + 0) * 64 + lane) * 8) + j, lo = ((((size_t)(ob * 4 + m) * 2 + 1) * 64 + lane) * 8) + j;
                    split_f16(W[(size_t)(16 * ob + (lane & 15)) * 128 + k], out + hi, out + lo);
                }
}
// encoder first layer W [128][n_feat]: MFMA step s covers features (2s, 2s+1); K padded to 48
void pack_enc1(const float* W, int n_feat, float* out) {
    for (int tp = 0; tp < 4; ++tp)
        for (int g = 0; g < 6; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = 32 * tp + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
                    out[(((tp * 6 + g) * 64 + lane) * 4) + j] = k < n_feat ? W[n * n_feat + k] : 0.f;
                }
}

// ---- packing for the 16-edge generic-width conv kernel (wide16.hip) --------------------------------------------------------
// K position p = 4 m + g of the 32-edge kernels' accumulation order <-> input feature kfeat(m, g); image [ob 8][m4 8][lane 64][c 4]
// = W[row(ob, lane & 15)][kfeat(4 m4 + c, lane >> 4)].  chained = true (W1, W2, W3): packed output row 16 ob + 4 g' + r' is feature
// kfeat(4 ob + r', g'), so the C/D registers of one GEMM are the B operands of the next; chained = false (W4): packed column n of
// block ob is feature 64 (ob >> 2) + 4 n + (ob & 3) (a lane of the F2 output owns four consecutive features).
int kfeat16x(int m, int g) { return 32 * (m >> 3) + 8 * ((m >> 1) & 3) + 4 * (g & 1) + (g >> 1) + 2 * (m & 1); }
void pack16x(const float* W, float* out, int ld, bool chained) {
    for (int ob = 0; ob < 8; ++ob)
        for (int m4 = 0; m4 < 8; ++m4)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 15, g = lane >> 4;
                const int row = chained ? kfeat16x(4 * ob + (i & 3), i >> 2) : 64 * (ob >> 2) + 4 * i + (ob & 3);
                for (int c = 0; c < 4; ++c)
                    out[(((size_t)ob * 8 + m4) * 64 + lane) * 4 + c] = W[(size_t)row * ld + kfeat16x(4 * m4 + c, g)];
            }
}

// ---- bf16 packing (config 5), layout of gamd_bf16.h --------------------------------------------
uint16_t f2bf(float x) {                       // round to nearest even
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
void pack128_bf16(const float* W, uint16_t* out, int ld = 128) {
    for (int tp = 0; tp < 4; ++tp)
        for (int t = 0; t < 4; ++t)
            for (int u = 0; u < 2; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * u + j, half = lane >> 5;
                        const int n = 32 * tp + (lane & 31), k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        out[((((tp * 4 + t) * 2 + u) * 64 + lane) * 8) + j] = f2bf(W[(size_t)n * ld + k]);
                    }
}
void pack_enc1_bf16(const float* W, int n_feat, uint16_t* out) {     // [tp][s][lane][8], K padded to 48
    for (int tp = 0; tp < 4; ++tp)
        for (int s = 0; s < 3; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * tp + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
                    out[(((tp * 3 + s) * 64 + lane) * 8) + j] = k < n_feat ? f2bf(W[n * n_feat + k]) : (uint16_t)0;
                }
}

// ---- split-fp16 packing (GAMD_EDGE_F16X3), layout of gamd_f16x3.h: [hi part | lo part] ------------------
void split_f16(float w, uint16_t* hi, uint16_t* lo) {
    const _Float16 h = (_Float16)w;                              // round to nearest even, subnormals kept
    const _Float16 l = (_Float16)(w - (float)h);
    memcpy(hi, &h, 2); memcpy(lo, &l, 2);
}
void pack128_f16x3(const float* W, uint16_t* out, int ld = 128) {  // 2 x 16384 halves = 64 KiB; W: a 128 x 128 block, row stride ld
    for (int tp = 0; tp < 4; ++tp)
        for (int t = 0; t < 4; ++t)
            for (int u = 0; u < 2; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * u + j, half = lane >> 5;
                        const int n = 32 * tp + (lane & 31), k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const size_t at = ((((tp * 4 + t) * 2 + u) * 64 + lane) * 8) + j;
                        split_f16(W[(size_t)n * ld + k], out + at, out + 16384 + at);
                    }
}
void pack_enc1_f16x3(const float* W, int n_feat, uint16_t* out) {   // 2 x [tp][s][lane][8], K padded to 48
    for (int tp = 0; tp < 4; ++tp)
        for (int s = 0; s < 3; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * tp + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
                    const size_t at = (((tp * 3 + s) * 64 + lane) * 8) + j;
                    split_f16(k < n_feat ? W[n * n_feat + k] : 0.f, out + at, out + 6144 + at);
                }
}

struct BlobBuilder {
    std::vector<float> host;
    size_t add(size_t n_floats) {
        const size_t off = host.size();
        host.resize(off + ((n_floats + 63) & ~(size_t)63), 0.f);
        return off;
    }
};

const HostTensor* find_w(gamd_handle* h, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = h->host_w.find(name);
    if (it == h->host_w.end()) { fail(-2, "missing weight '%s'", name.c_str()); return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) {
        std::string got;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        std::string want;
        for (auto d : shape) want += std::to_string(d) + ",";
        fail(-22, "weight '%s' has shape (%s), expected (%s) for this configuration", name.c_str(), got.c_str(),
             want.c_str());
        return nullptr;
    }
    return &it->second;
}

struct EdgeList { const int* centre; const int* neigh; long long n; };

// rigid-water block shared by both integrators; returns 0 or an error code (message set)
int fill_rigid(const gamd_handle* h, int rigid_water, float mass_o, float mass_h, float r_oh, float r_hh,
               int* use_rigid, RigidWater* g) {
    *use_rigid = 0;
    if (!rigid_water) return 0;
    if (h->n_per_box % 3 != 0) return fail(-22, "rigid_water needs O,H,H triples: n_atoms = %d is not a multiple of 3", h->n_per_box);
    if (!(mass_h > 0.f) || !(mass_o > 0.f)) return fail(-22, "rigid_water needs mass_amu (O) and mass_h_amu (H)");
    if (!(r_oh > 0.f) || !(r_hh > 0.f) || !(r_hh < 2.f * r_oh)) return fail(-22, "rigid_water needs 0 < r_hh < 2 r_oh");
    const double rc = 0.5 * (double)r_hh, t = std::sqrt((double)r_oh * r_oh - rc * rc);
    const double ra = t * 2.0 * mass_h / ((double)mass_o + 2.0 * mass_h);
    g->m_o = mass_o; g->m_h = mass_h;
    g->rc = (float)rc; g->ra = (float)ra; g->rb = (float)(t - ra);
    *use_rigid = 1;
    return 0;
}

int enqueue_forward(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, float* out_norm_dev,
                    float* out_denorm_dev, hipStream_t st, hipEvent_t* evs, int* n_ev, std::vector<std::string>* labels,
                    const EdgeList* el = nullptr, const MdFuse* fuse = nullptr, bool copy_counters = true, bool l0_reuse = false) {
    auto mark = [&](const char* label) {
        if (evs) { (void)hipEventRecord(evs[*n_ev], st); ++*n_ev; labels->push_back(label); }
    };
    int r;
    mark("begin");
    // Verlet-skin reuse: ping-pong counter blocks instead of a memset node per call; n <= 1024: 4 neighbour launches per call
    // with the integrator halves folded in (neighbor.hip)
    const bool pingpong = !el && h->skin > 0.f;
    int* counters_next = nullptr;
    if (pingpong) {
        h->cnt_parity ^= 1;
        h->cur_counters = h->cnt2.as<int>() + h->cnt_parity * CNT_COUNT;
        counters_next = h->cnt2.as<int>() + (1 - h->cnt_parity) * CNT_COUNT;
    } else {
        h->cur_counters = h->counters.as<int>();
    }
    NbrArgs na = nbr_args(h, pos_dev, species_dev);
    na.counters_next = counters_next;
    if (el) {
        if ((r = launch_csr_from_edges(na, el->centre, el->neigh, el->n, h->tmp_eid.as<int>(), st)))
            return fail(-1, "edge-list CSR launch failed (%d)", r);
        h->cand_valid = false;                                    // atom order changed under the candidate list
    } else if (h->skin > 0.f) {
        na.ref_pos = h->ref_pos.as<float4>();
        na.force_rebuild = h->cand_valid ? 0 : 1;
        // rebuilds so far (host-mapped counter, lags the stream by what is enqueued) against calls so far: the one-workgroup
        // cell build costs ~70 us more per rebuild at 10^4 atoms and saves three gated launches (~14 us) on every other step
        ++h->skin_calls;
        na.cells_one_wg = (h->skin_calls < 32 || 6ll * h->sticky_host[STICKY_REBUILDS] < h->skin_calls) ? 1 : 0;
        if (fuse && !pingpong) return fail(-1, "internal: integrator halves can only be fused into the skin path");
        if ((r = launch_neighbor_skin(na, st, fuse))) return fail(-1, "neighbor (skin) launch failed (%d)", r);
        h->cand_valid = true;
    } else if ((r = launch_neighbor_build(na, st))) return fail(-1, "neighbor build launch failed (%d)", r);
    mark("neighbor_build");

    EncArgs ea{};
    ea.counters = h->cur_counters;
    ea.devflags = h->devflags.as<int>();
    ea.pos_s = h->pos_s.as<float4>();
    ea.col = h->col.as<int>();
    ea.erow = h->erow.as<int>();
    ea.bond_nbr = h->has_bonds ? h->bond_nbr.as<int>() : nullptr;
    ea.perm = h->perm.as<int>();
    ea.row_ptr = h->row_ptr.as<int>();
    ea.self_loop = na.self_loop;
    ea.zero_row = h->n;
    for (int d = 0; d < 3; ++d) { ea.box[d] = h->box[d]; ea.half[d] = 0.5f * h->box[d]; }
    ea.bx = box_ref(h);
    ea.length_mean = h->length_mean;
    ea.length_std = h->length_std;
    ea.gamma = (float)(1.0 / 0.025);             // RBFExpansion(high=1, gap=0.025): gamma = 1/gap (nn_module.py:240)
    ea.ln_inv_width = 1.0f / (float)h->Eh_true;
    ea.ln_n_pad = (float)(h->Eh - h->Eh_true);
    ea.n_feat = h->n_feat;
    ea.n_ksteps = (h->n_feat + 1) / 2;
    ea.centers = h->centers;
    ea.rbf = h->rbf;
    ea.w1p = h->enc_w1p; ea.w2p = h->enc_w2p; ea.w3p = h->enc_w3p;
    ea.b1 = h->enc_b1; ea.b2 = h->enc_b2; ea.b3 = h->enc_b3;
    ea.ln_g = h->enc_lng; ea.ln_b = h->enc_lnb;
    ea.e_format = !h->wide_enc ? 0 : h->cfg.edge_dtype == GAMD_EDGE_F16X3 ? 2 : h->cfg.edge_dtype == GAMD_EDGE_BF16 ? 1 : 0;
    ea.e_frag = h->e_frag.as<float>();
    ea.e_cap = h->e_cap;
    ea.sticky = h->sticky_dev;
    ea.feat_dbg = h->cfg.keep_stages ? h->feat_dbg.as<float>() : nullptr;
    // fp32 path, few tiles (measured crossover ~500 tiles, half a tile per SIMD): the latency-oriented kernels, one tile
    // per 4-wave workgroup (conv_edge_small.hip, k_edge_encode_small).  They are bit-identical to the throughput kernels, so
    // the choice (from the last known edge count, or the density estimate before the first call) never shows in the results.
    int small_tiles = 0;
    if (h->cfg.edge_dtype == GAMD_EDGE_F32) {
        const long long e_est = el ? el->n + (na.self_loop ? h->n : 0)
                                   : (h->counters_host[CNT_E] > 0 ? (long long)h->counters_host[CNT_E] : (long long)((double)h->e_cap / 1.5));
        const long long tiles = (e_est + GAMD_TILE - 1) / GAMD_TILE;
        if (tiles <= h->small_tile_limit) small_tiles = (int)std::max<long long>(1, std::min<long long>(tiles + tiles / 8 + 1, 4096));
    }
    // Generic-width fp32 conv kernel on 16-edge work units (k_conv_edge_wide16, bit-identical to k_conv_edge_wide): opt-in through
    // GAMD_KSEL_FORCE_HALF_QUANTUM.  The idea — with t tiles per SIMD the launch takes ceil(2 t) / 2 tile quanta instead of ceil(t):
    // 1.5 instead of 2 at the DFT-water size — holds for the matrix time (61 against 82 us) but not for the launch: one wave per
    // SIMD pays the barrier, the 64 KiB weight copy and its vector work per 8 192-cycle phase instead of per 32 768-cycle phase
    // pair, ~2.7 us x 18 phases (110 us against 107, profiles/r06_experiments.md).  Not chosen automatically.
    const bool half_quantum = h->wide_conv && h->cfg.edge_dtype == GAMD_EDGE_F32 && (h->cfg.kernel_select & GAMD_KSEL_FORCE_HALF_QUANTUM) != 0;
    bool tev_full = false;
    auto tev_begin = [&](int kind) -> int {
        if (!h->timing) return 0;
        if (h->tev_used + 2 > h->tev.size()) {
            if (h->tev.size() >= 8 * gamd_handle::EVENT_POOL_CAP) { tev_full = true; return 0; }   // pool full: not timed any more
            for (int k = 0; k < 256; ++k) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->tev.push_back(e); h->tev_kind.push_back(0); }
        }
        h->tev_kind[h->tev_used] = kind;
        HIP_TRY(hipEventRecord(h->tev[h->tev_used], st));
        return 0;
    };
    auto tev_end = [&]() -> int {
        if (!h->timing || tev_full) return 0;
        HIP_TRY(hipEventRecord(h->tev[h->tev_used + 1], st));
        h->tev_used += 2;
        return 0;
    };
    if ((r = tev_begin(100))) return r;                         // kind 100: edge encoder
    r = h->DT > 1 ? launch_edge_encode_wide_d(ea, h->EHT, h->DT, h->n_cu, st)
        : h->wide_enc ? launch_edge_encode_wide(ea, h->EHT, h->n_cu, st)
        : h->cfg.edge_dtype == GAMD_EDGE_BF16 ? launch_edge_encode_bf16(ea, h->n_cu, st)
        : h->cfg.edge_dtype == GAMD_EDGE_F16X3 ? launch_edge_encode_f16x3(ea, h->n_cu, st)
        : small_tiles > 0 ? launch_edge_encode_small(ea, small_tiles, st) : launch_edge_encode(ea, h->n_cu, st);
    if (r) return fail(-1, "edge encode launch failed (%d)", r);
    if ((r = tev_end())) return r;
    mark("edge_encode");

    const size_t nh = (size_t)h->n * (size_t)h->H;
    auto node = [&](const NodeArgs& na_) {
        return h->DT > 1 ? launch_node_wide_d(na_, h->HT, h->DT, st) : h->wide_conv ? launch_node_wide(na_, h->HT, st) : launch_node(na_, st);
    };
    auto hptr = [&](int l) { return h->hbuf.as<float>() + (h->cfg.keep_stages ? (size_t)l * nh : (size_t)(l & 1) * nh); };

    NodeArgs no{};
#ifdef GAMD_PROFILING
    { static const bool node_time = getenv("GAMD_NODE_TIME") != nullptr; if (node_time) no.tdbg = h->tdbg.as<long long>(); }
#endif
    no.counters = h->cur_counters;
    no.devflags = h->devflags.as<int>();
    no.sticky = h->sticky_dev;
    no.n = h->n;
    no.pos_s = h->pos_s.as<float4>();
    no.node_emb = h->node_emb; no.enc_w = h->nenc_w; no.enc_b = h->nenc_b;
    no.row_ptr = h->row_ptr.as<int>(); no.na_excl = h->na_excl.as<int>(); no.deg = h->deg.as<int>();
    no.partial = h->partial.as<float>();
    no.piece_cap = h->piece_cap;
    no.P_in = h->P.as<float>();
    no.hn_perm = (!h->wide_conv && h->cfg.edge_dtype == GAMD_EDGE_F16X3) ? 1 : 0;
    no.tab16 = (!h->wide_conv && h->cfg.edge_dtype == GAMD_EDGE_BF16) ? 1 : 0;       // conv_edge_bf16.hip gathers fp16 rows
    no.hn_out = h->hn.as<float>(); no.S_out = h->S.as<float>(); no.D_out = h->D.as<float>(); no.P_out = h->P.as<float>();
    no.dec_w1p = h->dec_w1p; no.dec_b1 = h->dec_b1; no.dec_w2 = h->dec_w2; no.dec_b2 = h->dec_b2;
    no.ln_inv_width = 1.0f / (float)h->H_true;
    no.ln_n_pad = (float)(h->H - h->H_true);
    no.norm_bn = h->norm_bn;
    no.f16x3 = h->node_f16 ? 1 : 0;
    no.scale = (float)std::sqrt(h->scaler_var);
    no.shift = (float)h->scaler_mean;
    no.perm = h->perm.as<int>();
    no.forces_norm = out_norm_dev ? out_norm_dev : h->f_norm.as<float>();
    no.forces = out_denorm_dev;

    // layer 0 owns its tables in skin mode (see gamd_handle::l0_*); l0_reuse (steps of an enqueued MD run behind its first: same
    // species buffer, same weights): the launch returns at once unless this step rebuilt the candidate list
    const bool l0 = h->skin > 0.f && !el;
    float* const h0 = (l0 && !h->cfg.keep_stages) ? h->l0_h.as<float>() : hptr(0);
    no.mode = 0;
    no.pre = h->layers[0].node;
    no.h_out = h0;
    if (l0) { no.hn_out = h->l0_hn.as<float>(); no.S_out = h->l0_S.as<float>(); no.D_out = h->l0_D.as<float>(); no.P_out = h->l0_P.as<float>(); }
    no.l0_gate = (l0 && l0_reuse) ? 1 : 0;
    if ((r = node(no))) return fail(-1, "node(0) launch failed (%d)", r);
    no.l0_gate = 0;
    no.hn_out = h->hn.as<float>(); no.S_out = h->S.as<float>(); no.D_out = h->D.as<float>(); no.P_out = h->P.as<float>();
    mark("node_first");


    for (int l = 0; l < h->L; ++l) {
        ConvEdgeArgs ca{};
        ca.counters = h->cur_counters;
        ca.devflags = h->devflags.as<int>();
        ca.col = h->col.as<int>(); ca.erow = h->erow.as<int>();
        ca.chunk_piece = h->chunk_piece.as<int>(); ca.chunk_mask = h->chunk_mask.as<unsigned>();
        // update_edge_emb: layers after the first read the previous layer's LayerNorm(e_emb) (nn_module.py:145-146)
        ca.e_frag = (h->update_edge && l > 0) ? h->e_frag2.as<float>() : h->e_frag.as<float>();
        ca.emb_out = (h->update_edge && l + 1 < h->L) ? h->e_emb.as<float>() : nullptr;
        ca.hn = h->hn.as<float>(); ca.S = h->S.as<float>(); ca.D = h->D.as<float>();
        if (l0 && l == 0) { ca.hn = h->l0_hn.as<float>(); ca.S = h->l0_S.as<float>(); ca.D = h->l0_D.as<float>(); }
        const LayerDev& ld = h->layers[l];
        ca.w1p = ld.w1p; ca.w2p = ld.w2p; ca.w3p = ld.w3p; ca.w4p = ld.w4p; ca.w16p = ld.w16p;
        ca.b1 = ld.b1; ca.b3 = ld.b3; ca.b4 = ld.b4;
        ca.partial = h->partial.as<float>();
        ca.piece_cap = h->piece_cap;
        ca.sticky = h->sticky_dev;
        ca.e_cap = h->e_cap;
        ca.zero_row = h->n;
#ifdef GAMD_CHECKED
        // fault injection for tests/test_gpu_checked.py: the conv kernels are told that the node tables end at row 0, so the
        // first source index above it must be reported (GAMD_CHK_CONV_SRC) and clamped
        { static const bool inject = getenv("GAMD_CHK_INJECT") != nullptr; if (inject) ca.zero_row = 0; }
#endif
        ca.tdbg = h->tdbg.as<long long>();
        if ((r = tev_begin(l))) return r;                        // kind l: conv-layer edge kernel of layer l
        r = h->DT > 1 ? launch_conv_edge_wide_d(ca, h->EHT, h->HT, h->DT, h->n_cu, st)
            : h->wide_conv ? (h->cfg.edge_dtype == GAMD_EDGE_F16X3 ? launch_conv_edge_f16x3_wide(ca, h->EHT, h->HT, h->n_cu, st)
                            : h->cfg.edge_dtype == GAMD_EDGE_BF16 ? launch_conv_edge_bf16_wide(ca, h->EHT, h->HT, h->n_cu, st)
                            : (half_quantum && ca.w16p && !ca.emb_out) ? launch_conv_edge_wide16(ca, h->EHT, h->HT, h->n_cu, st)
                            : small_tiles > 0 ? launch_conv_edge_small_wide(ca, h->EHT, h->HT, small_tiles, st)
                                              : launch_conv_edge_wide(ca, h->EHT, h->HT, h->n_cu, st))
            : h->cfg.edge_dtype == GAMD_EDGE_BF16 ? launch_conv_edge_bf16(ca, h->n_cu, st)
            : h->cfg.edge_dtype == GAMD_EDGE_F16X3 ? launch_conv_edge_f16x3(ca, h->n_cu, st)
            : small_tiles > 0 ? launch_conv_edge_small(ca, small_tiles, st) : launch_conv_edge(ca, h->n_cu, st);
        if (r) return fail(-1, "conv edge launch failed (%d)", r);
        if ((r = tev_end())) return r;
        mark("conv_edge");
        if (ca.emb_out) {
            EdgeUpdateArgs ua{};
            ua.counters = h->cur_counters; ua.devflags = h->devflags.as<int>(); ua.e_cap = h->e_cap;
            ua.emb = ca.emb_out; ua.ln_g = ld.e_ln_g; ua.ln_b = ld.e_ln_b;
            ua.ln_inv_width = 1.0f / (float)h->Eh_true;
            ua.ln_n_pad = (float)(h->Eh - h->Eh_true);
            ua.e_frag_out = h->e_frag2.as<float>();
            if ((r = launch_edge_update(ua, h->HT, small_tiles > 0 ? std::max(1, small_tiles / 4) : 2 * h->n_cu, st)))
                return fail(-1, "edge update launch failed (%d)", r);
            mark("edge_update");
        }

        no.mode = (l == h->L - 1) ? 2 : 1;
        no.post = ld.node;
        if (l + 1 < h->L) no.pre = h->layers[l + 1].node;
        no.h_in = l == 0 ? h0 : hptr(l);
        no.P_in = (l0 && l == 0) ? h->l0_P.as<float>() : h->P.as<float>();
        no.h_out = hptr(l + 1);
        if ((r = node(no))) return fail(-1, "node launch failed (%d)", r);
        mark(no.mode == 2 ? "node_last_decode" : "node_mid");
    }
    // inside an enqueued MD run only the last step's counters are fetched (an overflow in any step reaches the host
    // through the mapped sticky flags): one copy node less per step
    if (copy_counters)
        HIP_TRY(hipMemcpyAsync(h->counters_host, h->cur_counters, sizeof(int) * CNT_COUNT, hipMemcpyDeviceToHost, st));
    return 0;
}

// checked build: a device-side range check (GAMD_CHK_RANGE) failed in some kernel since the last report
int check_traps(gamd_handle* h) {
    const int code = h->sticky_host[STICKY_CHECK_CODE];
    if (!code) return 0;
    const int value = h->sticky_host[STICKY_CHECK_VALUE], line = h->sticky_host[STICKY_CHECK_LINE];
    h->sticky_host[STICKY_CHECK_CODE] = 0;
    return fail(-35, "checked build: device-side range check %d failed (value %d, source line %d)", code, value, line);
}

int check_ready(gamd_handle* h) {
    if (!h) return fail(-22, "null handle");
    if (!h->finalized) return fail(-22, "weights not finalized (call gamd_finalize_weights)");
    return 0;
}

// what every force evaluation needs besides positions and box (shared by the forces, md_run and profile entry points)
int check_model_inputs(gamd_handle* h, const uint8_t* species_dev) {
    if (h->cfg.kind == GAMD_KIND_WATER && !species_dev && !h->feat_dev)
        return fail(-22, "water model needs species (or gamd_set_node_features)");
    if (h->cfg.use_bond && !h->has_bonds) return fail(-22, "use_bond set but no bonds given (gamd_set_bonds)");
    return 0;
}

// rigid_water integrates O,H,H triples: check once per species buffer that the layout really is 1,0,0 per molecule
int check_rigid_layout(gamd_handle* h, const uint8_t* species_dev, hipStream_t st) {
    if (!species_dev) return fail(-22, "rigid_water needs species (O = 1, H = 0, atoms ordered O,H,H)");
    if (h->rigid_checked == species_dev) return 0;
    std::vector<uint8_t> sp((size_t)h->n);
    HIP_TRY(hipMemcpyAsync(sp.data(), species_dev, sp.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < h->n; ++i)
        if ((sp[(size_t)i] != 0) != (i % 3 == 0))
            return fail(-22, "rigid_water needs atoms ordered O,H,H per molecule: species[%d] = %d", i, (int)sp[(size_t)i]);
    h->rigid_checked = species_dev;
    return 0;
}

int clear_devflags(gamd_handle* h) {
    const int init[2] = {0, -1};                             // FROZEN, FROZEN_AT; the rebuild counter behind them stays
    HIP_TRY(init_upload(h->devflags.p, init, sizeof(init)));
    return 0;
}

// live timing: a HIP event on the launch stream in front of the first kernel of an MD step (and behind the last kernel of
// the run); consecutive events bracket one step.  Events are created outside the timed region (gamd_timing_enable).
int step_event(gamd_handle* h, hipStream_t st, bool closes = false) {
    if (!h->timing) return 0;
    if (h->sev_used == h->sev.size()) {
        if (h->sev.size() >= gamd_handle::EVENT_POOL_CAP) return 0;          // pool full: the rest of the run is not timed
        for (int k = 0; k < 256; ++k) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->sev.push_back(e); h->sev_closes.push_back(0); }
    }
    h->sev_closes[h->sev_used] = closes ? 1 : 0;
    HIP_TRY(hipEventRecord(h->sev[h->sev_used++], st));
    return 0;
}

// steps [s_begin, n_steps) of the pending MD run; skip_first: the first half of step s_begin has already been done
int enqueue_md_steps(gamd_handle* h, long long s_begin, bool skip_first) {
    MdPending& p = h->pending;
    int r;
    // Skin mode, BAOAB (free atoms or rigid water): the B of step s-1 and the B A O A of step s ride in the first kernel of step s's force
    // evaluation (k_step_small / k_skin_check): 2 launches less per step; the last B is launched on its own.
    if (p.kind == 0 && h->skin > 0.f) {
        for (long long s = s_begin; s < p.n_steps; ++s) {
            if ((r = step_event(h, p.st))) return r;
            p.m.step = p.first_step + (unsigned long long)s;
            p.m.step_index = (int)s;
            int do_second = s > s_begin ? 1 : 0;
            const int do_first = (skip_first && s == s_begin) ? 0 : 1;
            if (p.m.com.enabled) {
                // COM motion removal sits between the B of step s-1 and the first half of step s and needs a sum over all
                // atoms: the B is launched on its own, then the momentum sums; only the first half rides in the neighbour kernel
                if (do_second && do_first) {                      // B of step s - 1 + the momentum sums of step s: one launch
                    MdArgs prev = p.m;
                    prev.step_index = (int)(s - 1);
                    if ((r = launch_baoab_second_com(prev, p.st))) return fail(-1, "integrator launch failed (%d)", r);
                    do_second = 0;
                } else {
                    if (do_second) {
                        MdArgs prev = p.m;
                        prev.step_index = (int)(s - 1);
                        if ((r = launch_baoab_second(prev, p.st))) return fail(-1, "integrator launch failed (%d)", r);
                        do_second = 0;
                    }
                    if (do_first && (r = launch_com_partial(p.m.com, p.m.v, p.m.species, p.m.inv_mass, p.m.inv_mass_h, p.m.n, p.m.bx,
                                                            p.m.devflags, p.m.use_rigid, p.st)))
                        return fail(-1, "integrator launch failed (%d)", r);
                }
            }
            const MdFuse fuse{&p.m, do_second, do_first};
            if ((r = enqueue_forward(h, p.x, p.species, nullptr, p.f, p.st, nullptr, nullptr, nullptr, nullptr, &fuse,
                                     s + 1 == p.n_steps, s > s_begin)))
                return r;
        }
        if (p.n_steps > s_begin) {
            p.m.step_index = (int)(p.n_steps - 1);
            if ((r = launch_baoab_second(p.m, p.st))) return fail(-1, "integrator launch failed (%d)", r);
        }
        return step_event(h, p.st, true);
    }
    for (long long s = s_begin; s < p.n_steps; ++s) {
        if ((r = step_event(h, p.st))) return r;
        const bool first = !(skip_first && s == s_begin);
        const bool last = s + 1 == p.n_steps;
        if (p.kind == 0) {
            p.m.step = p.first_step + (unsigned long long)s;
            p.m.step_index = (int)s;
            if (first && (r = launch_baoab_first(p.m, p.st))) return fail(-1, "integrator launch failed (%d)", r);
            if ((r = enqueue_forward(h, p.x, p.species, nullptr, p.f, p.st, nullptr, nullptr, nullptr, nullptr, nullptr, last, s > s_begin))) return r;
            if ((r = launch_baoab_second(p.m, p.st))) return fail(-1, "integrator launch failed (%d)", r);
        } else {
            p.a.step_index = (int)s;
            if (first && (r = launch_nhc_first(p.a, p.st))) return fail(-1, "integrator launch failed (%d)", r);
            if ((r = enqueue_forward(h, p.x, p.species, nullptr, p.f, p.st, nullptr, nullptr, nullptr, nullptr, nullptr, last, s > s_begin))) return r;
            if ((r = launch_nhc_second(p.a, p.st))) return fail(-1, "integrator launch failed (%d)", r);
        }
    }
    return step_event(h, p.st, true);
}

}  // namespace

extern "C" {

#ifdef GAMD_CHECKED
const char* gamd_version(void) { return "gamd_hip 0.1 (gfx950) checked"; }       // libgamd_hip_chk.so: device-side range checks
#else
const char* gamd_version(void) { return "gamd_hip 0.1 (gfx950)"; }
#endif
const char* gamd_last_error(void) { return g_err; }

int32_t gamd_create(const gamd_config* cfg, gamd_handle** out) {
    if (!cfg || !out) return fail(-22, "null argument");
    if (cfg->n_atoms <= 0) return fail(-22, "n_atoms must be positive");
    if (cfg->n_boxes < 0 || cfg->n_boxes > (1 << 20)) return fail(-22, "n_boxes out of range");
    const int n_boxes = cfg->n_boxes > 1 ? cfg->n_boxes : 1;
    // node-table rows are addressed with 32-bit byte offsets (row * 512 B, scalar base + offset loads) in the conv-layer edge
    // kernels: 2^23 - 1 rows including the zero row (a 288 GB device holds ~5e6 atoms of this model)
    if ((long long)cfg->n_atoms * n_boxes >= (1 << 23) - 1)
        return fail(-22, "n_atoms x n_boxes = %lld: at most %d atoms per handle", (long long)cfg->n_atoms * n_boxes, (1 << 23) - 2);
    if (cfg->n_layers <= 0 || cfg->n_layers > 16) return fail(-22, "n_layers out of range");
    if (!(cfg->cutoff > 0.f)) return fail(-22, "cutoff must be positive");
    if (cfg->edge_dtype != GAMD_EDGE_F32 && cfg->edge_dtype != GAMD_EDGE_BF16 && cfg->edge_dtype != GAMD_EDGE_F16X3)
        return fail(-22, "unknown edge_dtype");
    if (!(cfg->neighbor_skin >= 0.f)) return fail(-22, "neighbor_skin must be >= 0");
    if (cfg->self_loop_mode != GAMD_SELF_LOOP_DGL07_NOOP && cfg->self_loop_mode != GAMD_SELF_LOOP_APPEND_ZERO_FEATURE)
        return fail(-22, "unknown self_loop_mode %d", cfg->self_loop_mode);
    if (cfg->self_loop_mode != GAMD_SELF_LOOP_DGL07_NOOP && cfg->edge_dtype != GAMD_EDGE_F32)
        return fail(-22, "self_loop_mode 1 is built for the fp32 edge dtype only");
    if (cfg->kernel_select & ~(GAMD_KSEL_FORCE_GENERIC_WIDTH | GAMD_KSEL_FORCE_HALF_QUANTUM))
        return fail(-22, "unknown kernel_select bits 0x%x", cfg->kernel_select);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(-19, "no HIP device available: libgamd_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(-22, "device %d out of range (%d visible)", cfg->device, ndev);
    DeviceGuard guard(cfg->device);            // the caller's current device is restored on every return path
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    // Widths as build_model hands them to the model constructors (nn_module.py:561-601, :410-460, :266-320).  The kernels work
    // in 128-wide blocks: narrower (or in-between) widths are zero-padded by gamd_finalize_weights — padded features stay
    // exact zeros through every Linear / SiLU / GELU, and the two LayerNorms divide by the true width.
    const int H_true = cfg->encoding_size ? cfg->encoding_size : 128, Eh_true = cfg->edge_embedding_dim ? cfg->edge_embedding_dim : 128;
    const int D_true = cfg->hidden_dim ? cfg->hidden_dim : 128;
    if (H_true < 1 || H_true > 256 || Eh_true < 1 || Eh_true > 256)
        return fail(-22, "encoding_size and edge_embedding_dim must be in [1, 256] (got %d, %d)", H_true, Eh_true);
    if (D_true < 1 || D_true > 256) return fail(-22, "hidden_dim must be in [1, 256] (got %d)", D_true);
    const int Dp = D_true <= 128 ? 128 : 256;
    // hidden_dim above 128: two 128-blocks per D-wide operand, the fp32 kernels of wide_d.hip
    if (Dp > 128 && cfg->edge_dtype != GAMD_EDGE_F32)
        return fail(-22, "hidden_dim above 128 is built for edge_dtype f32 only (got hidden_dim %d)", D_true);
    const int H = H_true <= 128 ? 128 : 256, Eh = Eh_true <= 128 ? 128 : 256;
    const bool exact128 = H_true == 128 && Eh_true == 128 && D_true == 128;
    const bool generic = H != 128 || Eh != 128 || cfg->no_expand_edge;
    // reduced-precision edge MLPs (bf16, fp32-grade split-fp16) exist for every width and both feature sets: 128 / 128 / 128
    // expanded runs the specialised kernels, anything else wide.hip's encoder writing the operands + wide_lp.hip
    if (cfg->edge_dtype != GAMD_EDGE_F32 && H == 256 && (long long)cfg->n_atoms * n_boxes > (1ll << 22) - 2)
        return fail(-22, "bf16 / split-fp16 with encoding_size > 128: at most 2^22 - 2 atoms per handle (32-bit byte offsets into hn)");
    gamd_handle* h = new gamd_handle();
    h->cfg = *cfg;
    h->dev = cfg->device;
    if (hipStreamCreateWithFlags(&h->init_stream, hipStreamNonBlocking) != hipSuccess) {
        h->init_stream = nullptr;
        gamd_destroy(h);
        return fail(-12, "stream creation failed");
    }
    InitStream init(h->init_stream);
    h->n_boxes = n_boxes;
    h->n_per_box = cfg->n_atoms;
    h->n = cfg->n_atoms * n_boxes;
    h->L = cfg->n_layers;
    h->H = H; h->Eh = Eh; h->HT = H / 128; h->EHT = Eh / 128;
    h->H_true = H_true; h->Eh_true = Eh_true; h->D_true = D_true;
    h->Dp = Dp; h->DT = Dp / 128;
    h->skin = cfg->neighbor_skin;
    if (cfg->small_tile_limit != 0) h->small_tile_limit = cfg->small_tile_limit < 0 ? -1 : cfg->small_tile_limit;
    const bool forced = (cfg->kernel_select & GAMD_KSEL_FORCE_GENERIC_WIDTH) && cfg->edge_dtype == GAMD_EDGE_F32;
    // bf16 / split-fp16 outside 128 / 128 / 128 expanded: encoder, conv and node kernels all take the generic-width route
    const bool lp_generic = cfg->edge_dtype != GAMD_EDGE_F32 && (generic || !exact128);
    h->wide_enc = generic || forced || lp_generic || Dp > 128;
    h->wide_conv = H != 128 || Eh != 128 || forced || lp_generic || Dp > 128;
    h->n_feat = (cfg->no_expand_edge ? 4 : 44) + (cfg->use_bond ? 1 : 0);
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t n = (size_t)h->n;
    int r = 0;
    r |= h->pos_w.ensure(sizeof(float4) * n, true);
    // one extra row behind pos_s / perm for the padding edges that align the boxes of a batch (col = n): zero position,
    // original id -2 (matches no bond partner)
    r |= h->pos_s.ensure(sizeof(float4) * (n + 1), true);
    r |= h->cell_of.ensure(sizeof(int) * n, true);
    r |= h->perm.ensure(sizeof(int) * (n + 1), true);
    if (n_boxes > 1) {
        r |= h->boxes_dev.ensure(sizeof(float) * 12 * (size_t)n_boxes, true);
        r |= h->box_shift.ensure(sizeof(int) * ((size_t)n_boxes + 2), true);
    }
    r |= h->inv_perm.ensure(sizeof(int) * n, true);
    r |= h->deg.ensure(sizeof(int) * n, true);
    r |= h->row_ptr.ensure(sizeof(int) * (n + 1), true);
    r |= h->na_excl.ensure(sizeof(int) * (n + 1), true);
    const size_t nh = n * (size_t)H * sizeof(float), nd = n * (size_t)Dp * sizeof(float), zd = (size_t)Dp * sizeof(float);
    r |= h->hbuf.ensure(nh * (cfg->keep_stages ? (size_t)h->L + 1 : 2), true);
    // one extra, all-zero row (index n) behind the node tables the conv-layer edge kernel gathers from: the padding
    // slots of the last 32-edge tile point at it, so their messages are exact zeros without a per-element mask
    r |= h->hn.ensure(nh + (size_t)H * sizeof(float), true);
    r |= h->S.ensure(nd + zd, true);
    r |= h->D.ensure(nd + zd, true);
    r |= h->P.ensure(nd, true);
    if (cfg->neighbor_skin > 0.f) {
        r |= h->l0_hn.ensure(nh + (size_t)H * sizeof(float), true);
        r |= h->l0_S.ensure(nd + zd, true);
        r |= h->l0_D.ensure(nd + zd, true);
        r |= h->l0_P.ensure(nd, true);
        if (!cfg->keep_stages) r |= h->l0_h.ensure(nh, true);
    }
    r |= h->f_norm.ensure(sizeof(float) * 3 * n, true);
    r |= h->f_den.ensure(sizeof(float) * 3 * n, true);
    r |= h->tdbg.ensure(sizeof(long long) * 16 * 8 * 1024, true);
    r |= h->devflags.ensure(sizeof(int) * DEVFLAG_COUNT, true);
    r |= h->cnt2.ensure(sizeof(int) * 2 * CNT_COUNT, true);
    if (r) { gamd_destroy(h); return fail(-12, "device allocation failed"); }
    {
        const int no_atom = -2;
        if (init_upload(h->perm.as<int>() + n, &no_atom, sizeof(int)) != hipSuccess) {
            gamd_destroy(h);
            return fail(-1, "device copy failed");
        }
    }
    if ((r = clear_devflags(h))) { gamd_destroy(h); return r; }
    if (hipHostMalloc((void**)&h->counters_host, sizeof(int) * CNT_COUNT) != hipSuccess) {
        gamd_destroy(h);
        return fail(-12, "pinned allocation failed");
    }
    memset(h->counters_host, 0, sizeof(int) * CNT_COUNT);
    if (hipHostMalloc((void**)&h->sticky_host, sizeof(int) * STICKY_COUNT, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->sticky_dev, h->sticky_host, 0) != hipSuccess) {
        gamd_destroy(h);
        return fail(-12, "mapped host allocation failed");
    }
    memset(h->sticky_host, 0, sizeof(int) * STICKY_COUNT);
    if (h->skin > 0.f) {
        int rr = 0;
        rr |= h->ref_pos.ensure(sizeof(float4) * n, true);
        rr |= h->cand_deg.ensure(sizeof(int) * n, true);
        rr |= h->cand_ptr.ensure(sizeof(int) * (n + 1), true);
        if (rr) { gamd_destroy(h); return fail(-12, "device allocation failed"); }
    }
    {
        std::vector<float> all((size_t)n_boxes * 3);              // every box starts with the constructor's box
        for (int b = 0; b < n_boxes; ++b)
            for (int d = 0; d < 3; ++d) all[(size_t)3 * b + d] = cfg->box[d];
        if ((r = set_box(h, all.data()))) { gamd_destroy(h); return r; }
    }
    {
        // Small-system path (k_step_small: skin check, integrator halves and — on the steps that need it — the cell build of a
        // candidate rebuild in ONE workgroup; the candidate rows are written by the count pass behind it).  Worth it while
        // rebuilds are rare and cheap: a box of 3 cells per axis (the 774-atom DFT-water configuration: every atom is a
        // candidate of every other, a rebuild every ~7 steps) is as fast on the grid-wide path (measured: 0.8825 against
        // 0.884 ms per step), and a dilute box with thousands of cells is walked faster by 32 workgroups than by one.
        const double per_atom = std::min<double>(h->n, 27.0 * h->n / std::max(1, h->ncell));
        h->use_small = h->n <= 1024 && n_boxes <= 1 && (double)h->n * per_atom <= 3.0e5 && h->ncell <= 4096;
    }
    long long ecap = cfg->edge_capacity;
    if (ecap <= 0) {
        const double vol = (double)cfg->box[0] * cfg->box[1] * cfg->box[2];
        const double per_atom = 4.18879 * std::pow((double)cfg->cutoff, 3) * (double)h->n_per_box / vol + 1.0;
        ecap = (long long)(1.5 * per_atom * (double)h->n) + 1024 + 16ll * n_boxes;
    }
    if (cfg->self_loop_mode) ecap += h->n;
    if ((r = alloc_edges(h, ecap))) { gamd_destroy(h); return r; }
    if (hipStreamSynchronize(h->init_stream) != hipSuccess) { gamd_destroy(h); return fail(-1, "device synchronisation failed"); }
    *out = h;                                  // every initialising memset / copy has landed: any stream may use the handle
    return 0;
}

int32_t gamd_destroy(gamd_handle* h) {
    if (!h) return 0;
    DeviceGuard guard(h->dev);
    InitStream init(h->init_stream);
    h->devflags.release();
    h->cnt2.release();
    DevBuf* bufs[] = {&h->boxes_dev, &h->box_shift, &h->wblob, &h->pos_w, &h->pos_s, &h->cell_of, &h->perm, &h->inv_perm, &h->deg, &h->row_ptr,
                      &h->na_excl, &h->bond_nbr, &h->hbuf, &h->hn, &h->S, &h->D, &h->P, &h->l0_h, &h->l0_hn, &h->l0_S, &h->l0_D, &h->l0_P, &h->f_norm, &h->f_den,
                      &h->cell_cnt, &h->cell_fill, &h->cell_start, &h->col, &h->erow, &h->chunk_piece,
                      &h->chunk_mask, &h->e_frag, &h->e_emb, &h->e_frag2, &h->partial, &h->feat_dbg, &h->counters, &h->tdbg, &h->tmp_eid, &h->ke_partial, &h->com_partial,
                      &h->ref_pos, &h->cand_deg, &h->cand_ptr, &h->cand_col};
    for (DevBuf* b : bufs) b->release();
    h->pos_in.release();
    if (h->host_in) (void)hipHostFree(h->host_in);
    if (h->host_out) (void)hipHostFree(h->host_out);
    if (h->counters_host) (void)hipHostFree(h->counters_host);
    if (h->sticky_host) (void)hipHostFree(h->sticky_host);
    for (hipEvent_t e : h->tev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->sev) (void)hipEventDestroy(e);
    if (h->init_stream) (void)hipStreamDestroy(h->init_stream);
    delete h;
    return 0;
}

int32_t gamd_load_weight(gamd_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 2) return fail(-22, "bad argument to gamd_load_weight");
    HostTensor t;
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); cnt *= (size_t)shape[i]; }
    t.data.assign(data, data + cnt);
    h->host_w[name] = std::move(t);
    h->finalized = false;
    return 0;
}

int32_t gamd_finalize_weights(gamd_handle* h) {
    if (!h) return fail(-22, "null handle");
    DeviceGuard guard(h->dev);
    InitStream init(h->init_stream);
    const int F = h->n_feat, L = h->L;
    const int64_t H = h->H, Eh = h->Eh, HT = h->HT, EHT = h->EHT;         // padded widths the kernels work in
    const int64_t Ht = h->H_true, Et = h->Eh_true, Dt = h->D_true;        // the state_dict's widths
    const int64_t Dp = h->Dp, DT = h->DT;                                 // hidden_dim padded to 128-blocks (DT = 2: wide_d.hip)
    const bool expand = !h->cfg.no_expand_edge;
    BlobBuilder bb;
    struct Off { size_t w1p, w2p, w3p, w4p, w16p = 0, b1, b3, b4, elng = 0, elnb = 0, lng, lnb, wsp, wdp, wpdp, bS, bP, wpep, wphip, bphi; };
    std::vector<Off> lo(L);
    // get(name, true shape, padded shape): the tensor as the reference stores it, zero-padded to the kernels' block widths.
    // Padded output rows / input columns are zeros, so padded features are exact zeros through every layer.
    std::deque<HostTensor> padded;
    auto get = [&](const std::string& nm, std::initializer_list<int64_t> shp, std::initializer_list<int64_t> pad) -> const HostTensor* {
        const HostTensor* t = find_w(h, nm, shp);
        if (!t) return nullptr;
        const std::vector<int64_t> ps(pad);
        if (t->shape == ps) return t;
        HostTensor o;
        o.shape = ps;
        const int64_t r = ps.size() == 2 ? t->shape[0] : 1, c = ps.size() == 2 ? t->shape[1] : t->shape[0];
        const int64_t cp = ps.size() == 2 ? ps[1] : ps[0], rp = ps.size() == 2 ? ps[0] : 1;
        o.data.assign((size_t)(rp * cp), 0.f);
        for (int64_t i = 0; i < r; ++i) std::copy(t->data.begin() + i * c, t->data.begin() + (i + 1) * c, o.data.begin() + i * cp);
        padded.push_back(std::move(o));
        return &padded.back();
    };
    auto put_vec = [&](const HostTensor* t) { size_t o = bb.add(t->data.size()); std::copy(t->data.begin(), t->data.end(), bb.host.begin() + o); return o; };
    // [128 OB][128 KB] matrix -> OB*KB packed 128x128 blocks, block (ob, kb) at index ob*KB + kb
    auto put_blocks = [&](const HostTensor* t, int OB, int KB) {
        size_t o = bb.add((size_t)OB * KB * GAMD_WFRAG_FLOATS);
        for (int ob = 0; ob < OB; ++ob)
            for (int kb = 0; kb < KB; ++kb)
                pack128(t->data.data() + (size_t)128 * ob * 128 * KB + 128 * kb,
                        bb.host.data() + o + (size_t)(ob * KB + kb) * GAMD_WFRAG_FLOATS, 128 * KB);
        return o;
    };
    // [128 OB][128 KB] matrix -> OB*KB [hi | lo] fp16 images, block (ob, kb) at index ob*KB + kb (the order put_blocks uses)
    auto put_blocks_f16x3_fn = [&](const HostTensor* t, int OB, int KB) {
        size_t o = bb.add((size_t)OB * KB * GAMD_WFRAG_FLOATS);
        for (int ob = 0; ob < OB; ++ob)
            for (int kb = 0; kb < KB; ++kb)
                pack128_f16x3(t->data.data() + (size_t)128 * ob * 128 * KB + 128 * kb,
                              reinterpret_cast<uint16_t*>(bb.host.data() + o + (size_t)(ob * KB + kb) * GAMD_WFRAG_FLOATS), 128 * KB);
        return o;
    };
    // node-side matrices: the 128-wide node kernel (node.hip) runs on 16x16x4 MFMAs with its own operand order; the
    // generic-width node kernel of wide.hip keeps the 32x32x2 fragment blocks
    // Reduced-precision edge modes (bf16, split-fp16) on the 128-wide kernels: the node kernel's five GEMMs run in split-fp16 too
    // (fp32-grade results at 3/16 of the fp32 matrix time, node.hip and wide.hip's k_node_wide); GAMD_NODE_F32=1 keeps them on the fp32 pipe (A/B timing)
    h->node_f16 = h->cfg.edge_dtype != GAMD_EDGE_F32 && !getenv("GAMD_NODE_F32");
    auto put_node = [&](const HostTensor* t, int OB, int KB) {
        if (h->wide_conv) return h->node_f16 ? put_blocks_f16x3_fn(t, OB, KB) : put_blocks(t, OB, KB);
        size_t o = bb.add(GAMD_WFRAG_FLOATS);
        if (h->node_f16) pack16_f16x3(t->data.data(), reinterpret_cast<uint16_t*>(bb.host.data() + o));
        else pack16(t->data.data(), bb.host.data() + o);
        return o;
    };
    const bool bf16_edges = h->cfg.edge_dtype == GAMD_EDGE_BF16 && !h->wide_conv;     // 128 / 128 / 128 expanded: the specialised kernels
    const bool bf16_wide = h->cfg.edge_dtype == GAMD_EDGE_BF16 && h->wide_conv;       // any other width / feature set: wide_lp.hip
    const bool f16x3_edges = h->cfg.edge_dtype == GAMD_EDGE_F16X3 && !h->wide_conv;   // 128 / 128 / 128 expanded: the specialised kernels
    const bool f16x3_wide = h->cfg.edge_dtype == GAMD_EDGE_F16X3 && h->wide_conv;     // any other width / feature set: wide_lp.hip
    auto put_edge_f16x3 = [&](const HostTensor* t) {
        size_t o = bb.add(GAMD_WFRAG_FLOATS);
        pack128_f16x3(t->data.data(), reinterpret_cast<uint16_t*>(bb.host.data() + o));
        return o;
    };
    auto put_blocks_f16x3 = put_blocks_f16x3_fn;
    auto put_edge_bf16 = [&](const HostTensor* t) {
        size_t o = bb.add(GAMD_WFRAG_FLOATS / 2);
        pack128_bf16(t->data.data(), reinterpret_cast<uint16_t*>(bb.host.data() + o));
        return o;
    };

    // update_edge_emb=True checkpoints (WaterMDDynamicBoxNet(update_edge=True), nn_module.py:91-92): every conv layer owns an
    // edge_layer_norm; LayerNorm(in_edge_feats) is applied to e_emb of width in_node_feats (:141), so the two must agree.
    // Served by the generic-width kernels (wide.hip) for any width.
    const bool update_edge = h->host_w.count("graph_conv.conv.0.edge_layer_norm.weight") != 0;
    if (update_edge) {
        if (Ht != Et)
            return fail(-22, "update_edge_emb needs encoding_size == edge_embedding_dim (got %d, %d)", (int)Ht, (int)Et);
        if (h->cfg.edge_dtype != GAMD_EDGE_F32 || h->cfg.self_loop_mode)
            return fail(-22, "update_edge_emb is built for the fp32 edge MLP without appended self loops");
        h->wide_enc = h->wide_conv = true;
    }
    if (update_edge != h->update_edge) {
        h->update_edge = update_edge;
        if (alloc_edges(h, h->e_cap)) return -12;
    }
    // BatchNorm checkpoints carry running statistics next to norm_layers' weight and bias
    const bool norm_bn = h->host_w.count("graph_conv.norm_layers.0.running_mean") != 0;
    h->norm_bn = norm_bn ? 1 : 0;
    for (int l = 0; l < L; ++l) {
        const std::string p = "graph_conv.conv." + std::to_string(l);
        // edge_affine = MLP(Eh, hidden_dim, hidden_layer=2): its inner width is MLP's default 128 (nn_module.py:25,95)
        const HostTensor *ea0w = get(p + ".edge_affine.mlp_layer.0.weight", {128, Et}, {128, Eh}), *ea0b = get(p + ".edge_affine.mlp_layer.0.bias", {128}, {128});
        const HostTensor *ea2w = get(p + ".edge_affine.mlp_layer.2.weight", {Dt, 128}, {Dp, 128}), *ea2b = get(p + ".edge_affine.mlp_layer.2.bias", {Dt}, {Dp});
        const HostTensor *sw = get(p + ".src_affine.weight", {Dt, Ht}, {Dp, H}), *sb = get(p + ".src_affine.bias", {Dt}, {Dp});
        const HostTensor *dw = get(p + ".dst_affine.weight", {Dt, Ht}, {Dp, H}), *db = get(p + ".dst_affine.bias", {Dt}, {Dp});
        const HostTensor *t1w = get(p + ".theta_edge.mlp_layer.1.weight", {Dt, Dt}, {Dp, Dp}), *t1b = get(p + ".theta_edge.mlp_layer.1.bias", {Dt}, {Dp});
        const HostTensor *t3w = get(p + ".theta_edge.mlp_layer.3.weight", {Ht, Dt}, {H, Dp}), *t3b = get(p + ".theta_edge.mlp_layer.3.bias", {Ht}, {H});
        const HostTensor *pdw = get(p + ".phi_dst.weight", {Dt, Ht}, {Dp, H}), *pdb = get(p + ".phi_dst.bias", {Dt}, {Dp});
        const HostTensor *pew = get(p + ".phi_edge.weight", {Dt, Ht}, {Dp, H}), *peb = get(p + ".phi_edge.bias", {Dt}, {Dp});
        const HostTensor *phw = get(p + ".phi.mlp_layer.1.weight", {Ht, Dt}, {H, Dp}), *phb = get(p + ".phi.mlp_layer.1.bias", {Ht}, {H});
        const HostTensor *ng = get("graph_conv.norm_layers." + std::to_string(l) + ".weight", {Ht}, {H});
        const HostTensor *nb = get("graph_conv.norm_layers." + std::to_string(l) + ".bias", {Ht}, {H});
        if (!ea0w || !ea0b || !ea2w || !ea2b || !sw || !sb || !dw || !db || !t1w || !t1b || !t3w || !t3b || !pdw ||
            !pdb || !pew || !peb || !phw || !phb || !ng || !nb)
            return -2;
        if (norm_bn) {
            // use_layer_norm=False (the constructors' default, nn_module.py:171-196,579): norm_layers are nn.BatchNorm1d; the
            // rollout runs the model in eval mode, where it is the per-feature affine map torch's CPU kernel evaluates as
            // x * alpha + beta with alpha = weight / sqrt(running_var + eps), beta = bias - running_mean * alpha (float).
            // Folded here into the ln_g / ln_b slots; padded features keep alpha = beta = 0.
            const std::string np_ = "graph_conv.norm_layers." + std::to_string(l);
            const HostTensor *rm = get(np_ + ".running_mean", {Ht}, {H}), *rv = get(np_ + ".running_var", {Ht}, {H});
            if (!rm || !rv) return -2;
            HostTensor al, be;
            al.shape = be.shape = {H};
            al.data.assign((size_t)H, 0.f);
            be.data.assign((size_t)H, 0.f);
            for (int64_t i = 0; i < Ht; ++i) {
                const float invstd = 1.0f / std::sqrt(rv->data[i] + 1e-5f);
                al.data[i] = ng->data[i] * invstd;
                be.data[i] = nb->data[i] - rm->data[i] * al.data[i];
            }
            padded.push_back(std::move(al)); ng = &padded.back();
            padded.push_back(std::move(be)); nb = &padded.back();
        }
        // bf16 edge MLP (conv_edge_bf16.hip): the pre-activations of the layer's three SiLUs are computed times log2 e, so that
        // SiLU(x) log2 e = x' / (1 + 2^-x') needs no multiply in front of its exponential: W1, b1 (first SiLU), the S / D tables
        // (second: W2 sees the first SiLU's output times log2 e and needs no factor itself), b3 (third) carry log2 e, W4 ln 2
        if (bf16_edges) {
            auto scaled = [&](const HostTensor* t, double f) -> const HostTensor* {
                HostTensor o = *t;
                for (float& v : o.data) v = (float)((double)v * f);
                padded.push_back(std::move(o));
                return &padded.back();
            };
            const double LOG2E = 1.4426950408889634, LN2 = 0.6931471805599453;
            ea0w = scaled(ea0w, LOG2E); ea0b = scaled(ea0b, LOG2E); t1b = scaled(t1b, LOG2E); t3w = scaled(t3w, LN2);
            sw = scaled(sw, LOG2E); dw = scaled(dw, LOG2E);
            sb = scaled(sb, LOG2E); db = scaled(db, LOG2E); ea2b = scaled(ea2b, LOG2E);      // bS = (b_src + b_dst) + b_edge_affine.2
        }
        Off& o = lo[l];
        HostTensor b4_perm;
        // 128-wide fp32 and bf16 kernels (conv_edge.hip, conv_edge_small.hip, conv_edge_bf16.hip): output row 32 q + s of the
        // packed W4 is feature 4 s + q, so a lane of the last GEMM's F2 output holds four CONSECUTIVE features of each edge —
        // hn[src] is gathered and the pieces are stored 16 bytes at a time.  b4 is stored in the same order.
        HostTensor w4_perm;
        auto permute_w4 = [&]() {
            w4_perm = *t3w;
            b4_perm = *t3b;
            for (int q = 0; q < 4; ++q)
                for (int s = 0; s < 32; ++s) {
                    std::copy(t3w->data.begin() + (size_t)(4 * s + q) * 128, t3w->data.begin() + (size_t)(4 * s + q + 1) * 128,
                              w4_perm.data.begin() + (size_t)(32 * q + s) * 128);
                    b4_perm.data[32 * q + s] = t3b->data[4 * s + q];
                }
            t3b = &b4_perm;
        };
        if (bf16_edges) {
            permute_w4();
            o.w1p = put_edge_bf16(ea0w); o.w2p = put_edge_bf16(ea2w); o.w3p = put_edge_bf16(t1w); o.w4p = put_edge_bf16(&w4_perm);
        } else if (f16x3_edges) {
            o.w1p = put_edge_f16x3(ea0w); o.w2p = put_edge_f16x3(ea2w); o.w3p = put_edge_f16x3(t1w); o.w4p = put_edge_f16x3(t3w);
        } else if (bf16_wide) {
            // one contiguous run of 32 KiB bf16 images: W1[:, kb] (EHT) | W2 | W3 | W4[ob, :] (HT)
            auto put_blocks_bf16 = [&](const HostTensor* t, int OB, int KB) {
                size_t o = bb.add((size_t)OB * KB * (GAMD_WFRAG_FLOATS / 2));
                for (int ob = 0; ob < OB; ++ob)
                    for (int kb = 0; kb < KB; ++kb)
                        pack128_bf16(t->data.data() + (size_t)128 * ob * 128 * KB + 128 * kb,
                                     reinterpret_cast<uint16_t*>(bb.host.data() + o + (size_t)(ob * KB + kb) * (GAMD_WFRAG_FLOATS / 2)), 128 * KB);
                return o;
            };
            o.w1p = put_blocks_bf16(ea0w, 1, (int)EHT);
            o.w2p = put_blocks_bf16(ea2w, 1, 1);
            o.w3p = put_blocks_bf16(t1w, 1, 1);
            o.w4p = put_blocks_bf16(t3w, (int)HT, 1);
        } else if (f16x3_wide) {
            // one contiguous run of [hi | lo] images: W1[:, kb] (EHT) | W2 | W3 | W4[ob, :] (HT)
            o.w1p = put_blocks_f16x3(ea0w, 1, (int)EHT);
            o.w2p = put_blocks_f16x3(ea2w, 1, 1);
            o.w3p = put_blocks_f16x3(t1w, 1, 1);
            o.w4p = put_blocks_f16x3(t3w, (int)HT, 1);
        } else {
            // one contiguous run of blocks: W1[:, kb] (EHT) | W2 | W3 | W4[ob, :] (HT) -- the order the kernels stream them
            // (hidden_dim above 128, wide_d.hip: W1[:, kb] (EHT) | W2[db, :] (DT) | W3[ob][db] (DT x DT) | W4[ob][db] (HT x DT))
            o.w1p = put_blocks(ea0w, 1, (int)EHT);
            o.w2p = put_blocks(ea2w, (int)DT, 1);
            o.w3p = put_blocks(t1w, (int)DT, (int)DT);
            if (DT > 1) {
                o.w4p = put_blocks(t3w, (int)HT, (int)DT);
            } else if (h->wide_conv) {
                o.w4p = put_blocks(t3w, (int)HT, 1);
                // the same run of blocks for the opt-in 16-edge kernel (wide16.hip): W1[:, kb] | W2 | W3 chained, W4[ob, :] not
                o.w16p = bb.add((size_t)(EHT + 2 + HT) * GAMD_WFRAG_FLOATS);
                size_t at = o.w16p;
                for (int kb = 0; kb < (int)EHT; ++kb, at += GAMD_WFRAG_FLOATS) pack16x(ea0w->data.data() + 128 * kb, bb.host.data() + at, 128 * (int)EHT, true);
                pack16x(ea2w->data.data(), bb.host.data() + at, 128, true); at += GAMD_WFRAG_FLOATS;
                pack16x(t1w->data.data(), bb.host.data() + at, 128, true); at += GAMD_WFRAG_FLOATS;
                for (int ob = 0; ob < (int)HT; ++ob, at += GAMD_WFRAG_FLOATS) pack16x(t3w->data.data() + (size_t)128 * ob * 128, bb.host.data() + at, 128, false);
            } else {
                permute_w4();
                o.w4p = put_blocks(&w4_perm, 1, 1);
            }
        }
        o.b1 = put_vec(ea0b); o.b3 = put_vec(t1b); o.b4 = put_vec(t3b);
        o.lng = put_vec(ng); o.lnb = put_vec(nb);
        if (update_edge) {
            const HostTensor *eg = get(p + ".edge_layer_norm.weight", {Et}, {Eh}), *eb = get(p + ".edge_layer_norm.bias", {Et}, {Eh});
            if (!eg || !eb) return -2;
            o.elng = put_vec(eg); o.elnb = put_vec(eb);
        }
        o.wsp = put_node(sw, (int)DT, (int)HT); o.wdp = put_node(dw, (int)DT, (int)HT); o.wpdp = put_node(pdw, (int)DT, (int)HT);
        o.bS = bb.add((size_t)Dp);
        for (int i = 0; i < (int)Dp; ++i) bb.host[o.bS + i] = (sb->data[i] + db->data[i]) + ea2b->data[i];
        o.bP = bb.add((size_t)Dp);
        for (int i = 0; i < (int)Dp; ++i) bb.host[o.bP + i] = pdb->data[i] + peb->data[i];
        o.wpep = put_node(pew, (int)DT, (int)HT); o.wphip = put_node(phw, (int)HT, (int)DT); o.bphi = put_vec(phb);
    }
    const HostTensor *e0w = get("edge_encoder.mlp_layer.0.weight", {Dt, (int64_t)F}, {Dp, (int64_t)F}), *e0b = get("edge_encoder.mlp_layer.0.bias", {Dt}, {Dp});
    const HostTensor *e2w = get("edge_encoder.mlp_layer.2.weight", {Dt, Dt}, {Dp, Dp}), *e2b = get("edge_encoder.mlp_layer.2.bias", {Dt}, {Dp});
    const HostTensor *e4w = get("edge_encoder.mlp_layer.4.weight", {Et, Dt}, {Eh, Dp}), *e4b = get("edge_encoder.mlp_layer.4.bias", {Et}, {Eh});
    const HostTensor *elg = get("edge_layer_norm.weight", {Et}, {Eh}), *elb = get("edge_layer_norm.bias", {Et}, {Eh});
    const HostTensor *cen = expand ? get("edge_expand.centers", {40}, {40}) : nullptr;
    const HostTensor *lm = get("length_mean", {1}, {1}), *ls = get("length_std", {1}, {1});
    const HostTensor *d0w = get("graph_decoder.mlp_layer.0.weight", {Dt, Ht}, {Dp, H}), *d0b = get("graph_decoder.mlp_layer.0.bias", {Dt}, {Dp});
    const HostTensor *d2w = get("graph_decoder.mlp_layer.2.weight", {3, Dt}, {3, Dp}), *d2b = get("graph_decoder.mlp_layer.2.bias", {3}, {3});
    if (!e0w || !e0b || !e2w || !e2b || !e4w || !e4b || !elg || !elb || (expand && !cen) || !lm || !ls || !d0w || !d0b ||
        !d2w || !d2b)
        return -2;
    // (hidden_dim above 128: the DT 128-row images of the first Linear side by side in one 64 KiB slot, the streamed blocks of
    //  W2 and W3 right behind it — k_edge_encode_wide_d walks them as one run)
    const size_t o_e1 = bb.add(DT > 1 ? (size_t)GAMD_WFRAG_FLOATS : (size_t)(4 * 6 * 64 * 4));
    if (DT > 1) for (int64_t b = 0; b < DT; ++b) pack_enc1(e0w->data.data() + (size_t)(128 * b * F), F, bb.host.data() + o_e1 + (size_t)b * (4 * 6 * 64 * 4));
    else if (bf16_edges) pack_enc1_bf16(e0w->data.data(), F, reinterpret_cast<uint16_t*>(bb.host.data() + o_e1));
    else if (f16x3_edges) pack_enc1_f16x3(e0w->data.data(), F, reinterpret_cast<uint16_t*>(bb.host.data() + o_e1));
    else pack_enc1(e0w->data.data(), F, bb.host.data() + o_e1);
    // generic-width encoder in the reduced-precision modes: its two 128-wide GEMMs run in split-fp16 (wide.hip, e_format != 0)
    const bool enc_wide_f16 = h->wide_enc && h->cfg.edge_dtype != GAMD_EDGE_F32;
    const size_t o_e2 = bf16_edges ? put_edge_bf16(e2w) : f16x3_edges ? put_edge_f16x3(e2w) : enc_wide_f16 ? put_blocks_f16x3(e2w, 1, 1)
                                                                                                              : put_blocks(e2w, (int)DT, (int)DT);
    // fp32 path: the last encoder Linear is stored with its OUTPUT rows centred, W' = W - mean over rows, b' = b - mean(b)
    // (in double): y' = W' x + b' = y - mean(y) exactly in real arithmetic, so edge_layer_norm's mean subtraction
    // (nn_module.py:646) is done here once instead of per edge; the kernels only normalise the variance
    // (layernorm_chain_centered; wide.hip's generic LayerNorm sees a mean of ~0 and is unaffected).
    HostTensor e4w_c = *e4w, e4b_c = *e4b;
    if (!bf16_edges && !f16x3_edges) {
        // (the mean is over the Et true output rows; zero-padded rows stay zero)
        for (int64_t k = 0; k < Dp; ++k) {
            double m = 0.0;
            for (int64_t o = 0; o < Et; ++o) m += (double)e4w->data[(size_t)(o * Dp + k)];
            m /= (double)Et;
            for (int64_t o = 0; o < Et; ++o) e4w_c.data[(size_t)(o * Dp + k)] = (float)((double)e4w->data[(size_t)(o * Dp + k)] - m);
        }
        double mb = 0.0;
        for (int64_t o = 0; o < Et; ++o) mb += (double)e4b->data[(size_t)o];
        mb /= (double)Et;
        for (int64_t o = 0; o < Et; ++o) e4b_c.data[(size_t)o] = (float)((double)e4b->data[(size_t)o] - mb);
    }
    const size_t o_e3 = bf16_edges ? put_edge_bf16(e4w) : f16x3_edges ? put_edge_f16x3(e4w) : enc_wide_f16 ? put_blocks_f16x3(&e4w_c, (int)EHT, 1)
                                                                                                              : put_blocks(&e4w_c, (int)EHT, (int)DT);
    const size_t o_eb1 = put_vec(e0b), o_eb2 = put_vec(e2b), o_eb3 = put_vec(&e4b_c), o_elg = put_vec(elg), o_elb = put_vec(elb);
    const size_t o_cen = expand ? put_vec(cen) : bb.add(64);
    const size_t o_d1 = put_node(d0w, (int)DT, (int)HT), o_db1 = put_vec(d0b), o_d2 = put_vec(d2w), o_db2 = put_vec(d2b);
    size_t o_emb = 0, o_nw = 0, o_nb = 0;
    if (h->cfg.kind == GAMD_KIND_LJ) {
        const HostTensor* emb = get("node_emb", {1, Ht}, {1, H});
        if (!emb) return -2;
        o_emb = put_vec(emb);
    } else {
        const HostTensor *nw = get("node_encoder.weight", {Ht, 1}, {H, 1}), *nb = get("node_encoder.bias", {Ht}, {H});
        if (!nw || !nb) return -2;
        o_nw = put_vec(nw); o_nb = put_vec(nb);
    }
    h->length_mean = lm->data[0];
    h->length_std = ls->data[0];
    h->rbf = RbfGrid{};
    if (expand) {
        // RBFExpansion builds linspace(low, high, n) (nn_module.py:237-239): uniform up to fp32 rounding.  Anything else
        // (hand-edited centres) keeps the exact per-centre form.
        const double c0 = cen->data[0], delta = ((double)cen->data[39] - c0) / 39.0;
        bool uniform = delta > 0.0;
        for (int k = 0; k < 40 && uniform; ++k)
            uniform = std::fabs((double)cen->data[k] - (c0 + k * delta)) <= 2.5e-7 * std::max(1.0, std::fabs(c0 + k * delta));
        if (uniform) {
            const double gexp = -(1.0 / 0.025) * 1.4426950408889634, s = 2.0 * delta;     // -gamma log2(e), chain step
            h->rbf = RbfGrid{1, (float)c0, (float)delta, (float)(-2.0 * gexp * s), (float)(gexp * s * s),
                             (float)std::exp2(2.0 * gexp * s * s)};
        }
    }

    if (h->wblob.ensure(sizeof(float) * bb.host.size(), false)) return fail(-12, "weight blob allocation failed");
    HIP_TRY(init_upload(h->wblob.p, bb.host.data(), sizeof(float) * bb.host.size()));
    const float* B = h->wblob.as<float>();
    h->layers.assign(L, LayerDev{});
    for (int l = 0; l < L; ++l) {
        const Off& o = lo[l];
        LayerDev& d = h->layers[l];
        d.w1p = B + o.w1p; d.w2p = B + o.w2p; d.w3p = B + o.w3p; d.w4p = B + o.w4p;
        d.w16p = (h->wide_conv && h->cfg.edge_dtype == GAMD_EDGE_F32 && DT == 1) ? B + o.w16p : nullptr;
        d.b1 = B + o.b1; d.b3 = B + o.b3; d.b4 = B + o.b4;
        d.node.ln_g = B + o.lng; d.node.ln_b = B + o.lnb;
        if (update_edge) { d.e_ln_g = B + o.elng; d.e_ln_b = B + o.elnb; }
        d.node.wsp = B + o.wsp; d.node.wdp = B + o.wdp; d.node.wpdp = B + o.wpdp;
        d.node.bS = B + o.bS; d.node.bP = B + o.bP;
        d.node.wpep = B + o.wpep; d.node.wphip = B + o.wphip; d.node.bphi = B + o.bphi;
    }
    h->enc_w1p = B + o_e1; h->enc_w2p = B + o_e2; h->enc_w3p = B + o_e3;
    h->enc_b1 = B + o_eb1; h->enc_b2 = B + o_eb2; h->enc_b3 = B + o_eb3;
    h->enc_lng = B + o_elg; h->enc_lnb = B + o_elb; h->centers = B + o_cen;
    h->dec_w1p = B + o_d1; h->dec_b1 = B + o_db1; h->dec_w2 = B + o_d2; h->dec_b2 = B + o_db2;
    h->node_emb = h->cfg.kind == GAMD_KIND_LJ ? B + o_emb : nullptr;
    h->nenc_w = h->cfg.kind == GAMD_KIND_LJ ? nullptr : B + o_nw;
    h->nenc_b = h->cfg.kind == GAMD_KIND_LJ ? nullptr : B + o_nb;
    h->finalized = true;
    return 0;
}

int32_t gamd_set_scaler(gamd_handle* h, double mean, double var) {
    if (!h) return fail(-22, "null handle");
    if (!(var >= 0.0)) return fail(-22, "scaler variance must be non-negative");
    h->scaler_mean = mean;
    h->scaler_var = var;
    return 0;
}

int32_t gamd_set_bonds(gamd_handle* h, const int32_t* bonds, int64_t n_bonds) {
    if (!h || (!bonds && n_bonds > 0)) return fail(-22, "bad argument to gamd_set_bonds");
    DeviceGuard guard(h->dev);
    InitStream init(h->init_stream);
    std::vector<int> tab((size_t)h->n * 4, -1);
    auto add = [&](int i, int j) -> int {
        for (int k = 0; k < 4; ++k) {
            if (tab[(size_t)i * 4 + k] == j) return 0;
            if (tab[(size_t)i * 4 + k] < 0) { tab[(size_t)i * 4 + k] = j; return 0; }
        }
        return -1;
    };
    // several boxes: the bond list names atoms of ONE box (every box has the same topology) and is applied to each of them
    for (int box = 0; box < h->n_boxes; ++box)
        for (int64_t b = 0; b < n_bonds; ++b) {
            int i = bonds[2 * b], j = bonds[2 * b + 1];
            if (i < 0 || j < 0 || i >= h->n_per_box || j >= h->n_per_box)
                return fail(-22, "bond %lld references atom out of range", (long long)b);
            i += box * h->n_per_box; j += box * h->n_per_box;
            if (add(i, j) || add(j, i)) return fail(-22, "more than 4 bonded partners for one atom is not supported");
        }
    if (h->bond_nbr.ensure(sizeof(int) * tab.size(), false)) return fail(-12, "bond table allocation failed");
    HIP_TRY(init_upload(h->bond_nbr.p, tab.data(), sizeof(int) * tab.size()));
    h->has_bonds = n_bonds > 0;
    return 0;
}

int32_t gamd_set_node_features(gamd_handle* h, const float* feat_dev) {
    if (!h) return fail(-22, "null handle");
    if (feat_dev && h->cfg.kind != GAMD_KIND_WATER) return fail(-22, "node features belong to the water models (the LJ model has node_emb)");
    h->feat_dev = feat_dev;
    return 0;
}

int32_t gamd_get_device_flags(gamd_handle* h, int32_t flags[4]) {
    if (!h || !flags) return fail(-22, "null argument");
    flags[0] = h->sticky_host[STICKY_NONFINITE] ? 1 : 0;
    flags[1] = flags[2] = flags[3] = 0;
    h->sticky_host[STICKY_NONFINITE] = 0;
    return 0;
}

int32_t gamd_build_neighbors(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!pos_dev || !box) return fail(-22, "null argument");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if ((r = set_box(h, box, st))) return r;
        h->cur_counters = h->counters.as<int>();
        NbrArgs na = nbr_args(h, pos_dev, species_dev);
        h->cand_valid = false;                                    // the exact build below reorders the atoms
        if ((r = launch_neighbor_build(na, st))) return fail(-1, "neighbor build launch failed (%d)", r);
        HIP_TRY(hipMemcpyAsync(h->counters_host, h->counters.p, sizeof(int) * CNT_COUNT, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h->sticky_host[STICKY_EDGE_OVERFLOW] = 0;
        if (int t = check_traps(h)) return t;
        if (!h->counters_host[CNT_OVERFLOW]) return attempt ? 1 : 0;
        const long long need = (long long)(1.25 * (double)h->counters_host[CNT_E]) + 1024;
        if ((r = alloc_edges(h, need))) return r;
        if ((r = clear_devflags(h))) return r;
    }
    return fail(-34, "neighbor buffers still overflow after regrowing");
}

int32_t gamd_forces_async(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                          float* out_norm_dev, float* out_denorm_dev, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!pos_dev || !box) return fail(-22, "null argument");
    if ((r = check_model_inputs(h, species_dev))) return r;
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    if ((r = set_box(h, box, (hipStream_t)stream))) return r;
    h->pending.active = false;
    return enqueue_forward(h, pos_dev, species_dev, out_norm_dev, out_denorm_dev, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

int32_t gamd_sync_status(gamd_handle* h, void* stream) {
    if (!h) return fail(-22, "null handle");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    bool resumed = false;
    for (int attempt = 0; attempt < 5; ++attempt) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        // the sticky flags catch an overflow in ANY step enqueued since the last check (gamd_md_run), not only the last
        const bool cand_ovf = h->sticky_host[STICKY_CAND_OVERFLOW] != 0;
        const bool edge_ovf = h->counters_host[CNT_OVERFLOW] || h->sticky_host[STICKY_EDGE_OVERFLOW];
        if (int t = check_traps(h)) { h->pending.active = false; return t; }
        if (!cand_ovf && !edge_ovf) {
            h->pending.active = false;
            if (h->cfg.edge_dtype != GAMD_EDGE_F32 && h->sticky_host[STICKY_NONFINITE])
                return fail(-33, "non-finite forces with a reduced-precision edge dtype: an MFMA operand left the fp16 range "
                                 "(|x| > 65504) or the input positions are not finite");
            return resumed ? 1 : 0;
        }
        int r;
        long long need = 0;
        if (cand_ovf) {
            const long long seen = std::max<long long>(h->sticky_host[STICKY_NCAND], h->cand_cap);
            need = (long long)(1.25 * (double)seen) + 1024;
            if ((r = alloc_candidates(h, need))) return r;
        }
        if (edge_ovf) {
            const long long seen = std::max<long long>(h->counters_host[CNT_E], h->e_cap);
            need = (long long)(1.25 * (double)seen) + 1024;
            if ((r = alloc_edges(h, need))) return r;
            h->cand_valid = false;
        }
        h->sticky_host[STICKY_CAND_OVERFLOW] = 0;
        h->sticky_host[STICKY_EDGE_OVERFLOW] = 0;
        h->counters_host[CNT_OVERFLOW] = 0;
        int flags[DEVFLAG_COUNT] = {0, -1, 0, 0};
        HIP_TRY(hipMemcpyAsync(flags, h->devflags.p, sizeof(flags), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if ((r = clear_devflags(h))) return r;
        if (!h->pending.active)
            return fail(-34, "%s neighbour buffers overflowed; regrown to %lld, re-issue the call", cand_ovf ? "candidate" : "edge", need);
        // an enqueued MD run froze at (step, half): forces at the frozen positions, the rest of that step, the remaining steps
        const int at = flags[DEVFLAG_FROZEN_AT];
        if (at < 0) {
            h->pending.active = false;
            return fail(-34, "neighbour buffers overflowed inside an MD run but no integrator kernel recorded where it stopped");
        }
        if ((r = enqueue_md_steps(h, at / 2, (at & 1) != 0))) return r;
        resumed = true;
    }
    h->pending.active = false;
    return fail(-34, "neighbour buffers still overflow after regrowing four times");
}

int32_t gamd_forces(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                    float* out_norm_dev, float* out_denorm_dev, void* stream) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        int r = gamd_forces_async(h, pos_dev, species_dev, box, out_norm_dev, out_denorm_dev, stream);
        if (r) return r;
        r = gamd_sync_status(h, stream);
        if (r == 0) return attempt ? 1 : 0;
        if (r != -34) return r;
    }
    return fail(-34, "neighbour buffers still overflow after regrowing");
}

// The reference's host-array boundary (predict_forces, LJ/train_network_lj.py:133-157) in one call with one synchronisation
int32_t gamd_forces_host(gamd_handle* h, const float* pos_host, const uint8_t* species_dev, const float* box, float* out_host,
                         int32_t denormalize, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!pos_host || !out_host || !box) return fail(-22, "null argument");
    DeviceGuard guard(h->dev);
    const size_t bytes = sizeof(float) * 3 * (size_t)h->n;
    if (!h->host_in) {
        InitStream init((hipStream_t)stream);
        if (hipHostMalloc((void**)&h->host_in, bytes) != hipSuccess) { h->host_in = nullptr; return fail(-12, "pinned allocation failed"); }
        if (hipHostMalloc((void**)&h->host_out, bytes) != hipSuccess) {
            (void)hipHostFree(h->host_in);
            h->host_in = h->host_out = nullptr;
            return fail(-12, "pinned allocation failed");
        }
        if (h->pos_in.ensure(bytes, false)) return fail(-12, "device allocation failed");
    }
    std::memcpy(h->host_in, pos_host, bytes);
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIP_TRY(hipMemcpyAsync(h->pos_in.p, h->host_in, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        if ((r = gamd_forces_async(h, h->pos_in.as<float>(), species_dev, box, nullptr, denormalize ? h->f_den.as<float>() : nullptr, stream)))
            return r;
        HIP_TRY(hipMemcpyAsync(h->host_out, denormalize ? h->f_den.p : h->f_norm.p, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        r = gamd_sync_status(h, stream);                    // the one synchronisation; -34: regrown, replay
        if (r == 0) {
            std::memcpy(out_host, h->host_out, bytes);
            return attempt ? 1 : 0;
        }
        if (r != -34) return r;
    }
    return fail(-34, "neighbour buffers still overflow after regrowing");
}

int32_t gamd_forces_edges(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                          const int32_t* centre_dev, const int32_t* neigh_dev, int64_t n_edges, float* out_norm_dev,
                          float* out_denorm_dev, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!pos_dev || !box || n_edges < 0 || (n_edges > 0 && (!centre_dev || !neigh_dev))) return fail(-22, "bad argument");
    if ((r = check_model_inputs(h, species_dev))) return r;
    if (n_edges > 0x7fff0000ll) return fail(-22, "edge list too long");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    if ((r = set_box(h, box, (hipStream_t)stream))) return r;
    h->pending.active = false;
    int status = 0;
    // + one appended loop per atom, + at most 15 padding slots per box of a batch (neighbor.hip: d_box_align)
    const long long total = n_edges + (h->cfg.self_loop_mode ? h->n : 0) + (h->n_boxes > 1 ? 16ll * h->n_boxes : 0);
    if (total > h->e_cap) {                         // the count is known up front: grow before launching
        if ((r = alloc_edges(h, total + total / 8 + 1024))) return r;
        status = 1;
    }
    if (h->tmp_eid.ensure(sizeof(int) * ((size_t)std::max<long long>(total, 1) + 64), false))
        return fail(-12, "edge scratch allocation failed");
    EdgeList el{centre_dev, neigh_dev, (long long)n_edges};
    if ((r = enqueue_forward(h, pos_dev, species_dev, out_norm_dev, out_denorm_dev, (hipStream_t)stream, nullptr, nullptr,
                             nullptr, &el)))
        return r;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    h->sticky_host[STICKY_EDGE_OVERFLOW] = 0;
    if (h->counters_host[CNT_OVERFLOW]) {
        const int why = h->counters_host[CNT_OVERFLOW];
        h->counters_host[CNT_OVERFLOW] = 0;
        if ((r = clear_devflags(h))) return r;
        if (why == 2) return fail(-22, "edge list references an atom index outside [0, n_atoms)");
        return fail(-34, "edge buffers overflowed unexpectedly");
    }
    return status;
}

int32_t gamd_get_counts(gamd_handle* h, int64_t* n_edges, int64_t* n_pieces, int64_t* edge_capacity) {
    if (!h) return fail(-22, "null handle");
    if (n_edges) *n_edges = h->counters_host[CNT_E];
    if (n_pieces) *n_pieces = h->counters_host[CNT_PIECES];
    if (edge_capacity) *edge_capacity = h->e_cap;
    return 0;
}

int32_t gamd_get_skin_stats(gamd_handle* h, int64_t* n_rebuilds, int64_t* n_candidates, int64_t* candidate_capacity) {
    if (!h) return fail(-22, "null handle");
    if (n_rebuilds) *n_rebuilds = h->sticky_host[STICKY_REBUILDS];
    if (n_candidates) *n_candidates = h->sticky_host[STICKY_NCAND];
    if (candidate_capacity) *candidate_capacity = h->cand_cap;
    return 0;
}

int32_t gamd_debug_get(gamd_handle* h, int32_t what, void* host_out, size_t bytes) {
    if (!h || !host_out) return fail(-22, "null argument");
    DeviceGuard guard(h->dev);
    InitStream init(h->init_stream);
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = (size_t)h->n;
    const size_t E = (size_t)std::min<long long>(h->counters_host[CNT_E], h->e_cap);
    const void* src = nullptr;
    size_t avail = 0;
    if (what == GAMD_DBG_PERM) { src = h->perm.p; avail = sizeof(int) * n; }
    else if (what == GAMD_DBG_ROWPTR) { src = h->row_ptr.p; avail = sizeof(int) * (n + 1); }
    else if (what == GAMD_DBG_COL) { src = h->col.p; avail = sizeof(int) * E; }
    else if (what == GAMD_DBG_EFRAG) { src = h->e_frag.p; avail = sizeof(float) * 4096 * (size_t)h->EHT * ((E + 31) / 32); }
    else if (what == GAMD_DBG_FEAT) {
        if (!h->cfg.keep_stages) return fail(-22, "FEAT needs keep_stages=1");
        src = h->feat_dbg.p; avail = sizeof(float) * 48 * E;
    } else if (what >= GAMD_DBG_H0 && what <= GAMD_DBG_H0 + h->L) {
        if (!h->cfg.keep_stages) return fail(-22, "H_l needs keep_stages=1");
        src = h->hbuf.as<float>() + (size_t)(what - GAMD_DBG_H0) * n * (size_t)h->H; avail = sizeof(float) * n * (size_t)h->H;
    } else if (what == GAMD_DBG_PARTIAL) {
        src = h->partial.p; avail = sizeof(float) * (size_t)h->H * (size_t)std::max(0, h->counters_host[CNT_PIECES]);
    } else if (what == GAMD_DBG_CYCLES) { src = h->tdbg.p; avail = sizeof(long long) * 16 * 8 * (size_t)h->n_cu;
    } else return fail(-22, "unknown debug tensor %d", what);
    if (bytes < avail) return fail(-22, "host buffer too small: %zu < %zu", bytes, avail);
    HIP_TRY(hipMemcpy(host_out, src, avail, hipMemcpyDeviceToHost));
    return 0;
}

int32_t gamd_md_run(gamd_handle* h, float* x_dev, float* v_dev, float* f_dev, const uint8_t* species_dev,
                    const float* box, const gamd_md_params* p, int64_t n_steps, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!x_dev || !v_dev || !f_dev || !box || !p) return fail(-22, "null argument");
    if (n_steps < 0 || n_steps > 0x3fffffff) return fail(-22, "n_steps out of range");
    if ((r = check_model_inputs(h, species_dev))) return r;
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    if ((r = set_box(h, box, (hipStream_t)stream))) return r;
    hipStream_t st = (hipStream_t)stream;
    MdArgs m{};
    m.n = h->n; m.x = x_dev; m.v = v_dev; m.f = f_dev;
    if (!(p->mass_amu > 0.f)) return fail(-22, "mass_amu must be positive");
    m.species = species_dev;
    m.inv_mass = 1.0f / p->mass_amu;
    m.inv_mass_h = p->mass_h_amu > 0.f ? 1.0f / p->mass_h_amu : 0.f;
    m.len = p->length_per_nm > 0.f ? p->length_per_nm : 10.0f;
    m.dt = p->dt_ps;
    const double kB = 0.00831446261815324;                       // kJ/mol/K
    const double a = std::exp(-(double)p->gamma_per_ps * p->dt_ps);
    m.a = (float)a;
    m.b_len_kT = (float)(std::sqrt(1.0 - a * a) * (double)m.len * std::sqrt(kB * p->temperature_k));
    if ((r = fill_rigid(h, p->rigid_water, p->mass_amu, p->mass_h_amu, p->r_oh, p->r_hh, &m.use_rigid, &m.rigid)))
        return r;
    if (m.use_rigid && (r = check_rigid_layout(h, species_dev, st))) return r;
    for (int d = 0; d < 3; ++d) m.box[d] = box[d];
    m.bx = box_ref(h);
    if ((r = fill_com(h, p->remove_cm_motion, &m.com))) return r;
    m.seed = p->seed;
    m.devflags = h->devflags.as<int>();
    MdPending& pd = h->pending;
    pd.active = true; pd.kind = 0; pd.m = m; pd.first_step = p->first_step; pd.n_steps = n_steps;
    pd.x = x_dev; pd.f = f_dev; pd.species = species_dev; pd.st = st;
    return enqueue_md_steps(h, 0, false);
}

int32_t gamd_md_run_nhc(gamd_handle* h, float* x_dev, float* v_dev, float* f_dev, const uint8_t* species_dev,
                        const float* box, const gamd_nhc_params* p, double* chain_state_dev, int64_t n_steps, void* stream) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!x_dev || !v_dev || !f_dev || !box || !p || !chain_state_dev) return fail(-22, "null argument");
    if (n_steps < 0 || n_steps > 0x3fffffff) return fail(-22, "n_steps out of range");
    if (p->chain_length < 1 || p->chain_length > 16) return fail(-22, "chain_length must be in [1, 16]");
    if ((r = check_model_inputs(h, species_dev))) return r;
    static const double YS1[] = {1.0};
    static const double YS3[] = {0.8289815435887510, -0.6579630871775020, 0.8289815435887510};
    static const double YS5[] = {0.2967324292201065, 0.2967324292201065, -0.1869297168804260, 0.2967324292201065,
                                 0.2967324292201065};                       // hack_integrator.py:183-187
    const double* ys = p->num_yoshidasuzuki == 1 ? YS1 : p->num_yoshidasuzuki == 3 ? YS3 : p->num_yoshidasuzuki == 5 ? YS5 : nullptr;
    if (!ys) return fail(-22, "Invalid Yoshida-Suzuki value. Allowed values are: 1,3,5");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    if ((r = set_box(h, box, (hipStream_t)stream))) return r;
    hipStream_t st = (hipStream_t)stream;
    NhcArgs a{};
    a.n = h->n; a.x = x_dev; a.v = v_dev; a.f = f_dev;
    if (!(p->mass_amu > 0.f)) return fail(-22, "mass_amu must be positive");
    a.species = species_dev;
    a.mass = p->mass_amu; a.mass_h = p->mass_h_amu > 0.f ? p->mass_h_amu : 0.f;
    a.len = p->length_per_nm > 0.f ? p->length_per_nm : 10.0f;
    a.dt = p->dt_ps;
    if ((r = fill_rigid(h, p->rigid_water, p->mass_amu, p->mass_h_amu, p->r_oh, p->r_hh, &a.use_rigid, &a.rigid)))
        return r;
    if (a.use_rigid && (r = check_rigid_layout(h, species_dev, st))) return r;
    for (int d = 0; d < 3; ++d) a.box[d] = box[d];
    a.bx = box_ref(h);
    if ((r = fill_com(h, p->remove_cm_motion, &a.com))) return r;
    a.kT = 0.00831446261815324 * (double)p->temperature_k;
    a.freq = p->frequency_per_ps;
    a.ndf = p->ndf;
    a.M = p->chain_length; a.n_c = p->num_mts; a.n_ys = p->num_yoshidasuzuki;
    for (int i = 0; i < a.n_ys; ++i) a.w[i] = ys[i];
    a.state = chain_state_dev;
    a.n_blocks = std::min(256, (3 * h->n_per_box + 255) / 256);       // per box
    if (h->ke_partial.ensure(sizeof(double) * 256 * (size_t)h->n_boxes, true)) return fail(-12, "allocation failed");
    a.partial = h->ke_partial.as<double>();
    a.devflags = h->devflags.as<int>();
    if (p->reset) {
        const size_t stride = 3 * (size_t)a.M + 2;
        std::vector<double> init(stride * (size_t)h->n_boxes, 0.0);
        for (int b = 0; b < h->n_boxes; ++b)
            for (int i = 0; i < a.M; ++i) init[stride * b + 2 * a.M + i] = -a.freq * a.freq;      // G_i = -frequency^2 (:255)
        HIP_TRY(hipMemcpyAsync(chain_state_dev, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    MdPending& pd = h->pending;
    pd.active = true; pd.kind = 1; pd.a = a; pd.first_step = 0; pd.n_steps = n_steps;
    pd.x = x_dev; pd.f = f_dev; pd.species = species_dev; pd.st = st;
    return enqueue_md_steps(h, 0, false);
}

int32_t gamd_profile(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                     float* out_norm_dev, void* stream, char* names, size_t names_bytes, float* ms, int32_t max_ms,
                     int32_t* n_out) {
    int r;
    if ((r = check_ready(h))) return r;
    if (!pos_dev || !box || !names || !ms || !n_out) return fail(-22, "null argument");
    if ((r = check_model_inputs(h, species_dev))) return r;
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    if ((r = set_box(h, box, (hipStream_t)stream))) return r;
    h->pending.active = false;
    hipStream_t st = (hipStream_t)stream;
    const int max_ev = 64;
    hipEvent_t evs[max_ev];
    for (int i = 0; i < max_ev; ++i) HIP_TRY(hipEventCreate(&evs[i]));
    int n_ev = 0;
    std::vector<std::string> labels;
    r = enqueue_forward(h, pos_dev, species_dev, out_norm_dev, nullptr, st, evs, &n_ev, &labels);
    if (r == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        std::string all;
        int cnt = 0;
        for (int i = 1; i < n_ev && cnt < max_ms; ++i, ++cnt) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, evs[i - 1], evs[i]));
            ms[cnt] = t;
            all += labels[i];
            all += "\n";
        }
        *n_out = cnt;
        snprintf(names, names_bytes, "%s", all.c_str());
    }
    for (int i = 0; i < max_ev; ++i) (void)hipEventDestroy(evs[i]);
    return r;
}

int32_t gamd_timing_enable(gamd_handle* h, int32_t enable) {
    if (!h) return fail(-22, "null handle");
    DeviceGuard guard(h->dev);
    InitStream init(h->init_stream);
    h->timing = enable != 0;
    h->tev_used = 0;
    h->sev_used = 0;
    if (h->timing) {
        // a first pool of events HERE, not inside the region that is about to be timed (round-4 review: nothing but the
        // step's own launches belongs between the two synchronisation points); the pools still grow on demand
        while (h->tev.size() < 4096) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->tev.push_back(e); h->tev_kind.push_back(0); }
        while (h->sev.size() < 1024) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->sev.push_back(e); h->sev_closes.push_back(0); }
    }
    return 0;
}

int32_t gamd_timing_read_steps(gamd_handle* h, void* stream, float* step_ms, int64_t max_steps, int64_t* n_steps) {
    if (!h || !n_steps || (max_steps > 0 && !step_ms)) return fail(-22, "null argument");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    // events: one in front of every step, one behind the last step of each enqueue_md_steps call.  The interval that starts
    // at such a closing event ends at the first event of the NEXT run: the host's gap between two gamd_md_run calls (or the
    // regrow of a run that froze on a neighbour-buffer overflow), not a step, and is skipped.  A frozen-and-resumed run
    // contributes its frozen steps (cheap: their kernels return at once) and the resumed ones.
    int64_t n = 0;
    for (size_t i = 0; i + 1 < h->sev_used; ++i) {
        if (h->sev_closes[i]) continue;
        if (n < max_steps) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, h->sev[i], h->sev[i + 1]));
            step_ms[n] = t;
        }
        ++n;
    }
    *n_steps = n;
    return 0;
}

int32_t gamd_timing_read(gamd_handle* h, void* stream, double* total_ms, int64_t* n_launches) {
    if (!h || !total_ms || !n_launches) return fail(-22, "null argument");
    double ms[3];
    int64_t cnt[3];
    int r = gamd_timing_read_stages(h, stream, ms, cnt);
    if (r) return r;
    *total_ms = ms[0];
    *n_launches = cnt[0];
    return 0;
}

int32_t gamd_timing_read_stages(gamd_handle* h, void* stream, double total_ms[3], int64_t n[3]) {
    if (!h || !total_ms || !n) return fail(-22, "null argument");
    DeviceGuard guard(h->dev);
    InitStream init((hipStream_t)stream);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int k = 0; k < 3; ++k) { total_ms[k] = 0.0; n[k] = 0; }
    for (size_t i = 0; i + 1 < h->tev_used; i += 2) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, h->tev[i], h->tev[i + 1]));
        const int kind = h->tev_kind[i];
        const int slot = kind == 100 ? 1 : 0;
        total_ms[slot] += t;
        n[slot] += 1;
        // the node kernel between two conv layers of one force evaluation: stop of layer l -> start of layer l + 1
        if (kind != 100 && i + 3 < h->tev_used && h->tev_kind[i + 2] == kind + 1) {
            HIP_TRY(hipEventElapsedTime(&t, h->tev[i + 1], h->tev[i + 2]));
            total_ms[2] += t;
            n[2] += 1;
        }
    }
    return 0;
}

}  // extern "C"
