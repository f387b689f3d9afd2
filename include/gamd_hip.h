/* gamd_hip.h — C ABI of libgamd_hip.so, the MI355X (gfx950) implementation of GAMD's
 * force-inference hot path.
 *
 * The reference has no FFI layer: its hot path sits behind Python objects
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub); every entry point names the reference
 * interface it replaces (paths relative to /root/reference/code).
 *
 * Conventions
 *   - plain C types only; all `*_dev` pointers are caller-owned DEVICE pointers,
 *     all other pointers are HOST pointers; `stream` is a hipStream_t passed as void*
 *     (NULL = default stream).  Work is ordered on that stream only (it may be a non-blocking stream: nothing relies
 *     on the NULL stream's implicit ordering; what a call allocates and initialises has landed before it returns).
 *   - every function returns an int32 status: 0 = ok, 1 = ok after the neighbour
 *     buffers overflowed and were regrown (the analogue of jax-md's
 *     did_buffer_overflow -> re-allocate, graph_utils.py:41-42), < 0 = error
 *     (text via gamd_last_error).  No exceptions cross the ABI.
 *   - one handle per box (or per gamd_config.n_boxes independent boxes), any number of handles per GPU and per
 *     process.  A HANDLE is not thread-safe (one caller at a time; the reference is single-threaded too); DIFFERENT
 *     handles may be driven from different threads at the same time — the library keeps no mutable state outside the
 *     handle, and gamd_last_error is per thread.
 *   - atoms keep the CALLER's order at the boundary; internally they are renumbered
 *     in cell order every call.
 */
#ifndef GAMD_HIP_H
#define GAMD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gamd_handle gamd_handle;

enum { GAMD_EDGE_F32 = 0, GAMD_EDGE_BF16 = 1, GAMD_EDGE_F16X3 = 2 };

enum { GAMD_KIND_LJ = 0, GAMD_KIND_WATER = 1 };        /* SimpleMDNetNew | WaterMDNetNew / WaterMDDynamicBoxNet */
enum { GAMD_NBR_JAXMD = 0, GAMD_NBR_TORCH = 1 };        /* dr^2 < rc^2 + self pair | |dr| <= rc, no self */

/* Constructor arguments.  Replaces build_model() + NeighborSearcher(BOX_SIZE, cutoff):
 * LJ/train_network_lj.py:68-88,108-112; water/train_network_tip3p.py:75-97;
 * graph_utils.py:12-27.  Widths: whatever build_model passes (nn_module.py:561-601, --encoding_size / --hidden_dim /
 * --edge_embedding_dim, LJ/train_network_lj.py:394-396): encoding_size, edge_embedding_dim and hidden_dim in [1, 256]
 * (hidden_dim above 128: edge_dtype f32 only).  The kernels work in 128-wide blocks; other widths are zero-padded when the weights are packed (padded features
 * are exact zeros through every layer, the two LayerNorms divide by the true width).  128 / 128 / 128 (every shipped LJ / TIP
 * config, LJ/test_script/test_langevin.py:63-73) runs the specialised kernels, anything wider than 128 the generic-width
 * ones (the DFT-water config 256 / 128 / 256, water/test_script/test_nosehoover_hb.py:69-81; the trainers' defaults).
 * Normalisation between the conv layers (nn_module.py:171-196): LayerNorm (use_layer_norm=True, every rollout driver) or
 * eval-mode BatchNorm1d (use_layer_norm=False, the constructors' default) -- chosen by the weights: a state_dict that carries
 * graph_conv.norm_layers.<l>.running_mean / running_var is a BatchNorm checkpoint (num_batches_tracked is not needed).
 * update_edge=True models (SmoothConvLayerNew.update_edge_emb, nn_module.py:91-92, :140-146) are recognised the same way, by
 * their graph_conv.conv.<l>.edge_layer_norm.weight / .bias keys: fp32 edge MLP, encoding_size == edge_embedding_dim. */
typedef struct gamd_config {
    int32_t n_atoms;
    int32_t kind;            /* GAMD_KIND_*  */
    int32_t n_layers;        /* conv_layer (4 in build_model) */
    int32_t use_bond;        /* water: 45th edge feature from the bond graph (nn_module.py:450-454) */
    int32_t nbr_flavour;     /* GAMD_NBR_*  */
    int32_t device;          /* HIP device ordinal */
    float cutoff;            /* CUTOFF_RADIUS */
    float box[3];            /* initial box (may change per call) */
    int64_t edge_capacity;   /* 0 = estimate from density */
    int32_t keep_stages;     /* 1 = keep per-stage tensors for the debug getters */
    int32_t edge_dtype;      /* GAMD_EDGE_F32 (bit-exact fp32 MFMA, default) | GAMD_EDGE_BF16 (BASELINE config 5: edge-MLP
                                operands rounded to bf16, fp32 accumulate; S/D adds, SiLU, sums in fp32; the node side's GEMMs fp32-grade: split-fp16
                                as in F16X3)
                                | GAMD_EDGE_F16X3 (fp32-grade edge-MLP on the fp16 matrix pipe: every operand split into
                                hi + lo fp16, W x = Wh xh + (Wh xl + Wl xh), fp32 accumulate; meets F32's 1e-5 parity bar on every
                                shipped architecture and golden; 3-4 x the fp32 kernels' rounding error, so a deep model with
                                heavy cancellation can land just above it: opt-in, labelled with its own dtype).
                                Both exist for every width and feature set (at most 2^22 - 2 atoms per handle when
                                encoding_size > 128); fp32 only: self_loop_mode 1 and update_edge models */
    int32_t encoding_size;   /* node width H: 0 (= 128) or 1 .. 256 (build_model 'encoding_size') */
    int32_t edge_embedding_dim; /* edge-embedding width Eh: 0 (= 128) or 1 .. 256 ('edge_embedding_dim') */
    int32_t hidden_dim;      /* 0 (= 128) or 1 .. 256 ('hidden_dim'); above 128: GAMD_EDGE_F32 only (wide_d.hip) */
    int32_t no_expand_edge;  /* 1 = expand_edge=False: edge features are (unit vector, standardised length[, bond])
                                without the 40 RBFs (nn_module.py:329-336; --disable_expand_edge,
                                water/train_network_real_large.py:363) */
    float neighbor_skin;     /* 0 = exact cell-list rebuild every call (default).  > 0: Verlet-skin reuse like the reference's
                                jax-md list (graph_utils.py:21-25 dr_threshold = cutoff/6, :36-44 update): candidates
                                within cutoff + skin are rebuilt only when an atom has moved more than skin/2, the exact
                                cutoff is re-applied every call, so the edge SET is the same either way */
    int32_t self_loop_mode;  /* what `fluid_graph.add_self_loop()` with its result DISCARDED does (nn_module.py:650-652, :364,
                                :518).  0 = GAMD_SELF_LOOP_DGL07_NOOP (default): nothing — under the pinned DGL 0.7.0 (DGL >= 0.5)
                                add_self_loop is functional and returns a new graph, so the graph that is used has no extra
                                edges.  1 = GAMD_SELF_LOOP_APPEND_ZERO_FEATURE: what an in-place add_self_loop (DGL < 0.5) would
                                have done: one extra edge i -> i per atom whose embedding e is DGL's zero fill (it is appended
                                after edata['e'] was set).  The one reference semantic that cannot be executed in the build
                                container (DGL absent), hence the switch.  fp32 edge dtype only. */
    int32_t kernel_select;   /* 0 = automatic.  Bit flags for tests (never change results beyond fp32 rounding):
                                GAMD_KSEL_FORCE_GENERIC_WIDTH (1): run a 128/128 configuration on the generic-width kernels
                                of wide.hip; GAMD_KSEL_FORCE_HALF_QUANTUM (2): run the generic-width fp32 conv layer on 16-edge
                                work units (wide16.hip, v_mfma_f32_16x16x4_f32; bit-identical results, measured slower at every
                                size tried: never chosen automatically) */
    int32_t small_tile_limit;/* fp32 path: edge counts of at most this many 32-edge tiles run the latency-oriented conv kernel
                                (one tile per 4-wave workgroup, bit-identical results).  0 = default (512), -1 = never */
    int32_t n_boxes;         /* 0 or 1: one box (default).  B > 1: B INDEPENDENT boxes of n_atoms atoms each, evaluated and
                                integrated in one set of launches — the reference's several-graphs-per-forward
                                (build_graph_batches + dgl.batch, nn_module.py:655-661, :520-527, :676-679) and the "more
                                replicas than GPUs" half of the ensemble.  Every `[n][3]` / `[n]` device array of the entry
                                points below is then `[B][n][3]` / `[B][n]` (box-major, contiguous), every `box` argument is
                                HOST float [B][3] (a box per graph, as WaterMDDynamicBoxNet.forward's box_size_lst), bonds name
                                atoms of one box and apply to each, gamd_md_run's seed means seed + b for box b, and
                                gamd_md_run_nhc's chain_state_dev is [B][3*chain_length + 2] with ndf per box.  Atoms of
                                different boxes are never neighbours (the box index is folded into the cell index); results are
                                bit-identical to the boxes evaluated one by one (each box's CSR rows start on a 16-edge
                                boundary; gamd_get_counts / GAMD_DBG_COL include those <= 15 padding slots per box, source
                                index B*n).  A neighbour-buffer overflow in any box regrows the shared buffers. */
} gamd_config;
enum { GAMD_SELF_LOOP_DGL07_NOOP = 0, GAMD_SELF_LOOP_APPEND_ZERO_FEATURE = 1 };
enum { GAMD_KSEL_FORCE_GENERIC_WIDTH = 1, GAMD_KSEL_FORCE_HALF_QUANTUM = 2 };

const char* gamd_version(void);
const char* gamd_last_error(void);

int32_t gamd_create(const gamd_config* cfg, gamd_handle** out);
int32_t gamd_destroy(gamd_handle* h);

/* Weights contract = the reference state_dict: call once per key with the
 * reference's key name (e.g. "graph_conv.conv.0.src_affine.weight"), fp32 host data, row-major
 * torch shape; then gamd_finalize_weights packs them into MFMA fragment order on the device.
 * Replaces model.load_state_dict / load_from_checkpoint (LJ/train_network_lj.py:85-87,
 * LJ/test_script/test_langevin.py:74). */
int32_t gamd_load_weight(gamd_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
int32_t gamd_finalize_weights(gamd_handle* h);

/* scaler.npz mean/var.  Replaces load_training_stats (LJ/train_network_lj.py:119-123). */
int32_t gamd_set_scaler(gamd_handle* h, double mean, double var);

/* Bond list [n_bonds][2] (both directions are implied).  Replaces build_bond_graph
 * (nn_module.py:529-534) fed by create_water_bond (water/train_network_tip3p.py:38-42). */
int32_t gamd_set_bonds(gamd_handle* h, const int32_t* bonds, int64_t n_bonds);

/* Enqueue neighbour build + full network forward for positions `pos_dev` [n][3] fp32 (any periodic
 * image), optional species [n] (u8: O=1/H=0 -> node feature, water/test_script/test_nosehoover.py:82-89),
 * box[3].  Writes the NORMALISED network output [n][3] fp32 to out_norm_dev (what
 * pnet_model([pos],[edge_idx]) returns, nn_module.py:672-685 / :545-558) and, if out_denorm_dev is not NULL,
 * out*sqrt(var)+mean in fp32 (device-side copy of denormalize(), train_network_lj.py:128-131).
 * Replaces search_for_neighbor + get_edge_idx + model.forward (LJ/train_network_lj.py:135-147,166-199).
 * Does not synchronise; call gamd_sync_status afterwards. */
int32_t gamd_forces_async(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                          float* out_norm_dev, float* out_denorm_dev, void* stream);

/* Float node features for the water models: feat_dev [n] fp32 (device, caller-owned, must stay valid for the calls
 * that follow) is what the reference feeds to node_encoder = Linear(1 -> H) as `x` (nn_module.py:554, :403); NULL
 * (default) = use (float)species, i.e. the O = 1 / H = 0 flag the drivers build (water/test_script/test_nosehoover.py:82-89).
 * species_dev stays the integer type the integrators pick masses by. */
int32_t gamd_set_node_features(gamd_handle* h, const float* feat_dev);

/* Wait for the stream and report what the enqueued work ran into:
 *   0    nothing.
 *   1    a neighbour buffer overflowed inside an enqueued gamd_md_run / gamd_md_run_nhc: the device froze positions,
 *        velocities and thermostat chain at the last consistent point (integrator and node kernels return while the
 *        overflow flag is set), this call regrew the buffers, re-evaluated the forces there and finished the remaining
 *        steps: the trajectory is the one an ample buffer would have produced.
 *   -34  a neighbour buffer overflowed in gamd_forces_async: regrown, the caller must re-issue that call (gamd_forces
 *        does it for you).
 *   -33  non-finite forces in a reduced-precision edge dtype (bf16 / f16x3): an MFMA operand left the fp16 range
 *        (|x| > 65504) or the input was not finite.  (The fp32 path returns non-finite forces silently, like the
 *        reference; gamd_get_device_flags tells.) */
int32_t gamd_sync_status(gamd_handle* h, void* stream);

/* flags[0]: 1 if any force evaluation since the last call of this function produced a non-finite force component
 * (then cleared); flags[1..3] reserved (0). */
int32_t gamd_get_device_flags(gamd_handle* h, int32_t flags[4]);

/* gamd_forces_async + gamd_sync_status + automatic regrow-and-retry; returns 0 or 1. */
int32_t gamd_forces(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                    float* out_norm_dev, float* out_denorm_dev, void* stream);

/* The reference's host-array boundary in one call: predict_forces(pos) takes and returns HOST arrays
 * (LJ/train_network_lj.py:133-157, water/train_network_tip3p.py:142-159).  pos_host: float32 [n][3], wrapped and rounded as
 * :141-142 do (np.mod in float64, then float32), any host memory; out_host: float32 [n][3] — the normalised network output
 * (denormalize = 0: what pnet_model returns at :152; the caller denormalises in float64 as :155) or out * sqrt(var) + mean in
 * fp32 (denormalize = 1).  The positions go through a pinned staging buffer of the handle; the copy in, the kernels and the
 * copy out are enqueued on `stream`, the call synchronises ONCE and replays itself after a regrow.  species_dev / box as in
 * gamd_forces.  Returns 0, or 1 if a neighbour buffer was regrown.  PCIe-inclusive: never what bench.py reports as `value`. */
int32_t gamd_forces_host(gamd_handle* h, const float* pos_host, const uint8_t* species_dev, const float* box, float* out_host,
                         int32_t denormalize, void* stream);

/* Same network forward on a CALLER-SUPPLIED directed edge list instead of the built-in radius search:
 * centre_dev[e] / neigh_dev[e] (int32, device) = rows 0 / 1 of the reference's edge_idx tensor; messages flow
 * neigh -> centre.  Replaces the model-level call pnet_model([pos], [edge_idx]) / ([pos], feat, [edge_idx])
 * (nn_module.py:672-685, :545-558, build_graph :636-653).  Atoms are NOT renumbered on this path; every row
 * keeps the caller's edge order.  Synchronises; returns 0, or 1 if the edge buffers had to grow. */
int32_t gamd_forces_edges(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                          const int32_t* centre_dev, const int32_t* neigh_dev, int64_t n_edges,
                          float* out_norm_dev, float* out_denorm_dev, void* stream);

/* Neighbour build only (stage entry point for parity tests / profiling). */
int32_t gamd_build_neighbors(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                             void* stream);

/* n_edges = directed edge count of the last build (incl. self edges in the jax-md flavour). */
int32_t gamd_get_counts(gamd_handle* h, int64_t* n_edges, int64_t* n_pieces, int64_t* edge_capacity);

/* Verlet-skin bookkeeping: candidate-list rebuilds so far, size of the candidate list in use, its capacity.  The analogue of watching nbrs.did_buffer_overflow / re-allocation in graph_utils.py:36-44. */
int32_t gamd_get_skin_stats(gamd_handle* h, int64_t* n_rebuilds, int64_t* n_candidates, int64_t* candidate_capacity);

/* Debug / parity getters: copy a stage tensor of the last call to HOST memory (synchronises).
 * Need keep_stages = 1 for H / FEAT. */
enum {
    GAMD_DBG_PERM = 0,      /* int32 [n]      sorted -> original atom id */
    GAMD_DBG_ROWPTR = 1,    /* int32 [n+1]    CSR by destination, sorted ids */
    GAMD_DBG_COL = 2,       /* int32 [E]      source atom (sorted id) per CSR edge */
    GAMD_DBG_EFRAG = 3,     /* fp32  [ceil(E/32)][Eh/128][4][4][64][4]  e in fragment order */
    GAMD_DBG_FEAT = 4,      /* fp32  [E][48]  raw edge features (first 44|45, or 4|5 unexpanded, columns valid) */
    GAMD_DBG_CYCLES = 5,    /* int64 [n_cu][8][16]  per-wave cycle sums of the instrumented conv-edge kernel (profiling build) */
    GAMD_DBG_PARTIAL = 6,   /* fp32  [pieces][H]  the LAST conv layer's partial-sum pieces (one row per run of edges with the same
                               destination inside a 16-edge chunk), in CSR order */
    GAMD_DBG_H0 = 16        /* fp32  [n][H] residual stream h_l, sorted order: GAMD_DBG_H0 + l */
};
int32_t gamd_debug_get(gamd_handle* h, int32_t what, void* host_out, size_t bytes);

/* Split BAOAB Langevin step of the reference drivers, on device:
 *   first half  B A O A   HackLangevinIntegrator     hack_integrator.py:141-165
 *   force eval            predict_forces             LJ/test_script/test_langevin.py:108
 *   second half B         HackHalfVelocityIntegrator hack_integrator.py:175-178
 * x [n][3] Angstrom, v [n][3] Angstrom/ps, f [n][3] kJ/mol/nm (denormalised; in: forces at x, out: forces
 * at the new x).  Enqueues n_steps steps without synchronising; gamd_sync_status afterwards reports 0, or 1 when a
 * neighbour buffer overflowed on the way (state frozen on the device, buffers regrown, run resumed and finished). */
typedef struct gamd_md_params {
    float dt_ps;             /* 0.002 in the drivers */
    float mass_amu;          /* 39.9 for argon */
    float temperature_k;     /* 100 */
    float gamma_per_ps;      /* 25 */
    uint64_t seed;
    uint64_t first_step;     /* RNG counter of the first step */
    /* zero-initialised = the LJ behaviour (one mass, Angstrom, no constraints) */
    float mass_h_amu;        /* > 0 and species given: mass of the species-0 atoms (H, 1.008); mass_amu is then the O mass */
    float length_per_nm;     /* length unit of x, v, box per nm: 0 or 10 = Angstrom; 18.8972613 = bohr (DFT model:
                                positions in bohr, water/test_script/test_nosehoover_hb.py:106-109) */
    int32_t rigid_water;     /* 1 = atoms are O,H,H triples held rigid, as OpenMM does for the constrained water systems
                                of the water drivers at every addConstrainPositions / addConstrainVelocities of
                                hack_integrator.py:145-164,178,277-280,427-428 (SETTLE + analytic velocity constraint) */
    float r_oh, r_hh;        /* constraint lengths in the length unit (TIP3P: 0.9572, 1.5139 A) */
    int32_t remove_cm_motion;/* 1 = subtract the centre-of-mass velocity (sum m v / sum m, per box) at the top of every step, as
                                OpenMM's CMMotionRemover does through addUpdateContextState() (hack_integrator.py:142) when the
                                System carries one: the water drivers' openmmtools WaterBox does, the LJ fluid does not.  GNN
                                forces do not sum to zero, so without it the centre of mass random-walks in long rollouts */
} gamd_md_params;
int32_t gamd_md_run(gamd_handle* h, float* x_dev, float* v_dev, float* f_dev, const uint8_t* species_dev,
                    const float* box, const gamd_md_params* p, int64_t n_steps, void* stream);

/* Split Nose-Hoover-chain step of the reference drivers, on device:
 *   first half   propagateNHC; v += dt/2 f_last/m; x += dt v     HackNoseHooverIntegrator      hack_integrator.py:182-330
 *   force eval                                                    predict_forces                LJ/test_script/test_nosehoover.py:113
 *   second half  v += dt/2 f_gnn/m; propagateNHC                  HackHalfNoseHooverIntegrator  hack_integrator.py:334-493
 * One chain state (xi, vxi, G) is shared by both halves (the drivers copy it across every step,
 * test_nosehoover.py:104-118).  chain_state_dev: double [3*chain_length + 2] device buffer owned by the caller
 * (xi[M], vxi[M], G[M], last scale, last 2*KE); pass reset != 0 to initialise it (xi = vxi = 0, G = -freq^2). */
typedef struct gamd_nhc_params {
    float dt_ps;              /* 0.002 */
    float mass_amu;           /* 39.9 */
    float temperature_k;      /* 100 */
    float frequency_per_ps;   /* collision_frequency: 25 */
    int32_t chain_length;     /* 10 in the drivers (<= 16) */
    int32_t num_mts;          /* 5 */
    int32_t num_yoshidasuzuki;/* 1, 3 or 5 */
    int32_t reset;
    double ndf;               /* degrees of freedom (3N for the unconstrained LJ system; 6 per rigid water) */
    /* zero-initialised = the LJ behaviour (one mass, Angstrom, no constraints) */
    float mass_h_amu;        /* > 0 and species given: mass of the species-0 atoms (H, 1.008); mass_amu is then the O mass */
    float length_per_nm;     /* length unit of x, v, box per nm: 0 or 10 = Angstrom; 18.8972613 = bohr (DFT model:
                                positions in bohr, water/test_script/test_nosehoover_hb.py:106-109) */
    int32_t rigid_water;     /* 1 = atoms are O,H,H triples held rigid, as OpenMM does for the constrained water systems
                                of the water drivers at every addConstrainPositions / addConstrainVelocities of
                                hack_integrator.py:145-164,178,277-280,427-428 (SETTLE + analytic velocity constraint) */
    float r_oh, r_hh;        /* constraint lengths in the length unit (TIP3P: 0.9572, 1.5139 A) */
    int32_t remove_cm_motion;/* as in gamd_md_params, at the place hack_integrator.py:271-272 has it: propagateNHC() takes KE2 from
                                the velocities as they are and scales them, THEN addUpdateContextState() removes the centre-of-mass
                                velocity, then the kick (ndf is 3 smaller with a remover, :226-235) */
} gamd_nhc_params;
int32_t gamd_md_run_nhc(gamd_handle* h, float* x_dev, float* v_dev, float* f_dev, const uint8_t* species_dev,
                        const float* box, const gamd_nhc_params* p, double* chain_state_dev, int64_t n_steps, void* stream);

/* Event-timed replay of one force evaluation: per-kernel milliseconds of the last gamd_profile call.
 * names: newline-separated kernel labels; ms: one float per label.  For bench.py's roofline block. */
int32_t gamd_profile(gamd_handle* h, const float* pos_dev, const uint8_t* species_dev, const float* box,
                     float* out_norm_dev, void* stream, char* names, size_t names_bytes, float* ms, int32_t max_ms,
                     int32_t* n_out);

/* Live timing of the dominant kernel (conv-layer edge kernel) inside a timed region: while enabled,
 * every conv-edge launch is bracketed by HIP events on the launch stream.  gamd_timing_read
 * synchronises the stream and returns the summed duration and the launch count since the last
 * enable.  Used by bench.py for roofline.achieved. */
int32_t gamd_timing_enable(gamd_handle* h, int32_t enable);
int32_t gamd_timing_read(gamd_handle* h, void* stream, double* total_ms, int64_t* n_launches);
/* The same events split by stage: [0] conv-layer edge kernel (= gamd_timing_read), [1] edge encoder (its own event pair),
 * [2] the node kernel between two conv layers (from the stop event of layer l to the start event of layer l + 1, so it
 * includes the two kernel boundaries around it). */
int32_t gamd_timing_read_stages(gamd_handle* h, void* stream, double total_ms[3], int64_t n_launches[3]);
/* While timing is enabled, gamd_md_run / gamd_md_run_nhc also record one HIP event in front of the first kernel of every MD
 * step (the iteration of the drivers' loop, LJ/test_script/test_langevin.py:95-113) and one behind the last: step_ms[i] =
 * device time between consecutive events, in enqueue order, since the last gamd_timing_enable.  Writes at most max_steps
 * values; *n_steps = intervals available.  K runs of n steps give K * n intervals: the interval between the event behind a run's
 * last step and the first event of the next run (the host's gap between two calls, or the re-allocation of a run that froze
 * on a neighbour-buffer overflow) is not a step and is skipped.  A step that takes far longer than the median is a candidate
 * rebuild or a stall: bench.py reports min / p50 / p99 / max.  The event pools are bounded (65 536 step events): with timing
 * left on across a longer run the remaining steps are not timed. */
int32_t gamd_timing_read_steps(gamd_handle* h, void* stream, float* step_ms, int64_t max_steps, int64_t* n_steps);

#ifdef __cplusplus
}
#endif
#endif /* GAMD_HIP_H */
