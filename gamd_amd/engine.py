"""Host side of the MI355X force path: `GamdForce.forward(pos, box, species) -> forces`.

Mirrors what the reference's Python drivers call:
``ParticleNetLightning.predict_forces`` (LJ/train_network_lj.py:133-157,
water/train_network_tip3p.py:142-159) = neighbour search + model forward +
denormalise.  PyTorch is used for device memory and streams only; all compute
is in libgamd_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import GamdConfig, GamdMdParams, GamdNhcParams, check
from .weights import ModelConfig, infer_config, validate_state_dict

ArrayLike = Union[np.ndarray, torch.Tensor]

KIND = {"lj": 0, "water": 1, "dynbox": 1}
FLAVOUR = {"jaxmd": 0, "torch": 1}
# what `fluid_graph.add_self_loop()` with its result discarded does (nn_module.py:650-652)
SELF_LOOP = {"dgl07_noop": 0, "append_zero_feature_loops": 1}
KSEL_FORCE_GENERIC_WIDTH = 1


def _box3(box) -> np.ndarray:
    b = np.asarray(box, dtype=np.float64).reshape(-1)
    if b.size == 1:
        b = np.repeat(b, 3)
    if b.size != 3:
        raise ValueError("box must be a scalar or 3 values")
    return b.astype(np.float32)


def _boxes(box, n_boxes: int) -> np.ndarray:
    """scalar | [3] (ONE orthorhombic box, the same for every box) | [n_boxes, 1] or [n_boxes] cubic edge per box |
    [n_boxes, 3] -> float32 [n_boxes, 3].  A 1-D input of three numbers is always one box for all, also when n_boxes == 3:
    per-box cubic edges of a 3-box batch are spelled [3, 1] (or [3, 3])."""
    b = np.asarray(box, dtype=np.float64)
    if b.ndim == 0 or b.size == 1 or (b.ndim == 1 and b.size == 3):
        return np.tile(_box3(b), (n_boxes, 1))
    if b.ndim == 2 and b.shape == (n_boxes, 3):
        return b.astype(np.float32)
    if (b.ndim == 1 and b.size == n_boxes) or (b.ndim == 2 and b.shape == (n_boxes, 1)):
        return np.repeat(b.reshape(-1).astype(np.float32)[:, None], 3, axis=1)
    raise ValueError(f"box must be a scalar, 3 values, [{n_boxes}, 1] or [{n_boxes}, 3] ({n_boxes} boxes)")


class GamdForce:
    """One GPU-resident force model for a fixed atom count.

    Parameters mirror build_model()/ParticleNetLightning.__init__ of the reference:
    ``state_dict`` (reference key names), ``box``/``cutoff`` (BOX_SIZE / CUTOFF_RADIUS
    module constants there), ``bond`` (create_water_bond) and the scaler (mean, var).

    ``n_boxes`` > 1: that many INDEPENDENT boxes of ``n_atoms`` atoms each share every launch (the reference's
    several-graphs-per-forward, nn_module.py:655-661,676-679; a replica ensemble on one GPU).  Positions / species /
    forces are then [n_boxes * n_atoms, ...] (or [n_boxes, n_atoms, ...]), box-major; ``box`` may differ per box
    ([n_boxes, 3]); ``bond`` names atoms of one box.  Results are bit-identical to the boxes evaluated one by one.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], n_atoms: int, box, cutoff: float,
                 bond: Optional[np.ndarray] = None, scaler: Tuple[float, float] = (0.0, 1.0),
                 nbr_flavour: str = "jaxmd", device: int = 0, keep_stages: bool = False,
                 edge_capacity: int = 0, cfg: Optional[ModelConfig] = None, edge_dtype: str = "f32",
                 neighbor_skin: float = 0.0, self_loop_mode: str = "dgl07_noop", kernel_select: int = 0,
                 small_tile_limit: int = 0, n_boxes: int = 1):
        self._h = C.c_void_p()
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GamdError("GamdForce needs a HIP device (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        cfg = cfg or infer_config(state_dict)
        # build_model's widths (nn_module.py:561-601): anything up to 256 / 256 / 256 — the library zero-pads to its 128-wide
        # blocks and normalises over the true widths (hidden_dim above 128: fp32 edge MLP only, the library says so)
        if (not 1 <= cfg.encoding_size <= 256 or not 1 <= cfg.edge_embedding_dim <= 256 or not 1 <= cfg.hidden_dim <= 256
                or cfg.n_rbf not in (0, 40)):
            raise ValueError("the gfx950 kernels cover encoding_size / edge_embedding_dim / hidden_dim up to 256 "
                             "and the RBF expansion on (40 centres) or off "
                             f"(got enc={cfg.encoding_size} hidden={cfg.hidden_dim} edge={cfg.edge_embedding_dim} "
                             f"n_rbf={cfg.n_rbf})")
        validate_state_dict(state_dict, cfg)
        self.cfg = cfg
        self.n = int(n_atoms)                          # atoms per box
        self.n_boxes = max(1, int(n_boxes))
        self.n_total = self.n * self.n_boxes
        self.device = torch.device("cuda", device)
        self.box = _box3(box)                          # constructor box (every box starts with it)
        self.cutoff = float(cutoff)
        c = GamdConfig()
        c.n_boxes = self.n_boxes
        c.n_atoms, c.kind, c.n_layers = self.n, KIND[cfg.kind], cfg.conv_layer
        c.use_bond, c.nbr_flavour, c.device = int(cfg.use_bond), FLAVOUR[nbr_flavour], device
        c.cutoff = self.cutoff
        for d in range(3):
            c.box[d] = float(self.box[d])
        c.edge_capacity, c.keep_stages = int(edge_capacity), int(keep_stages)
        c.edge_dtype = {"f32": 0, "bf16": 1, "f16x3": 2}[edge_dtype]
        c.encoding_size, c.edge_embedding_dim, c.hidden_dim = cfg.encoding_size, cfg.edge_embedding_dim, cfg.hidden_dim
        c.no_expand_edge = int(cfg.n_rbf == 0)
        c.neighbor_skin = float(neighbor_skin)      # > 0: Verlet-skin reuse (jax-md uses cutoff/6, graph_utils.py:24)
        if self_loop_mode not in SELF_LOOP:
            raise ValueError(f"self_loop_mode must be one of {sorted(SELF_LOOP)}")
        c.self_loop_mode = SELF_LOOP[self_loop_mode]
        c.kernel_select, c.small_tile_limit = int(kernel_select), int(small_tile_limit)
        self.edge_dtype = edge_dtype
        check(self._lib.gamd_create(C.byref(c), C.byref(self._h)), "gamd_create")
        self.keep_stages = keep_stages
        for name, t in state_dict.items():
            if name.endswith("num_batches_tracked"):      # BatchNorm's step counter: not used at inference
                continue
            a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(self._lib.gamd_load_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                  f"gamd_load_weight({name})")
        check(self._lib.gamd_finalize_weights(self._h), "gamd_finalize_weights")
        self.set_scaler(*scaler)
        if cfg.use_bond:
            if bond is None:
                raise ValueError("use_bond model needs the bond list")
            b = np.ascontiguousarray(np.asarray(bond, dtype=np.int32))
            check(self._lib.gamd_set_bonds(self._h, b.ctypes.data_as(C.c_void_p), b.shape[0]), "gamd_set_bonds")
        self._feat = None
        self._out = torch.empty((self.n_total, 3), dtype=torch.float32, device=self.device)
        self._out_den = torch.empty((self.n_total, 3), dtype=torch.float32, device=self.device)
        self.last_status = 0

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gamd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------
    def set_scaler(self, mean, var):
        """load_training_stats (LJ/train_network_lj.py:119-123): mean/var of scaler.npz."""
        self.scaler_mean = np.asarray(mean, dtype=np.float64).reshape(-1)[:1]
        self.scaler_var = np.asarray(var, dtype=np.float64).reshape(-1)[:1]
        check(self._lib.gamd_set_scaler(self._h, float(self.scaler_mean[0]), float(self.scaler_var[0])),
              "gamd_set_scaler")

    # -- helpers ---------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev_pos(self, pos: ArrayLike) -> torch.Tensor:
        if isinstance(pos, np.ndarray):
            pos = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float32))
        pos = pos.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(pos.shape) == (self.n_boxes, self.n, 3):
            pos = pos.view(self.n_total, 3)
        if tuple(pos.shape) != (self.n_total, 3):
            raise ValueError(f"pos must be [{self.n_total}, 3]" + (f" or [{self.n_boxes}, {self.n}, 3]" if self.n_boxes > 1 else "")
                             + f", got {tuple(pos.shape)}")
        return pos

    def _md_state(self, *ts):
        for t in ts:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                    and tuple(t.shape) in ((self.n_total, 3), (self.n_boxes, self.n, 3))):
                raise ValueError(f"x, v, f must be contiguous float32 CUDA tensors of shape [{self.n_total}, 3]")

    def _dev_species(self, species) -> Optional[torch.Tensor]:
        """species / node feature [N] or [N,1] -> uint8 O=1/H=0 flags on the device (what the integrators pick masses
        by).  A floating-point input is ALSO handed to the library as the float node feature the reference feeds to
        node_encoder (nn_module.py:554): it need not be 0/1."""
        if species is None:
            self._set_features(None)
            return None
        if isinstance(species, np.ndarray):
            species = torch.from_numpy(species)
        # Host-side input (what the drivers pass): the device copy is kept by the engine and reused while the content stays the
        # same.  The library reads it from kernels that may still be in flight when an asynchronous md_run returns, so it must not
        # die with this call's locals; and a stable pointer lets the library skip its O,H,H layout check (a device -> host copy
        # and a stream synchronisation) on every call after the first.
        key = None
        if not species.is_cuda:
            c = species.detach().reshape(-1).contiguous()
            # (bytes through a uint8 view: dtypes numpy does not know, bfloat16, hash like any other)
            key = (str(c.dtype), c.numel(), hash(c.view(torch.uint8).numpy().tobytes()), torch.cuda.current_stream(self.device).cuda_stream)
            hit = getattr(self, "_species_cache", None)
            if hit is not None and hit[0] == key:
                self._set_features(hit[2])
                return hit[1]
        s = species.reshape(-1).to(device=self.device)
        if s.numel() == self.n and self.n_boxes > 1:
            s = s.repeat(self.n_boxes)                    # one box's species, the same for every box
        if s.numel() != self.n_total:
            raise ValueError("species must have one entry per atom")
        feat = s.to(torch.float32).contiguous() if s.dtype.is_floating_point and self.cfg.kind != "lj" else None
        self._set_features(feat)
        flags = (s != 0).to(torch.uint8).contiguous()
        # (a device tensor of the caller's is the caller's to keep alive; the flags derived from it are held until the next call)
        # The library skips its O,H,H layout check while the species POINTER is the one it validated last: new content must never
        # arrive at an address it has validated for other content.  The flags above were allocated while the previous buffer was
        # still alive, and the previous generation stays alive one change longer, so three consecutive generations are distinct.
        self._species_prev = getattr(self, "_species_cache", None)
        self._species_cache = (key, flags, feat)
        return flags

    def _set_features(self, feat: Optional[torch.Tensor]) -> None:
        if feat is None and self._feat is None:
            return
        self._feat = feat                             # keeps the device buffer alive while the library points at it
        check(self._lib.gamd_set_node_features(self._h, C.c_void_p(feat.data_ptr()) if feat is not None else None),
              "gamd_set_node_features")

    def _box_arg(self, box):
        b = _boxes(self.box if box is None else box, self.n_boxes)
        return (C.c_float * (3 * self.n_boxes))(*[float(x) for x in b.reshape(-1)])

    # -- the hot path ----------------------------------------------------------------------------
    def forward(self, pos: ArrayLike, box=None, species=None, denormalize: bool = False,
                inplace: bool = False) -> torch.Tensor:
        """pos [N,3] (any periodic image), box scalar/[3] (default: constructor box), species [N]
        (O=1/H=0, or the float node feature of the water models; ignored for LJ) -> network output [N,3] fp32 on
        the device, in the caller's atom order.  Normalised like ``pnet_model(...)`` unless ``denormalize`` (then
        fp32 out*sqrt(var)+mean).  Returns a FRESH tensor like the reference's ``pnet_model(...)``; ``inplace=True``
        returns the engine's persistent output buffer instead (overwritten by the next call: for MD loops)."""
        p = self._dev_pos(pos)
        s = self._dev_species(species)
        st = self._lib.gamd_forces(self._h, C.c_void_p(p.data_ptr()),
                                   C.c_void_p(s.data_ptr()) if s is not None else None,
                                   self._box_arg(box), C.c_void_p(self._out.data_ptr()),
                                   C.c_void_p(self._out_den.data_ptr()), self._stream())
        self.last_status = check(st, "gamd_forces")
        out = self._out_den if denormalize else self._out
        return out if inplace else out.clone()

    __call__ = forward

    def forward_host(self, pos: np.ndarray, box=None, species=None, denormalize: bool = False) -> np.ndarray:
        """The reference's host-array boundary (``predict_forces``: numpy positions in, numpy forces out,
        LJ/train_network_lj.py:133-157) through ``gamd_forces_host``: float64 -> float32 as ``torch.from_numpy(pos).float()``
        rounds, the library's pinned staging buffers both ways, copy in / kernels / copy out enqueued on the caller's stream
        and ONE synchronisation (the library replays the call if a neighbour buffer had to be regrown: ``last_status`` 1).
        Returns a float32 [N,3] array owned by the engine: valid until the next call."""
        pos = np.asarray(pos)
        if pos.shape == (self.n_boxes, self.n, 3):
            pos = pos.reshape(self.n_total, 3)
        if pos.shape != (self.n_total, 3):
            raise ValueError(f"pos must be [{self.n_total}, 3], got {tuple(pos.shape)}")
        p32 = np.ascontiguousarray(pos, dtype=np.float32)
        if getattr(self, "_host_out", None) is None:
            self._host_out = np.empty((self.n_total, 3), dtype=np.float32)
        s = self._dev_species(species)
        st = self._lib.gamd_forces_host(self._h, p32.ctypes.data_as(C.c_void_p), C.c_void_p(s.data_ptr()) if s is not None else None,
                                        self._box_arg(box), self._host_out.ctypes.data_as(C.c_void_p), 1 if denormalize else 0,
                                        self._stream())
        self.last_status = check(st, "gamd_forces_host")
        return self._host_out

    def forward_edges(self, pos: ArrayLike, edge_idx, box=None, species=None, denormalize: bool = False,
                      inplace: bool = False) -> torch.Tensor:
        """Model-level call of the reference, ``pnet_model([pos], [edge_idx])``: edge_idx [2,E] with row 0 =
        centre, row 1 = neighbour (LJ/train_network_lj.py:183-184); the built-in radius search is bypassed.
        Returns a fresh tensor unless ``inplace``."""
        p = self._dev_pos(pos)
        s = self._dev_species(species)
        if isinstance(edge_idx, np.ndarray):
            edge_idx = torch.from_numpy(edge_idx)
        e = edge_idx.to(device=self.device, dtype=torch.int32).contiguous()
        if e.dim() != 2 or e.shape[0] != 2:
            raise ValueError("edge_idx must be [2, E]")
        st = self._lib.gamd_forces_edges(self._h, C.c_void_p(p.data_ptr()),
                                         C.c_void_p(s.data_ptr()) if s is not None else None, self._box_arg(box),
                                         C.c_void_p(e[0].data_ptr()), C.c_void_p(e[1].data_ptr()), int(e.shape[1]),
                                         C.c_void_p(self._out.data_ptr()), C.c_void_p(self._out_den.data_ptr()),
                                         self._stream())
        self.last_status = check(st, "gamd_forces_edges")
        out = self._out_den if denormalize else self._out
        return out if inplace else out.clone()

    def build_neighbors(self, pos: ArrayLike, box=None, species=None) -> int:
        p = self._dev_pos(pos)
        s = self._dev_species(species)
        st = self._lib.gamd_build_neighbors(self._h, C.c_void_p(p.data_ptr()),
                                            C.c_void_p(s.data_ptr()) if s is not None else None,
                                            self._box_arg(box), self._stream())
        return check(st, "gamd_build_neighbors")

    def counts(self) -> Tuple[int, int, int]:
        e, p, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.gamd_get_counts(self._h, C.byref(e), C.byref(p), C.byref(c)), "gamd_get_counts")
        return e.value, p.value, c.value

    def skin_stats(self) -> Tuple[int, int, int]:
        """(candidate-list rebuilds so far, candidates in the last rebuilt list, candidate capacity)."""
        r, c, cap = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.gamd_get_skin_stats(self._h, C.byref(r), C.byref(c), C.byref(cap)), "gamd_get_skin_stats")
        return r.value, c.value, cap.value

    # -- stage getters for parity tests ----------------------------------------------------------
    def _dbg(self, what: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(self._lib.gamd_debug_get(self._h, what, out.ctypes.data_as(C.c_void_p), out.nbytes), "gamd_debug_get")
        return out

    def debug_partial(self) -> np.ndarray:
        """[pieces, H] partial-sum pieces of the LAST conv layer (one row per run of edges with the same destination inside a
        16-edge chunk), CSR order."""
        return self._dbg(6, (self.counts()[1], 128 * ((self.cfg.encoding_size + 127) // 128)), np.float32)

    def debug_perm(self) -> np.ndarray:
        return self._dbg(0, (self.n_total,), np.int32)

    def debug_csr(self) -> Tuple[np.ndarray, np.ndarray]:
        e = self.counts()[0]
        return self._dbg(1, (self.n_total + 1,), np.int32), self._dbg(2, (e,), np.int32)

    def debug_edges(self) -> np.ndarray:
        """[2,E] (centre, neighbour) in ORIGINAL atom ids (box-major over all boxes), CSR order.  The padding slots that
        align the boxes of a batch (source index n_total) are not edges and are left out."""
        perm = self.debug_perm().astype(np.int64)
        row_ptr, col = self.debug_csr()
        dst = np.repeat(np.arange(self.n_total), np.diff(row_ptr))
        real = col < self.n_total
        return np.stack([perm[dst[real]], perm[col[real]]])

    def debug_edge_rows(self) -> np.ndarray:
        """CSR slots that hold real edges (bool [E]): the rows of debug_e / debug_feat that debug_edges lists."""
        return self.debug_csr()[1] < self.n_total

    def debug_e(self) -> np.ndarray:
        """e [E, edge_embedding_dim] de-fragmented to CSR edge order."""
        e = self.counts()[0]
        nt = (e + 31) // 32
        nb = (self.cfg.edge_embedding_dim + 127) // 128               # widths below a 128-block are zero-padded on the device
        if self.edge_dtype in ("f16x3", "bf16"):
            # operand-form fragments: [tile][block][t][u][hi | lo][lane][8 halves] (split-fp16: the value is hi + lo) or
            # [tile][block][t][u][lane][8 bf16]; K step (t, u) value j of lane (slot, half) is feature
            # 128 block + 32 t + (r & 3) + 8 (r >> 2) + 4 half with r = 8 u + j
            raw = self._dbg(3, (nt, nb, 4096), np.float32)
            if self.edge_dtype == "f16x3":
                v = raw.view(np.float16).reshape(nt, nb, 4, 2, 2, 64, 8).astype(np.float32)
                val = v[:, :, :, :, 0] + v[:, :, :, :, 1]
            else:
                u16 = raw.reshape(-1).view(np.uint16)[:nt * nb * 4096].reshape(nt, nb, 4, 2, 64, 8)       # 8 KiB per (tile, block), dense
                val = (u16.astype(np.uint32) << 16).view(np.float32)
            lane = np.arange(64)
            slot, half = lane & 31, lane >> 5
            pi = 16 * ((slot >> 2) & 1) + (slot & 3) + 4 * (slot >> 3)
            rows = (np.arange(nt)[:, None] * 32 + pi[None, :])
            out = np.zeros((nt * 32, 128 * nb), dtype=np.float32)
            for blk in range(nb):
                for t in range(4):
                    for u in range(2):
                        for j in range(8):
                            r = 8 * u + j
                            feat = 128 * blk + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half
                            out[rows, feat[None, :]] = val[:, blk, t, u, :, j]
            return out[:e, :self.cfg.edge_embedding_dim]
        frag = self._dbg(3, (nt, nb, 4, 4, 64, 4), np.float32)
        lane = np.arange(64)
        slot, half = lane & 31, lane >> 5
        pi = 16 * ((slot >> 2) & 1) + (slot & 3) + 4 * (slot >> 3)
        out = np.zeros((nt * 32, 128 * nb), dtype=np.float32)
        rows = (np.arange(nt)[:, None] * 32 + pi[None, :])
        for b in range(nb):
            for t in range(4):
                for q in range(4):
                    for j in range(4):
                        feat = 128 * b + 32 * t + 8 * q + 4 * half + j           # per lane
                        out[rows, feat[None, :]] = frag[:, b, t, q, :, j]
        return out[:e, :self.cfg.edge_embedding_dim]

    def debug_feat(self, n_feat: int) -> np.ndarray:
        e = self.counts()[0]
        return self._dbg(4, (e, 48), np.float32)[:, :n_feat]

    def debug_h(self, layer: int) -> np.ndarray:
        """residual stream h_layer [N, encoding_size] in ORIGINAL atom order."""
        hp = 128 * ((self.cfg.encoding_size + 127) // 128)
        hs = self._dbg(16 + layer, (self.n_total, hp), np.float32)[:, :self.cfg.encoding_size]
        out = np.empty_like(hs)
        out[self.debug_perm()] = hs
        return out

    # -- on-device MD (split BAOAB of hack_integrator.py) ----------------------------------------
    def md_run(self, x: torch.Tensor, v: torch.Tensor, f: torch.Tensor, n_steps: int, dt_ps=0.002,
               mass_amu=39.9, temperature_k=100.0, gamma_per_ps=25.0, seed=0, first_step=0,
               box=None, species=None, sync: bool = True, mass_h_amu=0.0, length_per_nm=0.0,
               rigid_water: bool = False, r_oh=0.0, r_hh=0.0, remove_cm_motion: Optional[bool] = None) -> None:
        """Advance (x, v, f) in place by n_steps; f holds denormalised forces (kJ/mol/nm) at x.

        Water: ``mass_amu`` is the oxygen mass and ``mass_h_amu`` the mass of the species-0 atoms;
        ``rigid_water`` holds every O,H,H triple rigid at (r_oh, r_hh) like OpenMM's constrained water
        (positions must then be whole molecules; they are kept whole).  ``length_per_nm`` is the length
        unit of x/v/box (10 = Angstrom, the default; 18.8972613 = bohr for the DFT model).
        ``remove_cm_motion``: subtract the centre-of-mass velocity at the top of every step like the CMMotionRemover that
        hack_integrator.py:142 runs when the OpenMM System has one; default True with ``rigid_water`` (the water drivers'
        openmmtools WaterBox carries one), False otherwise (the LJ fluid does not)."""
        self._md_state(x, v, f)
        s = self._dev_species(species)
        if remove_cm_motion is None:
            remove_cm_motion = bool(rigid_water)
        p = GamdMdParams(dt_ps, mass_amu, temperature_k, gamma_per_ps, seed, first_step, mass_h_amu, length_per_nm,
                         int(rigid_water), r_oh, r_hh, int(bool(remove_cm_motion)))
        st = self._lib.gamd_md_run(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()),
                                   C.c_void_p(f.data_ptr()), C.c_void_p(s.data_ptr()) if s is not None else None,
                                   self._box_arg(box), C.byref(p), int(n_steps), self._stream())
        check(st, "gamd_md_run")
        if sync:
            self.last_status = check(self._lib.gamd_sync_status(self._h, self._stream()), "gamd_sync_status")

    def md_run_nhc(self, x: torch.Tensor, v: torch.Tensor, f: torch.Tensor, n_steps: int, chain_state: torch.Tensor = None,
                   dt_ps=0.002, mass_amu=39.9, temperature_k=100.0, frequency_per_ps=25.0, chain_length=10, num_mts=5,
                   num_yoshidasuzuki=5, ndf=None, box=None, species=None, sync: bool = True, mass_h_amu=0.0,
                   length_per_nm=0.0, rigid_water: bool = False, r_oh=0.0, r_hh=0.0,
                   remove_cm_motion: Optional[bool] = None) -> torch.Tensor:
        """Split Nose-Hoover-chain steps (hack_integrator.py:182-493).  Returns the chain state tensor
        (float64 [3*chain_length+2] on the device); pass it back in to continue a trajectory.
        Several boxes: one chain per box (state [n_boxes, 3*chain_length+2]), ``ndf`` is per box.
        ``ndf`` defaults to what hack_integrator.py:226-235 computes from the OpenMM System: 3 per particle, minus the
        constraints (three per rigid molecule), minus 3 when the System holds a CMMotionRemover.
        ``remove_cm_motion`` says whether it does: default True with ``rigid_water`` (the water drivers build an
        openmmtools WaterBox, whose System carries one), False otherwise (the LJ drivers' ndf is 3N).  When it does, the
        centre-of-mass velocity is also subtracted on the device where hack_integrator.py:271-272 does it: behind
        propagateNHC() of the first half (the chain sees the velocities as they are), in front of the kick."""
        self._md_state(x, v, f)
        reset = chain_state is None
        if reset:
            shape = (3 * chain_length + 2,) if self.n_boxes == 1 else (self.n_boxes, 3 * chain_length + 2)
            chain_state = torch.zeros(shape, dtype=torch.float64, device=self.device)
        assert chain_state.dtype == torch.float64 and chain_state.is_contiguous() \
            and chain_state.numel() == self.n_boxes * (3 * chain_length + 2)
        s = self._dev_species(species)
        if remove_cm_motion is None:
            remove_cm_motion = bool(rigid_water)
        if ndf is None:
            ndf = (2 * self.n if rigid_water else 3 * self.n) - (3 if remove_cm_motion else 0)
        p = GamdNhcParams(dt_ps, mass_amu, temperature_k, frequency_per_ps, chain_length, num_mts, num_yoshidasuzuki,
                          int(reset), float(ndf),
                          mass_h_amu, length_per_nm, int(rigid_water), r_oh, r_hh, int(bool(remove_cm_motion)))
        st = self._lib.gamd_md_run_nhc(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()),
                                       C.c_void_p(f.data_ptr()), C.c_void_p(s.data_ptr()) if s is not None else None,
                                       self._box_arg(box), C.byref(p), C.c_void_p(chain_state.data_ptr()), int(n_steps),
                                       self._stream())
        check(st, "gamd_md_run_nhc")
        if sync:
            self.last_status = check(self._lib.gamd_sync_status(self._h, self._stream()), "gamd_sync_status")
        return chain_state

    def sync_status(self) -> int:
        """0, or 1 when an enqueued MD run overflowed a neighbour buffer, froze on the device and was resumed."""
        return check(self._lib.gamd_sync_status(self._h, self._stream()), "gamd_sync_status")

    def nonfinite_seen(self) -> bool:
        """True if any force evaluation since the last call produced a non-finite component (clears the flag)."""
        flags = (C.c_int32 * 4)()
        check(self._lib.gamd_get_device_flags(self._h, flags), "gamd_get_device_flags")
        return bool(flags[0])

    def timing_enable(self, on: bool = True) -> None:
        check(self._lib.gamd_timing_enable(self._h, int(on)), "gamd_timing_enable")

    def timing_read(self) -> Tuple[float, int]:
        """(summed conv-edge kernel ms, launches) since timing_enable(True); synchronises."""
        tot, cnt = C.c_double(), C.c_int64()
        check(self._lib.gamd_timing_read(self._h, self._stream(), C.byref(tot), C.byref(cnt)), "gamd_timing_read")
        return tot.value, cnt.value

    def timing_read_stages(self):
        """{stage: (summed ms, count)} since timing_enable(True) for 'conv_edge', 'edge_encode' and 'node_mid' (the node
        kernel between two conv layers, kernel boundaries included); synchronises."""
        ms, cnt = (C.c_double * 3)(), (C.c_int64 * 3)()
        check(self._lib.gamd_timing_read_stages(self._h, self._stream(), ms, cnt), "gamd_timing_read_stages")
        return {k: (ms[i], cnt[i]) for i, k in enumerate(("conv_edge", "edge_encode", "node_mid"))}

    def timing_read_steps(self, max_steps: int = 1 << 16) -> np.ndarray:
        """Device milliseconds of every MD step enqueued by md_run / md_run_nhc since timing_enable(True) (one HIP event in
        front of each step's first kernel, one behind the last); synchronises."""
        buf = (C.c_float * max_steps)()
        n = C.c_int64()
        check(self._lib.gamd_timing_read_steps(self._h, self._stream(), buf, max_steps, C.byref(n)), "gamd_timing_read_steps")
        return np.ctypeslib.as_array(buf)[:min(n.value, max_steps)].astype(np.float64)

    def profile(self, pos: ArrayLike, box=None, species=None):
        """Event-timed single forward: list of (kernel label, ms)."""
        p = self._dev_pos(pos)
        s = self._dev_species(species)
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        n = C.c_int32()
        for _ in range(4):
            st = self._lib.gamd_profile(self._h, C.c_void_p(p.data_ptr()),
                                        C.c_void_p(s.data_ptr()) if s is not None else None, self._box_arg(box),
                                        C.c_void_p(self._out.data_ptr()), self._stream(), names, 4096, ms, 64, C.byref(n))
            check(st, "gamd_profile")
            st = self._lib.gamd_sync_status(self._h, self._stream())
            if st != -34:                          # -34: a neighbour buffer overflowed and was regrown -> replay
                check(st, "gamd_sync_status")
                break
        else:
            check(st, "gamd_sync_status")
        labels = names.value.decode().strip().split("\n")
        return list(zip(labels, [ms[i] for i in range(n.value)]))
