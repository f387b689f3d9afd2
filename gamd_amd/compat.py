"""Reference-shaped adapters: the call contract of the rollout drivers
(`code/LJ/test_script/test_langevin.py:74-77,91,108`,
`code/water/test_script/test_nosehoover.py:79-81,106,123`) on top of `GamdForce`.

    model = ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT); model.cuda(); model.eval()
    force = model.predict_forces(pos)            # LJ    (np f64 [N,3] -> np f64 [N,3])
    force = model.predict_forces(feat, pos)      # water (feat: torch [N,1], O=1/H=0)
    force = model.predict_forces(feat, pos, box_size)   # DFT water (bohr; box_size np [3] per call)

Physics constants that are module-level in the reference (BOX_SIZE, CUTOFF_RADIUS,
NUM_OF_ATOMS; LJ/train_network_lj.py:26-29, water/train_network_tip3p.py:24-29) are
constructor arguments here; `gamd_amd/dropin/` holds modules NAMED like the reference's
(`train_network_lj`, `train_network_tip3p`, `train_network_tip4p`, `train_network_real_large`) that carry
those constants and the reference's constructor signature, so a driver runs with no edit at all.
"""
from __future__ import annotations

import functools
import os
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .engine import GamdForce
from .weights import ModelConfig, load_checkpoint, load_scaler, make_state_dict, infer_config


def create_water_bond(total_atom_num: int) -> np.ndarray:
    """O-H1, O-H2 per molecule, atoms ordered O,H,H (water/train_network_tip3p.py:38-42)."""
    o = np.arange(0, total_atom_num, 3)
    return np.stack([np.repeat(o, 2), (o[:, None] + np.array([1, 2])).reshape(-1)], axis=1)


def _node_feature(feat):
    """The `x` / `feat` argument of the water models: a float [N,1] tensor that the reference feeds to node_encoder
    as is (nn_module.py:554, :403).  The drivers build it as O = 1 / H = 0 (water/test_script/test_nosehoover.py:82-89)
    but any float value is carried through (GamdForce hands floating-point input to the library as float features)."""
    if isinstance(feat, np.ndarray):
        feat = torch.from_numpy(feat)
    f = feat.reshape(-1)
    return f if f.dtype.is_floating_point else f.to(torch.float32)


class _ModelLevel:
    """`self.pnet_model` of the reference wrappers: the nn.Module called as ``model([pos], [edge_idx])``
    (SimpleMDNetNew.forward, nn_module.py:672-685) or ``model([pos], feat, [edge_idx])``
    (WaterMDNetNew.forward, :545-558) or ``model([pos], feat, [box_size], cutoff)`` (WaterMDDynamicBoxNet.forward,
    :391-407, which searches neighbours itself).  Returns the NORMALISED output; lists with several graphs are evaluated
    graph by graph (see _batched)."""

    def __init__(self, owner):
        self._owner = owner

    def __call__(self, pos_lst, *rest):
        if len(pos_lst) != 1:
            return self._batched(pos_lst, *rest)
        if len(rest) == 1:
            feat, edge_lst = None, rest[0]
        elif len(rest) == 2:
            feat, edge_lst = rest
        elif len(rest) == 3:
            feat, box_lst, cutoff = rest
            self._owner._size_for(int(pos_lst[0].shape[0]))
            eng = self._owner._get_engine()
            if abs(float(cutoff) - eng.cutoff) > 1e-6 * eng.cutoff:
                raise ValueError(f"cutoff {cutoff} differs from the one the engine was built with ({eng.cutoff})")
            return eng.forward(pos_lst[0], box=np.asarray(box_lst[0], dtype=np.float32), species=_node_feature(feat))
        else:
            raise TypeError("expected ([pos], [edge_idx]), ([pos], feat, [edge_idx]) or ([pos], feat, [box], cutoff)")
        species = None if feat is None else _node_feature(feat)
        # a given edge list needs no neighbour search: a graph of any size runs (dgl.graph takes whatever it is handed)
        return self._owner._get_engine(n_atoms=int(pos_lst[0].shape[0])).forward_edges(pos_lst[0], edge_lst[0], species=species)

    def _batched(self, pos_lst, *rest):
        """``len(pos_lst) > 1``: the reference builds one graph per entry and runs the network on their disjoint union
        (build_graph_batches + dgl.batch, nn_module.py:655-661, :520-527; edge indices are local to each graph, `feat`
        is the concatenation of the per-graph node features) and returns the outputs concatenated in order.  Here the
        graphs are the boxes of ONE batched engine (GamdForce(n_boxes=len(pos_lst))): a single set of launches, results
        bit-identical to the graphs evaluated one by one.  Graphs of DIFFERENT sizes (dgl.batch takes any, nn_module.py:655-661)
        on the edge-list forms: runs of equal-sized graphs share a batched engine, the outputs are concatenated in list order
        (independent graphs: the same numbers as one disjoint union); the dynamic-box form (its own neighbour search, one
        atom count per engine) takes equal sizes."""
        if len(pos_lst) == 0:
            raise ValueError("empty pos_lst")
        if len(rest) not in (1, 2, 3):
            raise TypeError("expected (pos_lst, edge_lst), (pos_lst, feat, edge_lst) or (pos_lst, feat, box_lst, cutoff)")
        # the per-graph list must be as long as pos_lst (a longer one would be truncated silently, a shorter one would
        # surface as a bare IndexError)
        name, lst = ("box_size_lst", rest[1]) if len(rest) == 3 else ("edge_lst", rest[-1])
        if len(lst) != len(pos_lst):
            raise ValueError(f"{name} has {len(lst)} entries for {len(pos_lst)} graphs in pos_lst")
        sizes = [int(p.shape[0]) for p in pos_lst]
        if len(rest) == 3 and len(set(sizes)) == 1:
            self._owner._size_for(sizes[0])
        if len(set(sizes)) > 1:
            if len(rest) == 3:
                raise ValueError(f"the dynamic-box form takes graphs of one size per call, got {sizes}")
            # maximal runs of equal-sized graphs, each as one batch, outputs in list order
            outs, i, off = [], 0, 0
            while i < len(sizes):
                j = i
                while j < len(sizes) and sizes[j] == sizes[i]:
                    j += 1
                cnt = sizes[i] * (j - i)
                # (the LJ call pnet_model(pos_lst, None, edge_lst) carries feat = None: forwarded as it is)
                sub_rest = (lst[i:j],) if len(rest) == 1 else (None if rest[0] is None else rest[0][off:off + cnt], lst[i:j])
                outs.append(self.__call__(pos_lst[i:j], *sub_rest))
                off += cnt
                i = j
            return torch.cat(outs, dim=0)
        n, nb = sizes[0], len(pos_lst)
        if len(rest) == 3 and n != self._owner.num_atoms:
            raise ValueError(f"every graph of a batch must have {self._owner.num_atoms} atoms (the wrapper's size), got {sizes}")
        eng = self._owner._get_engine(n_boxes=nb, n_atoms=n)
        dev = eng.device
        pos = torch.cat([(torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)) if isinstance(p, np.ndarray) else p)
                         .to(device=dev, dtype=torch.float32) for p in pos_lst], dim=0)
        species = None
        if len(rest) >= 2:
            species = _node_feature(rest[0])
            if species.shape[0] != n * nb:
                raise ValueError(f"feat has {species.shape[0]} rows for {n * nb} atoms")
        if len(rest) == 3:
            cutoff = rest[2]
            if abs(float(cutoff) - eng.cutoff) > 1e-6 * eng.cutoff:
                raise ValueError(f"cutoff {cutoff} differs from the one the engine was built with ({eng.cutoff})")
            boxes = np.stack([np.broadcast_to(np.asarray(b, dtype=np.float32).reshape(-1), (3,)) for b in rest[1]])
            return eng.forward(pos, box=boxes, species=species)
        # dgl.batch: graph i's node ids are shifted by the number of nodes in front of it
        edges = torch.cat([(torch.from_numpy(e) if isinstance(e, np.ndarray) else e).to(device=dev, dtype=torch.int64) + i * n
                           for i, e in enumerate(lst)], dim=1)
        return eng.forward_edges(pos, edges, species=species)

    forward = __call__


class _class_or_instance_method:
    """Binds (cls, instance-or-None): Lightning's load_from_checkpoint is a classmethod that the drivers call on an
    instance (`ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)`, LJ/test_script/test_langevin.py:74)."""

    def __init__(self, fn):
        self.fn = fn
        functools.update_wrapper(self, fn)

    def __get__(self, obj, cls=None):
        return functools.partial(self.fn, cls if cls is not None else type(obj), obj)


# args fields that name a width / depth of the network: a checkpoint of another size is refused, as load_state_dict(strict)
# refuses it in the reference (size mismatch)
_ARG_WIDTHS = (("encoding_size", "encoding_size"), ("hidden_dim", "hidden_dim"), ("edge_embedding_dim", "edge_embedding_dim"))


class _ForceFieldBase:
    # build_model of the LJ / TIP wrappers hard-codes 'conv_layer': 4 (LJ/train_network_lj.py:75,
    # water/train_network_tip3p.py:85; the drop-in modules set _FIXED_CONV_LAYER = 4); only
    # train_network_real_large.py:80 reads args.conv_layer
    _CONV_LAYER_FROM_ARGS = False
    _FIXED_CONV_LAYER = None

    def __init__(self, args=None, state_dict=None, *, num_atoms: int, box_size, cutoff: float,
                 bond=None, scaler_ckpt: Optional[str] = None, device: int = 0, edge_dtype: str = "f32",
                 self_loop_mode: str = "dgl07_noop"):
        # what a fresh instance of the same wrapper needs (load_from_checkpoint returns one)
        self._ctor_kw = dict(num_atoms=num_atoms, box_size=box_size, cutoff=cutoff, bond=bond, device=device,
                             edge_dtype=edge_dtype, self_loop_mode=self_loop_mode)
        self.args = args or SimpleNamespace()
        # num_atoms None (DFT flavour only): taken from the first call, and re-sized when a later call brings another count
        self.num_atoms = None if num_atoms is None else int(num_atoms)
        self.box_size, self.cutoff = box_size, float(cutoff)
        self.bond, self.device_index = bond, device
        self.edge_dtype = edge_dtype                 # "f32" (default) | "f16x3" (fp32-grade split-fp16 GEMMs) | "bf16"
        self.self_loop_mode = self_loop_mode         # what add_self_loop()'s discarded result means
        self._nbr_flavour = "jaxmd"                  # graph_utils.NeighborSearcher semantics ('<' on r^2, self edge kept)
        self._skin = self.cutoff / 6.0               # its dr_threshold (graph_utils.py:24): candidate list reused
                                                     # between calls, exact cutoff re-applied every call
        self.training_mean = np.array([0.])          # LJ/train_network_lj.py:105-106
        self.training_var = np.array([1.])
        self._sd = state_dict
        self._engines = {}                           # n_boxes -> GamdForce (1: the driver path; B: model-level batches)
        self.pnet_model = _ModelLevel(self)        # attribute name of the reference (train_network_lj.py:95)
        if scaler_ckpt is not None:
            self.load_training_stats(scaler_ckpt)

    # -- construction / loading (test_langevin.py:74-77) --------------------------------------
    @_class_or_instance_method
    def load_from_checkpoint(cls, self, path: str, args=None, allow_pickle: bool = False, **kw):
        """pl.LightningModule.load_from_checkpoint as the drivers use it,
        ``ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)`` (LJ/test_script/test_langevin.py:74): a
        classmethod (callable on the class or on an instance) that builds a NEW wrapper from ``args`` (+ ``kw``), loads the
        checkpoint's ``state_dict`` ('pnet_model.' prefix) into it and returns it; the instance it was called on is left
        alone.  Called on an instance, the new wrapper inherits its system size / box / cutoff / bond / device — NOT its
        scaler: like the module Lightning's classmethod returns, it starts from the constructor's training_mean = 0,
        training_var = 1, and the drivers call load_training_stats on it next (LJ/test_script/test_langevin.py:75-76).
        ``allow_pickle``: see weights.load_checkpoint (restricted unpickler unless opted out)."""
        sd = load_checkpoint(path, allow_pickle=allow_pickle)
        if self is not None:
            new = self._respawn(args if args is not None else self.args, **kw)
        else:
            new = cls(args, **kw)
        new.load_state_dict(sd)
        return new

    def _respawn(self, args, **kw):
        return type(self)(args, None, **{**self._ctor_kw, **kw})

    def load_state_dict(self, sd, strict: bool = True):
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self._check_against_args(sd)
        self._sd = sd
        self._drop_engines()
        return self

    def state_dict(self):
        return dict(self._sd) if self._sd is not None else {}

    def _check_against_args(self, sd):
        """The reference builds the network from ``args`` and then loads the checkpoint strictly: a checkpoint of other
        widths / depth raises there (size mismatch / missing keys).  Same here, for the fields ``args`` carries."""
        cfg = infer_config(sd)
        want = {f: getattr(self.args, a) for a, f in _ARG_WIDTHS if hasattr(self.args, a)}
        if self._CONV_LAYER_FROM_ARGS and hasattr(self.args, "conv_layer"):
            want["conv_layer"] = int(self.args.conv_layer)
        elif self._FIXED_CONV_LAYER is not None:
            want["conv_layer"] = self._FIXED_CONV_LAYER
        if hasattr(self.args, "use_layer_norm"):
            want["use_layer_norm"] = bool(self.args.use_layer_norm)
        bad = {f: (v, getattr(cfg, f)) for f, v in want.items() if getattr(cfg, f) != v}
        if bad:
            raise RuntimeError("Error(s) in loading state_dict: size mismatch between args and checkpoint: "
                               + ", ".join(f"{f}: args {a} vs checkpoint {c}" for f, (a, c) in bad.items()))

    def _drop_engines(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    @property
    def _engine(self) -> Optional[GamdForce]:
        return self._engines.get(1)

    def load_training_stats(self, scaler_ckpt):
        if scaler_ckpt is not None:
            self.training_mean, self.training_var = load_scaler(scaler_ckpt)
            for e in self._engines.values():
                e.set_scaler(self.training_mean, self.training_var)

    def cuda(self, device=None):
        return self

    def eval(self):
        return self

    def _size_for(self, n: int) -> None:
        """The dynamic-box model has no fixed atom count in the reference (md_module.get_neighbor searches whatever it is
        handed, water/train_network_real_large.py:148-162): the engines are (re)built for the count a call brings.  The
        fixed-box wrappers keep the count they were constructed with (NUM_OF_ATOMS is baked into the reference's jitted
        neighbour mask, LJ/train_network_lj.py:110-112)."""
        if self._nbr_flavour != "torch" or n == self.num_atoms:
            return
        self._drop_engines()
        self.num_atoms = int(n)

    def _get_engine(self, n_boxes: int = 1, n_atoms: Optional[int] = None) -> GamdForce:
        """n_boxes = 1: the engine behind predict_forces / single-graph model calls; n_boxes = B: the batched engine a
        model-level call with B graphs runs on (built on first use, kept).  ``n_atoms`` other than the wrapper's size: an
        engine for graphs of that size on the edge-list forms of the model-level call (no neighbour search involved; the bond
        table of a water model is rebuilt for that many atoms, O,H,H order)."""
        n = self.num_atoms if n_atoms is None else int(n_atoms)
        key = n_boxes if n == self.num_atoms else (n_boxes, n)
        if key not in self._engines:
            if self._sd is None:
                raise RuntimeError("no weights loaded: call load_from_checkpoint / load_state_dict first")
            bond = self.bond if (n == self.num_atoms or self.bond is None) else create_water_bond(n)
            self._engines[key] = GamdForce(
                self._sd, n, self.box_size, self.cutoff, bond=bond,
                scaler=(self.training_mean, self.training_var), device=self.device_index,
                nbr_flavour=self._nbr_flavour, neighbor_skin=self._skin if (n_boxes == 1 and n == self.num_atoms) else 0.0,
                edge_dtype=self.edge_dtype, self_loop_mode=self.self_loop_mode, n_boxes=n_boxes)
        return self._engines[key]

    def denormalize(self, normalized_force, var, mean):
        return normalized_force * np.sqrt(var) + mean         # LJ/train_network_lj.py:128-131

    def _predict(self, pos: np.ndarray, feat=None, verbose=False) -> np.ndarray:
        eng = self._get_engine()
        # enforce periodic boundary in f64 on the host, then f32 (train_network_lj.py:141-142)
        posw = np.mod(np.asarray(pos, dtype=np.float64), np.asarray(self.box_size, dtype=np.float64))
        species = None if feat is None else _node_feature(feat)
        if verbose or os.environ.get("GAMD_PREDICT_LEGACY"):
            x = torch.from_numpy(posw).float()
        if verbose:
            # the reference's two time.time() buckets (train_network_lj.py:134-151): neighbour search vs network forward.
            # Here both are stages of one enqueued call, so they are measured with HIP events on the stream
            stages = eng.profile(x, species=species)
            pred = eng._out.detach().cpu().numpy()
            nbr = sum(ms for label, ms in stages if label == "neighbor_build") * 1e-3
            force = sum(ms for label, ms in stages if label != "neighbor_build") * 1e-3
            print('=============================================')
            print(f'Nbr search used time: {nbr}')
            print(f'Force eval used time: {force}')
        elif os.environ.get("GAMD_PREDICT_LEGACY"):      # (A/B switch of tools/predict_forces_latency.py: the three-sync form)
            pred = eng.forward(x, species=species, inplace=True).detach().cpu().numpy()   # device -> host sync (:153)
        else:
            # pinned staging both ways, one stream synchronisation (:141-153 in one enqueue); float64 -> float32 as .float() rounds
            pred = eng.forward_host(posw, species=species)
        return self.denormalize(pred, self.training_var, self.training_mean)   # f64 result (:155), a fresh array


class ParticleNetLightningLJ(_ForceFieldBase):
    """LJ flavour (code/LJ/train_network_lj.py:91-157): BOX_SIZE 27.27, CUTOFF_RADIUS 7.5, 258 atoms."""

    def __init__(self, args=None, state_dict=None, *, num_atoms=258, box_size=27.27, cutoff=7.5, **kw):
        super().__init__(args, state_dict, num_atoms=num_atoms, box_size=box_size, cutoff=cutoff, **kw)

    def predict_forces(self, pos: np.ndarray, verbose=False) -> np.ndarray:
        return self._predict(pos, None, verbose)


class ParticleNetLightningWater(_ForceFieldBase):
    """TIP3P/TIP4P flavour (code/water/train_network_tip3p.py:100-159): box 20, cutoff 4.2, 258*3 atoms."""

    def __init__(self, args=None, state_dict=None, *, num_atoms=258 * 3, box_size=20.0, cutoff=4.2, bond=None, **kw):
        bond = create_water_bond(num_atoms) if bond is None else bond
        super().__init__(args, state_dict, num_atoms=num_atoms, box_size=box_size, cutoff=cutoff, bond=bond, **kw)

    def predict_forces(self, feat: torch.Tensor, pos: np.ndarray) -> np.ndarray:
        return self._predict(pos, feat)


# test_nosehoover_hb.py:109: predict_forces returns hartree/bohr; the driver converts to kJ/mol/nm
HARTREE_PER_BOHR_TO_KJ_PER_MOL_NM = 2625.5 / 0.0529177


class ParticleNetLightningDFT(_ForceFieldBase):
    """DFT-water flavour (code/water/train_network_real_large.py:103-162, driver
    water/test_script/test_nosehoover_hb.py:64-113): WaterMDDynamicBoxNet with encoding 256 / edge embedding 256 /
    hidden 128 / 5 layers, cutoff ``args.cutoff`` (9.5 bohr), positions and box in bohr, the box handed over per
    call; neighbour semantics of md_module.get_neighbor ('<=' on the norm, no self edges)."""

    _CONV_LAYER_FROM_ARGS = True                     # water/train_network_real_large.py:80

    def __init__(self, args=None, state_dict=None, *, num_atoms=258 * 3, box_size=None, cutoff=None, **kw):
        if cutoff is None:
            cutoff = getattr(args, "cutoff", 9.5)
        if box_size is None:                         # only sizes the first neighbour buffers; the real box comes per call
            box_size = 20.0 / 0.529177
        kw.pop("bond", None)                         # WaterMDDynamicBoxNet is built without a bond graph (:64)
        super().__init__(args, state_dict, num_atoms=num_atoms, box_size=box_size, cutoff=cutoff, **kw)
        self._nbr_flavour = "torch"
        self._skin = 0.0                             # md_module.get_neighbor searches from scratch every call

    def predict_forces(self, feat: torch.Tensor, pos: np.ndarray, box_size) -> np.ndarray:
        self._size_for(int(np.asarray(pos).shape[0]))
        eng = self._get_engine()
        box = np.asarray(box_size, dtype=np.float64).reshape(-1)
        posw = np.mod(np.asarray(pos, dtype=np.float64), box)                 # train_network_real_large.py:150
        pred = eng.forward_host(posw, box=box.astype(np.float32), species=_node_feature(feat))   # one synchronisation per call
        return self.denormalize(pred, self.training_var, self.training_mean)  # :160
