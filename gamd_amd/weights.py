"""Weights contract of the force-inference path (SURVEY.md §8b "Weights contract").

The reference keeps its parameters in a torch ``state_dict`` whose key names and
shapes are fixed by the module constructors in ``code/nn_module.py``
(SimpleMDNetNew :561-601, WaterMDNetNew :410-460, WaterMDDynamicBoxNet :266-320,
SmoothConvLayerNew :95-106, SmoothConvBlockNew :176-196, MLP :48-65).  This
module

* enumerates those keys (`state_dict_spec`) so a checkpoint can be validated,
* loads plain state_dicts, Lightning checkpoints (``ckpt['state_dict']`` with the
  ``pnet_model.`` prefix, LJ/train_network_lj.py:95) and ``scaler.npz``
  (keys ``mean``/``var``, LJ/train_network_lj.py:119-123,346-349),
* creates seeded random-initialised state_dicts of the same architecture for
  synthetic benchmarks (the pretrained ``checkpoint.ckpt`` blobs are absent from
  the reference mirror, ``.MISSING_LARGE_BLOBS``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class ModelConfig:
    """Constructor arguments of the reference models (build_model,
    LJ/train_network_lj.py:68-88; water/train_network_tip3p.py:75-97)."""
    kind: str = "lj"              # "lj" (SimpleMDNetNew) | "water" (WaterMDNetNew) | "dynbox" (WaterMDDynamicBoxNet)
    encoding_size: int = 128
    hidden_dim: int = 128
    edge_embedding_dim: int = 128
    conv_layer: int = 4
    in_feats: int = 1             # water / dynbox: node feature width
    out_feats: int = 3
    use_bond: bool = False        # water: 45th edge feature (bond flag)
    n_rbf: int = 40               # RBFExpansion(high=1, gap=0.025) -> ceil(1/0.025) = 40; 0 = expand_edge=False
                                  # (WaterMDDynamicBoxNet only, nn_module.py:278,296-297,329-336)
    use_layer_norm: bool = True   # build_model's args.use_layer_norm (True in every rollout driver).  False = the constructors'
                                  # default: BatchNorm1d between the conv layers (nn_module.py:171-196, use_batch_norm = not
                                  # use_layer_norm :579), which at inference is a per-feature affine map of the running statistics
    update_edge: bool = False     # WaterMDDynamicBoxNet(update_edge=True) (--update_edge, water/train_network_real_large.py:83,362):
                                  # every conv layer leaves edata['e'] = edge_layer_norm(e_emb) for the layers after it
                                  # (nn_module.py:91-92, :140-146); needs encoding_size == edge_embedding_dim

    @property
    def edge_in(self) -> int:
        return 3 + 1 + self.n_rbf + (1 if self.use_bond else 0)


def state_dict_spec(cfg: ModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key -> shape, in the order torch registers them in the reference modules."""
    H, D, Eh = cfg.encoding_size, cfg.hidden_dim, cfg.edge_embedding_dim
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    spec["length_mean"] = (1,)
    spec["length_std"] = (1,)
    if cfg.kind == "lj":
        spec["node_emb"] = (1, H)

    def lin(name, out_f, in_f):
        spec[name + ".weight"] = (out_f, in_f)
        spec[name + ".bias"] = (out_f,)

    for l in range(cfg.conv_layer):
        p = f"graph_conv.conv.{l}"
        if cfg.update_edge:                          # registered first (nn_module.py:91-92); LayerNorm(in_edge_feats)
            spec[p + ".edge_layer_norm.weight"] = (Eh,)
            spec[p + ".edge_layer_norm.bias"] = (Eh,)
        # MLP(in_edge_feats, hidden_dim, hidden_layer=2) at nn_module.py:95 does not pass
        # MLP's own hidden_dim, so its inner width is the MLP default 128 (nn_module.py:25)
        lin(p + ".edge_affine.mlp_layer.0", 128, Eh)
        lin(p + ".edge_affine.mlp_layer.2", D, 128)
        lin(p + ".src_affine", D, H)
        lin(p + ".dst_affine", D, H)
        lin(p + ".theta_edge.mlp_layer.1", D, D)
        lin(p + ".theta_edge.mlp_layer.3", H, D)
        lin(p + ".phi_dst", D, H)
        lin(p + ".phi_edge", D, H)
        lin(p + ".phi.mlp_layer.1", H, D)
    for l in range(cfg.conv_layer):
        spec[f"graph_conv.norm_layers.{l}.weight"] = (H,)
        spec[f"graph_conv.norm_layers.{l}.bias"] = (H,)
        if not cfg.use_layer_norm:                   # nn.BatchNorm1d buffers, in the order torch registers them
            spec[f"graph_conv.norm_layers.{l}.running_mean"] = (H,)
            spec[f"graph_conv.norm_layers.{l}.running_var"] = (H,)
            spec[f"graph_conv.norm_layers.{l}.num_batches_tracked"] = ()
    if cfg.n_rbf > 0:
        spec["edge_expand.centers"] = (cfg.n_rbf,)
    if cfg.kind != "lj":
        lin("node_encoder", H, cfg.in_feats)
    lin("edge_encoder.mlp_layer.0", D, cfg.edge_in)
    lin("edge_encoder.mlp_layer.2", D, D)
    lin("edge_encoder.mlp_layer.4", Eh, D)
    spec["edge_layer_norm.weight"] = (Eh,)
    spec["edge_layer_norm.bias"] = (Eh,)
    lin("graph_decoder.mlp_layer.0", D, H)
    lin("graph_decoder.mlp_layer.2", cfg.out_feats, D)
    return spec


def make_state_dict(cfg: ModelConfig, seed: int = 0, length_mean: float = 4.0,
                    length_std: float = 1.5) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random-init weights of the reference architecture.

    Linear layers use U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
    (torch's nn.Linear default scale); LayerNorm affines, ``node_emb`` and the
    edge-length statistics are set to *non-trivial* values so that parity tests
    exercise them.  Deterministic for a given (cfg, seed) on any
    host: a CPU torch.Generator drives every draw in key order.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in state_dict_spec(cfg).items():
        if name == "length_mean":
            t = torch.tensor([float(length_mean)])
        elif name == "length_std":
            t = torch.tensor([float(length_std)])
        elif name == "edge_expand.centers":
            t = torch.tensor(np.linspace(0.0, 1.0, cfg.n_rbf)).float()
        elif name == "node_emb":
            t = torch.randn(shape, generator=g)
        elif name.endswith(".num_batches_tracked"):
            sd[name] = torch.tensor(1000, dtype=torch.int64)
            continue
        elif name.endswith(".running_mean"):
            t = 0.3 * torch.randn(shape, generator=g)
        elif name.endswith(".running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "norm" in name and name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:                      # bias: fan_in of the matching weight
                fan_in = sd[name[:-len("bias")] + "weight"].shape[1]
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
        sd[name] = t.to(torch.float32).contiguous()
    return sd


def infer_config(sd: Dict[str, torch.Tensor]) -> ModelConfig:
    """Recover the constructor arguments from a state_dict's shapes."""
    n_layers = 0
    while f"graph_conv.conv.{n_layers}.src_affine.weight" in sd:
        n_layers += 1
    if n_layers == 0:
        raise ValueError("state_dict has no graph_conv.conv.* layers")
    D, H = sd["graph_conv.conv.0.src_affine.weight"].shape
    Eh = sd["edge_layer_norm.weight"].shape[0]
    n_rbf = sd["edge_expand.centers"].shape[0] if "edge_expand.centers" in sd else 0
    edge_in = sd["edge_encoder.mlp_layer.0.weight"].shape[1]
    if "node_emb" in sd:
        kind, in_feats = "lj", 1
    else:
        kind, in_feats = "water", sd["node_encoder.weight"].shape[1]
    use_bond = edge_in == 3 + 1 + n_rbf + 1
    if edge_in not in (3 + 1 + n_rbf, 3 + 1 + n_rbf + 1):
        raise ValueError(f"unexpected edge_encoder input width {edge_in}")
    return ModelConfig(kind=kind, encoding_size=H, hidden_dim=D, edge_embedding_dim=Eh,
                       conv_layer=n_layers, in_feats=in_feats,
                       out_feats=sd["graph_decoder.mlp_layer.2.weight"].shape[0],
                       use_bond=use_bond, n_rbf=n_rbf,
                       use_layer_norm="graph_conv.norm_layers.0.running_mean" not in sd,
                       update_edge="graph_conv.conv.0.edge_layer_norm.weight" in sd)


def validate_state_dict(sd: Dict[str, torch.Tensor], cfg: ModelConfig) -> None:
    """Raise with the list of missing / unexpected / mis-shaped keys (strict, like
    ``nn.Module.load_state_dict`` used at LJ/train_network_lj.py:85-87)."""
    spec = state_dict_spec(cfg)
    missing = [k for k in spec if k not in sd]
    unexpected = [k for k in sd if k not in spec]
    bad = [f"{k}: {tuple(sd[k].shape)} != {spec[k]}" for k in spec
           if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
    if missing or unexpected or bad:
        raise KeyError(f"state_dict mismatch: missing={missing} unexpected={unexpected} shape={bad}")


def strip_prefix(sd: Dict[str, torch.Tensor], prefix: str = "pnet_model.") -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in sd.items():
        out[k[len(prefix):] if k.startswith(prefix) else k] = v
    return out


def load_checkpoint(path: str, allow_pickle: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Load either a plain state_dict (build_model, train_network_lj.py:85-87) or a
    Lightning checkpoint (``ckpt['state_dict']``, keys prefixed ``pnet_model.``).

    The file is read with torch's restricted unpickler (``weights_only=True``: tensors, containers and the
    argparse / SimpleNamespace hyper-parameter objects a Lightning checkpoint of the reference carries, nothing
    executable).  The reference itself calls the unrestricted ``torch.load`` (train_network_lj.py:85-87); a
    checkpoint that needs it must be opted into with ``allow_pickle=True`` — unpickling runs arbitrary code
    from the file, so only do that for files you trust."""
    import argparse
    import types
    try:
        with torch.serialization.safe_globals([argparse.Namespace, types.SimpleNamespace, OrderedDict]):
            obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:                          # pickle.UnpicklingError and friends
        if not allow_pickle:
            raise RuntimeError(
                f"{path}: not loadable with torch.load(weights_only=True) ({type(exc).__name__}: {exc}). "
                "If you trust the file, pass allow_pickle=True (load_checkpoint / load_from_checkpoint): the full "
                "unpickler executes code stored in the checkpoint.") from exc
        obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]
    sd = strip_prefix(obj)
    return OrderedDict((k, v.detach().contiguous() if k.endswith("num_batches_tracked") else v.detach().to(torch.float32).contiguous())
                       for k, v in sd.items() if isinstance(v, torch.Tensor))


def load_scaler(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """``scaler.npz`` -> (mean, var) float64 arrays (train_network_lj.py:119-123)."""
    info = np.load(path)
    return np.asarray(info["mean"], dtype=np.float64), np.asarray(info["var"], dtype=np.float64)


# Values of the four scaler.npz files shipped with the reference
# (code/LJ/model_ckpt_lj, code/water/model_ckpt_{tip3p,tip4p,dft}); the synthetic
# benchmarks denormalise with them so integrated trajectories stay finite.
SHIPPED_SCALERS = {
    "lj": (np.array([-7.07544845e-10]), np.array([1010.00278026])),
    "tip3p": (np.array([-4.80747553e-08]), np.array([349136.50451337])),
    "tip4p": (np.array([2.25095791e-08]), np.array([416596.56870182])),
    "dft": (np.array([-2.61092397e-13]), np.array([0.0005677])),
}
