"""Shared by the drop-in modules: the reference's ParticleNetLightning constructor signature
(`LJ/train_network_lj.py:91-93`) on top of the gamd_amd.compat wrappers."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import gamd_amd  # noqa: F401
except ImportError:                                   # only this directory is on sys.path: add the repository root
    sys.path.insert(0, _ROOT)

import torch  # noqa: E402

from gamd_amd import compat  # noqa: E402
from gamd_amd.weights import load_checkpoint  # noqa: E402

_TRAINING_ONLY = ("training_step", "validation_step", "configure_optimizers", "train_dataloader", "val_dataloader",
                  "training_epoch_end", "validation_epoch_end", "fit")


def lightning_init(self, base, args, consts, num_device, epoch_num, batch_size, learning_rate, log_freq, model_weights_ckpt,
                   scaler_ckpt, **engine_kw):
    """Body of ParticleNetLightning.__init__(args, num_device=1, epoch_num=100, batch_size=1, learning_rate=3e-4,
    log_freq=1000, model_weights_ckpt=None, scaler_ckpt=None): the module constants size the system, the Lightning
    bookkeeping attributes are kept as plain attributes, ``model_weights_ckpt`` is a bare state_dict file loaded into
    ``pnet_model`` (build_model(args, ckpt): `model.load_state_dict(torch.load(ckpt))`, LJ/train_network_lj.py:85-87)."""
    base.__init__(self, args, None, scaler_ckpt=scaler_ckpt, **consts, **engine_kw)
    self.epoch_num, self.learning_rate, self.batch_size = epoch_num, learning_rate, batch_size
    self.num_device, self.log_freq = num_device, log_freq
    for name in ("rotate_aug", "data_dir", "use_part"):
        if hasattr(self.args, name):
            setattr(self, name, getattr(self.args, name))
    if hasattr(self.args, "loss"):
        self.loss_fn = self.args.loss
        assert self.loss_fn in ["mae", "mse"]                      # LJ/train_network_lj.py:117
    if model_weights_ckpt is not None:
        print('Loading model weights from: ', model_weights_ckpt)
        self.load_state_dict(load_checkpoint(model_weights_ckpt))


def training_only(name):
    def stub(self, *a, **kw):
        raise NotImplementedError(f"ParticleNetLightning.{name}: the gfx950 drop-in covers the force-inference path "
                                  "(predict_forces / forward); training stays with the reference (SURVEY.md section 8)")
    stub.__name__ = name
    return stub


def add_training_stubs(cls):
    for name in _TRAINING_ONLY:
        setattr(cls, name, training_only(name))
    return cls


def lightning_forward(self, pos, *rest):
    """ParticleNetLightning.forward: denormalize(pnet_model(...)) (LJ/train_network_lj.py:125-126,
    water/train_network_tip3p.py:136-137) — the model-level call on a given edge list, on the device."""
    pos = [pos] if torch.is_tensor(pos) else pos
    rest = tuple(r if i + 1 < len(rest) or not torch.is_tensor(r) else [r] for i, r in enumerate(rest))   # the edge tensor -> [edge_idx]
    out = self.pnet_model(pos, *rest)
    mean = torch.as_tensor(self.training_mean, dtype=torch.float64, device=out.device)
    var = torch.as_tensor(self.training_var, dtype=torch.float64, device=out.device)
    return out * torch.sqrt(var) + mean
