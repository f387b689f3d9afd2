"""Drop-in for the reference's ``code/LJ/train_network_lj.py`` on the force-inference path (MI355X, libgamd_hip.so).

Module constants as `LJ/train_network_lj.py:26-29`; ``ParticleNetLightning(args, ...)`` as `:91-117`; ``predict_forces(pos,
verbose=False)`` as `:133-157`.  See gamd_amd/dropin/__init__.py.
"""
import numpy as np  # noqa: F401  (the reference module exposes np / torch to `from train_network_lj import *` users)

from _gamd_dropin_common import compat, lightning_init, lightning_forward, add_training_stubs

CUTOFF_RADIUS = 7.5
BOX_SIZE = 27.27

NUM_OF_ATOMS = 258

LAMBDA1 = 100.
LAMBDA2 = 1e-3


def build_model(args, ckpt=None):
    """The nn.Module of the reference (SimpleMDNetNew) is `ParticleNetLightning(args).pnet_model` here: callable as
    ``model([pos], [edge_idx])``, state_dict-compatible.  conv_layer is 4 whatever args says (`:75`)."""
    return ParticleNetLightning(args, model_weights_ckpt=ckpt).pnet_model


@add_training_stubs
class ParticleNetLightning(compat.ParticleNetLightningLJ):
    _FIXED_CONV_LAYER = 4

    def __init__(self, args, num_device=1, epoch_num=100, batch_size=1, learning_rate=3e-4, log_freq=1000,
                 model_weights_ckpt=None, scaler_ckpt=None, **engine_kw):
        consts = dict(num_atoms=engine_kw.pop("num_atoms", NUM_OF_ATOMS), box_size=engine_kw.pop("box_size", BOX_SIZE),
                      cutoff=engine_kw.pop("cutoff", CUTOFF_RADIUS))
        engine_kw.pop("bond", None)
        lightning_init(self, compat.ParticleNetLightningLJ, args, consts, num_device, epoch_num, batch_size, learning_rate,
                       log_freq, model_weights_ckpt, scaler_ckpt, **engine_kw)

    def _respawn(self, args, **kw):
        return type(self)(args, **{**self._ctor_kw, **kw})

    def forward(self, pos, feat, edge_idx_tsr):
        return lightning_forward(self, pos, edge_idx_tsr)          # `:125-126` (feat is unused by SimpleMDNetNew's call there)
