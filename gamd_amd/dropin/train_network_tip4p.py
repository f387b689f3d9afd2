"""Drop-in for the reference's ``code/water/train_network_tip4p.py`` on the force-inference path: the TIP3P wrapper with
``NUM_OF_ATOMS = 251 * 3`` (`water/train_network_tip4p.py:30`; the M-site is dropped before the network,
train_utils.py:58-59).  See gamd_amd/dropin/__init__.py."""
import numpy as np  # noqa: F401

from _gamd_dropin_common import compat, lightning_init, lightning_forward, add_training_stubs
from train_network_tip3p import create_water_bond  # noqa: F401  (same function body in both reference modules)

# for water box
CUTOFF_RADIUS = 4.2
left_bound = 0.0
right_bound = 20.0
BOX_SIZE = right_bound - left_bound

NUM_OF_ATOMS = 251 * 3  # 258 *3

LAMBDA1 = 100.
LAMBDA2 = 1e-3


def build_model(args, ckpt=None):
    return ParticleNetLightning(args, model_weights_ckpt=ckpt).pnet_model


@add_training_stubs
class ParticleNetLightning(compat.ParticleNetLightningWater):
    _FIXED_CONV_LAYER = 4

    def __init__(self, args, num_device=1, epoch_num=100, batch_size=1, learning_rate=3e-4, log_freq=1000,
                 model_weights_ckpt=None, scaler_ckpt=None, **engine_kw):
        n = engine_kw.pop("num_atoms", NUM_OF_ATOMS)
        consts = dict(num_atoms=n, box_size=engine_kw.pop("box_size", BOX_SIZE), cutoff=engine_kw.pop("cutoff", CUTOFF_RADIUS),
                      bond=engine_kw.pop("bond", None))
        if consts["bond"] is None:
            consts["bond"] = create_water_bond(n)
        lightning_init(self, compat.ParticleNetLightningWater, args, consts, num_device, epoch_num, batch_size, learning_rate,
                       log_freq, model_weights_ckpt, scaler_ckpt, **engine_kw)

    def _respawn(self, args, **kw):
        return type(self)(args, **{**self._ctor_kw, **kw})

    def forward(self, pos, feat, edge_idx_tsr):
        return lightning_forward(self, pos, feat, edge_idx_tsr)
