"""Drop-in for the reference's ``code/water/train_network_tip3p.py`` on the force-inference path.

Module constants as `water/train_network_tip3p.py:24-32`; ``create_water_bond`` as `:38-42`; ``ParticleNetLightning(args,
...)`` as `:100-128`; ``predict_forces(feat, pos)`` as `:142-159`.  See gamd_amd/dropin/__init__.py.
"""
import numpy as np

from _gamd_dropin_common import compat, lightning_init, lightning_forward, add_training_stubs

# for water box
CUTOFF_RADIUS = 4.2
left_bound = 0.0
right_bound = 20.0
BOX_SIZE = right_bound - left_bound

NUM_OF_ATOMS = 258 * 3

LAMBDA1 = 100.
LAMBDA2 = 1e-3


def create_water_bond(total_atom_num):
    """[[O, H1], [O, H2]] per molecule, atoms ordered O,H,H."""
    o = np.arange(0, total_atom_num, 3)
    return np.stack([np.repeat(o, 2), (o[:, None] + np.array([1, 2])).reshape(-1)], axis=1)


def build_model(args, ckpt=None):
    """WaterMDNetNew of the reference = `ParticleNetLightning(args).pnet_model` here (``model([pos], feat, [edge_idx])``)."""
    return ParticleNetLightning(args, model_weights_ckpt=ckpt).pnet_model


@add_training_stubs
class ParticleNetLightning(compat.ParticleNetLightningWater):
    _FIXED_CONV_LAYER = 4                                          # `:85`

    def __init__(self, args, num_device=1, epoch_num=100, batch_size=1, learning_rate=3e-4, log_freq=1000,
                 model_weights_ckpt=None, scaler_ckpt=None, **engine_kw):
        n = engine_kw.pop("num_atoms", NUM_OF_ATOMS)
        consts = dict(num_atoms=n, box_size=engine_kw.pop("box_size", BOX_SIZE), cutoff=engine_kw.pop("cutoff", CUTOFF_RADIUS),
                      bond=engine_kw.pop("bond", None))
        if consts["bond"] is None:
            consts["bond"] = create_water_bond(n)                  # build_model: bond_info = create_water_bond(NUM_OF_ATOMS)
        lightning_init(self, compat.ParticleNetLightningWater, args, consts, num_device, epoch_num, batch_size, learning_rate,
                       log_freq, model_weights_ckpt, scaler_ckpt, **engine_kw)

    def _respawn(self, args, **kw):
        return type(self)(args, **{**self._ctor_kw, **kw})

    def forward(self, pos, feat, edge_idx_tsr):
        return lightning_forward(self, pos, feat, edge_idx_tsr)    # `:136-137`
