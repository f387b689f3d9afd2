"""Drop-in for the reference's ``code/water/train_network_real_large.py`` (DFT water, WaterMDDynamicBoxNet) on the
force-inference path.

No module-level box or atom count in the reference (`water/train_network_real_large.py:24-31` has them commented out): the
box arrives with every ``predict_forces(feat, pos, box_size)`` call (`:148-162`), the cutoff is ``args.cutoff`` (`:119`),
the network's depth ``args.conv_layer`` (`:80`) and the atom count is whatever the first call brings.  ``create_water_bond``
as `:36-40`.  See gamd_amd/dropin/__init__.py.
"""
import numpy as np

from _gamd_dropin_common import compat, lightning_init, lightning_forward, add_training_stubs

LAMBDA1 = 100.
LAMBDA2 = 0.5e-2


def create_water_bond(total_atom_num):
    o = np.arange(0, total_atom_num, 3)
    return np.stack([np.repeat(o, 2), (o[:, None] + np.array([1, 2])).reshape(-1)], axis=1)


def build_model(args, ckpt=None):
    """WaterMDDynamicBoxNet of the reference = `ParticleNetLightning(args).pnet_model` here
    (``model([pos], feat, [box_size], cutoff)``)."""
    return ParticleNetLightning(args, model_weights_ckpt=ckpt).pnet_model


@add_training_stubs
class ParticleNetLightning(compat.ParticleNetLightningDFT):
    def __init__(self, args, num_device=1, epoch_num=100, batch_size=1, learning_rate=3e-4, log_freq=1000,
                 model_weights_ckpt=None, scaler_ckpt=None, **engine_kw):
        consts = dict(num_atoms=engine_kw.pop("num_atoms", None), box_size=engine_kw.pop("box_size", None),
                      cutoff=engine_kw.pop("cutoff", None))
        engine_kw.pop("bond", None)
        lightning_init(self, compat.ParticleNetLightningDFT, args, consts, num_device, epoch_num, batch_size, learning_rate,
                       log_freq, model_weights_ckpt, scaler_ckpt, **engine_kw)

    def _respawn(self, args, **kw):
        return type(self)(args, **{**self._ctor_kw, **kw})

    def forward(self, pos, feat, edge_idx_tsr):
        return lightning_forward(self, pos, feat, edge_idx_tsr)    # `:135-136`
