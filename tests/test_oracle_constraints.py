"""The oracle's rigid-water constraint solvers (SHAKE / RATTLE iterated to convergence): the properties that
define the constrained update uniquely — constraints met, displacements along the reference bonds weighted by
1/m (so linear momentum is conserved), idempotence."""
import numpy as np

import gamd_oracle as orc
from gamd_amd import workloads as wl


def _setup(n_mol=27, seed=1):
    pos, box, species, bonds = wl.water_box(n_mol, seed=seed, jitter=0.0, wrap=False)
    mass = np.where(species == 1, wl.MASS_O, wl.MASS_H).astype(np.float64)
    pairs, lengths = orc.water_constraints(pos.shape[0], wl.TIP3P_R_OH, wl.TIP3P_R_HH)
    return pos, mass, pairs, lengths


def test_generated_molecules_are_rigid_tip3p():
    pos, mass, pairs, lengths = _setup()
    d = np.linalg.norm(pos[pairs[:, 0]] - pos[pairs[:, 1]], axis=1)
    assert np.abs(d - lengths).max() < 1e-12
    assert abs(wl.TIP3P_R_HH - 1.5139) < 1e-4


def test_shake_positions_properties():
    pos, mass, pairs, lengths = _setup()
    rng = np.random.default_rng(0)
    x1 = pos + rng.normal(0, 0.03, pos.shape)
    xc = orc.shake_positions(pos, x1, 1.0 / mass, pairs, lengths)
    d = np.linalg.norm(xc[pairs[:, 0]] - xc[pairs[:, 1]], axis=1)
    assert np.abs(d - lengths).max() < 1e-12
    # constraint displacements carry no net momentum per molecule
    dp = (mass[:, None] * (xc - x1)).reshape(-1, 3, 3).sum(axis=1)
    assert np.abs(dp).max() < 1e-10
    # ... and lie in the plane of the reference molecule (sums of reference bond vectors)
    ref = pos.reshape(-1, 3, 3)
    normal = np.cross(ref[:, 1] - ref[:, 0], ref[:, 2] - ref[:, 0])
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    off = np.einsum("mkd,md->mk", (xc - x1).reshape(-1, 3, 3), normal)
    assert np.abs(off).max() < 1e-12
    # already constrained input is a fixed point
    assert np.abs(orc.shake_positions(pos, xc, 1.0 / mass, pairs, lengths) - xc).max() < 1e-12


def test_rattle_velocities_properties():
    pos, mass, pairs, lengths = _setup()
    rng = np.random.default_rng(1)
    v = rng.normal(0, 5.0, pos.shape)
    vc = orc.rattle_velocities(pos, v, 1.0 / mass, pairs)
    r = pos[pairs[:, 0]] - pos[pairs[:, 1]]
    assert np.abs(np.sum(r * (vc[pairs[:, 0]] - vc[pairs[:, 1]]), axis=1)).max() < 1e-12
    dp = (mass[:, None] * (vc - v)).reshape(-1, 3, 3).sum(axis=1)
    assert np.abs(dp).max() < 1e-10
    # a rigid-body motion (translation + rotation about the centre of mass) is left untouched
    m3, x3 = mass.reshape(-1, 3, 1), pos.reshape(-1, 3, 3)
    com = (m3 * x3).sum(axis=1, keepdims=True) / m3.sum(axis=1, keepdims=True)
    omega = rng.normal(size=(x3.shape[0], 1, 3))
    vrig = (rng.normal(size=(x3.shape[0], 1, 3)) + np.cross(omega, x3 - com)).reshape(-1, 3)
    assert np.abs(orc.rattle_velocities(pos, vrig, 1.0 / mass, pairs) - vrig).max() < 1e-12
