"""Synthetic inputs of the BASELINE.json configs.  Pure numpy, seeded; used by
bench.py, smoke() and the tests so that every leg sees the same boxes."""
from __future__ import annotations

import numpy as np

KB = 0.00831446261815324        # kJ/mol/K
LJ_SIGMA = 3.4                  # Angstrom; openmmtools LennardJonesFluid argon default
                                # (dataset/generate_lj_data.py:55-56 uses reduced_density=0.5)


def lj_box(n_atoms: int = 10000, rho_star: float = 0.5, sigma: float = LJ_SIGMA, seed: int = 1234,
           jitter: float = 0.05):
    """C2/C4: simple-cubic lattice (first n sites) scaled to the cubic box of reduced density
    rho*, plus N(0, jitter*sigma) noise.  Returns (pos f64 [n,3] Angstrom, box_length)."""
    box = (n_atoms / rho_star) ** (1.0 / 3.0) * sigma
    m = int(np.ceil(n_atoms ** (1.0 / 3.0) - 1e-9))
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    pos = (g[:n_atoms] + 0.5) * (box / m)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, jitter * sigma, pos.shape)
    return np.mod(pos, box), float(box)


def maxwell_boltzmann(n_atoms: int, temperature_k: float = 100.0, mass_amu: float = 39.9, seed: int = 99):
    """velocities in Angstrom/ps (setVelocitiesToTemperature, LJ/test_script/test_langevin.py:53)."""
    rng = np.random.default_rng(seed)
    return rng.normal(0.0, 10.0 * np.sqrt(KB * temperature_k / mass_amu), (n_atoms, 3))


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)


TIP3P_R_OH = 0.9572
TIP3P_R_HH = 2.0 * 0.9572 * float(np.sin(np.deg2rad(104.52) / 2.0))       # 1.5139 A
MASS_O, MASS_H = 15.99943, 1.007947                                        # OpenMM tip3p.xml
BOHR_PER_NM = 18.8972613


def water_box(n_mol: int = 1390, mol_per_20A3: float = 258.0, seed: int = 2345, jitter: float = 0.02,
              wrap: bool = True):
    """C3 (1390 TIP3P molecules, 4170 atoms) / C5 (2000 molecules): rigid molecules (O-H 0.9572 A,
    104.52 deg) on a cubic lattice at the reference density (258 molecules per (20 A)^3,
    dataset/generate_tip3p_data.py:55-57), random orientations.  Atom order O,H,H
    (train_utils.py:25-26).  Returns (pos f64 [3*n_mol,3], box, species u8 [3*n_mol], bonds [2*n_mol,2]).
    ``jitter`` (A, per atom) makes the network input generic; ``jitter=0, wrap=False`` gives exactly rigid,
    whole molecules for the constrained integrators."""
    box = 20.0 * (n_mol / mol_per_20A3) ** (1.0 / 3.0)
    m = int(np.ceil(n_mol ** (1.0 / 3.0) - 1e-9))
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    com = (g[:n_mol] + 0.5) * (box / m)
    rng = np.random.default_rng(seed)
    r, ang = 0.9572, np.deg2rad(104.52)
    local = np.array([[0.0, 0.0, 0.0], [r, 0.0, 0.0], [r * np.cos(ang), r * np.sin(ang), 0.0]])
    rot = _random_rotations(rng, n_mol)
    pos = com[:, None, :] + np.einsum("nij,kj->nki", rot, local)
    pos = pos.reshape(-1, 3) + rng.normal(0.0, 0.02, (3 * n_mol, 3)) * (jitter / 0.02)
    species = np.tile(np.array([1, 0, 0], dtype=np.uint8), n_mol)
    o = np.arange(0, 3 * n_mol, 3)
    bonds = np.stack([np.repeat(o, 2), (o[:, None] + np.array([1, 2])).reshape(-1)], axis=1).astype(np.int32)
    return (np.mod(pos, box) if wrap else pos), float(box), species, bonds
