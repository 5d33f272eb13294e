"""Golden vectors of the REFERENCE's material loader (classes/Phonon.py:158-187 load_hdf_data / load_q_points / load_weights /
load_frequency / load_group_vel / load_temperature, :316-324 load_gamma, :515-564 expand_FBZ) on tests/golden/kappa-m999.hdf5.

    /opt/conda/bin/python3.9 tests/golden/make_fbz.py        (build container only; writes tests/golden/fbz.npz)

The reference's `load_base_properties` (:66-149) needs phonopy for two things only: reading the POSCAR and the crystal's
reciprocal point-group operations.  Both are handed in from this package (nanokappa_amd.crystal); everything between them and
the FBZ tables is the reference's own code, called on `Phonon.__new__(Phonon)` in the order of :86-113.  Stored: the expanded
q-points, frequency, omega, group velocity (rounded as :102), gamma without and with the isotope part, the temperature array
and the rotations that were used (data only)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))

import ref_harness as RH  # noqa: E402

RH._install_stubs()
sys.path.insert(0, RH.REF)
from classes.Phonon import Phonon  # noqa: E402  (the reference's class)

from nanokappa_amd import crystal  # noqa: E402


def load(isotope):
    p = Phonon.__new__(Phonon)
    p.pi = np.pi
    p.mat_index = 0
    p.args = types.SimpleNamespace(isotope_scat=[0] if isotope else [])
    cell = crystal.read_poscar(os.path.join(HERE, 'POSCAR_Si'))
    lattice = cell['lattice']
    reciprocal_lattice = np.linalg.inv(lattice) * 2 * np.pi                          # Phonon.py:72
    rotations = crystal.reciprocal_operations(lattice, cell['numbers'], cell['positions'])
    p.load_hdf_data(os.path.join(HERE, 'kappa-m999.hdf5'))                           # :90
    p.load_q_points()
    p.load_weights()
    p.load_frequency()
    q_fbz, frequency = p.expand_FBZ(0, p.weights, p.q_points, p.frequency, 0, rotations, reciprocal_lattice)
    p.frequency = frequency
    p.convert_to_omega()
    p.load_group_vel()
    _, group_vel = p.expand_FBZ(0, p.weights, p.q_points, p.group_vel, 1, rotations, reciprocal_lattice)
    group_vel = np.around(group_vel, decimals=10)                                     # :102
    p.load_temperature()
    p.load_gamma()
    _, gamma = p.expand_FBZ(1, p.weights, p.q_points, p.gamma, 0, rotations, reciprocal_lattice)
    return dict(q_points=q_fbz, frequency=frequency, omega=p.omega, group_vel=group_vel, gamma=gamma,
                temperature=p.temperature_array, data_mesh=p.data_mesh, rotations=np.asarray(rotations),
                weights=np.asarray(p.weights), q_ibz=np.asarray(p.q_points))


if __name__ == '__main__':
    a, b = load(False), load(True)
    out = dict(a)
    out['gamma_with_isotope'] = b['gamma']
    for k in ('q_points', 'frequency', 'group_vel'):
        assert np.array_equal(a[k], b[k])
    np.savez_compressed(os.path.join(HERE, 'fbz.npz'), **out)
    print('wrote fbz.npz:', {k: np.asarray(v).shape for k, v in out.items()})
