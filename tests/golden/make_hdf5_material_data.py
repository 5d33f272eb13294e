"""The datasets of tests/golden/kappa-m999.hdf5 (make_hdf5_material.py writes them with h5py): a synthetic material in the phono3py schema (irreducible q-points + weights;
datasets mesh, qpoint, weight, frequency, group_velocity, temperature, gamma, gamma_isotope -- reference Phonon.py:158-187,
:316-324), on the silicon POSCAR lattice written next to it.  Needs h5py:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_material.py

Data only (the real Si / Ge phono3py files are missing from the reference checkout)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from nanokappa_amd import crystal, synthetic  # noqa: E402

POSCAR = """Si
   1.0
     0.000000   2.734364   2.734364
     2.734364   0.000000   2.734364
     2.734364   2.734364   0.000000
   Si
   2
Direct
   0.875   0.875   0.875
   0.125   0.125   0.125
"""


def datasets(n=9):
    full = synthetic.make_material(n, 'Si', temperatures=np.arange(250.0, 351.0, 50.0))
    path = os.path.join(HERE, 'POSCAR_Si')
    open(path, 'w').write(POSCAR)
    cell = crystal.read_poscar(path)
    rot = crystal.reciprocal_operations(cell['lattice'], cell['numbers'], cell['positions'])
    reps, weights = crystal.reduce_to_IBZ(full['q_points'], rot)
    g = full['gamma'][:, reps, :]
    return dict(mesh=np.array([n, n, n]), qpoint=full['q_points'][reps], weight=weights, frequency=full['frequency'][reps],
                group_velocity=full['group_vel'][reps], temperature=full['temperature'], gamma=g,
                gamma_isotope=np.where(g > 0, 0.25 * g + 1e-4, 0.0))


