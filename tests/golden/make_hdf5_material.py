"""Writes tests/golden/kappa-m999.hdf5: a synthetic material in the phono3py schema (irreducible q-points + weights;
datasets mesh, qpoint, weight, frequency, group_velocity, temperature, gamma, gamma_isotope -- reference Phonon.py:158-187,
:316-324), on the silicon POSCAR lattice written next to it.  Needs h5py:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_material.py

Data only (the real Si / Ge phono3py files are missing from the reference checkout)."""
import os
import sys

import numpy as np
import h5py

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
sys.path.insert(0, HERE)

from make_hdf5_material_data import datasets  # noqa: E402


if __name__ == '__main__':
    d = datasets()
    with h5py.File(os.path.join(HERE, 'kappa-m999.hdf5'), 'w') as f:
        for k, v in d.items():
            f.create_dataset(k, data=v)
    print('wrote kappa-m999.hdf5:', {k: v.shape for k, v in d.items()})
