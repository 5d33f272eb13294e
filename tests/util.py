"""Shared helpers for the tests: golden loading and builders for the oracle structs."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d, prefix):
    p = prefix + '__'
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


def golden_material():
    """The test material exactly as handed to the reference (tests/golden/make_golden.py:material_small)."""
    g = golden('phonon')
    T = g['mat_temperature']
    gamma300 = g['mat_gamma_T300']
    # gamma = A*omega^2*T (synthetic.make_material); rebuilt bit-exactly from the stored omega
    from nanokappa_amd import synthetic
    gamma = 2.0e-9 * (g['mat_omega'] ** 2)[None, :, :] * T[:, None, None]
    gamma = np.where(gamma > 0, gamma, -1.0)
    assert np.array_equal(gamma[10], gamma300)
    return dict(data_mesh=g['mat_data_mesh'], q_points=g['mat_q_points'], omega=g['mat_omega'],
                frequency=g['mat_frequency'], group_vel=g['mat_group_vel'], temperature=T, gamma=gamma,
                reciprocal_lattice=g['mat_reciprocal_lattice'], volume_unitcell=float(g['mat_volume_unitcell']))


_PHONON = None


def golden_phonon():
    global _PHONON
    if _PHONON is None:
        from nanokappa_amd.phonon import Phonon
        _PHONON = Phonon(None, 0, material=golden_material())
    return _PHONON


def rel_err(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    scale = np.maximum(np.abs(b), 1e-300)
    with np.errstate(invalid='ignore'):
        e = np.abs(a - b) / scale
    e = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, e)
    return float(np.nanmax(e)) if e.size else 0.0
