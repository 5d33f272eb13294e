"""Host-side material tables vs the reference's Phonon (golden: tests/golden/phonon.npz)."""
import numpy as np

from util import golden, golden_phonon, rel_err


def test_wavevectors_and_counts():
    g = golden('phonon')
    ph = golden_phonon()
    assert ph.number_of_active_modes == int(g['number_of_active_modes'])
    assert np.array_equal(ph.inactive_modes_mask, g['inactive_modes_mask'])
    # |k| must agree; on the zone boundary several images tie to the last bit and the reference's
    # own pick depends on NumPy's rounding, so there the difference must be a reciprocal lattice vector
    assert np.allclose(np.linalg.norm(ph.wavevectors, axis=1), np.linalg.norm(g['wavevectors'], axis=1),
                       rtol=0, atol=1e-12)
    dq = ph.k_to_q(ph.wavevectors - g['wavevectors'])
    assert np.allclose(dq, np.round(dq), atol=1e-9)
    assert (np.abs(ph.wavevectors - g['wavevectors']).max(axis=1) > 1e-12).sum() < 0.05 * ph.number_of_qpoints


def test_lifetime_table_and_function():
    g = golden('phonon')
    ph = golden_phonon()
    assert rel_err(ph.lifetime, g['lifetime']) < 1e-14
    tau = ph.lifetime_function(np.vstack((g['s_T'], g['s_q'], g['s_j'])).T)
    assert rel_err(tau, g['s_tau']) < 1e-12


def test_occupation():
    g = golden('phonon')
    ph = golden_phonon()
    om = ph.omega[g['s_q'], g['s_j']]
    assert rel_err(ph.calculate_occupation(g['s_T'], om), g['s_occ']) < 1e-13
    Tz = g['s_T'].copy()
    Tz[:10] = 0.0
    assert rel_err(ph.calculate_occupation(Tz, om), g['s_occ_T0']) < 1e-13


def test_energy_temperature_tables():
    g = golden('phonon')
    ph = golden_phonon()
    assert rel_err(ph.zero_point, g['zero_point']) < 1e-14
    assert ph.energy_array.shape == g['energy_array'].shape
    assert rel_err(ph.energy_array, g['energy_array']) < 1e-13
    assert rel_err(ph.temperature_function(g['s_E']), g['s_T_of_E']) < 1e-9   # inverse amplifies E rounding
    assert rel_err(ph.crystal_energy_function(g['s_Tw']), g['s_E_of_T']) < 1e-13
    assert rel_err(ph.calculate_crystal_energy(g['s_T'][:50]), g['s_crystal_energy_exact']) < 1e-13


def test_find_min_k():
    g = golden('phonon')
    ph = golden_phonon()
    kmin, disp = ph.find_min_k(g['s_k'].copy(), return_disp=True)
    assert np.allclose(kmin, g['s_kmin'], atol=1e-12)
    assert np.allclose(disp, g['s_kdisp'], atol=1e-12)
