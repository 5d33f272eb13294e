"""The boundary of SURVEY 8b with the REFERENCE's objects on the other side: `nanokappa_amd.Population(args, geo, ph)`
was constructed in the build container from the reference's own `Geometry` and `Phonon` (tests/golden/make_dropin.py,
the INTEGRATION.md section 1 flow) with a recording engine; tests/golden/dropin.npz holds what it uploaded.
Here: those tables equal the ones this package's own Geometry / Phonon produce for the same arguments, the oracle
runs on them, and (GPU) the real engine fed with them follows the oracle particle by particle."""
import os
import sys

import numpy as np
import pytest

from util import case_from_dropin, case_tables, golden_phonon, make_engine, make_oracle_sim, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))


@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_tables_from_reference_objects_equal_own(case):
    ct = case_from_dropin(case)
    own = case_tables(case)
    t, o = ct['tables'], own['tables']
    for k in ('omega', 'group_vel', 'T_grid', 'lifetime', 'hbar', 'kb', 'QV', 'active_modes'):
        assert rel_err(t[k], o[k]) < 1e-14, k
    assert rel_err(t['T_array'], o['T_array']) < 1e-13 and rel_err(t['energy_array'], o['energy_array']) < 1e-12
    m, om = ct['mesh'], own['mesh']
    for k in ('face_normals', 'face_k', 'face_bounds', 'face_basis_matrix', 'face_origins', 'facets_normal', 'facet_centroid', 'bounds'):
        assert np.allclose(m[k], om[k], rtol=0, atol=1e-12), k
    assert np.array_equal(m['face_facets'], om['face_facets']) and np.array_equal(m['bound_cond'], om['bound_cond'])
    assert np.allclose(ct['centers'], own['centers']) and np.allclose(ct['volumes'], own['volumes'])
    assert np.array_equal(ct['res_facets'], own['res_facets']) and np.array_equal(ct['res_T'], own['res_T'])
    # the fixture was made at 2e4 particles, the set-up golden at 1e5: entry probabilities scale with the density
    scale = ct['particle_density'] / own['particle_density']
    assert rel_err(ct['enter_prob'], own['enter_prob'] * scale) < 1e-12
    if case == 'ttrrp':
        for k in ('specularity', 'roulette'):
            assert rel_err(ct['rough'][k], own['rough'][k]) < 1e-12, k
        assert np.array_equal(ct['rough']['true_spec'], own['rough']['true_spec'])
        assert np.array_equal(ct['rough']['spec_map'], own['rough']['spec_map'])
    # particles: inside the box, active modes, Bose-Einstein at the cold reservoir temperature (Population.py:280)
    ph = golden_phonon()
    b = m['bounds']
    assert ct['positions'].shape == (20000, 3) and np.all(ct['positions'] >= b[0]) and np.all(ct['positions'] <= b[1])
    assert not ph.inactive_modes_mask.ravel()[ct['mode']].any()
    assert rel_err(ct['occ'], ph.calculate_occupation(298.0, ph.omega.ravel()[ct['mode']])) < 1e-13


@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_oracle_runs_on_the_dropin_tables(case):
    ct = case_from_dropin(case)
    sim = make_oracle_sim(ct, ct['positions'], ct['mode'], ct['occ'], ct['counter'], seed=3, interp=ct['interp'])
    for _ in range(3):
        sim.run_timestep()
    assert int(sim.N_sv.sum()) > 15000 and np.all(np.abs(sim.T_sv - 298.0) < 4.5)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_engine_on_dropin_tables_follows_oracle(case):
    """25 steps of the real engine on the tables built from the reference's objects, against the oracle."""
    ct = case_from_dropin(case)
    sim = make_oracle_sim(ct, ct['positions'], ct['mode'], ct['occ'], ct['counter'], seed=3, interp=ct['interp'])
    eng = make_engine(ct, ct['positions'], ct['mode'], ct['occ'], ct['counter'], seed=3, interp=ct['interp'])
    t = eng.step(25)
    for s in range(25):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert np.allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=1e-8), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2]) and np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert np.allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=1e-10, atol=1e-8)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < 1e-8
