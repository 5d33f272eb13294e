"""The boundary of SURVEY 8b with the REFERENCE's objects on the other side: `nanokappa_amd.Population(args, geo, ph)`
was constructed in the build container from the reference's own `Geometry` and `Phonon` (tests/golden/make_dropin.py,
the INTEGRATION.md section 1 flow) with a recording engine; tests/golden/dropin.npz holds what it uploaded.
Here: those tables equal the ones this package's own Geometry / Phonon produce for the same arguments, the oracle
runs on them, and (GPU) the real engine fed with them follows the oracle particle by particle."""
import os
import sys

import numpy as np
import pytest

from util import case_from_dropin, case_tables, golden_phonon, make_engine, make_oracle_sim, rel_err, allclose, TOL_T, TOL_X, TOL_OCC

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
        assert allclose(m[k], om[k], rtol=0, atol=1e-12), k
    assert np.array_equal(m['face_facets'], om['face_facets']) and np.array_equal(m['bound_cond'], om['bound_cond'])
    assert allclose(ct['centers'], own['centers'], rtol=1e-12) and allclose(ct['volumes'], own['volumes'], rtol=1e-12)
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
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2]) and np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=0, atol=TOL_X)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


# ---- the constructor's DEVICE-builder branch with the reference's objects (round 3): tests/golden/make_dropin.py's
# DeviceRecorder accepted rough_begin / specular_pairs / rough_pairs / rough_finish / build_enter_prob / init_particles, so
# `Population(args, <reference Geometry>, <reference Phonon>)` took the branch a GPU run takes (reference flow
# classes/Population.py:80-123); the fixture holds the ARGUMENTS it handed to those builders.
def _dev(case):
    from util import golden, sub
    return sub(golden('dropin'), case + '_dev')


@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_device_branch_arguments_from_reference_objects(case):
    """What the constructor computed from the reference's Geometry / Phonon for the device builders equals what it computes
    from this package's own objects: inward normals, boundary thickness, subvolume shares, the rough facets' normals / eta /
    |k|, the pair search's inputs."""
    d = _dev(case)
    ct = case_from_dropin(case)
    own = case_tables(case)
    ph = golden_phonon()
    m = ct['mesh']
    ea = {k.split('__', 1)[1]: v for k, v in d.items() if k.startswith('enter_prob_args__')}
    assert allclose(ea['normal_in'], -m['facets_normal'][ct['res_facets']], rtol=0, atol=1e-12) and float(ea['dt']) == 1.0
    # thickness = M / (rho A): the fixture was made at 2e5 particles in the 200 A box with 200 x 200 facets
    rho = 200000 / 200.0 ** 3
    assert allclose(ea['thickness'], ph.number_of_active_modes / (rho * 200.0 * 200.0), rtol=1e-12)
    ip = {k.split('__', 1)[1]: v for k, v in d.items() if k.startswith('init_particles__')}
    assert int(ip['n']) == 200000 and int(ip['pid_lo']) == 0 and int(d['N_p']) == 200000
    active = np.nonzero(~ph.inactive_modes_mask.ravel())[0]
    assert np.array_equal(ip['unique_modes'], active)
    sf = ip['sv_first']
    assert sf.shape == (21,) and sf[0] == 0 and sf[-1] == 200000 and np.all(np.diff(sf) == 10000)   # 20 equal slices
    if case == 'ttrrp':
        assert bool(d['rough_on_device'])
        rb = {k.split('__', 1)[1]: v for k, v in d.items() if k.startswith('rough_begin__')}
        assert np.array_equal(rb['facets'], own['rough']['facets'])
        assert allclose(rb['normal_in'], -m['facets_normal'][rb['facets']], rtol=0, atol=1e-12)
        assert np.array_equal(rb['eta'], [5.0, 5.0])
        assert allclose(rb['k_norm'], np.sum(ph.wavevectors ** 2, axis=1) ** 0.5, rtol=1e-13)
        sb = {k.split('__', 1)[1]: v for k, v in d.items() if k.startswith('spec_begin__')}
        assert allclose(sb['group_vel'], ph.group_vel.reshape(-1, 3), rtol=0, atol=1e-12)
        assert allclose(sb['omega'], ph.omega.ravel(), rtol=1e-14)
        assert d['spec_pairs__normals'].shape == (2, 3) and np.array_equal(np.sort(d['rough_pairs__share_flat']), [0, 1])
    else:
        assert not bool(d['rough_on_device'])


@pytest.mark.gpu
def test_device_builders_replay_reference_arguments():
    """The recorded arguments through the REAL builders on the GPU: nk_build_enter_prob reproduces the enter_prob the NumPy
    branch computed from the same reference objects (scaled by the particle density), nk_rough_begin / nk_specular_pairs /
    nk_rough_pairs / nk_rough_finish reproduce its specularity, truly-specular mask, specular map and roulette, and
    nk_init_particles + nk_tally_state create exactly the recorded shares."""
    from nanokappa_amd.engine import Engine
    d = _dev('ttrrp')
    ct = case_from_dropin('ttrrp')
    g = lambda name: {k.split('__', 1)[1]: v for k, v in d.items() if k.startswith(name + '__')}
    eng = Engine(0, 7)
    eng.set_material(ct['tables'])
    eng.set_mesh(ct['mesh'])
    ea = g('enter_prob_args')
    ep = eng.build_enter_prob(ea['normal_in'], ea['thickness'], float(ea['dt']))
    scale = 200000 / 20000.0                                     # density of this fixture / of the NumPy-branch fixture
    assert rel_err(ep, ct['enter_prob'] * scale) < 1e-12
    sb, rb = g('spec_begin'), g('rough_begin')
    eng.specular_begin(sb['group_vel'], sb['omega'], sb['delta_omega'])
    eng.rough_begin(rb['facets'], rb['normal_in'], rb['eta'], rb['k_norm'])
    shares = np.split(d['rough_pairs__share_flat'], np.cumsum(d['rough_pairs__share_len'])[:-1])
    for nrm, share in zip(d['spec_pairs__normals'], shares):
        eng.specular_pairs(nrm, float(d['spec_pairs__crit']), download=False)
        eng.rough_pairs(share)
    eng.rough_finish()
    eng.specular_end()
    sp, ts, sm, ro = eng.rough_download()
    r = ct['rough']
    assert np.array_equal(ts.astype(bool), r['true_spec'].astype(bool))
    assert np.array_equal(sm[ts.astype(bool)], r['spec_map'][r['true_spec'].astype(bool)])
    assert rel_err(sp, r['specularity']) < 5e-13
    # the roulette is a cumulative sum of creation rates that depend on the density only through a common factor
    assert rel_err(ro / ro[:, -1:], r['roulette'] / r['roulette'][:, -1:]) < 1e-13
    # particles: the recorded shares, created on the device
    eng.set_subvolumes(ct['centers'], ct['volumes'], ct['kind'], ct['axis'], ct['interp'], ct['T_sv'])
    eng.set_reservoirs(ct['res_facets'], ct['res_T'], ep, np.random.default_rng(1).random(ep.shape))
    eng.set_params(dt=1.0, particle_density=ct['particle_density'] * scale)
    ip = g('init_particles')
    eng.init_particles(int(ip['n']), int(ip['capacity']), int(ip['pid_lo']), ip['unique_modes'], ip['sv_first'])
    eng.init_boundaries()
    E, N, F = eng.tally_state()
    assert np.array_equal(N, np.diff(ip['sv_first']).astype(float))
    p = eng.download()
    assert p['mode'].shape[0] == 200000 and set(np.unique(p['mode'])) == set(ip['unique_modes'].tolist())
    t = eng.step(3)
    assert t['N_sv'].sum(axis=1).min() > 190000
    eng.close()
