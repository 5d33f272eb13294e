"""Parity of the HIP engine (through the C ABI) against the reference goldens and the CPU oracle.
Needs a real MI355X: run with `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest

from util import golden, sub, golden_phonon, rel_err, case_tables, random_population, make_oracle_sim, make_engine, allclose, same_event_rule, TOL_T, TOL_X, TOL_X_LONG, TOL_NTS, TOL_OCC, TOL_OCC_GRID, TOL_E, TOL_RES

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


def base_engine(name, interp=1, seed=1):
    from nanokappa_amd.engine import Engine
    g = sub(golden('mesh'), name)
    ph = golden_phonon()
    eng = Engine(0, seed)
    eng.set_material(ph.tables())
    eng.set_mesh(g)
    eng.set_subvolumes(g['subvol_center'], g['subvol_volume'], 0, int(g['slice_axis']), interp,
                       np.full(g['subvol_center'].shape[0], 300.0))
    return eng, g, ph


@pytest.mark.parametrize('name', ['box200', 'box200ttp', 'box5000', 'cyl'])
def test_find_boundary(name):
    eng, g, ph = base_engine(name)
    xc, tc, fc = eng.find_boundary(g['ray_x'], g['ray_v'])
    assert np.array_equal(fc, g['ray_fc'])
    hit = fc >= 0
    assert rel_err(tc[hit], g['ray_tc'][hit]) < 1e-12
    assert np.all(np.isinf(tc[~hit]))
    assert allclose(xc[hit], g['ray_xc'][hit], rtol=1e-12, atol=1e-11)


LARGE_MESHES = {
    'wire1600': ['--geometry', 'cylinder', '--dimensions', '2000', '200', '400', '--subvolumes', 'slice', '20', '2'],
    # diagonal slivers everywhere: the star's flanks and cap fans are split into several tree references each
    'star': ['--geometry', 'star', '--dimensions', '600', '200', '90', '72', '--subvolumes', 'slice', '4', '2'],
}


@pytest.mark.parametrize('name', sorted(LARGE_MESHES))
def test_find_boundary_large_mesh_tree(name):
    """Meshes whose tables stay in global memory: every lane walks the 4-ary box tree over the (split) faces.
    Must equal the plain all-faces evaluation (this package's NumPy Mesh.find_boundary, itself checked against the
    reference goldens in test_host_geometry) ray by ray, including rays that start on the surface, outside, or run
    along an axis."""
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.engine import Engine
    argv = LARGE_MESHES[name] + ['--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                                 '--bound_values', '302', '298', '5'] + COMMON_ARGS
    args = initialise_parser().parse_args(argv)
    args.results_folder = ''
    geo = Geometry(args)
    assert geo.mesh.n_of_faces > 256
    ph = golden_phonon()
    eng = Engine(0, 1)
    eng.set_material(ph.tables())
    eng.set_mesh(geo.tables())
    eng.set_subvolumes(geo.subvol_center, geo.subvol_volume, 0, geo.slice_axis, 1, np.full(geo.n_of_subvols, 300.0))
    rng = np.random.default_rng(5)
    n = 4000
    b = geo.mesh.bounds
    x = geo.mesh.sample_volume(n, rng)
    v = rng.normal(size=(n, 3)) * 40.0
    v[:50, :2] = 0.0                                          # along the wire axis
    v[50:100, 2] = 0.0                                        # in the cross-section
    x[100:150] = b[1] + 7.0                                   # outside: mostly misses
    x[150:200, 2] = b[0, 2]                                   # on an end cap
    # from points of the surface itself (where reflected and entering particles start): first hits of other rays
    x0, t0, f0 = geo.mesh.find_boundary(x[1000:3000].copy(), v[1000:3000].copy())
    ok = np.isfinite(t0)
    x[1000:3000][ok] = x0[ok]
    v[1000:3000] = rng.normal(size=(2000, 3)) * 40.0
    xr, tr, fr = geo.mesh.find_boundary(x.copy(), v.copy())
    xc, tc, fc = eng.find_boundary(x, v)
    assert np.array_equal(fc, fr)
    hit = fr >= 0
    assert hit.sum() > 2500
    assert rel_err(tc[hit], tr[hit]) < 1e-12
    assert np.all(np.isinf(tc[~hit]))


def test_find_boundary_large_mesh_without_tree(monkeypatch):
    """The fallback of large meshes (NK_NO_TREE, also taken beyond 262 144 faces): all planes swept from global memory.
    Same answers as the tree walk, ray by ray."""
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.engine import Engine
    argv = LARGE_MESHES['star'] + ['--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                                   '--bound_values', '302', '298', '5'] + COMMON_ARGS
    args = initialise_parser().parse_args(argv)
    args.results_folder = ''
    geo = Geometry(args)
    ph = golden_phonon()
    rng = np.random.default_rng(12)
    x = geo.mesh.sample_volume(3000, rng)
    v = rng.normal(size=(3000, 3)) * 40.0
    out = []
    for no_tree in (False, True):
        if no_tree:
            monkeypatch.setenv('NK_NO_TREE', '1')
        eng = Engine(0, 1)
        eng.set_material(ph.tables())
        eng.set_mesh(geo.tables())
        eng.set_subvolumes(geo.subvol_center, geo.subvol_volume, 0, geo.slice_axis, 1, np.full(geo.n_of_subvols, 300.0))
        out.append(eng.find_boundary(x, v))
        eng.close()
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][1], out[1][1])
    assert np.all(out[0][2] >= 0)


@pytest.mark.parametrize('name', ['box200', 'box5000', 'cyl'])
def test_classifier(name):
    eng, g, ph = base_engine(name)
    assert np.array_equal(eng.classify(g['cls_x']), g['cls_id'])


def test_material_functions():
    eng, g, ph = base_engine('box200')
    gp = golden('phonon')
    mode = (gp['s_q'] * ph.number_of_branches + gp['s_j']).astype(np.int32)
    assert rel_err(eng.eval('occupation', gp['s_T'], mode), gp['s_occ']) < 1e-12
    assert rel_err(eng.eval('lifetime', gp['s_T'], mode), gp['s_tau']) < 1e-12
    assert rel_err(eng.eval('E_of_T', gp['s_Tw']), ph.crystal_energy_function(gp['s_Tw'])) < 1e-13
    assert rel_err(eng.eval('T_of_E', gp['s_E']), ph.temperature_function(gp['s_E'])) < 1e-12


def test_philox_matches_oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    import ctypes as C
    import nk_oracle as O
    eng, g, ph = base_engine('box200')
    for seed, pid, step, tag in [(0, 0, 0, 0), (1234, 77, 5, 0x10001), (2 ** 63 + 5, 2 ** 40 + 3, 999, 0x20002)]:
        a, b = C.c_double(), C.c_double()
        O.lib().nko_uniform2(C.c_uint64(seed), C.c_uint64(pid), C.c_uint32(step), C.c_uint32(tag), C.byref(a), C.byref(b))
        assert eng.uniform2(seed, pid, step, tag) == (a.value, b.value)


@pytest.mark.parametrize('model', ['velocity', 'k'])
def test_reflect(model):
    ct = case_tables('ttrrp', model)
    g = sub(golden('reflect'), model)
    gs = sub(golden('setup'), model)
    J = ct['J']
    from nanokappa_amd.engine import Engine
    eng = Engine(0, 3)
    eng.set_material(ct['tables'])
    eng.set_mesh(ct['mesh'])
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, g['subvol_temperature'])
    degen = None
    if model == 'k':
        di = gs['degen_index'].astype(int).ravel()
        dg = gs['degeneracies'].astype(int).reshape(-1, 3)
        degen = np.where(di > -1, dg[np.clip(di, 0, max(dg.shape[0] - 1, 0)), 2] if dg.shape[0] else -1, -1)
    r = ct['rough']
    eng.set_rough(r['facets'], r['specularity'], r['true_spec'], r['spec_map'], r['roulette'], degen)
    mo, no, oo = eng.reflect(g['facets'], g['in_modes'][:, 0] * J + g['in_modes'][:, 1], g['col_pos'], g['n_in'],
                             g['omega_in'], g['r_spec'], np.nan_to_num(g['r_deg']), np.nan_to_num(g['r_diff']))
    assert np.array_equal(mo, g['out_modes'][:, 0] * J + g['out_modes'][:, 1])
    assert rel_err(oo, g['omega_out']) < 1e-14
    assert rel_err(no, g['n_out']) < 1e-12


@pytest.mark.parametrize('variant', ['lin', 'near', 'fixed', 'tref'])
def test_frozen_step_vs_reference(variant):
    """One full timestep on a frozen reference state (no emission): the engine must reproduce the reference's
    drift -> boundary_scattering -> refresh_temperatures -> lifetime_scattering -> heat flux."""
    from nanokappa_amd.engine import Engine
    gm = sub(golden('mesh'), 'box200ttp')
    gs = sub(golden('step'), variant)
    ph = golden_phonon()
    J = ph.number_of_branches
    M = ph.number_of_qpoints * J
    interp = {'lin': 1, 'near': 0, 'fixed': 1, 'tref': 1}[variant]
    eng = Engine(0, 1)
    eng.set_material(ph.tables())
    eng.set_mesh(gm)
    eng.set_subvolumes(gm['subvol_center'], gm['subvol_volume'], 0, int(gm['slice_axis']), interp, gs['pre_subvol_temperature'])
    eng.set_reservoirs(gm['res_facets'], gs['res_facet_temperature'], np.zeros((2, M)), np.zeros((2, M)))
    eng.set_params(dt=1.0, norm_fixed=(variant == 'fixed'), particle_density=float(gs['particle_density']),
                   T_ref=(300.0 if variant == 'tref' else None), flux_every=1, contains_every=0, track_ids=True)
    eng.upload(gs['pre_positions'], gs['pre_modes'][:, 0] * J + gs['pre_modes'][:, 1], gs['pre_occupation'],
               gs['pre_n_timesteps'], gs['pre_collision_facets'])
    t = eng.step(1)
    assert np.array_equal(t['N_leaving'][0], gs['mid_N_leaving'])
    assert rel_err(t['res_energy'][0], gs['mid_res_energy_balance']) < TOL_RES
    assert allclose(t['res_flux'][0], gs['mid_res_heat_flux'], rtol=TOL_RES, atol=1e-14)
    assert np.array_equal(t['N_sv'][0], gs['post_subvol_N_p'])
    assert rel_err(t['E_sv'][0], gs['post_subvol_energy']) < TOL_E
    assert allclose(t['T_sv'][0], gs['post_subvol_temperature'], rtol=0, atol=TOL_T)
    # heat flux: Population.calculate_heat_flux scalings (Population.py:738-747)
    if variant == 'fixed':
        norm = ph.number_of_active_modes / (float(gs['particle_density']) * gm['subvol_volume'])
    else:
        norm = ph.number_of_active_modes / t['N_sv'][0]
    flux = t['flux_raw'][0] * norm[:, None] / (ph.number_of_qpoints * ph.volume_unitcell) * ph.eVpsa2_in_Wm2
    assert allclose(flux, gs['heat_flux'], rtol=0, atol=1e-13 * np.abs(gs['heat_flux']).max())     # (measured 1e-14 of the largest component)
    p = eng.download()          # flushes the deferred relaxation = lifetime_scattering
    n = gs['mid_positions'].shape[0]
    assert p['mode'].shape[0] == n
    # the engine reorders particles inside a segment; ids are the upload indices and np.delete keeps the reference's
    # survivors in index order, so sorting by id lines the two up
    o = np.argsort(p['pid'])
    assert np.array_equal(p['mode'][o], gs['mid_modes'][:, 0] * J + gs['mid_modes'][:, 1])
    assert np.array_equal(p['facet'][o], gs['mid_collision_facets'])
    assert allclose(p['positions'][o], gs['mid_positions'], rtol=0, atol=3e-12)
    assert allclose(p['n_timesteps'][o], gs['mid_n_timesteps'], rtol=0, atol=1e-12)
    assert rel_err(p['occupation'][o], gs['post_occupation']) < 2e-15


@pytest.mark.parametrize('store', ['box', 'cached', 'box-resident', 'cached-resident'])
@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_multistep_vs_oracle(case, store, monkeypatch):
    """Same seed, same counter-based RNG: engine and oracle must make the same decisions; compare the
    per-step tallies and the final particle set (matched by particle id).  Both layouts of the particle store: the box store
    (no cached next hit; these meshes are axis-aligned boxes) and, with NK_NO_BOX, the cached one that every other mesh uses.
    And the other way of stepping a small ensemble, NK_RESIDENT=1: many steps per launch with a grid barrier per step
    (k_resident; 'ttp' only -- rough facets keep the launch-per-step path).  It is opt-in because it measured slower; its
    results are the same."""
    if store.startswith('cached'):
        monkeypatch.setenv('NK_NO_BOX', '1')
    if store.endswith('resident'):
        if case == 'ttrrp':
            pytest.skip('rough facets never use the resident kernel')
        monkeypatch.setenv('NK_RESIDENT', '1')
    ct = case_tables(case)
    pos, mode, occ, counter = random_population(ct, 30000, seed=5)
    nsteps = 25
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=42)
    eng = make_engine(ct, pos, mode, occ, counter, seed=42)
    assert same_event_rule(eng, sim) == (1 if store.startswith('box') else 0)
    t = eng.step(nsteps)
    assert eng.timing()['emit_fused'] == (2 if store.endswith('resident') else 1)   # which path ran (1: the emission rides in the tail launch)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert np.array_equal(t['N_leaving'][s], sim.N_leaving[:2]), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
        assert rel_err(t['E_sv'][s], sim.E_sv) < TOL_E
    p = eng.download()
    n = sim.P.N
    assert p['pid'].shape[0] == n
    o1 = np.argsort(p['pid'])
    o2 = np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert np.array_equal(p['facet'][o1], sim.P.facet[:n][o2])
    assert allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=0, atol=TOL_X)
    assert allclose(p['n_timesteps'][o1], sim.P.n_ts[:n][o2], rtol=0, atol=TOL_NTS)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC
    # accumulated reservoir tallies
    assert rel_err(t['res_energy'].sum(axis=0), sim.res_energy[:2]) < TOL_RES


@pytest.mark.parametrize('gen', [1, 2])
def test_multistep_other_generators(gen):
    """fill_reservoirs 'fixed_rate' (Population.py:408-455) and 'one_to_one' (:457-489): same decisions as the oracle.
    one_to_one emits what left through each reservoir at the previous step, so the counts chain from step to step."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=6)
    nsteps = 20
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=43, gen=gen)
    eng = make_engine(ct, pos, mode, occ, counter, seed=43, gen=gen)
    t = eng.step(nsteps)
    emitted_prev = None
    for s in range(nsteps):
        n_before = sim.P.N
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert np.array_equal(t['N_leaving'][s], sim.N_leaving[:2]), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
        if gen == 2 and s > 0:
            assert t['N_emitted'][s] == t['N_leaving'][s - 1].sum()
    p = eng.download()
    n = sim.P.N
    assert p['pid'].shape[0] == n
    o1 = np.argsort(p['pid'])
    o2 = np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=0, atol=TOL_X)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


COMMON_ARGS = ['--poscar_file', 'POSCAR', '--hdf_file', 'synthetic', '--temp_interp', 'linear', '--timestep', '1',
               '--energy_normal', 'mean', '--particles', 'total', '30000']
EXTRA_CASES = {
    # BASELINE config 3 in small: Ge-like cross-plane film, 2000 A thick, periodic sides
    'ge_film': (['--geometry', 'box', '--dimensions', '2000', '500', '500', '--subvolumes', 'slice', '20', '0',
                 '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '--bound_cond', 'T', 'T', 'P',
                 '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
                 '--bound_values', '302', '298'], 'Ge'),
    # BASELINE config 4 in small: wire with rough side facets (tables in LDS: 64 faces, 18 facets)
    'wire16': (['--geometry', 'cylinder', '--dimensions', '500', '100', '16', '--subvolumes', 'slice', '10', '2',
                '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                '--bound_values', '302', '298', '5'], 'Si'),
    # non-convex wires: corrugated (alternating radii) and castle (flat annular lids), rough everywhere but the ends
    'corrugated': (['--geometry', 'corrugated', '--dimensions', '80', '60', '35', '10', '5', '--subvolumes', 'slice', '8', '2',
                    '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                    '--bound_values', '302', '298', '5'], 'Si'),
    'castle': (['--geometry', 'castle', '--dimensions', '90', '40', '70', '45', '8', '5', '1', '--subvolumes', 'slice', '8', '2',
                '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                '--bound_values', '302', '298', '5'], 'Si'),
    # same with 72 sides: 288 faces -> the ray-casting tables stay in global memory (large-mesh code path)
    'wire72': (['--geometry', 'cylinder', '--dimensions', '500', '100', '72', '--subvolumes', 'slice', '10', '2',
                '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                '--bound_values', '302', '298', '5'], 'Si'),
}


@pytest.mark.parametrize('name', ['ge_film', 'wire16', 'wire72', 'corrugated', 'castle'])
def test_other_geometries_vs_oracle(name):
    """Film with periodic sides, and wires with many rough facets (LDS and global-memory table paths): engine and
    oracle from the same state and seed, compared step by step and particle by particle."""
    from util import case_from_args, population_in_mesh
    argv, species = EXTRA_CASES[name]
    ct = case_from_args(argv + COMMON_ARGS, species)
    pos, mode, occ, counter = population_in_mesh(ct, 30000, seed=8)
    nsteps = 15
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=4242, cap=120000)
    eng = make_engine(ct, pos, mode, occ, counter, seed=4242)
    t = eng.step(nsteps)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    assert p['pid'].shape[0] == n
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=0, atol=TOL_X_LONG)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


@pytest.mark.parametrize('interp', [2, 3])
def test_grid_subvolumes_vs_oracle(interp):
    """'grid' subvolumes: 3-D nearest-centre classification and nearest-centre particle temperatures on the device
    (nk_classify general branch) against the oracle, step by step."""
    from util import case_from_args
    argv = ['--geometry', 'box', '--dimensions', '200', '200', '200', '--subvolumes', 'grid', '3', '3', '2',
            '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '--bound_cond', 'T', 'T', 'P',
            '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
            '--bound_values', '302', '298', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic', '--temp_interp', 'nearest',
            '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', '30000']
    ct = case_from_args(argv, 'Si')
    assert ct['kind'] == 1 and ct['centers'].shape[0] == 18
    pos, mode, occ, counter = random_population(ct, 30000, seed=21)
    nsteps = 20
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=5, interp=interp)
    eng = make_engine(ct, pos, mode, occ, counter, seed=5, interp=interp)
    if interp == 3:      # the device evaluation of the cubic-RBF temperature field against the NumPy statement of scipy's
        from nanokappa_amd import setup_tables as ST
        rbf = ST.rbf_system(ct['centers'])
        Tt = 300.0 + np.random.default_rng(1).normal(size=18)
        eng.set_subvol_temperature(Tt)
        x = np.random.default_rng(2).random((2000, 3)) * 200.0
        got = eng.eval('interp_T', x)
        assert np.abs(got - ST.rbf_evaluate(*rbf, ct['centers'], Tt, x)).max() < 1e-9
        eng.set_subvol_temperature(np.full(18, 298.0))
    t = eng.step(nsteps)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC_GRID


@pytest.mark.parametrize('sides,interp', [(72, 2), (72, 3), (16, 3)])
def test_rough_wire_grid_subvolumes_vs_oracle(sides, interp):
    """The remaining sweep variants: rough facets AND grid subvolumes with nearest-centre / cubic-RBF particle
    temperatures, on a 288-face wire (tables in global memory, face-tree ray caster: k_sweep<2, true, false / true>) and
    on a 64-face wire (tables in LDS: k_sweep<1, true, true>)."""
    from util import case_from_args, population_in_mesh
    argv = ['--geometry', 'cylinder', '--dimensions', '500', '100', str(sides), '--subvolumes', 'grid', '2', '2', '4',
            '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
            '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic', '--temp_interp', 'nearest',
            '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', '30000']
    ct = case_from_args(argv, 'Si')
    assert ct['kind'] == 1 and ct['mesh']['face_normals'].shape[0] == 4 * sides
    pos, mode, occ, counter = population_in_mesh(ct, 30000, seed=23)
    nsteps = 12
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=6, interp=interp, cap=120000)
    eng = make_engine(ct, pos, mode, occ, counter, seed=6, interp=interp)
    t = eng.step(nsteps)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC_GRID


def test_specular_pairs_on_device_equal_host_builder():
    """nk_specular_pairs (find_specular_correspondences 'velocity', Population.py:1241-1454, one thread per in-mode)
    returns exactly the pair set of the NumPy builder, which test_host_geometry pins to the reference's goldens --
    including the pairs the reference loses to arccos(>1) = NaN."""
    import ref_harness_args as A
    from nanokappa_amd import setup_tables as ST
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.engine import Engine
    args = initialise_parser().parse_args(A.argv_for('ttrrp', 20000))
    args.results_folder = ''
    geo = Geometry(args)
    ph = golden_phonon()
    corr_h, ts_h = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets)
    eng = Engine(0, 1)
    corr_d, ts_d = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets, engine=eng)
    assert corr_h.shape[0] > 1000
    assert np.array_equal(ts_h, ts_d)
    assert np.array_equal(corr_h, corr_d)
    g = sub(golden('setup'), 'velocity')
    assert np.array_equal(ts_d, g['true_specular'])
    # a tilted normal (no symmetry with the q-mesh axes)
    class G2(object):
        facets_normal = np.array([[0.6, 0.64, 0.48]])
    a, ta = ST.specular_correspondences_velocity(G2, ph, np.array([0]))
    b, tb = ST.specular_correspondences_velocity(G2, ph, np.array([0]), engine=eng)
    assert np.array_equal(a, b) and np.array_equal(ta, tb)
    eng.close()


@pytest.mark.parametrize('case,n,new_cap', [('ttrrp', 30000, 400000), ('ttp', 1100000, 6000000)])
def test_reserve_mid_run_preserves_state(case, n, new_cap):
    """nk_reserve between steps re-lays the particle store out -- through the host when the number of segments changes
    (new segment count, mode-sorted), on the device when it does not (1.1e6 particles: one segment per resident wave
    before and after, every segment grows in place): the deferred relaxation and the prepared emission of the next
    step must survive, i.e. the run continues exactly like the oracle."""
    ct = case_tables(case)
    pos, mode, occ, counter = random_population(ct, n, seed=15)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=8)
    eng = make_engine(ct, pos, mode, occ, counter, seed=8)
    t1 = eng.step(7)
    slots0 = eng.timing()['slots']
    eng.reserve(new_cap)
    assert eng.timing()['slots'] >= new_cap > slots0
    t2 = eng.step(8)
    T = np.concatenate((t1['T_sv'], t2['T_sv']))
    N = np.concatenate((t1['N_sv'], t2['N_sv']))
    for s in range(15):
        sim.run_timestep()
        assert np.array_equal(N[s], sim.N_sv), 'step %d' % s
        assert allclose(T[s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


def test_closed_periodic_box_conserves_particles():
    """No reservoirs at all (every facet periodic): nk_set_reservoirs is never called, nothing enters or leaves, and the
    oracle agrees step by step."""
    from util import case_from_args
    argv = ['--geometry', 'box', '--dimensions', '200', '200', '200', '--subvolumes', 'slice', '10', '0',
            '--bound_pos', 'relative', '0', '.5', '.5', '--bound_cond', 'P', 'P',
            '--connect_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1', '.5',
            '.5', '.5', '0', '.5', '.5', '1'] + COMMON_ARGS
    ct = case_from_args(argv, 'Si')
    assert ct['res_facets'].shape[0] == 0
    pos, mode, occ, counter = random_population(ct, 30000, seed=2)
    from nanokappa_amd.engine import Engine
    eng = Engine(0, 3)
    eng.set_material(ct['tables'])
    eng.set_mesh(ct['mesh'])
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, np.full(10, 298.0))
    eng.set_params(dt=1.0, particle_density=ct['particle_density'])
    eng.upload(pos, mode, occ)
    eng.init_boundaries()
    t = eng.step(30)
    assert np.all(t['N_sv'].sum(axis=1) == 30000)
    assert np.all(t['N_emitted'] == 0)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=3)
    for s in range(30):
        sim.run_timestep(emit=False)
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T)


def test_empty_start_fills_from_reservoirs():
    """An ensemble that starts with no particle at all: everything comes from the reservoirs."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 16, seed=1)
    from nanokappa_amd.engine import Engine
    eng = Engine(0, 5)
    eng.set_material(ct['tables'])
    eng.set_mesh(ct['mesh'])
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, np.full(ct['centers'].shape[0], 298.0))
    eng.set_reservoirs(ct['res_facets'], ct['res_T'], ct['enter_prob'], counter)
    eng.set_params(dt=1.0, particle_density=ct['particle_density'], track_ids=True)
    eng.reserve(200000)
    eng.upload(pos[:0], mode[:0], occ[:0])
    eng.init_boundaries()
    t = eng.step(40)
    live = t['N_sv'].sum(axis=1)
    assert live[0] > 0 and live[-1] > live[0]
    assert np.all(t['N_emitted'] > 0)
    p = eng.download()
    assert p['pid'].shape[0] == live[-1]
    assert np.unique(p['pid']).shape[0] == p['pid'].shape[0]


def test_error_paths_raise():
    """Misuse of the C ABI comes back as a negative status + message (NkError in the wrapper), never as a crash."""
    from nanokappa_amd.engine import Engine, NkError
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 1000, seed=1)
    eng = Engine(0, 1)
    with pytest.raises(NkError, match='not configured'):
        eng.step(1)
    eng.set_material(ct['tables'])
    with pytest.raises(NkError, match='mode index out of range'):
        eng.upload(pos, mode + 10 ** 6, occ)
    eng.set_mesh(ct['mesh'])
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, np.full(ct['centers'].shape[0], 298.0))
    eng.set_params(dt=1.0, particle_density=ct['particle_density'])
    with pytest.raises(NkError, match="has BC 'T'"):          # reservoir facets without nk_set_reservoirs
        eng.upload(pos, mode, occ)
        eng.step(1)
    with pytest.raises(NkError, match='one_to_one needs n_leaving'):
        eng.set_reservoirs(ct['res_facets'], ct['res_T'], ct['enter_prob'], counter, gen=2)
    eng.set_reservoirs(ct['res_facets'], ct['res_T'], ct['enter_prob'], counter)
    eng.upload(pos, mode, occ)
    eng.init_boundaries()
    eng.step(2)
    eng.close()


def test_long_run_keeps_segments_balanced():
    """Regression: entering particles are dealt in whole 64-particle tiles; the remainder once always landed in the last
    segment, which filled up after a few dozen steps.  Few entering particles per step and many steps."""
    from util import case_from_args, population_in_mesh
    argv, species = EXTRA_CASES['wire16']
    ct = case_from_args(argv + COMMON_ARGS[:-1] + ['200000'], species)
    pos, mode, occ, counter = population_in_mesh(ct, 200000, seed=3)
    eng = make_engine(ct, pos, mode, occ, counter, seed=11)
    for _ in range(4):
        t = eng.step(100)
    live = t['N_sv'][-1].sum()
    assert 0.8 * 200000 < live < 1.2 * 200000


def test_k_reflection_model_vs_oracle():
    """--bound_scat k: wavevector-mirror specular pairs plus the degenerate-branch coin flip (Population.py:963-969);
    tables from setup_tables.specular_correspondences_k (equal to the reference's, test_host_geometry), engine and
    oracle stepped from the same state."""
    import ref_harness_args as A
    from util import case_from_args
    ct = case_from_args(A.argv_for('ttrrp', 30000), 'Si', scat_model='k')
    assert (ct['rough']['degen_j2'] > -1).sum() > 0
    pos, mode, occ, counter = random_population(ct, 30000, seed=12)
    nsteps = 20
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=77)
    eng = make_engine(ct, pos, mode, occ, counter, seed=77)
    t = eng.step(nsteps)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    p = eng.download()
    n = sim.P.N
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


def test_rccl_path_single_rank(monkeypatch):
    """The multi-GPU code path (dlopen'ed RCCL, in-stream all-reduce of the tally vector) with a 1-rank communicator:
    results must equal the run without a communicator."""
    from nanokappa_amd.engine import comm_unique_id
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 20000, seed=9)
    ref = make_engine(ct, pos, mode, occ, counter, seed=1)
    t0 = ref.step(6)
    monkeypatch.setenv('NK_FORCE_COMM', '1')
    eng = make_engine(ct, pos, mode, occ, counter, seed=1)
    eng.comm_init(comm_unique_id(), 0, 1)
    t1 = eng.step(6)
    assert np.array_equal(t0['N_sv'], t1['N_sv'])
    assert allclose(t0['T_sv'], t1['T_sv'], rtol=0, atol=TOL_T)     # (the two runs sum their tally rows in different orders)


@pytest.mark.parametrize('case', ['ttp', 'ttrrp', 'wire72'])
def test_two_rank_sharding_on_one_gpu(case, monkeypatch):
    """The engine's rank-dependent code -- particle ids offset by the shard, every rank advancing all reservoir counters
    and keeping the entering particles it owns -- with two contexts on one GPU (NK_COMM_DRYRUN: no communicator, so the
    tallies, hence the temperatures and occupations, stay local).  Trajectories do not depend on the temperatures
    (in 'ttrrp': which mode a diffuse reflection draws does not, only its occupation), so the union of the two shards
    must hold exactly the single-rank run's particles: same ids, modes and positions."""
    from nanokappa_amd.sharding import shard_range
    n = 40000
    if case == 'wire72':                               # large mesh: split sweep, entering particles through the event queue
        from util import case_from_args, population_in_mesh
        argv, species = EXTRA_CASES['wire72']
        ct = case_from_args(argv + COMMON_ARGS, species)
        pos, mode, occ, counter = population_in_mesh(ct, n, seed=3)
    else:
        ct = case_tables(case)
        pos, mode, occ, counter = random_population(ct, n, seed=3)
    ref = make_engine(ct, pos, mode, occ, counter, seed=5)
    t = ref.step(12)
    p = ref.download()
    ref.close()
    monkeypatch.setenv('NK_COMM_DRYRUN', '1')
    parts, emitted = [], 0
    for r in (0, 1):
        lo, hi = shard_range(n, r, 2)
        e = make_engine(ct, pos[lo:hi], mode[lo:hi], occ[lo:hi], counter, seed=5, pid_offset=lo, comm=(bytes(128), r, 2))
        tr = e.step(12)
        emitted = emitted + tr['N_emitted']
        parts.append(e.download())
        e.close()
    assert np.array_equal(emitted, t['N_emitted'])
    pid = np.concatenate([q['pid'] for q in parts])
    o1, o2 = np.argsort(p['pid']), np.argsort(pid)
    assert np.array_equal(p['pid'][o1], pid[o2])
    assert np.array_equal(p['mode'][o1], np.concatenate([q['mode'] for q in parts])[o2])
    assert np.array_equal(p['positions'][o1], np.concatenate([q['positions'] for q in parts])[o2])


def test_rough_tables_of_the_k_model_built_on_device():
    """SURVEY 8f row 1, 'k' / wavevector model (Population.py:1056-1240): the pair search as a kernel (nk_kspec_pairs) and the
    tables built from it on the device, against the NumPy builders (pinned to the reference's goldens by
    tests/test_host_geometry.py) and against the reference's own tables; then 10 steps of the engine on the tables it built
    against the oracle on the host-built ones."""
    from util import case_from_args
    from nanokappa_amd import setup_tables as ST
    from nanokappa_amd.engine import Engine
    import ref_harness_args as A
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    args = initialise_parser().parse_args(A.argv_for('ttrrp', 100000) + ['--bound_scat', 'k'])
    args.results_folder = ''
    geo = Geometry(args)
    ph = golden_phonon()
    # zone-boundary q-points have several equally short images and numpy versions break the tie differently; the k model
    # mirrors wavevectors, so it is checked on the reference's own choice (as tests/test_host_geometry.py does)
    ph.wavevectors = golden('phonon')['wavevectors'].copy()
    Q, J = ph.omega.shape
    eng = Engine(0, 9)
    eng.set_material(ph.tables())
    eng.set_mesh(geo.tables())
    deg, didx = ST.find_degeneracies(ph)
    corr = ST.rough_tables_device_k(eng, geo, ph, geo.rough_facets, geo.rough_facets_values, deg, didx)
    sp, ts, sm, ro = eng.rough_download()
    corr_h, ts_h = ST.specular_correspondences_k(geo, ph, geo.rough_facets)
    spec_h = ts_h.astype(int) * ST.fbz_specularity(geo, ph, geo.rough_facets, geo.rough_facets_values)
    sm_h = ST.specular_map(corr_h, geo, geo.rough_facets, Q, J)
    _, ro_h = ST.diffuse_roulette(geo, ph, geo.rough_facets, spec_h, corr_h, scat_model='k', degeneracies=deg)
    assert corr.shape[0] > 1000 and np.array_equal(corr, corr_h)
    assert np.array_equal(ts.astype(bool), ts_h.reshape(-1, Q * J))
    assert np.array_equal(sm, sm_h.reshape(-1, Q * J))
    assert rel_err(sp, spec_h.reshape(-1, Q * J)) < 5e-13
    assert allclose(ro, ro_h, rtol=0, atol=1e-13)
    gs = sub(golden('setup'), 'k')
    assert np.array_equal(ts.astype(bool), gs['true_specular'].reshape(-1, Q * J))
    assert allclose(ro, gs['creation_roulette'].reshape(-1, Q * J), rtol=0, atol=1e-13)
    assert np.array_equal(ST.specular_rows_device_k(eng, geo, ph, geo.rough_facets), corr_h)
    ct = case_from_args(A.argv_for('ttrrp', 30000), 'Si', scat_model='k')
    pos, mode, occ, counter = random_population(ct, 20000, seed=5)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=9)
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, np.full(ct['centers'].shape[0], 298.0))
    eng.set_reservoirs(ct['res_facets'], ct['res_T'], ct['enter_prob'], counter)
    eng.set_params(dt=1.0, particle_density=ct['particle_density'])
    eng.upload(pos, mode, occ)
    eng.init_boundaries()
    t = eng.step(10)
    for s in range(10):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    eng.close()


def test_rough_tables_built_on_device_equal_host_builders_and_goldens():
    """specularity, truly-specular mask, specular map and creation roulette of the T T R R P box, built on
    the device (nk_rough_begin / nk_rough_pairs / nk_rough_finish), against the NumPy builders (which the reference's
    goldens pin) and against the reference's own tables; the pairs' rows against the reference's correspondent_modes;
    enter_probability on the device against the golden."""
    from util import case_from_args
    from nanokappa_amd import setup_tables as ST
    from nanokappa_amd.engine import Engine
    import ref_harness_args as A
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    args = initialise_parser().parse_args(A.argv_for('ttrrp', 100000))
    args.results_folder = ''
    geo = Geometry(args)
    ph = golden_phonon()
    Q, J = ph.omega.shape
    eng = Engine(0, 9)                        # the seed of the run at the end
    eng.set_material(ph.tables())
    eng.set_mesh(geo.tables())
    corr = ST.rough_tables_device(eng, geo, ph, geo.rough_facets, geo.rough_facets_values)
    sp, ts, sm, ro = eng.rough_download()
    gs = sub(golden('setup'), 'velocity')
    # host builders on the same inputs
    corr_h, ts_h = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets)
    spec_h = ts_h.astype(int) * ST.fbz_specularity(geo, ph, geo.rough_facets, geo.rough_facets_values)
    sm_h = ST.specular_map(corr_h, geo, geo.rough_facets, Q, J)
    _, ro_h = ST.diffuse_roulette(geo, ph, geo.rough_facets, spec_h, corr_h)
    assert np.array_equal(corr, corr_h)
    assert np.array_equal(ts.astype(bool), ts_h.reshape(-1, Q * J))
    assert np.array_equal(sm, sm_h.reshape(-1, Q * J))
    assert rel_err(sp, spec_h.reshape(-1, Q * J)) < 5e-13
    assert allclose(ro, ro_h, rtol=0, atol=1e-13)
    # ... and the reference's own tables
    assert np.array_equal(ts.astype(bool), gs['true_specular'].reshape(-1, Q * J))
    assert rel_err(sp, gs['specularity'].reshape(-1, Q * J)) < 5e-13
    assert allclose(ro, gs['creation_roulette'].reshape(-1, Q * J), rtol=0, atol=1e-13)
    g = gs['spec_map']
    assert np.array_equal(sm, np.where(g[..., 0] >= 0, g[..., 0] * J + g[..., 1], -1).reshape(-1, Q * J))
    # enter_probability
    thick = ph.number_of_active_modes / (float(gs['particle_density']) * geo.facets_area[geo.res_facets])
    ep = eng.build_enter_prob(-geo.facets_normal[geo.res_facets, :], thick, 1.0)
    assert rel_err(ep, gs['enter_prob'].reshape(2, -1)) < 1e-12
    # the engine runs on the tables it built: 10 steps against the oracle on the host-built ones
    ct = case_tables('ttrrp')
    pos, mode, occ, counter = random_population(ct, 20000, seed=5)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=9)
    eng.set_subvolumes(ct['centers'], ct['volumes'], 0, ct['axis'], 1, np.full(ct['centers'].shape[0], 298.0))
    eng.set_reservoirs(ct['res_facets'], ct['res_T'], ct['enter_prob'], counter)
    eng.set_params(dt=1.0, particle_density=ct['particle_density'])
    eng.upload(pos, mode, occ)
    eng.init_boundaries()
    t = eng.step(10)
    for s in range(10):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s


def test_two_rank_shards_grow_with_rough_walls(monkeypatch):
    """Rough walls on several ranks with a store that is too small (NK_TIGHT_STORE, six times the entry rate): the sweeps'
    halt requests and -- new in round 3 -- the 'a segment cannot take the migrants in its inbox' flag travel with the tally
    vector, so a rank grows its store and delivers instead of giving up (round 2 returned NK_ERR_CAPACITY there as soon as
    nranks > 1).  Two contexts on one GPU (NK_COMM_DRYRUN: no communicator); trajectories do not depend on the
    temperatures, so the union of the shards must be the single-rank run's particles."""
    from nanokappa_amd.sharding import shard_range
    monkeypatch.setenv('NK_TIGHT_STORE', '1')
    ct = case_tables('ttrrp')
    n = 24000
    pos, mode, occ, counter = random_population(ct, n, seed=13)
    ref = make_engine(ct, pos, mode, occ, counter, seed=5, emit_scale=6.0)
    slots0 = ref.timing()['slots']
    ref.step(70)
    p = ref.download()
    assert ref.timing()['slots'] > slots0 and ref.timing()['regrows'] > 0
    ref.close()
    monkeypatch.setenv('NK_COMM_DRYRUN', '1')
    parts, grew = [], 0
    for r in (0, 1):
        lo, hi = shard_range(n, r, 2)
        e = make_engine(ct, pos[lo:hi], mode[lo:hi], occ[lo:hi], counter, seed=5, emit_scale=6.0, pid_offset=lo, comm=(bytes(128), r, 2))
        e.step(70)                           # must not raise: every halt is served by growing the store
        grew += e.timing()['regrows']
        parts.append(e.download())
        e.close()
    assert grew > 0
    pid = np.concatenate([q['pid'] for q in parts])
    o1, o2 = np.argsort(p['pid']), np.argsort(pid)
    assert np.array_equal(p['pid'][o1], pid[o2])
    assert np.array_equal(p['mode'][o1], np.concatenate([q['mode'] for q in parts])[o2])
    assert np.array_equal(p['positions'][o1], np.concatenate([q['positions'] for q in parts])[o2])
