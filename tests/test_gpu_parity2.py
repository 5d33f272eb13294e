"""More parity of the HIP engine against the CPU oracle: the branches short runs never reach -- contains_check with
escaped particles (Population.py:1712-1722, Mesh.py:890-904), runs across the 100-step bookkeeping boundaries, a store
that must grow by itself, temperature ranges beyond the packed tables, an STL-imported 5000-triangle wire, and the
configuration that tracks no particle ids.  Needs a real MI355X: run with `pytest -m gpu`."""
import os
import sys
import tempfile

import numpy as np
import pytest

from util import rel_err, case_tables, random_population, make_oracle_sim, make_engine, allclose, same_event_rule, TOL_T, TOL_X, TOL_X_LONG, TOL_NTS, TOL_OCC

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


def compare_by_pid(p, sim, pos_atol=TOL_X_LONG):
    n = sim.P.N
    assert p['pid'].shape[0] == n
    o1, o2 = np.argsort(p['pid']), np.argsort(sim.P.pid[:n])
    assert np.array_equal(p['pid'][o1], sim.P.pid[:n][o2])
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert np.array_equal(p['facet'][o1], sim.P.facet[:n][o2])
    assert allclose(p['positions'][o1], sim.P.pos[:n][o2], rtol=0, atol=pos_atol)
    assert allclose(p['n_timesteps'][o1], sim.P.n_ts[:n][o2], rtol=0, atol=TOL_NTS)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


def compare_by_state(p, sim):
    """Without ids: the two ensembles as multisets, lined up by (mode, x, y, z)."""
    n = sim.P.N
    assert p['mode'].shape[0] == n and not p['pid'].any()
    o1 = np.lexsort((p['positions'][:, 2], p['positions'][:, 1], p['positions'][:, 0], p['mode']))
    q = sim.P.pos[:n]
    o2 = np.lexsort((q[:, 2], q[:, 1], q[:, 0], sim.P.mode[:n]))
    assert np.array_equal(p['mode'][o1], sim.P.mode[:n][o2])
    assert np.array_equal(p['facet'][o1], sim.P.facet[:n][o2])
    assert allclose(p['positions'][o1], q[o2], rtol=0, atol=TOL_X)
    assert rel_err(p['occupation'][o1], sim.P.occ[:n][o2]) < TOL_OCC


def steps_agree(eng, sim, nsteps, chunk=50):
    same_event_rule(eng, sim)
    done = 0
    while done < nsteps:
        k = min(chunk, nsteps - done)
        t = eng.step(k)
        for s in range(k):
            sim.run_timestep()
            assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % (done + s)
            assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % (done + s)
        done += k


@pytest.mark.parametrize('ids', [True, False])
def test_contains_check_resamples_escapees(ids):
    """Particles placed outside the bounding box: contains_check (step 0) draws them a new position in the volume
    (simplex by volume, Dirichlet weights) and a new first boundary; with and without particle ids (then the draws are
    keyed on the particle's state, in engine and oracle alike)."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=21)
    b = ct['mesh']['bounds']
    rng = np.random.default_rng(2)
    esc = rng.choice(30000, 700, replace=False)
    pos[esc] += (rng.integers(0, 2, (700, 3)) * 2 - 1) * (b[1] - b[0]) * rng.uniform(1.05, 3.0, (700, 3))
    assert np.all(np.any((pos[esc] < b[0]) | (pos[esc] > b[1]), axis=1))
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=12, ids_from_state=not ids)
    eng = make_engine(ct, pos, mode, occ, counter, seed=12, track_ids=ids)
    steps_agree(eng, sim, 3)
    p = eng.download()
    inside = np.all((p['positions'] >= b[0] - 1e-9) & (p['positions'] <= b[1] + 1e-9), axis=1)
    assert inside.mean() > 0.999                       # the escapees are back (a resampled particle may sit on the hull)
    (compare_by_pid if ids else compare_by_state)(p, sim)


def test_without_ids_matches_oracle_multiset():
    """The default layout of a mesh without rough facets stores no particle ids (44 bytes per particle): 40 steps of the
    T T P box against the oracle, ensembles compared as multisets."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=5)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=77)
    eng = make_engine(ct, pos, mode, occ, counter, seed=77, track_ids=False)
    steps_agree(eng, sim, 40)
    compare_by_state(eng.download(), sim)


@pytest.mark.parametrize('case', ['ttrrp', 'wire72'])
def test_long_run_vs_oracle(case):
    """260 steps: across two contains_check / bookkeeping boundaries (steps 100, 200), 26 heat-flux tallies, and, for the
    wire, the face-tree ray caster with rough walls.  Engine and oracle step for step, then particle for particle."""
    if case == 'ttrrp':
        ct = case_tables('ttrrp')
        pos, mode, occ, counter = random_population(ct, 30000, seed=31)
    else:
        from util import case_from_args, population_in_mesh
        from test_gpu_parity import EXTRA_CASES, COMMON_ARGS
        argv, species = EXTRA_CASES['wire72']
        ct = case_from_args(argv + COMMON_ARGS, species)
        pos, mode, occ, counter = population_in_mesh(ct, 30000, seed=31)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=2024, cap=200000)
    eng = make_engine(ct, pos, mode, occ, counter, seed=2024)
    steps_agree(eng, sim, 260, chunk=65)               # chunks that do not line up with the 100-step boundaries
    compare_by_pid(eng.download(), sim)


@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_store_grows_by_itself(case, monkeypatch):
    """Six times the entry rate into a store with hardly any head room (NK_TIGHT_STORE): the ensemble outgrows it several
    times over.  nk_step must stop before a step that could drop a particle, grow the segments on the device and carry
    on -- the run equals the oracle's throughout.  'ttrrp': with rough walls the particles also change segment (a
    reflection changes the mode), through inboxes that must keep up."""
    monkeypatch.setenv('NK_TIGHT_STORE', '1')
    ct = case_tables(case)
    pos, mode, occ, counter = random_population(ct, 20000, seed=9)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=3, cap=600000, emit_scale=6.0)
    eng = make_engine(ct, pos, mode, occ, counter, seed=3, emit_scale=6.0)
    slots0 = eng.timing()['slots']
    steps_agree(eng, sim, 90, chunk=45)
    tm = eng.timing()
    assert tm['live'] > slots0 and tm['slots'] > slots0            # it did outgrow the first allocation
    compare_by_pid(eng.download(), sim)


def test_dense_emission_into_a_tight_store(monkeypatch):
    """enter_prob >> 1 per (reservoir, mode) entry -- every entry emits dozens of particles per step -- into a store
    with hardly any head room.  The next step's emission runs ahead in the tail launch (k_tail); when the step before it asks
    for a halt, the segment may not hold the ahead emission: that is no loss (it is run again after the store has grown) and
    must not surface as 'particles were dropped'.  Equal to the oracle throughout."""
    monkeypatch.setenv('NK_TIGHT_STORE', '1')
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 20000, seed=9)
    scale = 25.0 / float(np.max(ct['enter_prob']))                 # the largest entry emits 25 particles per step
    assert scale > 10.0
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=4, cap=3000000, emit_scale=scale)
    eng = make_engine(ct, pos, mode, occ, counter, seed=4, emit_scale=scale)
    slots0 = eng.timing()['slots']
    steps_agree(eng, sim, 24, chunk=8)
    tm = eng.timing()
    assert tm['live'] > 5 * 20000 and tm['slots'] > slots0 and tm['halts'] > 0
    compare_by_pid(eng.download(), sim)


@pytest.mark.parametrize('case', ['ttp', 'ttrrp'])
def test_store_placement_choice_moves_the_store(case, monkeypatch):
    """nk_place_store (the store's allocation is timed against further candidates, the particles move into the fastest):
    with the test hooks every store -- the first one and each one the growing ensemble forces -- is timed however small it
    is, and the particles always move into the last candidate.  The run equals the oracle's throughout."""
    monkeypatch.setenv('NK_TIGHT_STORE', '1')
    monkeypatch.setenv('NK_PLACE_MIN_MB', '0')
    monkeypatch.setenv('NK_PLACE_TRIES', '3')
    monkeypatch.setenv('NK_PLACE_FORCE', '1')
    ct = case_tables(case)
    pos, mode, occ, counter = random_population(ct, 20000, seed=19)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=5, cap=600000, emit_scale=6.0)
    eng = make_engine(ct, pos, mode, occ, counter, seed=5, emit_scale=6.0)
    tm = eng.timing()
    slots0 = tm['slots']
    assert tm['place_tries'] == 3 and tm['place_gbps'] > 0.0
    steps_agree(eng, sim, 60, chunk=30)
    tm = eng.timing()
    assert tm['slots'] > slots0 and tm['regrows'] > 0 and tm['place_tries'] == 3     # grown, and placed again
    compare_by_pid(eng.download(), sim)


@pytest.mark.parametrize('gen', [1, 2])
def test_other_generators_on_a_large_mesh(gen):
    """'fixed_rate' and 'one_to_one' reservoirs on the 72-sided wire (288 faces: split sweep, face tree): their entering
    particles take their first ray cast in k_events' walks, from the segment's event queue, like those of 'constant' --
    engine and oracle step by step, then particle by particle."""
    from util import case_from_args, population_in_mesh
    from test_gpu_parity import EXTRA_CASES, COMMON_ARGS
    argv, species = EXTRA_CASES['wire72']
    ct = case_from_args(argv + COMMON_ARGS, species)
    pos, mode, occ, counter = population_in_mesh(ct, 30000, seed=21)
    nsteps = 20
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=77, cap=120000, gen=gen)
    eng = make_engine(ct, pos, mode, occ, counter, seed=77, gen=gen)
    t = eng.step(nsteps)
    for s in range(nsteps):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
        if gen == 2 and s > 0:
            assert t['N_emitted'][s] == t['N_leaving'][s - 1].sum()
    assert t['N_emitted'].sum() > 0
    compare_by_pid(eng.download(), sim)


def test_wide_temperature_range_vs_oracle():
    """Reservoirs at 340 K and 290 K, start at 340 K: the subvolume temperatures sweep a range wider than the two grid
    intervals packed into the mode records and further from the reference temperature of the precomputed exponentials
    than their short series allows -- the full-table lifetime lookup, the general exponential and the rebuilds of the
    records as the range moves must all agree with the oracle."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=41, T0=340.0)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=8, T0=340.0, res_T=[340.0, 290.0])
    eng = make_engine(ct, pos, mode, occ, counter, seed=8, T0=340.0, res_T=[340.0, 290.0])
    steps_agree(eng, sim, 120, chunk=30)
    assert sim.T_sv.max() - sim.T_sv.min() > 8.0 and sim.T_sv.max() < 330.0
    compare_by_pid(eng.download(), sim)


def test_stl_wire_5000_faces_vs_oracle():
    """BASELINE config 4's mesh -- the cylinder primitive with 1250 sides written as ASCII STL and imported again: 5000
    triangles, 1250 rough facets, caps at 302 / 298 K -- at 1.5e5 particles against the oracle's brute-force ray casts."""
    from util import case_from_args, population_in_mesh
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    tail = ['--subvolumes', 'slice', '20', '2', '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1',
            '--bound_cond', 'T', 'T', 'R', '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
            '--temp_interp', 'linear', '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', '150000']
    prim = initialise_parser().parse_args(['--geometry', 'cylinder', '--dimensions', '2000', '200', '1250'] + tail)
    prim.results_folder = ''
    g0 = Geometry(prim)
    tmp = tempfile.mkdtemp()
    g0.mesh.export_stl('wire', tmp)
    ct = case_from_args(['--geometry', os.path.join(tmp, 'wire.stl'), '--dimensions', '1', '1', '1'] + tail)
    assert ct['mesh']['face_normals'].shape[0] == 5000 and ct['rough']['facets'].shape[0] == 1250
    pos, mode, occ, counter = population_in_mesh(ct, 150000, seed=13)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=99, cap=400000)
    eng = make_engine(ct, pos, mode, occ, counter, seed=99)
    steps_agree(eng, sim, 10)
    compare_by_pid(eng.download(), sim)


def test_mesh_crossings_on_the_device_equal_the_host_count(monkeypatch):
    """Set-up helper nk_mesh_crossings (ray-parity inside tests of nanokappa_amd.mesh on large meshes) against the NumPy
    all-pairs count it replaces: random rays through the 5000-triangle wire, rays along mesh edges and vertices (equal
    distances counted once), and the faces' own centroid rays of the orientation pass (ray i ignores face i); then the
    whole geometry built both ways has the same orientation, tetrahedra and volume."""
    import nanokappa_amd.mesh as M
    from nanokappa_amd.mesh import Mesh
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry

    def wire_mesh():
        a = initialise_parser().parse_args(['--geometry', 'cylinder', '--dimensions', '2000', '200', '1250', '--subvolumes', 'slice', '20', '2',
                                            '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
                                            '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
                                            '--particles', 'total', '1000'])
        a.results_folder = ''
        return Geometry(a).mesh

    monkeypatch.setattr(M, '_DEVICE_HELPER', None)
    mesh = wire_mesh()
    assert M._device_helper() and mesh.faces.shape[0] == 5000
    v = mesh.vertices[mesh.faces]
    v0, e1, e2 = v[:, 0], v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    rng = np.random.default_rng(4)
    lo, hi = mesh.bounds[0], mesh.bounds[1]
    o = lo + (hi - lo) * (rng.random((3000, 3)) * 1.4 - 0.2)
    d = rng.normal(size=(3000, 3))
    # some rays aimed exactly at vertices and edge midpoints of the mesh
    tgt = np.vstack((mesh.vertices[rng.integers(0, mesh.vertices.shape[0], 300)], (v[:300, 0] + v[:300, 1]) / 2))
    d[:600] = tgt - o[:600]
    host = Mesh._count_crossings_host(o, d, v0, e1, e2)
    dev = mesh._count_crossings(o, d)
    assert np.array_equal(host, dev) and host.max() >= 2
    cen, nrm = v.mean(axis=1), mesh.face_normals
    own = np.arange(cen.shape[0])
    assert np.array_equal(Mesh._count_crossings_host(cen, nrm, v0, e1, e2, own), mesh._count_crossings(cen, nrm, skip_self=True))
    # the geometry as a whole, built with and without the helper
    monkeypatch.setattr(M, '_DEVICE_HELPER', False)
    ref = wire_mesh()
    assert np.array_equal(ref.faces, mesh.faces) and np.array_equal(ref.simplices, mesh.simplices) and ref.volume == mesh.volume


@pytest.mark.gpu
def test_engine_from_a_file_loaded_material_vs_oracle_on_reference_tables(tmp_path, monkeypatch):
    """SURVEY 8 row f2 on the GPU: `Phonon(args)` reads the phono3py datasets of tests/golden/kappa-m999.hdf5 (their .npz
    twin: h5py is not installed for the system interpreter) + POSCAR through THIS package's IBZ -> FBZ loader (reference
    Phonon.load_base_properties Phonon.py:66-149, expand_FBZ :515-564) -> `Population` -> 25 steps of the engine; the oracle
    runs the same ensemble on the tables the REFERENCE's loader produced from the same file (tests/golden/fbz.npz, written by
    tests/golden/make_fbz.py).  Particle by particle (as multisets: this configuration stores no ids)."""
    import shutil
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, golden_dir)
    import make_hdf5_material_data as M
    import ref_harness_args as A
    from util import golden
    from nanokappa_amd import crystal, setup_tables as ST
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population
    np.savez(tmp_path / 'kappa.npz', **M.datasets())
    shutil.copy(os.path.join(golden_dir, 'POSCAR_Si'), tmp_path / 'POSCAR')
    monkeypatch.setenv('NK_HOST_INIT', '1')                  # particles made on the host and uploaded: the oracle gets the same ones
    argv = A.argv_for('ttp', 30000)
    argv[argv.index('--hdf_file') + 1] = 'kappa.npz'
    i = argv.index('--bound_values')
    argv[i + 1:i + 3] = ['305', '295']                       # the file's temperature grid is 250 / 300 / 350 K
    args = initialise_parser().parse_args(argv + ['--mat_folder', str(tmp_path), '--seed', '17', '--isotope_scat', '0'])
    args.results_folder = ''
    geo = Geometry(args)
    ph = Phonon(args, 0)                                     # the loader: irreducible wedge -> full zone
    assert ph.omega.shape == (729, 6)
    pop = Population(args, geo, ph)
    eng = pop.engine
    p0 = eng.download()
    assert p0['mode'].shape[0] == 30000 and not p0['pid'].any()
    # ---- the oracle on the REFERENCE's expansion of the same file
    g = golden('fbz')
    cell = crystal.read_poscar(os.path.join(golden_dir, 'POSCAR_Si'))
    rec = np.around(np.linalg.inv(cell['lattice']) * 2 * np.pi, decimals=6)
    gam = np.where(g['gamma_with_isotope'] > 0, g['gamma_with_isotope'], -1)
    ph_ref = Phonon(None, 0, material=dict(data_mesh=g['data_mesh'], q_points=g['q_points'], omega=g['omega'], frequency=g['frequency'],
                                           group_vel=g['group_vel'], temperature=g['temperature'], gamma=gam,
                                           reciprocal_lattice=rec, volume_unitcell=abs(np.linalg.det(cell['lattice']))))
    Q, J = ph_ref.omega.shape
    Mm = Q * J
    density = 30000 / geo.volume
    ct = dict(ph=ph_ref, J=J, M=Mm, mesh=geo.tables(), tables=ph_ref.tables(), centers=geo.subvol_center, volumes=geo.subvol_volume,
              axis=geo.slice_axis, kind=0, res_facets=geo.res_facets, res_T=np.asarray(geo.res_values, dtype=float),
              enter_prob=ST.enter_probability(geo, ph_ref, geo.res_facets, density, 1.0).reshape(-1, Mm),
              particle_density=density, rough=None)
    assert rel_err(pop.enter_prob.reshape(-1, Mm), ct['enter_prob']) < 1e-12
    sim = make_oracle_sim(ct, p0['positions'], p0['mode'], p0['occupation'], pop.res_counter.reshape(-1, Mm), seed=17,
                          T0=float(pop.subvol_temperature[0]), ids_from_state=True)
    assert np.all(pop.subvol_temperature == pop.subvol_temperature[0])
    t = eng.step(25)
    for s in range(25):
        sim.run_timestep()
        assert np.array_equal(t['N_sv'][s], sim.N_sv), 'step %d' % s
        assert allclose(t['T_sv'][s], sim.T_sv, rtol=0, atol=TOL_T), 'step %d' % s
    assert t['N_emitted'].sum() > 0 and t['N_leaving'].sum() > 0
    compare_by_state(eng.download(), sim)


@pytest.mark.gpu
def test_stepping_one_by_one_equals_one_call():
    """The reference driver's granularity (nanokappa.py:91-98: one run_timestep per iteration) against one library call for all
    steps.  The tail launch of a step also runs the NEXT step's emission, now across calls too (nk_engine.hip emitted_for): a
    driver that steps one by one never launches k_emit.  Anything that touches the store in between -- a download, a regrow --
    must leave the run unchanged (the ahead emission is then simply run again)."""
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=5)
    a = make_engine(ct, pos, mode, occ, counter, seed=42)
    b = make_engine(ct, pos, mode, occ, counter, seed=42)
    ta = a.step(40)
    rows = []
    for s in range(40):
        rows.append(b.step(1))
        if s == 10:
            b.download()                                   # flushes the deferred relaxation, reads the store
        if s == 20:
            b.reserve(int(b.timing()['slots'] * 2))        # regrow: the particles k_tail appended ahead are dropped and made again
    for k in ('N_sv', 'N_emitted', 'N_leaving'):
        assert np.array_equal(ta[k], np.concatenate([r[k] for r in rows])), k
    # (not bit for bit: the stand-alone relaxation of the download and the re-dealt tiles after the regrow round differently)
    assert allclose(ta['T_sv'], np.concatenate([r['T_sv'] for r in rows]), rtol=0, atol=TOL_T)
    pa, pb = a.download(), b.download()
    oa, ob = np.argsort(pa['pid']), np.argsort(pb['pid'])
    assert np.array_equal(pa['pid'][oa], pb['pid'][ob])
    assert np.array_equal(pa['positions'][oa], pb['positions'][ob]) and rel_err(pa['occupation'][oa], pb['occupation'][ob]) < TOL_OCC


@pytest.mark.gpu
@pytest.mark.parametrize('store', ['box', 'cached'])
def test_alternating_walk_equals_walking_upwards_only(monkeypatch, store):
    """The fused sweeps of small meshes (box store, and the cached store that NK_NO_BOX=1 forces here): the sweeps alternate between walking their segments upwards and downwards (NkDev::seg_lo / down, nk_device.h), so
    that a sweep starts with what the one before wrote last.  The order in which a wave meets its particles decides nothing: the
    same run with NK_NO_ALTERNATE=1 (every sweep upwards from slot 0) has the same counts step by step and the same particles at
    the end.  160 steps in calls of 1, 7 and 50 steps: the lower end of the segments wanders up and is brought back (an UP sweep
    that starts its output at slot 0 again, k_anchor before the contains_check step at 100 and before the download in between), and
    a second population uploaded into the same store starts from slot 0 again."""
    if store == 'cached':
        monkeypatch.setenv('NK_NO_BOX', '1')
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 30000, seed=5)
    a = make_engine(ct, pos, mode, occ, counter, seed=42)
    assert a.timing()['box_store'] == (1 if store == 'box' else 0)
    monkeypatch.setenv('NK_NO_ALTERNATE', '1')
    b = make_engine(ct, pos, mode, occ, counter, seed=42)
    tb = b.step(160)
    monkeypatch.delenv('NK_NO_ALTERNATE')
    rows, done = [], 0
    for n in [1, 1, 7, 50, 1, 50, 50]:
        rows.append(a.step(n))
        done += n
        if done == 60:
            a.download()
    assert done == 160
    for k in ('N_sv', 'N_emitted', 'N_leaving'):
        assert np.array_equal(tb[k], np.concatenate([r[k] for r in rows])), k
    assert allclose(tb['T_sv'], np.concatenate([r['T_sv'] for r in rows]), rtol=0, atol=TOL_T)
    pa, pb = a.download(), b.download()
    oa, ob = np.argsort(pa['pid']), np.argsort(pb['pid'])
    assert np.array_equal(pa['pid'][oa], pb['pid'][ob]) and np.array_equal(pa['mode'][oa], pb['mode'][ob])
    assert allclose(pa['positions'][oa], pb['positions'][ob], rtol=0, atol=TOL_X) and rel_err(pa['occupation'][oa], pb['occupation'][ob]) < TOL_OCC
    # the same stores, a new population (it fits: no new allocation), both ways again
    pos2, mode2, occ2, _ = random_population(ct, 28000, seed=9)
    a.upload(pos2, mode2, occ2)
    a.init_boundaries()
    t2a = a.step(12)
    monkeypatch.setenv('NK_NO_ALTERNATE', '1')
    b.upload(pos2, mode2, occ2)
    b.init_boundaries()
    t2b = b.step(12)
    assert np.array_equal(t2a['N_sv'], t2b['N_sv']) and allclose(t2a['T_sv'], t2b['T_sv'], rtol=0, atol=TOL_T)
