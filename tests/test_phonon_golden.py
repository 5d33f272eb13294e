"""Host-side material tables vs the reference's Phonon (golden: tests/golden/phonon.npz)."""
import numpy as np
import os

import pytest

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


def diamond_poscar(path, species='Si'):
    """Primitive diamond-structure cell on the synthetic materials' fcc lattice, as a POSCAR file."""
    from nanokappa_amd import crystal, synthetic
    crystal.write_poscar(str(path), synthetic.fcc_lattice(species), [species], [2], [[0.75, 0.75, 0.75], [0.5, 0.5, 0.5]])
    return str(path)


def test_poscar_and_point_group(tmp_path):
    """POSCAR -> lattice: reciprocal lattice and cell volume equal the values the reference obtained through phonopy
    (tests/golden/phonon.npz); the symmetry finder returns the 48 operations of the diamond structure, a closed group
    that contains the inversion (time reversal)."""
    from nanokappa_amd import crystal
    cell = crystal.read_poscar(diamond_poscar(tmp_path / 'POSCAR'))
    g = golden('phonon')
    rec = np.around(np.linalg.inv(cell['lattice']) * 2 * np.pi, decimals=6)
    assert np.allclose(rec, g['mat_reciprocal_lattice'], rtol=0, atol=1e-6)
    assert abs(abs(np.linalg.det(cell['lattice'])) / float(g['mat_volume_unitcell']) - 1) < 1e-9
    R = crystal.reciprocal_operations(cell['lattice'], cell['numbers'], cell['positions'])
    assert R.shape == (48, 3, 3)
    S = {tuple(r.ravel()) for r in R}
    assert all(tuple((a @ b).ravel()) in S for a in R for b in R)
    assert tuple((-np.eye(3, dtype=int)).ravel()) in S


def test_irreducible_wedge_round_trip(tmp_path):
    """phono3py-style input (irreducible q-points + weights) -> expand_FBZ (Phonon.py:515-564) reproduces the full-mesh
    tables it was reduced from: same q-point set, frequencies, rotated group velocities and linewidths."""
    from nanokappa_amd import crystal, synthetic
    from nanokappa_amd.phonon import material_from_phono3py
    poscar = diamond_poscar(tmp_path / 'POSCAR')
    full = synthetic.make_material(9, 'Si', temperatures=np.arange(250.0, 351.0, 50.0))
    cell = crystal.read_poscar(poscar)
    rot = crystal.reciprocal_operations(cell['lattice'], cell['numbers'], cell['positions'])
    reps, weights = crystal.reduce_to_IBZ(full['q_points'], rot)
    assert weights.sum() == 729 and reps.shape[0] < 60
    data = dict(mesh=np.array([9, 9, 9]), qpoint=full['q_points'][reps], weight=weights,
                frequency=full['frequency'][reps], group_velocity=full['group_vel'][reps],
                temperature=full['temperature'], gamma=full['gamma'][:, reps, :])
    m = material_from_phono3py(data, poscar)
    assert m['q_points'].shape == (729, 3)

    def keyed(q):
        q = np.around(np.mod(q, 1.0), 6)
        q = np.where(q == 1.0, 0.0, q)
        return np.lexsort(q.T[::-1]), q

    o1, q1 = keyed(m['q_points'])
    o2, q2 = keyed(full['q_points'])
    assert np.allclose(q1[o1], q2[o2], atol=1e-6)
    assert np.allclose(m['frequency'][o1], full['frequency'][o2], rtol=0, atol=1e-12)
    assert np.allclose(m['gamma'][:, o1, :], full['gamma'][:, o2, :], rtol=0, atol=1e-15)
    # group velocities: rotated copies of the representative's; q-points on the zone boundary have several equally
    # short images and the synthetic table picked one of them, so those are compared by length only
    dv = np.abs(m['group_vel'][o1] - full['group_vel'][o2]).max(axis=(1, 2))
    interior = dv < 1e-8
    assert interior.mean() > 0.9
    assert np.allclose(np.linalg.norm(m['group_vel'][o1], axis=2), np.linalg.norm(full['group_vel'][o2], axis=2), atol=1e-8)


def test_isotope_scattering_option(tmp_path):
    """--isotope_scat adds the dataset gamma_isotope to gamma for the listed materials (load_gamma, Phonon.py:316-323)
    and fails loudly when the file has none."""
    from nanokappa_amd import crystal, synthetic
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.phonon import Phonon, material_from_phono3py
    poscar = diamond_poscar(tmp_path / 'POSCAR')
    full = synthetic.make_material(5, 'Si', temperatures=np.arange(250.0, 351.0, 50.0))
    cell = crystal.read_poscar(poscar)
    rot = crystal.reciprocal_operations(cell['lattice'], cell['numbers'], cell['positions'])
    reps, weights = crystal.reduce_to_IBZ(full['q_points'], rot)
    g = full['gamma'][:, reps, :]
    iso = np.where(g > 0, 0.25 * g + 1e-4, 0.0)            # the synthetic table marks modes without scattering with -1
    data = dict(mesh=np.array([5, 5, 5]), qpoint=full['q_points'][reps], weight=weights,
                frequency=full['frequency'][reps], group_velocity=full['group_vel'][reps],
                temperature=full['temperature'], gamma=full['gamma'][:, reps, :], gamma_isotope=iso)
    np.savez(tmp_path / 'kappa.npz', **data)
    base = ['--mat_folder', str(tmp_path), '--hdf_file', 'kappa.npz', '--poscar_file', 'POSCAR']
    plain = Phonon(initialise_parser().parse_args(base), 0)
    with_iso = Phonon(initialise_parser().parse_args(base + ['--isotope_scat', '0']), 0)
    other = Phonon(initialise_parser().parse_args(base + ['--isotope_scat', '1']), 0)       # another material's index
    assert np.array_equal(other.gamma, plain.gamma)
    ref = material_from_phono3py(dict(data, gamma=data['gamma'] + iso), poscar)['gamma']
    assert np.allclose(with_iso.gamma, ref, rtol=0, atol=1e-15) and np.all(with_iso.gamma >= plain.gamma)
    assert np.all(with_iso.lifetime[plain.lifetime > 0] < plain.lifetime[plain.lifetime > 0])
    del data['gamma_isotope']
    np.savez(tmp_path / 'kappa.npz', **data)
    with pytest.raises(Exception, match='gamma_isotope'):
        Phonon(initialise_parser().parse_args(base + ['--isotope_scat', '0']), 0)


def test_hdf5_loader_branch(tmp_path):
    """The phono3py HDF5 container itself (reference Phonon.py:66-149, :158-187): tests/golden/kappa-m999.hdf5 (written by
    tests/golden/make_hdf5_material.py) through `Phonon`'s .hdf5 branch must give the tables of the same datasets
    handed over as .npz.  h5py is not installed for the system interpreter of this image, so the loading runs in whichever
    interpreter has it (this one, or the image's conda python); skipped where neither does."""
    import shutil
    import subprocess
    import sys
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    root = os.path.abspath(os.path.join(golden_dir, '..', '..'))
    shutil.copy(os.path.join(golden_dir, 'kappa-m999.hdf5'), tmp_path / 'kappa-m999.hdf5')
    shutil.copy(os.path.join(golden_dir, 'POSCAR_Si'), tmp_path / 'POSCAR')
    code = ("import sys; sys.path.insert(0, %r); import numpy as np\n"
            "from nanokappa_amd.argument_parser import initialise_parser\n"
            "from nanokappa_amd.phonon import Phonon\n"
            "a = initialise_parser().parse_args(['--mat_folder', %r, '--hdf_file', 'kappa-m999.hdf5', '--poscar_file', 'POSCAR', '--isotope_scat', '0'])\n"
            "p = Phonon(a, 0)\n"
            "np.savez(%r, q_points=p.q_points, omega=p.omega, group_vel=p.group_vel, gamma=p.gamma, T=p.temperature_array, lifetime=p.lifetime)\n"
            % (root, str(tmp_path), str(tmp_path / 'from_hdf5.npz')))
    interp = None
    for cand in (sys.executable, '/opt/conda/bin/python3.9'):
        if os.path.exists(cand) and subprocess.run([cand, '-c', 'import h5py'], capture_output=True).returncode == 0:
            interp = cand
            break
    if interp is None:
        pytest.skip('no interpreter with h5py')
    r = subprocess.run([interp, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(tmp_path / 'from_hdf5.npz')
    # the same datasets as .npz through the other branch
    sys.path.insert(0, golden_dir)
    import make_hdf5_material_data as M
    d = M.datasets()
    np.savez(tmp_path / 'kappa.npz', **d)
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.phonon import Phonon
    ref = Phonon(initialise_parser().parse_args(['--mat_folder', str(tmp_path), '--hdf_file', 'kappa.npz', '--poscar_file', 'POSCAR',
                                                 '--isotope_scat', '0']), 0)
    assert got['q_points'].shape == (729, 3)
    for k, v in (('q_points', ref.q_points), ('omega', ref.omega), ('group_vel', ref.group_vel), ('gamma', ref.gamma),
                 ('T', ref.temperature_array), ('lifetime', ref.lifetime)):
        assert np.allclose(got[k], v, rtol=1e-13, atol=1e-300), k


def test_expand_FBZ_equals_the_reference(tmp_path):
    """The loader against the REFERENCE's own (tests/golden/fbz.npz, written by tests/golden/make_fbz.py: the reference's
    load_* functions and expand_FBZ, Phonon.py:158-187, :316-324, :515-564, on tests/golden/kappa-m999.hdf5 with this
    package's reciprocal operations handed in as `rotations`): same q-point SEQUENCE (the star of every irreducible point in
    the reference's order: np.unique over the rotated points rounded to 6 decimals), same star representative for the
    tensors, frequencies / rotated group velocities / linewidths equal to 1e-12, with and without the isotope part."""
    import sys
    from nanokappa_amd import crystal
    from nanokappa_amd.phonon import material_from_phono3py
    g = golden('fbz')
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, golden_dir)
    import make_hdf5_material_data as M
    d = M.datasets()                                   # the datasets kappa-m999.hdf5 was written from
    assert np.array_equal(d['qpoint'], g['q_ibz']) and np.array_equal(d['weight'], g['weights'])
    poscar = os.path.join(golden_dir, 'POSCAR_Si')
    cell = crystal.read_poscar(poscar)
    rot = crystal.reciprocal_operations(cell['lattice'], cell['numbers'], cell['positions'])
    assert np.array_equal(rot, g['rotations'])         # the operations the reference's expansion was given
    for iso, gam in ((False, g['gamma']), (True, g['gamma_with_isotope'])):
        m = material_from_phono3py(d, poscar, isotope=iso)
        assert m['q_points'].shape == g['q_points'].shape == (729, 3)
        assert np.array_equal(m['q_points'], g['q_points'])                       # bit for bit, in the reference's order
        assert np.array_equal(np.asarray(m['data_mesh']), g['data_mesh'])
        assert np.allclose(m['frequency'], g['frequency'], rtol=0, atol=1e-12)
        assert np.allclose(m['omega'], g['omega'], rtol=1e-15, atol=1e-12)
        assert np.allclose(m['group_vel'], g['group_vel'], rtol=0, atol=1e-12)
        assert np.allclose(m['temperature'], g['temperature'], rtol=0, atol=0)
        # load_gamma's "gamma > 0 else -1" (:323) is applied by Phonon._ingest here; compare after the same rule
        mg = np.where(m['gamma'] > 0, m['gamma'], -1)
        assert np.allclose(mg, gam, rtol=1e-15, atol=1e-12)
    assert not np.array_equal(g['gamma'], g['gamma_with_isotope'])
