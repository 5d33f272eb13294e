"""Text outputs against the reference's own sample files (tests/golden/ref_outputs, from readme_fig/test_white_0 of the
reference: 10 slices, 2 reservoirs): convergence.txt header byte for byte and every field of a data row at the same
width, subvolumes.txt header lines and field formats, residue.txt row format.  Also the kappa / reservoir-balance
formulas (Population.py:749-788, :1685-1693) against the deterministic goldens of the frozen step."""
import os
import re

import numpy as np
import pytest

from util import golden, sub, golden_phonon, allclose

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, 'golden', 'ref_outputs')


class _Args(object):
    hdf_file = ['kappa-m313131.hdf5']
    poscar_file = ['POSCAR']


class _Geo(object):
    subvol_type = 'slice'
    slice_axis = 0
    n_of_subvol_con = 0

    def __init__(self, S, bounds):
        self.bounds = np.asarray(bounds, dtype=float)
        ext = self.bounds[1] - self.bounds[0]
        c = np.tile(self.bounds[0] + ext / 2, (S, 1))
        c[:, 0] = self.bounds[0, 0] + (np.arange(S) + 0.5) * ext[0] / S
        self.subvol_center = c
        self.facets_area = np.array([ext[1] * ext[2]] * 2 + [ext[0] * ext[2]] * 2 + [ext[0] * ext[1]] * 2)


def bare_population(S, R, tmp):
    """A Population object with the attributes its writers read, without an engine (the writers are pure formatting)."""
    from nanokappa_amd.population import Population, _Stats
    from nanokappa_amd.constants import Constants
    pop = Population.__new__(Population)
    Constants.__init__(pop)
    rng = np.random.default_rng(3)
    pop.args = _Args()
    pop.results_folder_name = str(tmp)
    pop.rank, pop.nranks = 0, 1
    pop.n_of_subvols, pop.n_of_reservoirs = S, R
    pop.subvol_type = 'slice'
    pop.slice_axis = 0
    pop.current_timestep, pop.t, pop.dt = 10, 10.0, 1.0
    pop.total_energy = 1.234e-3
    pop.res_energy_balance = rng.normal(size=R)
    pop.res_heat_flux = rng.normal(size=(R, 3)) * 1e8
    pop.N_p = 100012
    pop.subvol_temperature = 300 + rng.normal(size=S)
    pop.subvol_energy = rng.random(S)
    pop.subvol_heat_flux = rng.normal(size=(S, 3)) * 1e8
    pop.subvol_N_p = np.full(S, 10001)
    pop.subvol_kappa = rng.random(S) * 80
    pop.kappa = 74.3
    pop.subvol_volume = np.full(S, 5e8)
    pop.n_mean = 10
    pop.conv_rows = []
    pop.f = None
    return pop


def widths(line):
    """(start, end) of every whitespace-separated field."""
    return [(m.start(), m.end()) for m in re.finditer(r'\S+', line)]


def test_convergence_txt_layout(tmp_path):
    ref = open(os.path.join(REF, 'convergence_head.txt')).read().split('\n')
    S, R = 10, 2
    geo = _Geo(S, [[0, 0, 0], [5000, 1000, 1000]])
    pop = bare_population(S, R, tmp_path)
    pop.open_convergence(geo)
    pop.write_convergence(geo)
    got = open(tmp_path / 'convergence.txt').read().split('\n')
    assert got[0] == ref[0]                                       # header: byte for byte (84 columns)
    # a data row: same number of fields, every field ends at the same column as the reference's (right-aligned formats);
    # field 0 is the ISO timestamp
    wr, wg = widths(ref[1]), widths(got[1])
    assert len(wr) == len(wg) == 84
    assert [e for _, e in wr] == [e for _, e in wg]
    assert re.match(r'\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\.\d{6} ', got[1])


def test_subvolumes_txt_layout(tmp_path):
    from nanokappa_amd.population import _Stats
    ref = open(os.path.join(REF, 'subvolumes.txt')).read().split('\n')
    S, R = 10, 2
    geo = _Geo(S, [[0, 0, 0], [5000, 1000, 1000]])
    pop = bare_population(S, R, tmp_path)
    pop.current_timestep = 100
    rng = np.random.default_rng(5)
    v = _Stats(pop)
    v.mean_T, v.std_T = 300 + rng.random(S), rng.random(S) * 1e-2
    v.mean_sv_phi, v.std_sv_phi = rng.normal(size=3 * S) * 1e8, rng.random(3 * S) * 1e7
    v.mean_sv_k, v.std_sv_k = rng.random(S) * 70, rng.random(S)
    pop.view = v
    pop.particles = lambda: dict(modes=np.zeros((3, 2), dtype=int), positions=np.zeros((3, 3)), occupation=np.ones(3))
    pop.write_final_state(geo)
    got = open(tmp_path / 'subvolumes.txt').read().split('\n')
    assert got[0] == ref[0] and got[2] == ref[2] and got[3] == ref[3]      # title, file names, column names
    assert got[1].startswith('# Date and time: ')
    num = lambda s: re.sub(r'[0-9]', '9', re.sub(r'-', '', s))             # digits -> 9, signs dropped: the format of a row
    assert len(got) == len(ref)
    for a, b in zip(got[4:4 + S], ref[4:4 + S]):
        fa, fb = a.split(', '), b.split(', ')
        assert len(fa) == len(fb) == 15
        for x, y in zip(fa[1:], fb[1:]):
            assert ('e' in x) == ('e' in y) and len(num(x).split('e')[0]) == len(num(y).split('e')[0])
    # particle_data.txt: header + '%d, %d, %.3f, %.3f, %.3f, %.6e' rows (Population.py:2071-2091)
    pd = open(tmp_path / 'particle_data.txt').read().split('\n')
    assert pd[0] == '# Particles final state data ' and pd[3] == '# q-point, branch, pos x [angs], pos y [angs], pos z [angs], occupation'
    assert re.match(r'^0, 0, 0\.000, 0\.000, 0\.000, 1\.000000e\+00$', pd[4])


def test_residue_txt_layout(tmp_path):
    ref = open(os.path.join(REF, 'residue_head.txt')).read().split('\n')
    S, R = 10, 2
    geo = _Geo(S, [[0, 0, 0], [5000, 1000, 1000]])
    pop = bare_population(S, R, tmp_path)
    pop.conv_crit, pop.conv_count_min = 0.0, 10
    pop.initialise_residue(geo)
    from nanokappa_amd.population import _Stats
    rng = np.random.default_rng(7)
    v = _Stats(pop)
    v.mean_T, v.std_T = 297 + rng.random(S), rng.random(S) * 1e-2
    v.mean_sv_phi, v.std_sv_phi = rng.normal(size=3 * S), rng.random(3 * S) * 1e-3
    v.mean_en_res, v.std_en_res = rng.normal(size=R), rng.random(R) * 1e-3
    v.mean_sv_k, v.std_sv_k = np.full(S, np.nan), np.full(S, np.nan)
    pop.view = v
    pop.update_residue(geo)
    got = open(tmp_path / 'residue.txt').read().split('\n')
    fr, fg = ref[0].split(), got[0].split()
    assert len(fr) == len(fg) == 3 * S + R                        # T, phi along the slice axis, en_res, kappa per slice
    assert [e for _, e in widths(ref[0])] == [e for _, e in widths(got[0])]
    assert ref[0].endswith(' ') and got[0].endswith(' ')


@pytest.mark.parametrize('variant', ['lin', 'near', 'fixed', 'tref'])
def test_kappa_and_reservoir_balance_formulas(variant, tmp_path):
    """calculate_kappa and adjust_reservoir_balance on the frozen step's tallies: the reference's deterministic values."""
    gs = sub(golden('step'), variant)
    gm = sub(golden('mesh'), 'box200ttp')
    ph = golden_phonon()
    S, R = 20, 2
    geo = _Geo(S, gm['bounds'])
    pop = bare_population(S, R, tmp_path)
    pop.subvol_temperature = gs['post_subvol_temperature']
    pop.subvol_heat_flux = gs['heat_flux']
    pop.subvol_N_p = gs['post_subvol_N_p']
    pop.N_p = int(gs['post_subvol_N_p'].sum())
    pop.res_facet_temperature = gs['res_facet_temperature']
    pop.calculate_kappa(geo)
    assert allclose(pop.subvol_kappa, gs['subvol_kappa'], rtol=1e-12, atol=0)
    assert np.isclose(pop.kappa, float(gs['kappa']), rtol=1e-12, atol=0)
    pop.res_facet = gm['res_facets']
    pop.res_energy_balance = gs['mid_res_energy_balance'].copy()
    pop.res_heat_flux = gs['mid_res_heat_flux'].copy()
    pop.particle_density = float(gs['particle_density'])
    pop.n_dt_to_conv = 10
    geo.facets_area = gm['facets_area']
    pop.adjust_reservoir_balance(geo, ph)
    assert allclose(pop.res_energy_balance, gs['adj_res_energy_balance'], rtol=1e-12, atol=0)
    assert allclose(pop.res_heat_flux, gs['adj_res_heat_flux'], rtol=1e-12, atol=1e-300)
