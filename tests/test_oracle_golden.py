"""Pins the CPU oracle (oracle/nk_oracle.c) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from util import golden, sub, golden_phonon, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
import nk_oracle as O  # noqa: E402

c_dp, c_ip = O.c_dp, O.c_ip


def P(a, t=c_dp):
    return a.ctypes.data_as(t)


@pytest.mark.parametrize('name', ['box200', 'box200ttp', 'box5000', 'cyl'])
def test_find_boundary(name):
    g = sub(golden('mesh'), name)
    mesh = O.make_mesh(g)
    x = np.ascontiguousarray(g['ray_x']); v = np.ascontiguousarray(g['ray_v'])
    n = x.shape[0]
    xc = np.zeros((n, 3)); tc = np.zeros(n); fc = np.zeros(n, dtype=np.int32)
    O.lib().nko_find_boundary(C.byref(mesh), C.c_int64(n), P(x), P(v), P(xc), P(tc), P(fc, c_ip))
    assert np.array_equal(fc, g['ray_fc'])
    hit = fc >= 0
    assert rel_err(tc[hit], g['ray_tc'][hit]) < 1e-12
    assert np.all(np.isinf(tc[~hit]))
    assert np.allclose(xc[hit], g['ray_xc'][hit], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize('name', ['box200', 'box5000', 'cyl'])
def test_classifier(name):
    g = sub(golden('mesh'), name)
    sv = O.make_subvols(g['subvol_center'], g['subvol_volume'], 0, int(g['slice_axis']), 1)
    x = np.ascontiguousarray(g['cls_x'])
    out = np.zeros(x.shape[0], dtype=np.int32)
    O.lib().nko_classify(C.byref(sv), C.c_int64(x.shape[0]), P(x), P(out, c_ip))
    assert np.array_equal(out, g['cls_id'])


def test_material_functions():
    g = golden('phonon')
    ph = golden_phonon()
    mat = O.make_material(ph.tables())
    n = g['s_T'].shape[0]
    L = O.lib()
    mode = np.ascontiguousarray(g['s_q'] * ph.number_of_branches + g['s_j'], dtype=np.int32)
    om = np.ascontiguousarray(ph.omega.ravel()[mode])
    out = np.zeros(n)
    L.nko_occupation(C.byref(mat), C.c_int64(n), P(np.ascontiguousarray(g['s_T'])), P(om), P(out))
    assert rel_err(out, g['s_occ']) < 1e-13
    L.nko_lifetime(C.byref(mat), C.c_int64(n), P(np.ascontiguousarray(g['s_T'])), P(mode, c_ip), P(out))
    assert rel_err(out, g['s_tau']) < 1e-12
    # feed the reference's own E(T) table so the inverse is compared like for like
    t = ph.tables()
    t['energy_array'] = g['energy_array']
    mat2 = O.make_material(t)
    L.nko_T_of_E(C.byref(mat2), C.c_int64(n), P(np.ascontiguousarray(g['s_E'])), P(out))
    assert rel_err(out, g['s_T_of_E']) < 1e-13
    L.nko_E_of_T(C.byref(mat2), C.c_int64(n), P(np.ascontiguousarray(g['s_Tw'])), P(out))
    assert rel_err(out, g['s_E_of_T']) < 1e-13


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    L = O.lib()
    u32 = C.c_uint32

    def run(ctr, key):
        c = (u32 * 4)(*ctr); k = (u32 * 2)(*key); o = (u32 * 4)()
        L.nko_philox4x32_10(c, k, o)
        return [int(v) for v in o]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _empty_rough():
    z = np.zeros(0)
    return O.make_rough(np.zeros(0, dtype=np.int32), z, np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.int32), z)


def oracle_from_step_golden(variant):
    gm = sub(golden('mesh'), 'box200ttp')
    gs = sub(golden('step'), variant)
    ph = golden_phonon()
    J = ph.number_of_branches
    mat = O.make_material(ph.tables())
    mesh = O.make_mesh(gm)
    interp = {'lin': 1, 'near': 0, 'fixed': 1, 'tref': 1}[variant]
    sv = O.make_subvols(gm['subvol_center'], gm['subvol_volume'], 0, int(gm['slice_axis']), interp)
    M = ph.number_of_qpoints * J
    res = O.make_reservoirs(gm['res_facets'], gs['res_facet_temperature'], np.zeros((2, M)), np.zeros((2, M)))
    par = O.make_params(dt=1.0, norm_fixed=(variant == 'fixed'), particle_density=float(gs['particle_density']),
                        T_ref=(300.0 if variant == 'tref' else None), seed=1)
    n = gs['pre_positions'].shape[0]
    store = O.ParticleStore(n + 16)
    store.load(gs['pre_positions'], gs['pre_modes'][:, 0] * J + gs['pre_modes'][:, 1], gs['pre_occupation'],
               gs['pre_n_timesteps'], gs['pre_collision_facets'])
    sim = O.OracleSim(mat, mesh, sv, res, _empty_rough(), par, store, gs['pre_subvol_temperature'])
    return sim, gs, gm, ph


@pytest.mark.parametrize('variant', ['lin', 'near', 'fixed', 'tref'])
def test_frozen_step(variant):
    """drift -> boundary_scattering -> refresh_temperatures -> lifetime_scattering -> heat flux on a frozen
    reference state (BCs T T P: no random numbers are consumed)."""
    sim, gs, gm, ph = oracle_from_step_golden(variant)
    L, r = sim.L, sim.ref
    J = ph.number_of_branches
    L.nko_drift(r(sim.mat), r(sim.p), r(sim.P.s))
    L.nko_boundary_scattering(r(sim.mat), r(sim.mesh), r(sim.sv), r(sim.res), r(sim.rough), r(sim.p),
                              P(sim.T_sv), C.c_int64(0), r(sim.P.s), P(sim.N_leaving, O.c_lp),
                              P(sim.res_energy), P(sim.res_flux))
    n = sim.P.N
    assert n == gs['mid_positions'].shape[0]
    assert np.array_equal(sim.N_leaving, gs['mid_N_leaving'])
    assert np.array_equal(sim.P.mode[:n], gs['mid_modes'][:, 0] * J + gs['mid_modes'][:, 1])
    assert np.array_equal(sim.P.facet[:n], gs['mid_collision_facets'])
    assert np.allclose(sim.P.pos[:n], gs['mid_positions'], rtol=1e-12, atol=1e-9)
    assert np.allclose(sim.P.n_ts[:n], gs['mid_n_timesteps'], rtol=1e-9, atol=1e-9)
    assert rel_err(sim.P.occ[:n], gs['mid_occupation']) < 1e-13
    assert rel_err(sim.res_energy, gs['mid_res_energy_balance']) < 1e-10
    assert np.allclose(sim.res_flux, gs['mid_res_heat_flux'], rtol=1e-10, atol=1e-12)

    L.nko_refresh_temperatures(r(sim.mat), r(sim.sv), r(sim.p), r(sim.P.s), P(sim.T_sv), P(sim.E_sv),
                               P(sim.N_sv, O.c_lp), P(sim.E_raw))
    assert np.array_equal(sim.N_sv, gs['post_subvol_N_p'])
    assert np.array_equal(sim.P.sv[:n], gs['post_subvol_id'])
    assert np.allclose(sim.P.energy[:n], gs['energies'], rtol=1e-9, atol=1e-16)
    assert rel_err(sim.E_sv, gs['post_subvol_energy']) < 1e-12
    assert np.allclose(sim.T_sv, gs['post_subvol_temperature'], rtol=0, atol=1e-7)
    assert np.allclose(sim.P.temp[:n], gs['post_temperatures'], rtol=0, atol=1e-7)

    L.nko_lifetime_scattering(r(sim.mat), r(sim.p), r(sim.P.s))
    assert rel_err(sim.P.occ[:n], gs['post_occupation']) < 1e-9
    L.nko_heat_flux(r(sim.mat), r(sim.sv), r(sim.p), r(sim.P.s), P(sim.N_sv, O.c_lp), P(sim.flux))
    assert np.allclose(sim.flux, gs['heat_flux'], rtol=1e-8, atol=1e-3)


def rough_from_setup(model, J):
    gs = sub(golden('setup'), model)
    gm = sub(golden('mesh'), 'box200')
    sm = gs['spec_map']
    flat = np.where(sm[..., 0] >= 0, sm[..., 0] * J + sm[..., 1], -1)
    degen = None
    if model == 'k':
        di = gs['degen_index'].astype(int)
        dg = gs['degeneracies'].astype(int).reshape(-1, 3)
        degen = np.where(di > -1, dg[np.clip(di, 0, max(dg.shape[0] - 1, 0)), 2] if dg.shape[0] else -1, -1)
    return O.make_rough(gm['rough_facets'], gs['specularity'], gs['true_specular'], flat,
                        gs['creation_roulette'], degen), gm, gs


@pytest.mark.parametrize('model', ['velocity', 'k'])
def test_reflect(model):
    """select_reflected_modes / pick_diffuse_modes with the uniforms the reference drew."""
    ph = golden_phonon()
    J = ph.number_of_branches
    rough, gm, _ = rough_from_setup(model, J)
    g = sub(golden('reflect'), model)
    mat = O.make_material(ph.tables())
    mesh = O.make_mesh(gm)
    sv = O.make_subvols(gm['subvol_center'], gm['subvol_volume'], 0, int(gm['slice_axis']), 1)
    n = g['facets'].shape[0]
    mode_in = np.ascontiguousarray(g['in_modes'][:, 0] * J + g['in_modes'][:, 1], dtype=np.int32)
    fac = np.ascontiguousarray(g['facets'], dtype=np.int32)
    r_deg = np.nan_to_num(g['r_deg'], nan=0.0)
    r_diff = np.nan_to_num(g['r_diff'], nan=0.0)
    mo = np.zeros(n, dtype=np.int32); no = np.zeros(n); oo = np.zeros(n)
    T_sv = np.ascontiguousarray(g['subvol_temperature'])
    O.lib().nko_reflect(C.byref(mat), C.byref(mesh), C.byref(sv), C.byref(rough), P(T_sv), C.c_int64(n),
                        P(fac, c_ip), P(mode_in, c_ip), P(np.ascontiguousarray(g['col_pos'])),
                        P(np.ascontiguousarray(g['n_in'])), P(np.ascontiguousarray(g['omega_in'])),
                        P(np.ascontiguousarray(g['r_spec'])), P(np.ascontiguousarray(r_deg)),
                        P(np.ascontiguousarray(r_diff)), P(mo, c_ip), P(no), P(oo))
    assert np.array_equal(mo, g['out_modes'][:, 0] * J + g['out_modes'][:, 1])
    assert rel_err(oo, g['omega_out']) < 1e-14
    assert rel_err(no, g['n_out']) < 1e-12


@pytest.mark.parametrize('scale', ['lo', 'hi'])
def test_emission(scale):
    """fill_reservoirs('constant') + add_reservoir_particles.  Deterministic parts (counters, which modes
    enter and how many, the level-1 entry time, the map (x0, v, dt_in) -> particle) must equal the
    reference; the uniformly drawn parts are checked as distributions."""
    g = sub(golden('emission'), scale)
    gm = sub(golden('mesh'), 'box200ttp')
    ph = golden_phonon()
    J = ph.number_of_branches
    M = ph.number_of_qpoints * J
    mat = O.make_material(ph.tables())
    mesh = O.make_mesh(gm)
    res = O.make_reservoirs(gm['res_facets'], [302.0, 298.0], g['enter_prob'], g['counter_pre'].copy())
    cap = g['res_modes'].shape[0] + 64
    O.attach_emission_taps(res, cap)
    par = O.make_params(seed=99)
    store = O.ParticleStore(cap)
    n = O.lib().nko_emit(C.byref(mat), C.byref(mesh), C.byref(res), C.byref(par), C.c_int64(5), C.c_int32(0),
                         C.c_int32(1), C.byref(store.s))
    assert n == g['res_modes'].shape[0]
    assert np.allclose(res.counter_array.reshape(g['counter_post'].shape), g['counter_post'], rtol=0, atol=1e-15)
    # same multiset of (reservoir, mode)
    ref_r = np.searchsorted(gm['res_facets'], g['res_facet_id'])
    ref_key = np.sort(ref_r * M + g['res_modes'][:, 0] * J + g['res_modes'][:, 1])
    my_key = np.sort(res.tap_res[:n].astype(np.int64) * M + store.mode[:n])
    assert np.array_equal(ref_key, my_key)
    # level-1 entry times: the reference lists, per reservoir, levels c_max..1; the level-1 block is last
    dt_ref = {}
    for r in range(2):
        sel = np.nonzero(ref_r == r)[0]
        c = np.floor(g['enter_prob'][r]) + (g['counter_pre'][r] + g['enter_prob'][r] - np.floor(g['enter_prob'][r]) >= 1)
        n1 = int((c >= 1).sum())
        blk = sel[-n1:]
        for i in blk:
            dt_ref[(r, int(g['res_modes'][i, 0] * J + g['res_modes'][i, 1]))] = g['res_dt_in'][i]
    lvl1 = np.nonzero(res.tap_level[:n] == 1)[0]
    assert len(lvl1) == len(dt_ref)
    mine = np.array([res.tap_dt_in[i] for i in lvl1])
    theirs = np.array([dt_ref[(int(res.tap_res[i]), int(store.mode[i]))] for i in lvl1])
    assert np.allclose(mine, theirs, rtol=1e-12, atol=1e-13)
    # random levels: same law (dt_in = dt*(1-(level-1+u)/p)), compare moments with the reference's draws
    hi = res.tap_level[:n] > 1
    if hi.any():
        p = g['enter_prob'].reshape(2, -1)[res.tap_res[:n][hi], store.mode[:n][hi]]
        u = (1.0 - res.tap_dt_in[:n][hi]) * p - (res.tap_level[:n][hi] - 1)
        assert u.min() >= 0 and u.max() < 1
        assert abs(u.mean() - 0.5) < 4 * (1 / 12 / u.size) ** 0.5
        assert abs(res.tap_dt_in[:n].mean() - g['res_dt_in'].mean()) < 0.02
    # positions: on the facet, uniform (compare first two moments with the reference's sample)
    for r, facet in enumerate(gm['res_facets']):
        x0 = res.tap_x0[:n][res.tap_res[:n] == r]
        xr = g['res_positions'][ref_r == r]
        assert np.allclose(x0[:, 0], gm['facet_centroid'][facet, 0], atol=1e-9)
        assert x0[:, 1:].min() >= 0 and x0[:, 1:].max() <= 200
        se = 200 / 12 ** 0.5 / x0.shape[0] ** 0.5
        assert np.all(np.abs(x0[:, 1:].mean(axis=0) - 100) < 5 * se)
        assert np.all(np.abs(x0[:, 1:].std(axis=0) - xr[:, 1:].std(axis=0)) < 10 * se)
    # deterministic map (x0, v, dt_in) -> particle state, against the oracle's own primitives and the
    # reference's add_reservoir_particles outputs
    vg = ph.group_vel.reshape(-1, 3)
    v = vg[store.mode[:n]]
    assert np.allclose(store.pos[:n], res.tap_x0[:n] + v * res.tap_dt_in[:n, None], rtol=1e-13, atol=1e-11)
    vr = vg[g['res_modes'][:, 0] * J + g['res_modes'][:, 1]]
    xr = np.ascontiguousarray(g['res_positions']); vr = np.ascontiguousarray(vr)
    m = xr.shape[0]
    xc = np.zeros((m, 3)); tc = np.zeros(m); fc = np.zeros(m, dtype=np.int32)
    O.lib().nko_find_boundary(C.byref(mesh), C.c_int64(m), P(xr), P(vr), P(xc), P(tc), P(fc, c_ip))
    assert np.array_equal(fc, g['new_collision_facets'])
    assert np.allclose(tc / 1.0 - g['res_dt_in'] / 1.0, g['new_n_timesteps'], rtol=1e-12, atol=1e-12)
    assert np.allclose(xr + vr * g['res_dt_in'][:, None], g['new_positions'], rtol=1e-13, atol=1e-11)
    occ = ph.calculate_occupation(np.array([302.0, 298.0])[res.tap_res[:n]], ph.omega.ravel()[store.mode[:n]])
    assert rel_err(store.occ[:n], occ) < 1e-13


def test_one_to_one_generator_counts_and_modes():
    """fill_reservoirs 'one_to_one' (Population.py:457-489) in the oracle: the first step emits round(sum enter_prob)
    per reservoir (:344), every later step what left through the facet at the previous one (:466); the drawn modes
    follow the cumulative enter_prob (:467-472): only modes that can enter, frequencies ~ probabilities."""
    from util import case_tables, random_population, make_oracle_sim, first_n_leaving
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 20000, seed=3)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=9, gen=2)
    first = first_n_leaving(ct['enter_prob'])
    left_prev = first.copy()
    M = ct['enter_prob'].reshape(2, -1).shape[1]
    seen = np.zeros((2, M))
    for s in range(12):
        n0 = sim.P.N
        sim.run_timestep()
        new = sim.P.pid[:sim.P.N] >> 40 == ((s + 1) & 0xFFFFFF)
        # emitted particles of this step that are still alive + those absorbed again cannot exceed what was due
        assert new.sum() <= left_prev.sum()
        pid = sim.P.pid[:sim.P.N][new]
        r = (pid >> 32) & 0xFF
        np.add.at(seen, (r.astype(int), sim.P.mode[:sim.P.N][new]), 1)
        left_prev = sim.N_leaving[:2].copy()
        assert np.array_equal(sim.res.n_leaving_array, left_prev)
    ep = ct['enter_prob'].reshape(2, -1)
    assert seen[ep == 0].sum() == 0                       # a mode that cannot enter is never drawn
    # chi-square-ish: group modes into 20 probability-ordered bins per reservoir
    for r in range(2):
        order = np.argsort(ep[r])
        bins = np.array_split(order, 20)
        obs = np.array([seen[r, b].sum() for b in bins])
        exp = np.array([ep[r, b].sum() for b in bins]) / ep[r].sum() * obs.sum()
        ok = exp > 20
        assert np.all(np.abs(obs[ok] - exp[ok]) < 6 * np.sqrt(exp[ok]) + 0.12 * exp[ok])


@pytest.mark.parametrize('scale', ['lo', 'hi'])
def test_emission_fixed_rate_replay(scale):
    """fill_reservoirs('fixed_rate') (Population.py:408-455) pinned deterministically: the oracle's fixed_rate branch, fed the
    dice array the reference drew (tests/golden/emission_fixed.npz, the first rand call of the step, :410), must make the
    reference's decisions -- which (reservoir, mode) entries emit and how many (:415-417), the level-1 entry time
    dt (1 - dice / p) (:440) -- and hand the particles to add_reservoir_particles the same way (:525-552)."""
    g = sub(golden('emission_fixed'), scale)
    gm = sub(golden('mesh'), 'box200ttp')
    ph = golden_phonon()
    J = ph.number_of_branches
    M = ph.number_of_qpoints * J
    mat = O.make_material(ph.tables())
    mesh = O.make_mesh(gm)
    ep = g['enter_prob'].reshape(2, M)
    res = O.make_reservoirs(gm['res_facets'], [302.0, 298.0], ep, np.zeros_like(ep), gen=1)
    O.attach_dice(res, g['dice'].reshape(2, M))
    cap = g['res_modes'].shape[0] + 64
    O.attach_emission_taps(res, cap)
    par = O.make_params(seed=99)
    store = O.ParticleStore(cap)
    n = O.lib().nko_emit(C.byref(mat), C.byref(mesh), C.byref(res), C.byref(par), C.c_int64(5), C.c_int32(0),
                         C.c_int32(1), C.byref(store.s))
    assert n == g['res_modes'].shape[0]
    # same multiset of (reservoir, mode): every emitting entry and its particle count
    ref_r = np.searchsorted(gm['res_facets'], g['res_facet_id'])
    ref_key = np.sort(ref_r * M + g['res_modes'][:, 0] * J + g['res_modes'][:, 1])
    my_key = np.sort(res.tap_res[:n].astype(np.int64) * M + store.mode[:n])
    assert np.array_equal(ref_key, my_key)
    # level-1 entry times: per reservoir the reference lists the levels c_max .. 1, the level-1 block last (:430-447)
    dice = g['dice'].reshape(2, M)
    dt_ref = {}
    for r in range(2):
        sel = np.nonzero(ref_r == r)[0]
        c = np.floor(ep[r]) + (dice[r] <= ep[r] - np.floor(ep[r]))
        n1 = int((c >= 1).sum())
        for i in sel[-n1:]:
            dt_ref[(r, int(g['res_modes'][i, 0] * J + g['res_modes'][i, 1]))] = g['res_dt_in'][i]
    lvl1 = np.nonzero(res.tap_level[:n] == 1)[0]
    assert len(lvl1) == len(dt_ref)
    mine = np.array([res.tap_dt_in[i] for i in lvl1])
    theirs = np.array([dt_ref[(int(res.tap_res[i]), int(store.mode[i]))] for i in lvl1])
    assert np.allclose(mine, theirs, rtol=1e-12, atol=1e-13)
    # higher levels: the law dt (1 - (level - 1 + u) / p) with u in [0, 1)
    hi = res.tap_level[:n] > 1
    if hi.any():
        p = ep[res.tap_res[:n][hi], store.mode[:n][hi]]
        u = (1.0 - res.tap_dt_in[:n][hi]) * p - (res.tap_level[:n][hi] - 1)
        assert u.min() >= -1e-12 and u.max() < 1 + 1e-12
    # the deterministic map (x0, v, dt_in) -> particle state against the reference's add_reservoir_particles outputs
    vg = ph.group_vel.reshape(-1, 3)
    vr = np.ascontiguousarray(vg[g['res_modes'][:, 0] * J + g['res_modes'][:, 1]])
    xr = np.ascontiguousarray(g['res_positions'])
    m = xr.shape[0]
    xc = np.zeros((m, 3)); tc = np.zeros(m); fc = np.zeros(m, dtype=np.int32)
    O.lib().nko_find_boundary(C.byref(mesh), C.c_int64(m), P(xr), P(vr), P(xc), P(tc), P(fc, c_ip))
    assert np.array_equal(fc, g['new_collision_facets'])
    assert np.allclose(tc - g['res_dt_in'], g['new_n_timesteps'], rtol=1e-12, atol=1e-12)
    assert np.allclose(xr + vr * g['res_dt_in'][:, None], g['new_positions'], rtol=1e-13, atol=1e-11)
    occ = ph.calculate_occupation(np.array([302.0, 298.0])[res.tap_res[:n]], ph.omega.ravel()[store.mode[:n]])
    assert rel_err(store.occ[:n], occ) < 1e-13
    assert rel_err(np.sort(store.occ[:n]), np.sort(g['res_occupation'])) < 1e-12
