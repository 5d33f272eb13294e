"""BASELINE configurations at their FULL size on the GPU, where the oracle cannot follow (1e7 particles x 178 746 modes):
size-independent properties of Population.run_timestep (Population.py:1724-1769) instead of a particle-by-particle check.

  balance       N(t) - N(t-1) = entered(t) - left(t): fill_reservoirs :356-455 / add_reservoir_particles :525-552 against the
                absorptions of boundary_scattering :1568-1608, every step;
  census        the subvolume counts of calculate_energy (:704-717) add up to the live particles of the store;
  state         every particle lies in the solid's box (or, a reference quirk, just behind a reservoir face flying in) with a finite, non-negative occupation, a valid next facet and a
                non-negative time to it (timesteps_to_boundary :797-830), and its mode is an active one;
  determinism   a second engine built from the same arguments gives the same integer tallies step by step (counts,
                entered, left) and the same temperatures to rounding (the LDS atomics add in any order);
  sharding      two ranks' shards (ids, emission ownership; NK_COMM_DRYRUN: no communicator on one GPU) hold together exactly
                the single-rank run's counts of entered particles, and the union of their censuses is the whole one.
"""
import os
import sys

import numpy as np
import pytest

from util import allclose

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def build(cfg, total, comm=None):
    import bench
    from nanokappa_amd import synthetic
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population
    argv, species, _ = bench.config_argv(cfg, total, 200.0)
    args = initialise_parser().parse_args(argv + ['--seed', '2025', '--device', '0'])
    args.results_folder = ''
    geo = bench.quiet(Geometry, args)
    ph = Phonon(args, 0, material=synthetic.make_material(31, species, temperatures=np.arange(200.0, 401.0, 10.0)))
    pop = bench.quiet(Population, args, geo, ph, None, comm)
    return pop, geo, ph


@pytest.mark.parametrize('cfg,total', [('c2', 10000000), ('c3', 10000000), ('c5', 12500000)])
def test_full_size_properties(cfg, total, monkeypatch):
    nsteps = 12
    pop, geo, ph = build(cfg, total)
    eng = pop.engine
    assert pop._init_on_device(geo, ph)
    n0 = int(pop.N_p)
    assert n0 == total
    t = eng.step(nsteps)
    N = t['N_sv'].sum(axis=1)
    prev = np.concatenate(([n0], N[:-1]))
    assert np.array_equal(N - prev, t['N_emitted'] - t['N_leaving'].sum(axis=1)), 'particle balance'
    assert t['N_emitted'].min() > 0 and t['N_leaving'].min() > 0
    assert int(N[-1]) == int(eng.timing()['live']), 'census against the store'
    assert np.all(np.isfinite(t['T_sv'])) and t['T_sv'].min() > 290.0 and t['T_sv'].max() < 310.0
    p = eng.download()
    x = p['positions']
    assert x.shape[0] == int(N[-1])
    lo, hi = geo.mesh.bounds[0] - 1e-6, geo.mesh.bounds[1] + 1e-6
    out = np.any((x < lo) | (x > hi), axis=1)
    # the reference's entry times can be negative for the last particle of a (reservoir, mode) entry ((c - 1 + r) / p > 1,
    # SURVEY quirk list): such a particle starts behind its reservoir face, less than one step's flight away, flying in
    assert out.mean() < 0.01
    v = ph.group_vel.reshape(-1, 3)[p['mode'][out]]
    xo = x[out]
    behind = np.where(xo[:, 0] < lo[0], lo[0] - xo[:, 0], np.where(xo[:, 0] > hi[0], xo[:, 0] - hi[0], 0.0))
    inward = np.where(xo[:, 0] < lo[0], v[:, 0], -v[:, 0])
    assert np.all(behind > 0) and np.all(inward > 0)                                   # only behind the reservoir faces (x)
    tau = behind / inward                                                               # flight time back to the face
    assert np.all(tau <= 1.0 + 1e-9)                                                    # dt = 1 ps
    on_face = xo + v * tau[:, None]
    assert np.all((on_face[:, 1:] >= lo[1:] - 1e-6) & (on_face[:, 1:] <= hi[1:] + 1e-6))
    assert np.all(np.isfinite(p['occupation'])) and p['occupation'].min() >= 0.0
    assert p['facet'].min() >= 0 and p['facet'].max() < geo.n_of_facets
    assert np.all(p['n_timesteps'] >= 0.0) and np.all(np.isfinite(p['n_timesteps']))
    active = ~ph.inactive_modes_mask.ravel()
    assert np.all(active[p['mode']])
    del p, x
    eng.close()
    # determinism
    pop2, _, _ = build(cfg, total)
    t2 = pop2.engine.step(nsteps)
    pop2.engine.close()
    for k in ('N_sv', 'N_emitted', 'N_leaving'):
        assert np.array_equal(t[k], t2[k]), k
    assert allclose(t['T_sv'], t2['T_sv'], rtol=0, atol=0)        # (fixed summation order: bitwise reproducible)
    # sharding (counts only: with local tallies the temperatures differ, trajectories do not depend on them)
    monkeypatch.setenv('NK_COMM_DRYRUN', '1')
    em, census = 0, 0
    for r in (0, 1):
        pr, _, _ = build(cfg, total, comm=(bytes(128), r, 2))
        tr = pr.engine.step(nsteps)
        em = em + tr['N_emitted']
        census = census + tr['N_sv']
        pr.engine.close()
    assert np.array_equal(em, t['N_emitted']) and np.array_equal(census, t['N_sv'])


def test_config4_full_size_balance():
    """BASELINE config 4 at full size (STL-imported 5000-triangle wire, 1250 rough facets, 5e7 particles): the split sweep
    with k_events drawing from all queues -- particle balance and census every step, temperatures between the caps',
    and the same integer tallies from a second engine (rough reflections draw from the particles' ids: deterministic)."""
    import bench
    from nanokappa_amd import synthetic
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population
    total, nsteps = 50000000, 8
    runs = []
    for _ in range(2):
        args, geo = bench.wire_geometry(total)
        args.seed, args.device = [2025], [0]
        ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
        pop = bench.quiet(Population, args, geo, ph, None, None)
        n0 = int(pop.N_p)
        t = pop.engine.step(nsteps)
        live = int(pop.engine.timing()['live'])
        pop.engine.close()
        N = t['N_sv'].sum(axis=1)
        assert np.array_equal(N - np.concatenate(([n0], N[:-1])), t['N_emitted'] - t['N_leaving'].sum(axis=1)), 'particle balance'
        assert int(N[-1]) == live and n0 == total
        assert np.all(np.isfinite(t['T_sv'])) and t['T_sv'].min() > 290.0 and t['T_sv'].max() < 310.0
        runs.append(t)
        del pop, geo, ph
    for k in ('N_sv', 'N_emitted', 'N_leaving'):
        assert np.array_equal(runs[0][k], runs[1][k]), k
