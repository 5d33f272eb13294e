"""ONE step clock: the library tallies the heat flux on ITS absolute step
((step + 1) % flux_every == 0, nk_step) while Population writes a convergence row on current_timestep % n_dt_to_conv
(reference Population.py:1762-1767, n_dt_to_conv :41).  A caller that also steps the engine directly -- bench.py's ramp,
warm-up and timed regions: 525 steps with the driver's arguments -- used to leave the two apart, and every row then read a
NaN flux (BENCH_r03: kappa_mean NaN).  Population.run now takes the engine's counter before it steps.
CPU test: the engine is a stand-in with the library's flux cadence and NaN rows (nk_engine.hip: nk_step)."""
import os
import sys

import numpy as np
import pytest

from util import golden_material

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))


class ClockEngine:
    """Stand-in for nanokappa_amd.engine.Engine: accepts every set-up call, keeps an absolute step counter and returns
    tallies whose flux rows are NaN except on the steps the library tallies them."""
    flux_every = 10

    def __init__(self):
        self.stepno = 0
        self.S = self.R = 0
        self.N = 0

    def set_subvolumes(self, centers, volumes, kind, axis, interp, T_sv, rbf=None):
        self.S = len(T_sv)
        self.T0 = np.asarray(T_sv, dtype=float).copy()

    def set_reservoirs(self, facets, T, enter_prob, counter, gen=0, n_leaving=None):
        self.R = len(facets)

    def upload(self, positions, mode, occ, **kw):
        self.N = len(occ)

    def set_params(self, **kw):
        self.flux_every = kw.get('flux_every', 10)

    def timing(self):
        return dict(slots=0, live=self.N)

    def get_step(self):
        return self.stepno

    def step(self, nsteps=1):
        S, R = self.S, self.R
        n_sv = np.full((nsteps, S), self.N // S, dtype=float)
        flux = np.full((nsteps, S, 3), np.nan)
        T = np.tile(np.linspace(301.0, 299.0, S), (nsteps, 1))
        for s in range(nsteps):
            if (self.stepno + s + 1) % self.flux_every == 0:
                flux[s] = 0.0
                flux[s, :, 0] = 1e-3 * (self.stepno + s + 1)
        self.stepno += nsteps
        return dict(T_sv=T, E_sv=np.ones((nsteps, S)), E_raw=np.ones((nsteps, S)), N_sv=n_sv, flux_raw=flux,
                    N_leaving=np.ones((nsteps, R)), res_energy=np.ones((nsteps, R)), res_flux=np.ones((nsteps, R, 3)),
                    N_emitted=np.ones(nsteps))

    def __getattr__(self, name):              # set_material, set_mesh, reserve, init_boundaries, ...: accepted
        if name.startswith('__') or name in ('rough_begin', 'specular_begin', 'kspec_begin', 'build_enter_prob',
                                             'init_particles', 'comm_info'):
            raise AttributeError(name)        # (the host builders run: no device here)
        return lambda *a, **k: None


def build(tmp_path=None):
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population
    import ref_harness_args as A
    args = initialise_parser().parse_args(A.argv_for('ttp', 20000) + ['--seed', '3'])
    args.results_folder = str(tmp_path) if tmp_path else ''
    geo = Geometry(args)
    ph = Phonon(args, 0, material=golden_material())
    eng = ClockEngine()
    pop = Population(args, geo, ph, eng)
    return pop, geo, ph, eng


@pytest.mark.parametrize('behind_the_back', [525, 7, 0, 1000])
def test_rows_land_on_flux_steps_after_direct_engine_steps(behind_the_back):
    """The driver's bench invocation steps the engine 400 + 5 + 20 + 5 x 20 = 525 times before Population.run: every
    convergence row must still carry a finite flux and kappa."""
    pop, geo, ph, eng = build()
    if behind_the_back:
        eng.step(behind_the_back)
    n0 = len(pop.conv_rows)
    pop.run(100, geo, ph)
    rows = pop.conv_rows[n0:]
    assert pop.current_timestep == behind_the_back + 100 == eng.get_step()
    assert len(rows) == (behind_the_back + 100) // 10 - behind_the_back // 10
    for r in rows:
        assert r['step'] % 10 == 0
        assert np.all(np.isfinite(r['phi'])) and np.isfinite(r['kappa']), 'row of step %d has no flux' % r['step']
        assert r['phi'][0, 0] != 0.0


def test_mixed_stepping_keeps_the_hundred_step_bookkeeping():
    """eng.step(7) between two Population.run calls: the second run resumes on the engine's counter, cuts its library calls
    at the engine's multiples of 100 and normalises the interrupted reservoir window by the steps it covers."""
    pop, geo, ph, eng = build()
    pop.run(50, geo, ph)
    assert pop.current_timestep == 50
    eng.step(7)
    n0 = len(pop.conv_rows)
    pop.run(53, geo, ph)
    assert pop.current_timestep == 110
    steps = [r['step'] for r in pop.conv_rows[n0:]]
    assert steps == [60, 70, 80, 90, 100, 110]
    assert all(np.isfinite(r['kappa']) for r in pop.conv_rows[n0:])
    # reservoir balance: the stand-in leaves 1 per step and reservoir; the row of step 60 covers 3 steps (58..60), the others 10:
    # normalised by the steps covered, all rows agree
    en = np.array([r['en_res'] for r in pop.conv_rows[n0:]])
    assert np.allclose(en, en[-1], rtol=1e-12)


def test_untouched_flow_is_unchanged():
    """Without direct engine steps nothing moves: rows at 10, 20, ..., the reference's normalisation by n_dt_to_conv."""
    pop, geo, ph, eng = build()
    for _ in range(30):
        pop.run_timestep(geo, ph)
    assert [r['step'] for r in pop.conv_rows] == [0, 10, 20, 30]
    assert pop._bal_steps == 0
