"""The engine's box store (round 4) keeps no cached next hit on axis-aligned box meshes: an event is read off the particle's
end-of-step position and its hit is evaluated then (nk_device.h nk_box_out / nk_box_first_hit).  The oracle has the same
rule as an option (nk_oracle.h nko_params::box).  Here, on the CPU: the oracle WITH the rule against the oracle with the
reference's rule -- the one the goldens pin (tests/test_oracle_golden.py): the cached n_timesteps, decremented every step
(Population.py:795, :1551) -- on the same ensembles: the same events, hence identical integer tallies and particle sets at
every step, reals equal to rounding."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from util import case_tables, random_population, make_oracle_sim, case_from_args, population_in_mesh


@pytest.mark.parametrize('case,gen', [('ttp', 0), ('ttrrp', 0), ('ttp', 2), ('ttp', 1)])
def test_box_rule_equals_cached_rule(case, gen):
    ct = case_tables(case)
    pos, mode, occ, counter = random_population(ct, 20000, seed=5)
    a = make_oracle_sim(ct, pos, mode, occ, counter, seed=21, gen=gen, box=False)
    b = make_oracle_sim(ct, pos, mode, occ, counter, seed=21, gen=gen, box='auto')
    assert a.p.box == 0 and b.p.box == 1
    # the walls as the box rule sees them: facets 0..5 = -x +x -y +y -z +z, k = -n.o
    assert list(b.p.box_facet) == [0, 1, 2, 3, 4, 5]
    assert np.allclose(list(b.p.box_k), [0.0, -200.0, 0.0, -200.0, 0.0, -200.0], rtol=0, atol=1e-9)
    for s in range(130):                                  # across a contains_check (step 100)
        a.run_timestep()
        b.run_timestep()
        assert np.array_equal(a.N_sv, b.N_sv), 'step %d' % s
        assert np.array_equal(a.N_leaving, b.N_leaving), 'step %d' % s
        assert np.abs(a.T_sv - b.T_sv).max() < 1e-11, 'step %d' % s
    n = a.P.N
    assert b.P.N == n
    assert np.array_equal(a.P.pid[:n], b.P.pid[:n]) and np.array_equal(a.P.mode[:n], b.P.mode[:n])
    assert np.abs(a.P.pos[:n] - b.P.pos[:n]).max() < 1e-10
    assert np.abs(a.P.occ[:n] - b.P.occ[:n]).max() < 1e-13 * max(1.0, np.abs(a.P.occ[:n]).max())
    # the carried next hit: same facet; times equal to the rounding of the drift (the box rule re-derives a hit from where the
    # particle stands when the event is due, the reference from where its flight began)
    assert np.array_equal(a.P.facet[:n], b.P.facet[:n])
    assert np.abs(a.P.n_ts[:n] - b.P.n_ts[:n]).max() < 1e-10


def test_box_rule_only_on_boxes_and_only_for_particles_inside():
    # a film (box 2000 x 500 x 500) qualifies, a cylinder does not
    import ref_harness_args as A
    ct = case_from_args(A.argv_for('film', 20000), species='Ge')
    pos, mode, occ, counter = population_in_mesh(ct, 5000, seed=2)
    assert make_oracle_sim(ct, pos, mode, occ, counter, seed=1).p.box == 1
    ctw = case_from_args(['--geometry', 'cylinder', '--dimensions', '600', '100', '16'] + A.argv_for('wire', 20000)[4:])
    pw, mw, ow, cw = population_in_mesh(ctw, 3000, seed=2)
    assert make_oracle_sim(ctw, pw, mw, ow, cw, seed=1).p.box == 0
    # particles planted OUTSIDE the box with a wall ahead of them: the reference runs their event when they reach it from
    # behind; the box rule cannot express that, so it is switched off for the run (the engine re-deals into the cached layout)
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 2000, seed=5)
    pos[:50, 0] = -30.0
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=3)
    v = ct['ph'].group_vel.reshape(-1, 3)[mode[:50]]
    assert (sim.P.facet[:50][v[:, 0] > 0] >= 0).any()
    assert sim.p.box == 0


def test_no_box_switch(monkeypatch):
    ct = case_tables('ttp')
    pos, mode, occ, counter = random_population(ct, 1000, seed=5)
    monkeypatch.setenv('NK_NO_BOX', '1')
    assert make_oracle_sim(ct, pos, mode, occ, counter, seed=3).p.box == 0
