"""bench.py's CPU legs (the oracle as baseline) on a small material: they run, and report what they used."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def test_cpu_baselines_run():
    import bench
    from nanokappa_amd import synthetic
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.phonon import Phonon
    args = initialise_parser().parse_args(bench.workload_argv(1000000, 200.0) + ['--seed', '1'])
    args.results_folder = ''
    geo = bench.quiet(Geometry, args)
    ph = Phonon(args, 0, material=synthetic.make_material(5, 'Si', temperatures=np.arange(200.0, 401.0, 50.0)))
    psteps, dt, steps = bench._oracle_run(geo, ph, 20000, 3, 0.5, 5)
    assert steps >= 1 and psteps > 10000 and dt > 0
    r = bench.cpu_baseline_all_cores(geo, ph, 5, 'box 200 A, T T P', seconds_target=0.5)
    if r is not None:                      # a single-core host has no such leg
        assert r['cores'] >= 2 and r['value'] > 0 and r['kind'] == 'port' and 'worker processes' in r['sample']
    one = bench.cpu_baseline(geo, ph, 5, 'box 200 A, T T P', n=20000, seconds_target=0.5)
    assert one['cores'] == 1 and one['value'] > 0 and '20000 particles' in one['sample']
