"""Long runs of several configurations through the Population front end: no error, finite tallies, stable particle count
(developer soak test; the 100-step bookkeeping -- contains_check, residue, store growth -- runs many times)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
import numpy as np
import bench
import ref_harness_args as A
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population

GRID = ['--geometry', 'box', '--dimensions', '200', '200', '200', '--subvolumes', 'grid', '3', '3', '2',
        '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '--bound_cond', 'T', 'T', 'P',
        '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
        '--bound_values', '302', '298']
WIRE = ['--geometry', 'cylinder', '--dimensions', '500', '100', '16', '--subvolumes', 'slice', '10', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5']
WIRE400 = ['--geometry', 'cylinder', '--dimensions', '600', '100', '100', '--subvolumes', 'slice', '20', '2',
           '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
           '--bound_values', '302', '298', '5']            # 400 triangles: tables in global memory, face-tree ray caster
STAR = ['--geometry', 'star', '--dimensions', '600', '200', '90', '72', '--subvolumes', 'slice', '8', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5']               # 576 triangles, diagonal slivers (split tree references)
CASTLE = ['--geometry', 'castle', '--dimensions', '90', '40', '70', '45', '8', '5', '1', '--subvolumes', 'slice', '8', '2',
          '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
          '--bound_values', '302', '298', '5']


def common(interp):
    c = list(A.COMMON)
    c[c.index('--temp_interp') + 1] = interp
    return c


CASES = {
    'ttp': (A.BOX_TTP + common('linear'), 1000000, 3000),
    'ttrrp': (A.BOX_TTRRP + common('linear'), 1000000, 3000),
    'ttrrp_k': (A.BOX_TTRRP + common('linear') + ['--bound_scat', 'k'], 300000, 2000),
    'o2o': (A.BOX_TTP + common('nearest') + ['--reservoir_gen', 'one_to_one'], 500000, 3000),
    'fixed_rate': (A.BOX_TTP + common('linear') + ['--reservoir_gen', 'fixed_rate'], 500000, 2000),
    'grid_rbf': (GRID + common('radial'), 300000, 2000),
    'wire': (WIRE + common('linear'), 300000, 3000),
    'castle': (CASTLE + common('linear'), 200000, 2000),
    'wire400': (WIRE400 + common('linear'), 500000, 3000),
    'star': (STAR + common('linear'), 500000, 2000),
    'hot_start': (A.BOX_TTP + common('linear') + ['--temp_dist', 'hot'], 300000, 2000),
}
which = sys.argv[1:] or list(CASES)
for name in which:
    argv, n, steps = CASES[name]
    args = initialise_parser().parse_args(argv + ['--particles', 'total', str(n), '--iterations', str(steps), '--seed', '11'])
    args.results_folder = ''
    t0 = time.time()
    geo = bench.quiet(Geometry, args)
    ph = Phonon(args, 0, material=synthetic.make_material(9, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
    pop = bench.quiet(Population, args, geo, ph)
    n0 = pop.N_p
    bench.quiet(pop.run, steps, geo, ph)
    T = np.asarray(pop.subvol_temperature)
    ok = np.all(np.isfinite(T)) and 290 < T.min() and T.max() < 310 and 0.5 * n0 < pop.N_p < 2.0 * n0
    print('%-10s steps %5d  N_p %8d -> %8d  T %.3f..%.3f  kappa %s  %.1f s  %s' % (
        name, steps, n0, pop.N_p, T.min(), T.max(), ('%.3f' % pop.kappa) if hasattr(pop, 'kappa') else 'n/a',
        time.time() - t0, 'OK' if ok else 'SUSPECT'), flush=True)
    pop.engine.close()
