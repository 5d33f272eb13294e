"""Does the converged state depend on the number of particles?  (developer probe)  Golden material (9^3 x 6 modes), box
T T P as in the reference's statistical goldens; window = convergence rows 50..100 (steps 500-1000)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
import numpy as np
import bench
from util import golden, golden_material
import ref_harness_args as A
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population


def run(n, seed):
    args = initialise_parser().parse_args(A.argv_for('ttp', n) + ['--seed', str(seed)])
    args.results_folder = ''
    geo = bench.quiet(Geometry, args)
    ph = Phonon(args, 0, material=golden_material())
    pop = bench.quiet(Population, args, geo, ph)
    T, k = [], []
    for i in range(100):
        bench.quiet(pop.run, 10, geo, ph)
        if i >= 50:
            T.append(pop.subvol_temperature.copy()); k.append(pop.kappa)
    pop.engine.close()
    return np.mean(T, axis=0), np.mean(k)


g = golden('stats_ttp')
rows = g['rows']
Tr = rows[:, 50:, 3:23].mean(axis=1); kr = rows[:, 50:, 2].mean(axis=1)
print('reference 1e5 x %d: T0 %.4f +- %.4f  T1 %.4f  T18 %.4f  T19 %.4f +- %.4f  kappa %.4f +- %.4f' % (
    Tr.shape[0], Tr[:, 0].mean(), Tr[:, 0].std(ddof=1) / np.sqrt(Tr.shape[0]), Tr[:, 1].mean(), Tr[:, 18].mean(), Tr[:, 19].mean(),
    Tr[:, 19].std(ddof=1) / np.sqrt(Tr.shape[0]), kr.mean(), kr.std(ddof=1) / np.sqrt(kr.size)), flush=True)
for n, seeds in ((30000, 32), (100000, 32), (300000, 8), (1000000, 4), (3000000, 2)):
    res = [run(n, 500 + s) for s in range(seeds)]
    T = np.array([r[0] for r in res]); k = np.array([r[1] for r in res])
    se = lambda a: a.std(ddof=1) / np.sqrt(len(a)) if len(a) > 1 else float('nan')
    print('engine %8d x %2d: T0 %.4f +- %.4f  T1 %.4f  T18 %.4f  T19 %.4f +- %.4f  kappa %.4f +- %.4f' % (
        n, seeds, T[:, 0].mean(), se(T[:, 0]), T[:, 1].mean(), T[:, 18].mean(), T[:, 19].mean(), se(T[:, 19]), k.mean(), se(k)), flush=True)
