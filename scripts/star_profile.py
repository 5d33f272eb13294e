"""cProfile of Population.__init__ on the soak's star case (developer probe)."""
import sys, os, time, cProfile, pstats
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
argv = ['--geometry', 'star', '--dimensions', '600', '200', '90', '72', '--subvolumes', 'slice', '8', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5'] + list(A.COMMON) + ['--particles', 'total', '500000', '--seed', '11']
args = initialise_parser().parse_args(argv)
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(9, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
pop = bench.quiet(Population, args, geo, ph)
pr.disable(); print('population %.1f s' % (time.time() - t0), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(16)
t0 = time.time(); bench.quiet(pop.run, 200, geo, ph); print('200 steps %.2f s' % (time.time() - t0))
