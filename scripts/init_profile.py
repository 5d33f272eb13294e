"""Where does Population's set-up time go on the bench workload?  (developer probe: cProfile of Population.__init__)
usage: init_profile.py PARTICLES"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
n = float(sys.argv[1])
args = initialise_parser().parse_args(bench.workload_argv(int(n), 200.0) + ['--seed', '2025'])
args.results_folder = ''
geo = bench.quiet(Geometry, args)
t0 = time.time()
ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
print('phonon %.1f s' % (time.time() - t0), flush=True)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
pop = bench.quiet(Population, args, geo, ph)
pr.disable()
print('population %.1f s' % (time.time() - t0), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
