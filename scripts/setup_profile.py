"""Where does Population's set-up time go on a many-facet wire?  (developer probe: cProfile of Population.__init__)
usage: setup_profile.py N_SIDES PARTICLES MESH_N"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
ns, n, mesh_n = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
argv = ['--geometry', 'cylinder', '--dimensions', '2000', '200', str(ns), '--subvolumes', 'slice', '20', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
        '--temp_interp', 'linear', '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', str(int(n)),
        '--seed', '3']
args = initialise_parser().parse_args(argv)
args.results_folder = ''
t0 = time.time()
geo = bench.quiet(Geometry, args)
print('geometry %.1f s' % (time.time() - t0), flush=True)
ph = Phonon(args, 0, material=synthetic.make_material(mesh_n, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
pop = bench.quiet(Population, args, geo, ph)
pr.disable()
print('population %.1f s' % (time.time() - t0), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
