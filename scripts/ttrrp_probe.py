"""Box with two rough walls (C1b of SURVEY 8d) at bench scale: set-up time and step time (developer probe)."""
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
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
mesh_n = int(sys.argv[2]) if len(sys.argv) > 2 else 31
args = initialise_parser().parse_args(A.argv_for('ttrrp', int(n)) + ['--seed', '7'])
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(mesh_n, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
t0 = time.time()
pop = bench.quiet(Population, args, geo, ph)
print('population set-up %.1f s' % (time.time() - t0), flush=True)
eng = pop.engine
eng.step(10)
for k in (20, 50):
    t0 = time.perf_counter(); eng.step(k); w = time.perf_counter() - t0
    tm = eng.timing()
    print('steps %3d wall %.3f ms/step sweep %.3f ms live %d -> %.3e phonon-steps/s' % (k, 1e3 * w / k, tm['step_kernel_ms'], tm['live'], tm['live'] * k / w), flush=True)
