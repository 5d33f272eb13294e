"""Wall vs stream time of nk_step on the bench workload (developer probe)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
args = initialise_parser().parse_args(bench.workload_argv(n) + ['--seed', '2025'])
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pop = bench.quiet(Population, args, geo, ph)
eng = pop.engine
eng.step(10)
for k in (10, 50, 50, 200):
    t0 = time.perf_counter(); eng.step(k); w = time.perf_counter() - t0
    tm = eng.timing()
    print('steps %4d wall %.3f ms/step  stream %.3f ms/step  sweep %.3f emit_count %.3f tail %.3f' % (
        k, 1e3 * w / k, tm['total_ms'] / k, tm['step_kernel_ms'], tm['emit_kernel_ms'], tm['events_kernel_ms']))
