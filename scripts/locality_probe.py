"""k_sweep time step by step right after a (mode-sorted) upload (developer probe)."""
import sys, os
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
for k in range(40):
    eng.step(1)
    tm = eng.timing()
    if k < 12 or k % 4 == 0:
        print('step %3d sweep %.3f ms emit %.3f tail %.3f live %d' % (k, tm['step_kernel_ms'], tm['emit_kernel_ms'], tm['events_kernel_ms'], tm['live']))
