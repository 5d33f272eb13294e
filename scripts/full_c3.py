"""BASELINE config 3: Ge-like cross-plane film (2000 A thick, 500 x 500 A periodic cell), 1e7 particles, 31^3 q-points."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
argv = bench.workload_argv(n)
i = argv.index('--dimensions')
argv[i + 1:i + 4] = ['2000', '500', '500']
args = initialise_parser().parse_args(argv + ['--seed', '2025', '--iterations', str(steps)])
args.results_folder = ''
t0 = time.time()
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(31, 'Ge', temperatures=np.arange(200.0, 401.0, 10.0)))
pop = bench.quiet(Population, args, geo, ph)
t1 = time.time()
psteps = 0
for _ in range(steps // 500):
    bench.quiet(pop.run, 500, geo, ph)
    psteps += 500 * pop.N_p
    print('step %6d  N_p %d  T %.3f..%.3f  kappa %.3f  elapsed %.1f s' % (pop.current_timestep, pop.N_p, pop.subvol_temperature.min(),
          pop.subvol_temperature.max(), pop.kappa, time.time() - t1), flush=True)
tm = pop.engine.timing()
print('set-up %.1f s; %.3e phonon-steps/s end to end; k_sweep %.3f ms' % (t1 - t0, psteps / (time.time() - t1), tm['step_kernel_ms']))
