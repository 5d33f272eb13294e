"""BASELINE config 2 in full: 1e7 particles, 10 000 timesteps, 31^3 q-points, through the Population front end."""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
args = initialise_parser().parse_args(bench.workload_argv(n) + ['--seed', '2025', '--iterations', str(steps)])
args.results_folder = ''
t0 = time.time()
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pop = bench.quiet(Population, args, geo, ph)
t1 = time.time()
psteps = 0
done = 0
while done < steps:
    bench.quiet(pop.run, 1000, geo, ph)
    done += 1000
    psteps += 1000 * pop.N_p
    print('step %6d  N_p %d  T %.3f..%.3f  kappa %.3f  elapsed %.1f s' % (pop.current_timestep, pop.N_p, pop.subvol_temperature.min(),
          pop.subvol_temperature.max(), pop.kappa, time.time() - t1), flush=True)
t2 = time.time()
rows = pop.conv_rows[-500:]
k = np.array([r['kappa'] for r in rows])
print('set-up %.1f s; %d steps in %.1f s = %.3e phonon-steps/s (front end included); kappa over the last 5000 steps %.3f +- %.3f W/mK'
      % (t1 - t0, steps, t2 - t1, psteps / (t2 - t1), k.mean(), k.std() / np.sqrt(k.size)))
