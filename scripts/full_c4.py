"""BASELINE config 4: Si-like nanowire imported from an ASCII STL (cylinder primitive with 1250 sides = 5000 triangles,
L = 2000 A, R = 200 A, written by Mesh.export_stl and read back), caps T 302/298 K, side wall rough (eta = 5 A),
slice 20 subvolumes along the axis, 31^3 q-points.  usage: full_c4.py [PARTICLES=5e7] [STEPS=300] [MESH_N=31]"""
import sys, os, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
n = float(sys.argv[1]) if len(sys.argv) > 1 else 5e7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mesh_n = int(sys.argv[3]) if len(sys.argv) > 3 else 31
tail = ['--subvolumes', 'slice', '20', '2', '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1',
        '--bound_cond', 'T', 'T', 'R', '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
        '--temp_interp', 'linear', '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', str(int(n)),
        '--seed', '2025', '--iterations', str(steps)]
import threading
def _ticker():
    while True:
        time.sleep(60)
        print('  ... %.0f s' % (time.time() - t0), flush=True)
t0 = time.time()
threading.Thread(target=_ticker, daemon=True).start()
prim = initialise_parser().parse_args(['--geometry', 'cylinder', '--dimensions', '2000', '200', '1250'] + tail)
prim.results_folder = ''
g0 = bench.quiet(Geometry, prim)
tmp = tempfile.mkdtemp()
g0.mesh.export_stl('wire', tmp)
stl = os.path.join(tmp, 'wire.stl')
args = initialise_parser().parse_args(['--geometry', stl, '--dimensions', '1', '1', '1'] + tail)
args.results_folder = ''
geo = bench.quiet(Geometry, args)
print('geometry (primitive -> STL %.1f MB -> import) %.1f s: faces %d facets %d rough %d volume %.4e A^3' % (
    os.path.getsize(stl) / 1e6, time.time() - t0, geo.mesh.n_of_faces, geo.mesh.n_of_facets, len(geo.rough_facets), geo.volume), flush=True)
ph = Phonon(args, 0, material=synthetic.make_material(mesh_n, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
t1 = time.time()
pop = bench.quiet(Population, args, geo, ph)
t2 = time.time()
print('population (specular pairs, roulettes, upload) %.1f s' % (t2 - t1), flush=True)
psteps = 0
block = 100
for _ in range(max(1, steps // block)):
    bench.quiet(pop.run, block, geo, ph)
    psteps += block * pop.N_p
    print('step %6d  N_p %d  T %.3f..%.3f  kappa %.3f  elapsed %.1f s' % (pop.current_timestep, pop.N_p, pop.subvol_temperature.min(),
          pop.subvol_temperature.max(), pop.kappa, time.time() - t2), flush=True)
tm = pop.engine.timing()
print('%.3e phonon-steps/s end to end; k_sweep %.3f ms' % (psteps / (time.time() - t2), tm['step_kernel_ms']))
