"""Tree-walk visit counts / kernel time of the ray caster on a strongly non-convex mesh (star wire, 576 triangles)
(developer probe; NK_LIBNAME=libnanokappa_hip_stats.so for the counts, NK_VERBOSE=1 for the kernel time)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
argv = ['--geometry', 'star', '--dimensions', '600', '200', '90', '72', '--subvolumes', 'slice', '8', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1', '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic']
args = initialise_parser().parse_args(argv)
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(5, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
eng = Engine(0, 1)
eng.set_material(ph.tables())
eng.set_mesh(geo.tables())
eng.set_subvolumes(geo.subvol_center, geo.subvol_volume, 0, geo.slice_axis, 1, np.full(geo.n_of_subvols, 300.0))
rng = np.random.default_rng(1)
x = geo.mesh.sample_volume(n, rng)
v = rng.normal(size=(n, 3)) * 40.0
STATS = 'stats' in os.environ.get('NK_LIBNAME', '')
for it in range(2):
    xc, tc, fc = eng.find_boundary(x, v)
    if STATS:
        ne, nl = tc, fc
        tot = (ne + nl)[: n // 64 * 64].reshape(-1, 64)
        print('box families per ray: mean %.1f p99 %d max %d | leaves: mean %.1f p99 %d max %d | wave iterations mean %.1f' % (
            ne.mean(), np.percentile(ne, 99), ne.max(), nl.mean(), np.percentile(nl, 99), nl.max(), tot.max(axis=1).mean()), flush=True)
    else:
        print('mean flight %.1f A, hits %.3f' % (np.mean(tc[np.isfinite(tc)]) * 40.0 * 1.6, np.mean(fc >= 0)), flush=True)
