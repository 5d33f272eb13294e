"""Ray caster alone on a many-facet wire (developer probe): kernel time per class of ray, printed by the library under
NK_VERBOSE.  usage: NK_VERBOSE=1 ray_probe.py N_SIDES RAYS"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.engine import Engine
ns, n = int(sys.argv[1]), int(float(sys.argv[2]))
argv = ['--geometry', 'cylinder', '--dimensions', '2000', '200', str(ns), '--subvolumes', 'slice', '20', '2',
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
b = geo.mesh.bounds
c = 0.5 * (b[0] + b[1])
R = 0.5 * (b[1, 0] - b[0, 0])


STATS = 'stats' in os.environ.get('NK_LIBNAME', '')      # `make stats` build: the tap returns visit counts


def report(res):
    if not STATS:
        return
    ne, nl = res[1], res[2]
    tot = (ne + nl)[: len(ne) // 64 * 64].reshape(-1, 64)
    print('   box families entered per ray: mean %.1f  p50 %d  p99 %d  max %d | leaves: mean %.1f p99 %d max %d' % (
        ne.mean(), np.percentile(ne, 50), np.percentile(ne, 99), ne.max(), nl.mean(), np.percentile(nl, 99), nl.max()))
    print('   per wave of 64 rays: iterations (max over lanes of families + leaves) mean %.1f  max %d' % (tot.max(axis=1).mean(), tot.max()), flush=True)


def disc(m, rmax):
    r = rmax * np.sqrt(rng.random(m)); th = 2 * np.pi * rng.random(m)
    return np.stack([c[0] + r * np.cos(th), c[1] + r * np.sin(th), np.zeros(m)], axis=1)


for it in range(2):
    print('--- interior points, random directions', flush=True)
    x = disc(n, 0.99 * R); x[:, 2] = b[0, 2] + (b[1, 2] - b[0, 2]) * rng.random(n)
    v = rng.normal(size=(n, 3)) * 40.0
    xc, tc, fc = eng.find_boundary(x, v)
    report((xc, tc, fc))
    if not STATS:
        print('   hits on caps %.3f' % np.mean(fc < 2), flush=True)
    print('--- from the lower cap, upwards', flush=True)
    x = disc(n, 0.99 * R); x[:, 2] = b[0, 2]
    v = rng.normal(size=(n, 3)) * 40.0; v[:, 2] = np.abs(v[:, 2])
    report(eng.find_boundary(x, v))
    print('--- from the side wall (previous hit points), inwards', flush=True)
    th = 2 * np.pi * rng.random(n)
    ra = R * np.cos(np.pi / ns) * (1.0 - 1e-9)              # just inside the polygon's flats
    x = np.stack([c[0] + ra * np.cos(th), c[1] + ra * np.sin(th), b[0, 2] + (b[1, 2] - b[0, 2]) * rng.random(n)], axis=1)
    v = rng.normal(size=(n, 3)) * 40.0
    rad = x[:, :2] - c[:2]
    flip = np.sum(rad * v[:, :2], axis=1) > 0
    v[flip, :2] *= -1.0
    report(eng.find_boundary(x, v))
    print('--- interior points, towards a cap (|vz| large)', flush=True)
    x = disc(n, 0.99 * R); x[:, 2] = b[0, 2] + (b[1, 2] - b[0, 2]) * rng.random(n)
    v = rng.normal(size=(n, 3)) * 5.0; v[:, 2] = 60.0
    report(eng.find_boundary(x, v))
