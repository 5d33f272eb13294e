"""Specular-pair search of the set-up tables at 31^3 q-points: GPU (nk_specular_pairs) vs the NumPy builder (developer probe)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
import numpy as np
import bench
import ref_harness_args as A
from nanokappa_amd import synthetic, setup_tables as ST
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
args = initialise_parser().parse_args(A.argv_for('ttrrp', 20000))
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(n, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
eng = Engine(0, 1)
t0 = time.time(); cd, td = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets, engine=eng); tg = time.time() - t0
print('GPU  : %.3f s for %d normal(s), %d pairs' % (tg, 1, cd.shape[0]), flush=True)
t0 = time.time(); cd2, td2 = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets, engine=eng); tg = time.time() - t0
print('GPU  : %.3f s (second call)' % tg, flush=True)
if '--host' in sys.argv:
    t0 = time.time(); ch, th = ST.specular_correspondences_velocity(geo, ph, geo.rough_facets); tc = time.time() - t0
    print('NumPy: %.3f s; identical: %s' % (tc, np.array_equal(ch, cd) and np.array_equal(th, td)), flush=True)
