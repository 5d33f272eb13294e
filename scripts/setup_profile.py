"""Where config 4's set-up time goes (host side): cProfile of geometry + material + Population construction.
   gpurun -- 'python scripts/setup_profile.py [particles]'"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench


def build(n):
    t0 = time.time()
    args, geo = bench.wire_geometry(n)
    args.seed, args.device = [2025], [0]
    t1 = time.time()
    from nanokappa_amd import synthetic
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population
    ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
    t2 = time.time()
    pop = bench.quiet(Population, args, geo, ph, None, None)
    t3 = time.time()
    pop.engine.step(2)
    t4 = time.time()
    sys.stderr.write('geometry %.1f s | material %.1f s | population %.1f s | first steps %.1f s\n' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))


if __name__ == '__main__':
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50000000
    pr = cProfile.Profile()
    pr.enable()
    build(n)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
    sys.stderr.write(s.getvalue())
