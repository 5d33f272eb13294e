import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
import numpy as np, bench
from nanokappa_amd import synthetic
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population
args, geo = bench.wire_geometry(5000000)
args.seed, args.device = [2025], [0]
ph = Phonon(args, 0, material=synthetic.make_material(31, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pop = bench.quiet(Population, args, geo, ph, None, None)
n_prev = int(pop.N_p)
for b in range(10):
    t = pop.engine.step(200)
    N = t['N_sv'].sum(axis=1)
    assert np.array_equal(N - np.concatenate(([n_prev], N[:-1])), t['N_emitted'] - t['N_leaving'].sum(axis=1)), 'balance'
    n_prev = int(N[-1])
    print(b, n_prev, t['T_sv'][-1].min(), t['T_sv'][-1].max(), flush=True)
p = pop.engine.download()
assert p['positions'].shape[0] == n_prev and np.all(np.isfinite(p['positions']))
r = np.hypot(p['positions'][:, 0] - geo.mesh.bounds[:, 0].mean(), p['positions'][:, 1] - geo.mesh.bounds[:, 1].mean())
print('max radius', r.max(), 'ok')
