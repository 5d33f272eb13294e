"""Long run of a BASELINE configuration with the particle balance checked over every batch of steps (and the state at the
end): python scripts/soak_balance.py [config] [particles] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from test_gpu_fullsize import build

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
pop, geo, ph = build(cfg, total)
eng = pop.engine
n_prev = int(pop.N_p)
done = 0
while done < steps:
    k = min(1000, steps - done)
    t = eng.step(k)
    N = t['N_sv'].sum(axis=1)
    prev = np.concatenate(([n_prev], N[:-1]))
    assert np.array_equal(N - prev, t['N_emitted'] - t['N_leaving'].sum(axis=1)), 'balance broken in steps %d..%d' % (done, done + k)
    assert np.all(np.isfinite(t['T_sv'])) and t['T_sv'].min() > 280 and t['T_sv'].max() < 320
    n_prev = int(N[-1])
    done += k
    print('step %d: %d particles, T %.3f .. %.3f' % (done, n_prev, t['T_sv'][-1].min(), t['T_sv'][-1].max()), flush=True)
p = eng.download()
assert p['positions'].shape[0] == n_prev and np.all(np.isfinite(p['positions'])) and np.all(np.isfinite(p['occupation'])) and p['occupation'].min() >= 0
print('ok')
