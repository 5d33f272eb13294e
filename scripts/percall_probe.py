import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from test_gpu_fullsize import build
pop, geo, ph = build('c2', 10000000)
eng = pop.engine
eng.step(400)
for n in (1,):
    t0 = time.perf_counter()
    for _ in range(500): eng.step(1)
    dt = time.perf_counter() - t0
    print('eng.step(1) x500: %.4f ms per call' % (1e3 * dt / 500))
t0 = time.perf_counter(); eng.step(500); dt = time.perf_counter() - t0
print('eng.step(500): %.4f ms per step' % (1e3 * dt / 500))
import ctypes as C
lib = eng._lib if hasattr(eng, '_lib') else None
t0 = time.perf_counter()
for _ in range(500): pop.run_timestep(geo, ph)
dt = time.perf_counter() - t0
print('pop.run_timestep x500: %.4f ms per call' % (1e3 * dt / 500))
