"""Does the sweep's time depend on WHERE the particle store sits in memory?  One process, config 2: after every block of steps
the store is grown a little (nk_reserve -> a new allocation, the particles copied on the device), and the mean k_sweep time
of the next block is printed.  Same code, same ensemble: what changes is the placement of the store.
    gpurun -- 'python scripts/placement_probe.py'"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
argv, species, desc = bench.config_argv(cfg, 10000000, 200.0)
args = initialise_parser().parse_args(argv + ['--seed', '2025'])
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(31, species, temperatures=np.arange(200.0, 401.0, 10.0)))
pop = bench.quiet(Population, args, geo, ph)
eng = pop.engine
eng.step(50)
cap = eng.timing()['slots']
nalloc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fine = len(sys.argv) > 3 and sys.argv[3] == 'fine'
pads = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else None     # KB: the store shifted inside its allocation
nseg = None
for k in range(nalloc):
    t = []
    for _ in range(2 if fine else 3):
        eng.step(20)
        t.append(eng.timing()['step_kernel_ms'])
    slots = eng.timing()['slots']
    if os.environ.get('NK_PROBE_COPY'):
        eng.calibrate_stream(1)                 # prints the copy probe's time of this placement on stderr
    print('allocation %d: slots %d  (%s)  k_sweep %s ms' % (k, slots, ' / '.join('%d blocks per segment at %d segments' % (slots // 64 // n, n) for n in (2048, 3072, 4096) if slots % (64 * n) == 0), ' '.join('%.4f' % x for x in t)), flush=True)
    cap = slots + 3072 * 64 if fine else int(cap * 1.02) + 4096 * 64      # fine: one more block per segment
    if pads:
        os.environ['NK_STORE_PAD_KB'] = str(pads[(k + 1) % len(pads)])
        print('   next pad %s KB' % os.environ['NK_STORE_PAD_KB'])
    eng.reserve(cap)
