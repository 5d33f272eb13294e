"""The multi-GPU launch path in one rank (developer probe): torch imported FIRST (as bench.py does under torchrun, for the
gloo rendezvous), so the process holds torch's bundled HIP runtime and RCCL when libnanokappa_hip.so is loaded; then a
1-rank RCCL communicator (NK_FORCE_COMM) and a few steps, against the same run without a communicator."""
import os, sys
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
dist.init_process_group('gloo', rank=0, world_size=1)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from nanokappa_amd import synthetic
from nanokappa_amd.argument_parser import initialise_parser
from nanokappa_amd.engine import comm_unique_id
from nanokappa_amd.geometry import Geometry
from nanokappa_amd.phonon import Phonon
from nanokappa_amd.population import Population


def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if ('hip' in l or 'rccl' in l or 'hsa' in l) and '.so' in l))


args = initialise_parser().parse_args(bench.workload_argv(1000000, 200.0) + ['--seed', '2025', '--device', '0'])
args.results_folder = ''
geo = bench.quiet(Geometry, args)
ph = Phonon(args, 0, material=synthetic.make_material(9, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
pop0 = bench.quiet(Population, args, geo, ph)
t0 = pop0.engine.step(20)
pop0.engine.close()
os.environ['NK_FORCE_COMM'] = '1'
buf = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).clone()
dist.broadcast(buf, 0)
pop1 = bench.quiet(Population, args, geo, ph, None, (bytes(buf.numpy().tobytes()), 0, 1))
t1 = pop1.engine.step(20)
print('loaded:', *maps(), sep='\n  ')
print('N_sv equal:', np.array_equal(t0['N_sv'], t1['N_sv']), ' max |dT|:', np.abs(t0['T_sv'] - t1['T_sv']).max())
print('timing with communicator:', {k: round(v, 4) for k, v in pop1.engine.timing().items() if k.endswith('_ms')})
dist.barrier(); dist.destroy_process_group()
