"""N > 1 on real hardware: two rank processes, one GPU each, a real RCCL communicator -- the per-step tally
vector is all-reduced in the engine's stream.  The union of the two shards must equal the single-rank run particle for
particle (ids, modes, positions), and both ranks must report the same, whole-ensemble tallies.  Skipped where fewer than
two GPUs are visible (the one-GPU development box); the same scheme runs on CPU in test_sharding_gloo.py and, without a
communicator, on one GPU in test_gpu_parity.py::test_two_rank_sharding_on_one_gpu."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from util import allclose

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NSTEPS = 12


def box_case(box):
    """'ttp' tables of the goldens; box = 1000: BASELINE config 5's box, built with this package's own Geometry."""
    from util import case_tables, case_from_args
    if box == 200:
        return case_tables('ttp')
    argv = ['--geometry', 'box', '--dimensions', '1000', '1000', '1000', '--subvolumes', 'slice', '20', '0',
            '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '--bound_cond', 'T', 'T', 'P',
            '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
            '--bound_values', '302', '298', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic', '--temp_interp', 'linear',
            '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', '40000']
    return case_from_args(argv, 'Si')


def population(ct, n, seed):
    from util import random_population, population_in_mesh
    return population_in_mesh(ct, n, seed) if 'geo' in ct else random_population(ct, n, seed)


def _rank(rank, world, key, box, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, '..'))
    from nanokappa_amd.sharding import NodeRendezvous, shard_range
    from nanokappa_amd.engine import comm_unique_id
    from util import make_engine
    rdv = NodeRendezvous(rank, world, key, timeout=120)
    ct = box_case(box)
    n = 40000
    pos, mode, occ, counter = population(ct, n, seed=3)
    lo, hi = shard_range(n, rank, world)
    uid = rdv.broadcast(comm_unique_id() if rank == 0 else b'')
    eng = make_engine(ct, pos[lo:hi], mode[lo:hi], occ[lo:hi], counter, seed=5, device=rank, pid_offset=lo, comm=(uid, rank, world))
    t = eng.step(NSTEPS)
    p = eng.download()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), pid=p['pid'], mode=p['mode'], pos=p['positions'], occ=p['occupation'],
             T=t['T_sv'], N_sv=t['N_sv'], N_emitted=t['N_emitted'])
    rdv.barrier()
    eng.close()
    rdv.close()


@pytest.mark.parametrize('box', [200, 1000])
def test_two_ranks_rccl_union_equals_single_rank(box, tmp_path):
    from nanokappa_amd.engine import device_count
    if device_count() < 2:
        pytest.skip('needs two GPUs')
    ctx = mp.get_context('spawn')                      # fresh processes: nothing has touched a GPU before the ranks start
    key = 'pytest_rccl_%d_%d' % (os.getpid(), box)
    procs = [ctx.Process(target=_rank, args=(r, 2, key, box, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    from util import make_engine
    ct = box_case(box)
    pos, mode, occ, counter = population(ct, 40000, seed=3)
    ref = make_engine(ct, pos, mode, occ, counter, seed=5)
    t = ref.step(NSTEPS)
    p = ref.download()
    # both ranks saw the same, whole-ensemble tallies: the all-reduce really summed over two ranks
    assert np.array_equal(r0['T'], r1['T']) and np.array_equal(r0['N_sv'], r1['N_sv'])
    assert np.array_equal(r0['N_sv'], t['N_sv'])
    assert allclose(r0['T'], t['T_sv'], rtol=0, atol=2e-11)      # (tallies summed in another order: the all-reduce)
    assert np.array_equal(r0['N_emitted'], t['N_emitted'])
    pid = np.concatenate((r0['pid'], r1['pid']))
    assert pid.shape[0] == p['pid'].shape[0] and np.unique(pid).shape[0] == pid.shape[0]
    o1, o2 = np.argsort(p['pid']), np.argsort(pid)
    assert np.array_equal(p['pid'][o1], pid[o2])
    assert np.array_equal(p['mode'][o1], np.concatenate((r0['mode'], r1['mode']))[o2])
    assert allclose(p['positions'][o1], np.concatenate((r0['pos'], r1['pos']))[o2], rtol=0, atol=1e-11)
    assert allclose(p['occupation'][o1], np.concatenate((r0['occ'], r1['occ']))[o2], rtol=1e-13, atol=0)
    assert abs(r0['pid'].shape[0] - r1['pid'].shape[0]) < 0.05 * pid.shape[0]
