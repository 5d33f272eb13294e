"""The N > 1 path on CPU: two processes (torch.distributed, gloo) each advance their shard of the ensemble with the
oracle, summing the per-step tally vector with an all-reduce -- the scheme the HIP engine runs over RCCL.  The union
of the two shards must reproduce the single-process run particle for particle."""
import os
import sys

import numpy as np
import pytest

from util import case_tables, random_population, make_oracle_sim

HERE = os.path.dirname(os.path.abspath(__file__))
NSTEPS = 12


def _worker(rank, world, port, out_dir, gen=0):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from nanokappa_amd.sharding import shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ct = case_tables('ttrrp')
    pos, mode, occ, counter = random_population(ct, 12000, seed=3)
    lo, hi = shard_range(pos.shape[0], rank, world)
    sim = make_oracle_sim(ct, pos[lo:hi], mode[lo:hi], occ[lo:hi], counter, seed=17, cap=30000, gen=gen)
    sim.P.pid[:hi - lo] = np.arange(lo, hi, dtype=np.uint64)
    sim.rank, sim.nranks = rank, world

    def allreduce(vec):
        t = torch.from_numpy(vec)
        dist.all_reduce(t)

    T_hist = []
    for _ in range(NSTEPS):
        sim.run_timestep_sharded(allreduce)
        T_hist.append(sim.T_sv.copy())
    n = sim.P.N
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), pid=sim.P.pid[:n], pos=sim.P.pos[:n], occ=sim.P.occ[:n],
             mode=sim.P.mode[:n], T=np.array(T_hist), N_sv=sim.N_sv)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,gen', [(2, 0), (2, 2), (4, 0), (4, 2)])
def test_shard_union_equals_single_run(tmp_path, world, gen):
    """Two and four ranks.  gen 0: 'constant' reservoirs (every rank advances all counters, keeps its share); gen 2:
    'one_to_one' (the all-reduced N_leaving of a step drives the next step's emission on every rank)."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + 7 * gen + 13 * world
    mp.spawn(_worker, args=(world, port, str(tmp_path), gen), nprocs=world, join=True)
    rs = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    # single-process reference
    ct = case_tables('ttrrp')
    pos, mode, occ, counter = random_population(ct, 12000, seed=3)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=17, cap=30000, gen=gen)
    T_hist = []
    for _ in range(NSTEPS):
        sim.run_timestep()
        T_hist.append(sim.T_sv.copy())
    n = sim.P.N
    # every rank saw the same (global) temperatures, equal to the single run up to summation order
    for r in rs[1:]:
        assert np.array_equal(rs[0]['T'], r['T'])
    assert np.allclose(rs[0]['T'], np.array(T_hist), rtol=0, atol=1e-11)
    assert np.array_equal(rs[0]['N_sv'], sim.N_sv)
    pid = np.concatenate([r['pid'] for r in rs])
    assert pid.shape[0] == n and np.unique(pid).shape[0] == n          # disjoint shards, nothing lost
    o = np.argsort(pid)
    o1 = np.argsort(sim.P.pid[:n])
    assert np.array_equal(pid[o], sim.P.pid[:n][o1])
    assert np.array_equal(np.concatenate([r['mode'] for r in rs])[o], sim.P.mode[:n][o1])
    assert np.allclose(np.concatenate([r['pos'] for r in rs])[o], sim.P.pos[:n][o1], rtol=0, atol=1e-10)
    assert np.allclose(np.concatenate([r['occ'] for r in rs])[o], sim.P.occ[:n][o1], rtol=1e-12, atol=0)
    # load balance of the emission split
    sizes = [r['pid'].shape[0] for r in rs]
    assert max(sizes) - min(sizes) < 0.05 * n


def _halt_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from nanokappa_amd.sharding import shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ct = case_tables('ttrrp')
    pos, mode, occ, counter = random_population(ct, 8000, seed=5)
    lo, hi = shard_range(pos.shape[0], rank, world)
    sim = make_oracle_sim(ct, pos[lo:hi], mode[lo:hi], occ[lo:hi], counter, seed=17, cap=30000)
    sim.P.pid[:hi - lo] = np.arange(lo, hi, dtype=np.uint64)
    sim.rank, sim.nranks = rank, world

    def allreduce(vec):
        dist.all_reduce(torch.from_numpy(vec))

    done, seen = 0, None
    for s in range(NSTEPS):
        # rank 1's store "cannot take its migrants" during step 4, rank 0 asks for head room during step 7: whichever comes
        # first stops BOTH ranks after that very step (the engine: halt[0] raised by the update on every rank)
        req = (1 if (rank == 0 and s == 7) else 0, 1 if (rank == 1 and s == 4) else 0)
        halts = sim.run_timestep_sharded(allreduce, halt_requests=req)
        done += 1
        if halts[0] > 0 or halts[1] > 0:
            seen = halts
            break
    n = sim.P.N
    np.savez(os.path.join(out_dir, 'halt%d.npz' % rank), done=done, seen=np.array(seen), pid=sim.P.pid[:n], pos=sim.P.pos[:n])
    dist.barrier()
    dist.destroy_process_group()


def test_halt_requests_ride_on_the_tally_vector(tmp_path):
    """The two halt requests of the engine's step (k_reduce -> all-reduce -> update, csrc/nk_kernels.h) as entries of the
    summed vector, with rough walls ('ttrrp') on two ranks: a request raised by ONE rank during a step is seen by BOTH after
    that step's all-reduce, so they stop at the same step with a consistent ensemble -- which the host then grows."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_halt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'halt0.npz'), np.load(tmp_path / 'halt1.npz')
    assert int(r0['done']) == int(r1['done']) == 5                       # rank 1's request at step index 4 stops both after it
    assert np.array_equal(r0['seen'], [0.0, 1.0]) and np.array_equal(r1['seen'], [0.0, 1.0])
    ct = case_tables('ttrrp')
    pos, mode, occ, counter = random_population(ct, 8000, seed=5)
    sim = make_oracle_sim(ct, pos, mode, occ, counter, seed=17, cap=30000)
    for _ in range(5):
        sim.run_timestep()
    n = sim.P.N
    pid = np.concatenate((r0['pid'], r1['pid']))
    o, o1 = np.argsort(pid), np.argsort(sim.P.pid[:n])
    assert np.array_equal(pid[o], sim.P.pid[:n][o1])
    assert np.allclose(np.concatenate((r0['pos'], r1['pos']))[o], sim.P.pos[:n][o1], rtol=1e-10, atol=1e-8)


def test_sharding_rules():
    from nanokappa_amd.sharding import shard_range, emission_owner, emission_pid
    n = 1000003
    spans = [shard_range(n, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
    owners = [emission_owner(rm, lv, 5, 8) for rm in range(64) for lv in range(1, 5)]
    assert sorted(set(owners)) == list(range(8))
    assert emission_pid(3, 2, 0) == (1 << 40) | (3 << 12) | 2
