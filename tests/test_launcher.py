"""bench.py's launch path and the rank rendezvous (host plumbing of SURVEY 8e), on CPU: ranks meet, exchange the
unique id, barrier and max; `--gpus N` is honoured or refused, never silently reduced to one rank."""
import multiprocessing as mp
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _rank(r, n, key, q):
    sys.path.insert(0, ROOT)
    from nanokappa_amd.sharding import NodeRendezvous
    rv = NodeRendezvous(r, n, key, timeout=60)
    uid = rv.broadcast(bytes(range(128)) if r == 0 else b'')
    rv.barrier()
    m = rv.max(10.0 + r)
    parts = rv.allgather(b'x' * r)
    rv.barrier()
    rv.close()
    q.put((r, uid, m, [len(p) for p in parts]))


def test_rendezvous_three_ranks():
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    key = 'pytest_%d' % os.getpid()
    procs = [ctx.Process(target=_rank, args=(r, 3, key, q)) for r in (2, 0, 1)]     # rank 0 need not come first
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r, uid, m, lens in res:
        assert uid == bytes(range(128)) and m == 12.0 and lens == [0, 1, 2]


def test_gpus_must_match_world_size():
    """Under a launcher (RANK set) a --gpus that disagrees with WORLD_SIZE is refused, world == 1 included."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and 'does not match WORLD_SIZE' in p.stderr


def test_self_launch_starts_n_ranks_and_reports_failure():
    """Without a launcher `--gpus 2` starts two rank processes itself; here there is no GPU, so both fail in nk_create
    (no CPU fallback) and the parent must exit non-zero naming them -- not print a one-rank line."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--mesh-n', '5', '--particles', '2000', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                       timeout=300)
    from nanokappa_amd.engine import device_count
    if device_count() == 0:
        assert p.returncode != 0 and ('rank 0 rc' in p.stderr or 'rank 1 rc' in p.stderr)
        assert '"n_gpus"' not in p.stdout


def test_self_launch_stops_the_survivors_when_a_rank_dies():
    """A rank that dies while its peer waits (in a collective, on real hardware) must not leave the parent waiting: the
    parent stops the survivor, exits non-zero and names the dead rank.  NK_BENCH_TEST_HOOK makes rank 1 exit with code 3 at
    once and rank 0 sleep for ten minutes."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env['NK_BENCH_TEST_HOOK'] = 'rank1_dies_rank0_hangs'
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and 'rank 1 rc 3' in p.stderr and 'stopped' in p.stderr
    assert time.time() - t0 < 60


def test_self_launch_deadline():
    """Every rank hanging: the overall deadline ends the run."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env['NK_BENCH_TEST_HOOK'] = 'all_hang'
    env['NK_BENCH_DEADLINE'] = '3'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and 'deadline' in p.stderr
