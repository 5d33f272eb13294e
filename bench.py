#!/usr/bin/env python
"""bench.py -- phonon-steps/s of the Population timestep loop on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...                     # starts the N rank processes itself (before anything touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W       # or under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): synthetic Si (31^3 q-points x 6 branches), box
200 x 200 x 200 angstrom, 20 slices along x, BCs T T P (302 K / 298 K reservoirs on +-x, periodic sides), dt = 1 ps,
1e7 particles PER GPU (weak scaling: the ensemble grows with N, the per-step tally is all-reduced over RCCL).
`--config` selects the other BASELINE workloads (c1b rough box, c3 Ge film, c4 STL wire, c5 the 1000 A box's share of
one GPU); the default line is c2.
A "step" is one Population.run_timestep: relax -> drift -> reservoir emission -> boundary events -> tally -> T update
(kernels k_sweep and k_reduce with the fused update; with RCCL the all-reduce sits between k_reduce and k_update).
Particles are resident in HBM before the timed region.  The timed region (exactly --steps steps between barriers, the
stream drained on both sides) is repeated --repeats times; the line reports the MEDIAN repeat and the spread.
One JSON line is printed by rank 0.  At N = 1 the line also carries the CPU legs, measured on this host with the oracle
(a C port of the reference loop): `cpu_baseline` (one thread, 1e6 particles, after the GPU part) and
`cpu_baseline_all_cores` (one worker process per core, forked before anything touches the GPU), plus
`cpu_reference_numpy`, the reference's own NumPy loop as measured in the build container (a labelled constant: the
reference cannot travel to the GPU box).  --no-cpu-baseline skips the measured legs.

No PyTorch anywhere: the ranks meet on a UNIX socket (nanokappa_amd.sharding.NodeRendezvous) to hand round the RCCL
unique id, for the barriers and for the max of the elapsed times; the compute path is libnanokappa_hip.so + RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_PHONON_STEP = 68.0      # x,y,z read+write (48) + mode read (4) + occupation read+write (16)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s


def workload_argv(particles, box=200.0):
    b = str(box)
    return ['--geometry', 'box', '--dimensions', b, b, b, '--subvolumes', 'slice', '20', '0',
            '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5', '--bound_cond', 'T', 'T', 'P',
            '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
            '--bound_values', '302', '298', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
            '--reference_temp', 'local', '--temp_dist', 'cold', '--temp_interp', 'linear',
            '--part_dist', 'random_subvol', '--timestep', '1', '--n_mean', '10', '--conv_crit', '0', '10',
            '--output', 'screen', '--energy_normal', 'mean', '--particles', 'total', str(int(particles))]


def quiet(fn, *a, **k):
    """The Population constructor prints progress like the reference; keep stdout for the JSON line."""
    old = sys.stdout
    sys.stdout = open(os.devnull, 'w')
    try:
        return fn(*a, **k)
    finally:
        sys.stdout.close()
        sys.stdout = old


def _oracle_run(geo, ph, n, seed, seconds_target, max_steps, start=None, rough=None):
    """n particles of the bench workload through the CPU oracle (oracle/nk_oracle.c, a scalar C port of the reference
    loop): same geometry / BCs / material.  `rough` = (facets, specularity, true_specular, spec_map, roulette) of the
    rough facets, as the Population built them.  Returns (phonon-steps, seconds, steps)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import nk_oracle as O
    from nanokappa_amd import setup_tables as ST
    density = n / geo.volume
    mat = O.make_material(ph.tables())
    g = geo.tables()
    g['bound_cond'] = np.array([ord(c) for c in geo.bound_cond], dtype=np.int8)
    mesh = O.make_mesh(g)
    sv = O.make_subvols(geo.subvol_center, geo.subvol_volume, 0, geo.slice_axis, 1)
    Q, J = ph.omega.shape
    ep = ST.enter_probability(geo, ph, geo.res_facets, density, 1.0).reshape(-1, Q * J)
    rng = np.random.default_rng(seed)
    res = O.make_reservoirs(geo.res_facets, geo.res_values, ep, rng.random(ep.shape))
    if rough is None:
        z = np.zeros(0)
        rgh = O.make_rough(np.zeros(0, dtype=np.int32), z, np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.int32), z)
    else:
        rgh = O.make_rough(*rough)
    par = O.make_params(dt=1.0, particle_density=density, seed=seed)
    pos = geo.mesh.sample_volume(n, rng)
    active = np.nonzero(~ph.inactive_modes_mask.ravel())[0]
    mode = active[np.arange(n) % active.shape[0]].astype(np.int32)
    occ = ph.calculate_occupation(298.0, ph.omega.ravel()[mode])
    store = O.ParticleStore(int(1.3 * n) + 4096)
    store.load(pos, mode, occ)
    sim = O.OracleSim(mat, mesh, sv, res, rgh, par, store, np.full(geo.n_of_subvols, 298.0), box='auto')   # the engine's event rule
    sim.init_boundaries()
    sim.run_timestep()                       # warm-up step
    if start is not None:
        start.wait()                         # all workers set up: measure while all of them run
    t0 = time.time()
    steps, psteps = 0, 0
    while steps < 1 or (time.time() - t0 < seconds_target and steps < max_steps):
        sim.run_timestep()
        psteps += int(sim.N_sv.sum())
        steps += 1
    return psteps, time.time() - t0, steps


def cpu_baseline(geo, ph, mesh_n, desc, n=1000000, seconds_target=15.0, rough=None):
    """The CPU oracle timed on this host on a bounded sample of the same workload (default 1e6 particles), one thread."""
    psteps, dt, steps = _oracle_run(geo, ph, n, 1, seconds_target, 200, rough=rough)
    return dict(value=psteps / dt, unit='phonon-steps/s', cores=1, kind='port',
                sample='%d particles x %d steps of the same workload (%s, %d^3 x 6 modes), oracle/nk_oracle.c, 1 thread'
                       % (n, steps, desc, mesh_n))


def _all_cores_worker(geo, ph, n, seed, seconds_target, start, out):
    try:
        out.put(_oracle_run(geo, ph, n, seed, seconds_target, 100000, start))
    except Exception as e:                   # the parent reports the failure; never hang the barrier
        try:
            start.abort()
        except Exception:
            pass
        out.put(('error', repr(e), 0))


def cpu_baseline_all_cores(geo, ph, mesh_n, desc, seconds_target=8.0):
    """The same oracle on every core this process may use: one worker process per core (forked BEFORE anything touches
    the GPU), each with its own share of 1e6 particles -- the particle shards of the multi-rank scheme, without the
    exchange of tallies.  Sum of the workers' rates while all of them run.  None if it cannot be measured."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # a one-GPU box's share of the host
    if cores < 2:
        return None
    ctx = mp.get_context('fork')
    n = 1000000 // cores
    start, out = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_all_cores_worker, args=(geo, ph, n, 100 + i, seconds_target, start, out), daemon=True)
             for i in range(cores)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(out.get(timeout=180))
    except Exception:
        res = []
    for p in procs:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
    if len(res) != cores or any(r[0] == 'error' for r in res):
        return None
    rate = sum(r[0] / r[1] for r in res)
    return dict(value=rate, unit='phonon-steps/s', cores=cores, kind='port',
                sample='%d worker processes x %d particles x %d-%d steps of the same workload (%s), oracle/nk_oracle.c, one thread each, '
                       'tallies not exchanged' % (cores, n, min(r[2] for r in res), max(r[2] for r in res), desc))


CONFIGS = ('c2', 'c1b', 'c3', 'c4', 'c5')

# The reference's own NumPy loop cannot run on the GPU box (it does not travel); measured in the build container
# (SURVEY.md section 6 probes; tests/golden/stats_*.npz 'rate'): /opt/conda/bin/python3.9, numpy 1.26.4, plots stubbed.
CPU_REFERENCE_NUMPY = dict(value=3.65e5, unit='phonon-steps/s', cores=1, kind='reference',
                           host='build container (Xeon 2.1 GHz, 8 vCPU, 1 core effective), NOT this host',
                           sample='reference Population.run_timestep, synthetic 9^3 x 6 modes, box 200 A, T T P, 1e6 particles '
                                  'x 10 steps (4.1e5 at 1e5 particles x 20 steps)')


def config_argv(cfg, total, box):
    """Argument list (Nano-kappa flags) of a BASELINE workload; returns (argv, species, description)."""
    argv = workload_argv(total, box)
    if cfg == 'c2':
        return argv, 'Si', 'box %gx%gx%g A, slice 20 subvols, BCs T T P' % (box, box, box)
    if cfg == 'c5':
        argv = workload_argv(total, 1000.0)
        return argv, 'Si', 'box 1000x1000x1000 A, slice 20 subvols, BCs T T P'
    if cfg == 'c3':
        i = argv.index('--dimensions')
        argv[i + 1:i + 4] = ['2000', '500', '500']
        return argv, 'Ge', 'film 2000 A thick, 500x500 A periodic cell, slice 20 subvols, BCs T T P'
    if cfg == 'c1b':
        i, j = argv.index('--bound_pos'), argv.index('--bound_values')
        argv[i:j + 3] = ['--bound_pos', 'relative', '-0.1', '0.5', '0.5', '1.1', '0.5', '0.5', '0.5', '0.5', '-0.1', '0.5', '0.5', '1.1',
                         '--bound_cond', 'T', 'T', 'R', 'R', 'P', '--connect_pos', 'relative', '0.5', '-0.1', '0.5', '0.5', '1.1', '0.5',
                         '--bound_values', '302', '298', '5', '5']
        return argv, 'Si', 'box %gx%gx%g A, slice 20 subvols, BCs T T R R P (eta 5 A)' % (box, box, box)
    raise SystemExit('unknown --config %r' % cfg)


def wire_geometry(total):
    """BASELINE config 4: cylinder primitive (1250 sides = 5000 triangles, L 2000 A, R 200 A) written as ASCII STL and
    imported again; caps T 302 / 298 K, rough side wall (eta 5 A), 20 slices along the axis."""
    import tempfile
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.geometry import Geometry
    tail = ['--subvolumes', 'slice', '20', '2', '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1',
            '--bound_cond', 'T', 'T', 'R', '--bound_values', '302', '298', '5', '--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
            '--temp_interp', 'linear', '--timestep', '1', '--energy_normal', 'mean', '--particles', 'total', str(int(total)),
            '--seed', '2025']
    prim = initialise_parser().parse_args(['--geometry', 'cylinder', '--dimensions', '2000', '200', '1250'] + tail)
    prim.results_folder = ''
    g0 = quiet(Geometry, prim)
    tmp = tempfile.mkdtemp()
    g0.mesh.export_stl('wire', tmp)
    args = initialise_parser().parse_args(['--geometry', os.path.join(tmp, 'wire.stl'), '--dimensions', '1', '1', '1'] + tail)
    args.results_folder = ''
    return args, quiet(Geometry, args)


def launch_ranks(n, deadline_s=None):
    """`bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) BEFORE anything touches a GPU,
    relay rank 0's JSON line.  The children are watched together: the first one that exits non-zero, or the overall
    deadline (NK_BENCH_DEADLINE seconds, default 1500), ends the others -- a rank that died after ncclCommInitRank would
    otherwise leave its peers in the all-reduce for ever -- and the parent exits non-zero naming the rank."""
    import socket
    import subprocess
    import tempfile
    if deadline_s is None:
        deadline_s = float(os.environ.get('NK_BENCH_DEADLINE', '1500'))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NK_RDV_KEY='%d_%d' % (os.getpid(), port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        out = tempfile.TemporaryFile() if r == 0 else None     # a file, not a pipe: nobody has to drain it while we poll
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(out if r == 0 else subprocess.DEVNULL)))
    t0, failed = time.time(), None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = 'rank(s) failed: %s' % ', '.join('rank %d rc %d' % b for b in bad)
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() - t0 > deadline_s:
            failed = 'deadline of %.0f s passed with rank(s) %s still running' % (deadline_s, [r for r, rc in enumerate(rcs) if rc is None])
            break
        time.sleep(0.2)
    if failed:
        for p in procs:                      # exactly the children started above, nothing by pattern
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t1 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs[0].seek(0)
    sys.stdout.write(outs[0].read().decode())
    sys.stdout.flush()
    if failed:
        raise SystemExit('bench.py: ' + failed + ' (the other ranks were stopped)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; the median is reported')
    ap.add_argument('--ramp', type=int, default=400,
                    help='untimed steps right after set-up, before the --warmup steps: the device comes out of ~10 s of host-side set-up '
                         'in a low power state and its first few hundred steps (0.1 s) run 3-6 %% slower than the sustained rate')
    ap.add_argument('--config', default='c2', choices=CONFIGS)
    ap.add_argument('--particles', type=float, default=None, help='particles per GPU (default: the config\'s own)')
    ap.add_argument('--mesh-n', type=int, default=31, help='q-mesh of the synthetic material (31 -> 29791 q-points)')
    ap.add_argument('--box', type=float, default=200.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sustained', type=int, default=None,
                    help='steps of the `sustained` leg (BASELINE config 2 is quoted on 10000 iterations); default 10000 for '
                         'the default line (c2, one GPU), 0 otherwise')
    ap.add_argument('--per-call', type=int, default=None,
                    help='Population.run_timestep calls of the `per_call` leg (the reference driver\'s granularity); '
                         'default 500 for the default line, 0 otherwise')
    ap.add_argument('--small', type=int, default=None,
                    help='steps per region of the `small_ensemble` leg (BASELINE config 1 at its own 1e5 particles, launch-per-step path '
                         'and resident kernel); default 500 for the default line, else 0')
    ap.add_argument('--calibrate', action='store_true',
                    help='after the timed region run 3 known-traffic sweeps (k_cal_stream) for PMC calibration')
    a = ap.parse_args()
    if a.gpus < 1 or a.steps < 1 or a.repeats < 1:
        raise SystemExit('--gpus, --steps and --repeats must be positive')

    if 'RANK' not in os.environ and a.gpus > 1:
        return launch_ranks(a.gpus)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world != a.gpus:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (a.gpus, world))
    hook = os.environ.get('NK_BENCH_TEST_HOOK')          # tests/test_launcher.py: what launch_ranks does about dead / hung ranks
    if hook == 'rank1_dies_rank0_hangs' and world > 1:
        if rank == 1:
            raise SystemExit(3)
        time.sleep(600)
    if hook == 'all_hang' and world > 1:
        time.sleep(600)
    from nanokappa_amd.sharding import NodeRendezvous
    rdv = NodeRendezvous(rank, world, os.environ.get('NK_RDV_KEY', os.environ.get('MASTER_PORT', 'solo')))

    from nanokappa_amd import synthetic
    from nanokappa_amd.argument_parser import initialise_parser
    from nanokappa_amd.engine import comm_unique_id
    from nanokappa_amd.geometry import Geometry
    from nanokappa_amd.phonon import Phonon
    from nanokappa_amd.population import Population

    per_gpu = a.particles if a.particles is not None else {'c4': 5e7, 'c5': 1.25e7}.get(a.config, 1e7)
    total = int(per_gpu) * world
    os.environ.setdefault('NK_MESH_DEVICE', str(local_rank))     # the host geometry's inside tests run on this rank's GPU too
    if a.config == 'c4':
        args, geo = wire_geometry(total)
        species, desc = 'Si', 'STL-imported wire, 5000 triangles, L 2000 A, R 200 A, caps T, rough side (eta 5 A), slice 20 subvols'
        args.seed, args.device = [2025], [local_rank]
    else:
        argv, species, desc = config_argv(a.config, total, a.box)
        args = initialise_parser().parse_args(argv + ['--seed', '2025', '--device', str(local_rank)])
        args.results_folder = ''
        geo = quiet(Geometry, args)
    ph = Phonon(args, 0, material=synthetic.make_material(a.mesh_n, species, temperatures=np.arange(200.0, 401.0, 10.0)))

    default_line = world == 1 and a.config == 'c2' and a.particles is None
    n_sustained = a.sustained if a.sustained is not None else (10000 if default_line else 0)
    n_per_call = a.per_call if a.per_call is not None else (500 if default_line else 0)
    n_small = a.small if a.small is not None else (500 if default_line else 0)
    cpu_legs = world == 1 and not a.no_cpu_baseline
    rough_cfg = a.config in ('c1b', 'c4')
    all_cores = None
    if cpu_legs and not rough_cfg:
        # forks: must come before the GPU is touched -- the box geometries stay below the size at which the host mesh
        # code would use the GPU (nk_mesh_crossings), and the engine library is not loaded yet
        assert 'nanokappa_amd.engine' not in sys.modules or sys.modules['nanokappa_amd.engine']._lib is None, \
            'cpu_baseline_all_cores must fork before the HIP library is loaded'
        all_cores = cpu_baseline_all_cores(geo, ph, a.mesh_n, desc)
    comm = None
    if world > 1:
        uid = rdv.broadcast(comm_unique_id() if rank == 0 else b'')
        comm = (uid, rank, world)
    from nanokappa_amd.engine import device_count
    nd = device_count()                     # a launcher may show every rank one device only: index modulo what is visible
    if nd > 0:
        args.device = [local_rank % nd]
    pop = quiet(Population, args, geo, ph, None, comm)
    eng = pop.engine
    # what RCCL itself says about the communicator (not WORLD_SIZE): every rank reports, rank 0 prints
    ci = eng.comm_info()

    def housekeeping(tm0, tm1):
        return {k: tm1[k] - tm0[k] for k in ('regrows', 'halts', 'tau_rebuilds', 'batches')}

    for _ in range(max(a.ramp, 0) // 100):
        eng.step(100)                      # the ramp (see --ramp): the same steps, untimed
    if a.warmup > 0:
        eng.step(a.warmup)                 # nk_step returns after the stream has drained (hipStreamSynchronize)
    # one untimed region with the timed regions' own call pattern (same --steps): first-use costs of that pattern (history
    # buffer, event pool, clocks) land here, not in the first timed region
    eng.step(a.steps)
    runs = []
    for _ in range(a.repeats):
        tm0 = eng.timing()
        rdv.barrier()
        t0 = time.perf_counter()
        t = eng.step(a.steps)              # enqueues the steps, drains the stream, copies the tallies back
        rdv.barrier()
        elapsed = rdv.max(time.perf_counter() - t0)
        tm = eng.timing()
        runs.append(dict(elapsed=elapsed, psteps=float(t['N_sv'].sum()), tm=tm, hk=housekeeping(tm0, tm)))   # N_sv: all ranks (all-reduced)
    if a.calibrate:
        cal = eng.calibrate_stream(3)
        if rank == 0:
            sys.stderr.write('calibration: k_cal_stream reads %d B and writes %d B per launch\n' % cal)
    order = sorted(range(len(runs)), key=lambda i: runs[i]['elapsed'])
    med = runs[order[len(order) // 2]]
    elapsed, psteps, tm = med['elapsed'], med['psteps'], med['tm']
    value = psteps / elapsed
    per_rank = rdv.allgather(json.dumps(dict(rank=rank, comm_rank=ci['comm_rank'], comm_nranks=ci['comm_nranks'],
                                              selftest_sum=ci['selftest_sum'], device=ci['device'], pci_bus_id=ci['pci_bus_id'],
                                              live=tm['live'], sweep_ms=tm['step_kernel_ms'])).encode())

    # ---- the reference driver's own granularity: one Population.run_timestep per step (nanokappa.py:91-98)
    per_call = None
    if n_per_call > 0:
        n0 = pop.current_timestep
        t0 = time.perf_counter()
        ps = 0.0
        for _ in range(n_per_call):
            quiet(pop.run_timestep, geo, ph)
            ps += pop.N_p
        dt_pc = time.perf_counter() - t0
        per_call = dict(calls=n_per_call, seconds=dt_pc, ms_per_call=1e3 * dt_pc / n_per_call, value=ps / dt_pc, unit='phonon-steps/s',
                        what='%d x Population.run_timestep (one nk_step(1) each: launches + stream drain + tally copy + host '
                             'bookkeeping), steps %d..%d' % (n_per_call, n0, pop.current_timestep))
    # ---- BASELINE's own run length in one go (config 2: 10 000 iterations), through Population.run
    sustained = None
    if n_sustained > 0:
        tm0 = eng.timing()
        n0 = pop.current_timestep
        t0 = time.perf_counter()
        quiet(pop.run, n_sustained, geo, ph)
        dt_s = time.perf_counter() - t0
        rows = [r for r in pop.conv_rows if r['step'] > n0 + n_sustained // 2]
        kap = np.array([r['kappa'] for r in rows], dtype=float)
        npart = np.array([r['N_p'] for r in pop.conv_rows if r['step'] > n0], dtype=float)
        sustained = dict(steps=n_sustained, seconds=dt_s, ms_per_step=1e3 * dt_s / n_sustained,
                         value=float(npart.mean()) * n_sustained / dt_s, unit='phonon-steps/s',
                         kappa_mean=float(np.nanmean(kap)), kappa_std=float(np.nanstd(kap)), kappa_samples=int(kap.size),
                         kappa_unit='W/m/K (synthetic material)', N_p_mean=float(npart.mean()),
                         housekeeping=housekeeping(tm0, eng.timing()),
                         what='Population.run(%d) in one go (100-step library calls, every tenth step a convergence row); kappa over '
                              'the rows of the second half' % n_sustained)

    # ---- BASELINE config 1 at its own size (1e5 particles): the launch-per-step path and the resident kernel (NK_RESIDENT=1), same box
    small = None
    if n_small > 0:
        argv_s, _, desc_s = config_argv('c2', 100000, a.box)
        args_s = initialise_parser().parse_args(argv_s + ['--seed', '2025', '--device', str(args.device[0])])
        args_s.results_folder = ''
        geo_s = quiet(Geometry, args_s)
        small = dict(particles=100000, steps=n_small, unit='phonon-steps/s',
                     what='%d^3x6 modes, %s, 1e5 particles (BASELINE config 1); median of 3 regions of %d steps after %d warm-up steps; '
                          'resident_kernel: one launch per library call (k_resident, opt-in NK_RESIDENT=1)' % (a.mesh_n, desc_s, n_small, n_small))
        for leg, env in (('launch_per_step', None), ('resident_kernel', '1')):
            if env is None:
                os.environ.pop('NK_RESIDENT', None)
            else:
                os.environ['NK_RESIDENT'] = env
            pop_s = quiet(Population, args_s, geo_s, ph, None, None)
            es = pop_s.engine
            es.step(n_small)
            reg = []
            for _ in range(3):
                t0 = time.perf_counter()
                ts = es.step(n_small)
                reg.append((time.perf_counter() - t0, float(ts['N_sv'].sum())))
            reg.sort()
            dt_r, ps_r = reg[1]
            small[leg] = dict(ms_per_step=1e3 * dt_r / n_small, value=ps_r / dt_r, emit_fused=int(es.timing().get('emit_fused', 0)))
            es.close()
        os.environ.pop('NK_RESIDENT', None)

    if rank == 0:
        live_rank = tm['live'] / world if world > 1 else tm['live']
        k_ms = float(np.median([r['tm']['step_kernel_ms'] for r in runs]))
        achieved = BYTES_PER_PHONON_STEP * live_rank / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM bytes of one k_sweep launch from the PMC counters: they cannot be collected inside this run (rocprofv3 wraps the
        # process), so the line carries the committed figure of the round's profile of this same command and says so
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')     # written from scripts/profile_round.sh's PMC passes
        if os.path.exists(tf) and world == 1 and a.config == 'c2' and int(per_gpu) == 10000000 and a.mesh_n == 31:
            try:
                tj = json.load(open(tf))
                traffic, traffic_source = tj.get('k_sweep_hbm_bytes_per_launch'), tj.get('source')
            except Exception:
                traffic = None
        ms = [1e3 * r['elapsed'] / a.steps for r in runs]
        ranks = [json.loads(b.decode()) for b in per_rank]
        out = {
            'metric': 'phonon-steps/sec (whole node)', 'value': value, 'unit': 'phonon-steps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'repeats': a.repeats, 'ms_per_step_min': min(ms), 'ms_per_step_max': max(ms),
            'ms_per_step_repeats': ms, 'sweep_ms_repeats': [r['tm']['step_kernel_ms'] for r in runs],
            'housekeeping_repeats': [r['hk'] for r in runs],
            # the same regions on the GPU's own clock (HIP events around the enqueued steps): a repeat that is long on the wall
            # but not here lost its time on the host (launch thread descheduled), not in a kernel
            'stream_ms_per_step_repeats': [r['tm']['total_ms'] / a.steps for r in runs],
            'timing': 'median of %d timed regions of %d steps each (barrier + drained stream on both sides, max over ranks); %d warm-up '
                      'steps and one untimed region of %d steps before them, behind an untimed ramp of %d steps after set-up (--ramp: the '
                      'device leaves the host-side set-up in a low power state)' % (a.repeats, a.steps, a.warmup, a.steps, max(a.ramp, 0) // 100 * 100),
            'ramp_steps': max(a.ramp, 0) // 100 * 100,
            'config': {'workload': '%s-like synthetic %d^3x6 modes, %s, dt 1 ps, %.3g particles per GPU (BASELINE config %s)'
                                   % (species, a.mesh_n, desc, per_gpu, a.config),
                       'particles_total': total, 'live_particles_end': tm['live'], 'parallelism': 'particle-shard x%d' % world},
            # the communicator as RCCL reports it (ncclCommCount / ncclCommUserRank, and the all-reduce of ones at nk_comm_init);
            # one GPU: no communicator (nranks 0)
            'rccl': {'nranks': ranks[0]['comm_nranks'], 'selftest_sum': ranks[0]['selftest_sum'],
                     'all_ranks_agree': all(r['comm_nranks'] == ranks[0]['comm_nranks'] for r in ranks),
                     'distinct_devices': len(set(r['pci_bus_id'] for r in ranks)), 'ranks': ranks},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': 'k_sweep', 'kernel_ms': k_ms, 'emit_count_kernel_ms': tm['emit_kernel_ms'],
                         'reduce_update_ms': tm['events_kernel_ms'], 'stream_ms_per_step': tm['total_ms'] / a.steps,
                         # True: the next step's emission runs inside the tail launch (k_tail), beside the reduce / update --
                         # reduce_update_ms then includes it and emit_count_kernel_ms covers only a call's first step
                         'emission_in_tail_launch': bool(tm.get('emit_fused', 0)),
                         'frac_whole_step': BYTES_PER_PHONON_STEP * value / 1e9 / HBM_PEAK_GBS / max(world, 1),
                         'algorithmic_bytes_per_launch': BYTES_PER_PHONON_STEP * live_rank},
            # where the particle store lies in memory decides between two speeds of the sweep (profiles/r03_notes.txt (9), (17)):
            # the library times an in-place copy over candidate allocations and keeps the fastest
            'store_placement': {'allocations_timed': tm.get('place_tries', 0), 'kept_copy_GBps': tm.get('place_gbps', 0.0),
                                'slowest_copy_GBps': tm.get('place_worst_gbps', 0.0)},
        }
        if world > 1 and not os.environ.get('NK_COMM_DRYRUN') and (ranks[0]['comm_nranks'] != world or not out['rccl']['all_ranks_agree']):
            raise SystemExit('bench.py: RCCL reports %d ranks, the launcher %d' % (ranks[0]['comm_nranks'], world))
        if per_call is not None:
            out['per_call'] = per_call
        if small is not None:
            out['small_ensemble'] = small
        if sustained is not None:
            out['sustained'] = sustained
            # kappa is half of BASELINE's metric: a line whose sustained leg recorded convergence rows but no finite kappa
            # (BENCH_r03: the engine's flux phase and Population's row phase apart) is not a result
            if sustained['kappa_samples'] > 0 and not np.isfinite(sustained['kappa_mean']):
                print(json.dumps(out))
                raise SystemExit('bench.py: %d convergence rows in the sustained leg and no finite kappa' % sustained['kappa_samples'])
        if cpu_legs:
            rough = None
            if rough_cfg:                    # the tables the Population built (on the device), handed to the oracle
                sp, ts, sm, ro = pop.rough_tables()
                rough = (np.asarray(pop.rough_facets, dtype=np.int32), sp, ts.astype(np.uint8), sm.astype(np.int32), ro)
            n_cpu = {'c4': 20000, 'c1b': 300000}.get(a.config, 1000000)
            out['cpu_baseline'] = cpu_baseline(geo, ph, a.mesh_n, desc, n=n_cpu, rough=rough)
            if all_cores is not None:
                out['cpu_baseline_all_cores'] = all_cores
        out['cpu_reference_numpy'] = CPU_REFERENCE_NUMPY
        print(json.dumps(out))
        sys.stdout.flush()
    rdv.barrier()
    rdv.close()


if __name__ == '__main__':
    main()
