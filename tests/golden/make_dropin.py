"""Drop-in check of the boundary, run in the build container only:

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_dropin.py

Builds the REFERENCE's own `Geometry` and `Phonon` objects (nanokappa.py:71-87), hands them to
`nanokappa_amd.Population` exactly as the reference driver would (nanokappa.py:89), with a recording stand-in for the
device engine, and stores every table the constructor would upload as tests/golden/dropin.npz.  The tests then feed
that fixture to the real engine (GPU) and to the oracle (CPU): the constructor's path through reference objects is
thereby executed, and what it produces is pinned.  Only arrays are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

ref = H.import_reference()
from nanokappa_amd.synthetic import make_material  # noqa: E402
from nanokappa_amd.population import Population  # noqa: E402
from nanokappa_amd.argument_parser import initialise_parser  # noqa: E402


class Recorder(object):
    """Stands in for nanokappa_amd.engine.Engine: records what the constructor uploads."""

    def __init__(self):
        self.calls = {}
        self.J = 0

    def _rec(self, name, **kw):
        self.calls[name] = kw

    def set_material(self, t):
        self._rec('material', **{k: np.asarray(v) for k, v in t.items()})
        self.J = int(np.asarray(t['omega']).shape[1])

    def set_mesh(self, g):
        d = {}
        for k, v in g.items():
            if k == 'facets':
                d['facets_flat'] = np.concatenate(v)
                d['facets_len'] = np.array([len(f) for f in v])
            elif k == 'bound_cond':
                d[k] = np.array([ord(str(c)[0]) for c in v], dtype=np.int8)
            else:
                d[k] = np.asarray(v)
        self._rec('mesh', **d)

    def set_subvolumes(self, centers, volumes, kind, axis, interp, T_sv, rbf=None):
        self._rec('subvols', centers=np.asarray(centers), volumes=np.asarray(volumes), kind=np.array(kind), axis=np.array(axis),
                  interp=np.array(interp), T_sv=np.asarray(T_sv))

    def set_reservoirs(self, facets, T, enter_prob, counter, gen=0, n_leaving=None):
        self._rec('res', facets=np.asarray(facets), T=np.asarray(T), enter_prob=np.asarray(enter_prob),
                  counter=np.asarray(counter), gen=np.array(gen))

    def set_rough(self, facets, specularity, true_spec, spec_map, roulette, degen_j2=None):
        self._rec('rough', facets=np.asarray(facets), specularity=np.asarray(specularity),
                  true_spec=np.asarray(true_spec).astype(np.uint8), spec_map=np.asarray(spec_map), roulette=np.asarray(roulette))

    def set_params(self, **kw):
        self._rec('params', **{k: np.array(np.nan if v is None else v) for k, v in kw.items()})

    def reserve(self, capacity):
        pass

    def upload(self, positions, mode, occ, pid_offset=0, **kw):
        self._rec('particles', positions=np.asarray(positions), mode=np.asarray(mode), occ=np.asarray(occ))

    def init_boundaries(self):
        pass

    def timing(self):
        return dict(slots=0, live=0)


class DeviceRecorder(Recorder):
    """The same stand-in with the methods of the engine's DEVICE builders (nk_rough_begin / nk_specular_pairs / nk_rough_pairs /
    nk_rough_finish, nk_build_enter_prob, nk_init_particles, nk_tally_state): `Population(args, <reference geo>, <reference
    phonon>)` then takes the branch a GPU run takes (population.py: _build_rough_tables, initialise_reservoirs,
    _init_on_device), and what it hands to those builders -- normals, eta, k-norms, thickness, subvolume shares -- is
    recorded.  Values the constructor needs back are computed with plain NumPy (enter_prob) or are placeholders (t = 0 tally)."""

    def __init__(self):
        super(DeviceRecorder, self).__init__()
        self.normals, self.shares = [], []

    def specular_begin(self, group_vel, omega, delta_omega):
        self._rec('spec_begin', group_vel=np.asarray(group_vel), omega=np.asarray(omega), delta_omega=np.asarray(delta_omega))

    def specular_pairs(self, normal, crit=1e-3, download=True):
        self.normals.append(np.asarray(normal, dtype=float))
        self._rec('spec_pairs', normals=np.array(self.normals), crit=np.array(crit))
        return (np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)) if download else None

    def specular_end(self):
        pass

    def rough_begin(self, facets, normal_in, eta, k_norm):
        self._rec('rough_begin', facets=np.asarray(facets), normal_in=np.asarray(normal_in), eta=np.asarray(eta), k_norm=np.asarray(k_norm))

    def rough_pairs(self, idx):
        self.shares.append(np.asarray(idx))
        self._rec('rough_pairs', share_flat=np.concatenate(self.shares), share_len=np.array([len(a) for a in self.shares]))

    def rough_finish(self, degeneracies=None, degen_j2=None):
        self._rec('rough_finish', has_degen=np.array(degeneracies is not None))

    def build_enter_prob(self, normal_in, thickness, dt):
        self._rec('enter_prob_args', normal_in=np.asarray(normal_in), thickness=np.asarray(thickness), dt=np.array(dt))
        vg = self.calls['material']['group_vel']
        p = np.einsum('rd,qjd->rqj', np.asarray(normal_in), vg) * dt / np.asarray(thickness).reshape(-1, 1, 1)
        return np.where(p < 0, 0, p).reshape(len(thickness), -1)

    def init_particles(self, n, capacity, pid_lo, unique_modes, sv_first=None):
        self._rec('init_particles', n=np.array(n), capacity=np.array(capacity), pid_lo=np.array(pid_lo), unique_modes=np.asarray(unique_modes),
                  sv_first=np.asarray(sv_first if sv_first is not None else np.zeros(0)))
        self._sv_first = None if sv_first is None else np.asarray(sv_first)

    def tally_state(self):
        S = self.calls['subvols']['centers'].shape[0]
        N = np.diff(self._sv_first).astype(float) if self._sv_first is not None else np.zeros(S)
        return np.zeros(S), N, np.zeros((S, 3))

    def comm_info(self):
        return dict(comm_nranks=0)


def build_device(case, particles):
    """The constructor's DEVICE-builder branch with the reference's objects: arguments only (prefix <case>_dev__)."""
    argv = H.argv_for(case, particles)
    args = H.make_args(ref, argv)
    geo = ref.Geometry(args)
    ph = H.make_phonon(ref, args, make_material(9, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))
    args.results_folder = ''
    np.random.seed(4321)
    rec = DeviceRecorder()
    pop = Population(args, geo, ph, engine=rec)
    out = {}
    for name in ('spec_begin', 'spec_pairs', 'rough_begin', 'rough_pairs', 'rough_finish', 'enter_prob_args', 'init_particles'):
        for k, v in rec.calls.get(name, {}).items():
            out['%s_dev__%s__%s' % (case, name, k)] = v
    out['%s_dev__rough_on_device' % case] = np.array(bool(getattr(pop, '_rough_on_device', False)))
    out['%s_dev__N_p' % case] = np.array(pop.N_p)
    return out


def build(case, particles):
    argv = H.argv_for(case, particles)
    args = H.make_args(ref, argv)
    geo = ref.Geometry(args)                                   # nanokappa.py:71
    ph = H.make_phonon(ref, args, make_material(9, 'Si', temperatures=np.arange(200.0, 401.0, 10.0)))   # :87 (synthetic tables)
    # flags the reference's parser does not know (--seed, --device, --checkpoint) are read with getattr defaults
    args.results_folder = ''
    np.random.seed(4321)                                       # the reference's Mesh.sample_volume draws from np.random
    rec = Recorder()
    pop = Population(args, geo, ph, engine=rec)                # nanokappa.py:89 with the replacement class
    out = {}
    for name, kw in rec.calls.items():
        for k, v in kw.items():
            out['%s__%s__%s' % (case, name, k)] = v
    out['%s__N_p' % case] = np.array(pop.N_p)
    out['%s__subvol_temperature' % case] = np.asarray(pop.subvol_temperature)
    out['%s__subvol_energy' % case] = np.asarray(pop.subvol_energy)
    return out


if __name__ == '__main__':
    out = {}
    out.update(build('ttp', 20000))
    out.update(build('ttrrp', 20000))
    out.update(build_device('ttp', 200000))         # enough particles for tiled modes: nk_init_particles is taken
    out.update(build_device('ttrrp', 200000))
    np.savez_compressed(os.path.join(HERE, 'dropin.npz'), **out)
    print('wrote dropin.npz: %d arrays' % len(out))
