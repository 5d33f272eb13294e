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
    np.savez_compressed(os.path.join(HERE, 'dropin.npz'), **out)
    print('wrote dropin.npz: %d arrays' % len(out))
