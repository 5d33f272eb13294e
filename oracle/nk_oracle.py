"""ctypes binding of the CPU oracle (oracle/nk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by nanokappa_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)
c_bp = C.POINTER(C.c_int8)
c_up = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


class Material(C.Structure):
    _fields_ = [('Q', C.c_int32), ('J', C.c_int32), ('NT', C.c_int32),
                ('omega', c_dp), ('group_vel', c_dp), ('T_grid', c_dp), ('lifetime', c_dp),
                ('nE', C.c_int32), ('T_fill_lo', C.c_double), ('T_fill_hi', C.c_double),
                ('T_array', c_dp), ('energy_array', c_dp),
                ('hbar', C.c_double), ('kb', C.c_double), ('QV', C.c_double),
                ('active_modes', C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [('F', C.c_int32), ('normals', c_dp), ('k', c_dp), ('bounds_lo', c_dp), ('bounds_hi', c_dp),
                ('basis', c_dp), ('origins', c_dp), ('face_facet', c_ip), ('vertices', c_dp),
                ('face_area', c_dp), ('Fc', C.c_int32), ('facet_bc', c_bp), ('facet_partner', c_ip),
                ('facet_centroid', c_dp), ('facet_normal', c_dp), ('facet_face_off', c_ip),
                ('facet_face_idx', c_ip), ('tol', C.c_double), ('bbox', C.c_double * 6),
                ('nS', C.c_int32), ('simplex_pts', c_dp), ('simplex_vol', c_dp)]


class Subvols(C.Structure):
    _fields_ = [('S', C.c_int32), ('kind', C.c_int32), ('axis', C.c_int32), ('interp', C.c_int32),
                ('centers', c_dp), ('volumes', c_dp), ('rbf_inv', c_dp), ('rbf_shift', c_dp), ('rbf_scale', c_dp),
                ('rbf_used', C.c_int32 * 3)]


class Reservoirs(C.Structure):
    _fields_ = [('R', C.c_int32), ('facet', c_ip), ('T', c_dp), ('enter_prob', c_dp), ('counter', c_dp),
                ('gen', C.c_int32), ('dbg_dt_in', c_dp), ('dbg_x0', c_dp), ('dbg_level', c_ip), ('dbg_res', c_ip),
                ('n_leaving', c_lp), ('dice', c_dp)]


class Rough(C.Structure):
    _fields_ = [('Fr', C.c_int32), ('facet', c_ip), ('specularity', c_dp), ('true_spec', c_up),
                ('spec_map', c_ip), ('roulette', c_dp), ('degen_j2', c_ip)]


class Params(C.Structure):
    _fields_ = [('dt', C.c_double), ('norm_fixed', C.c_int32), ('particle_density', C.c_double),
                ('T_ref_local', C.c_int32), ('T_ref', C.c_double), ('seed', C.c_uint64), ('ids_from_state', C.c_int32),
                ('box', C.c_int32), ('box_k', C.c_double * 6), ('box_facet', C.c_int32 * 6), ('box_face0', C.c_int32 * 6)]


class Particles(C.Structure):
    _fields_ = [('N', C.c_int64), ('cap', C.c_int64), ('pos', c_dp), ('mode', c_ip), ('occ', c_dp),
                ('n_ts', c_dp), ('facet', c_ip), ('pid', c_u64p), ('energy', c_dp), ('temp', c_dp),
                ('sv', c_ip)]


def build(force=False):
    so = os.path.join(_HERE, 'libnk_oracle.so')
    src = os.path.join(_HERE, 'nk_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.nko_emit.restype = C.c_int64
        L.nko_contains_check.restype = C.c_int64
        L.nko_init_boundaries.restype = C.c_int64
        L.nko_box_detect.restype = C.c_int32
    return _LIB


def _keep(obj, *arrays):
    """Pin numpy buffers to the struct that points into them."""
    if not hasattr(obj, '_keep'):
        obj._keep = []
    obj._keep.extend(arrays)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def make_material(tables):
    """tables: dict from nanokappa_amd.phonon.Phonon.tables() (omega, group_vel, T_grid, lifetime,
    T_array, energy_array, hbar, kb, QV, active_modes)."""
    m = Material()
    om = _d(tables['omega']); vg = _d(tables['group_vel']); tg = _d(tables['T_grid'])
    lt = _d(tables['lifetime']); ta = _d(tables['T_array']); ea = _d(tables['energy_array'])
    m.Q, m.J = om.shape
    m.NT = tg.shape[0]
    m.omega, m.group_vel, m.T_grid, m.lifetime = _p(om, c_dp), _p(vg, c_dp), _p(tg, c_dp), _p(lt, c_dp)
    m.nE = ta.shape[0]
    m.T_fill_lo = float(tg.min()); m.T_fill_hi = float(tg.max())
    m.T_array, m.energy_array = _p(ta, c_dp), _p(ea, c_dp)
    m.hbar, m.kb, m.QV = float(tables['hbar']), float(tables['kb']), float(tables['QV'])
    m.active_modes = int(tables['active_modes'])
    _keep(m, om, vg, tg, lt, ta, ea)
    return m


def make_mesh(g):
    """g: dict with the reference's mesh/geometry attribute names (face_normals, face_k, face_bounds,
    face_basis_matrix, face_origins, face_facets, vertices, faces, face_areas, bound_cond codes,
    connected_facets, facet_centroid, facets_normal, facets (list) or facets_flat/facets_len, bounds,
    simplices_points, simplices, simplices_volumes)."""
    m = Mesh()
    nrm = _d(g['face_normals']); k = _d(g['face_k'])
    fb = np.asarray(g['face_bounds'], dtype=np.float64)
    lo = _d(fb[0]); hi = _d(fb[1])
    basis = _d(g['face_basis_matrix']); org = _d(g['face_origins']); ff = _i(g['face_facets'])
    verts = _d(np.asarray(g['vertices'])[np.asarray(g['faces'])])          # (F,3,3)
    area = _d(g['face_areas'])
    m.F = nrm.shape[0]
    m.normals, m.k, m.bounds_lo, m.bounds_hi = _p(nrm, c_dp), _p(k, c_dp), _p(lo, c_dp), _p(hi, c_dp)
    m.basis, m.origins, m.face_facet, m.vertices, m.face_area = (_p(basis, c_dp), _p(org, c_dp), _p(ff, c_ip),
                                                                _p(verts, c_dp), _p(area, c_dp))
    bc = np.asarray(g['bound_cond'])
    if bc.dtype.kind in 'US':
        bc = np.array([ord(str(c)[0]) for c in bc])
    bc = np.ascontiguousarray(bc, dtype=np.int8)
    Fc = bc.shape[0]
    partner = -np.ones(Fc, dtype=np.int32)
    for a, b in np.asarray(g.get('connected_facets', np.zeros((0, 2))), dtype=int).reshape(-1, 2):
        partner[a] = b
        partner[b] = a
    cen = _d(g['facet_centroid']); fn = _d(g['facets_normal'])
    if 'facets_flat' in g:
        flat = _i(g['facets_flat']); ln = np.asarray(g['facets_len'], dtype=int)
    else:
        flat = _i(np.concatenate(g['facets'])); ln = np.array([len(f) for f in g['facets']])
    off = _i(np.concatenate(([0], np.cumsum(ln))))
    m.Fc = Fc
    m.facet_bc, m.facet_partner, m.facet_centroid, m.facet_normal = (_p(bc, c_bp), _p(partner, c_ip),
                                                                    _p(cen, c_dp), _p(fn, c_dp))
    m.facet_face_off, m.facet_face_idx = _p(off, c_ip), _p(flat, c_ip)
    m.tol = 1e-10
    b = np.asarray(g['bounds'], dtype=np.float64)
    for d in range(3):
        m.bbox[d] = b[0, d]
        m.bbox[3 + d] = b[1, d]
    sp = _d(np.asarray(g['simplices_points'])[np.asarray(g['simplices'], dtype=int)])   # (nS,4,3)
    sv = _d(g['simplices_volumes'])
    m.nS = sv.shape[0]
    m.simplex_pts, m.simplex_vol = _p(sp, c_dp), _p(sv, c_dp)
    _keep(m, nrm, k, lo, hi, basis, org, ff, verts, area, bc, partner, cen, fn, off, flat, sp, sv)
    return m


def make_subvols(centers, volumes, kind, axis, interp, rbf=None):
    """interp 3 (cubic RBF) needs rbf = (inv, shift, scale, used) from nanokappa_amd.setup_tables.rbf_system."""
    s = Subvols()
    c = _d(centers); v = _d(volumes)
    s.S = c.shape[0]
    s.kind, s.axis, s.interp = int(kind), int(axis), int(interp)
    s.centers, s.volumes = _p(c, c_dp), _p(v, c_dp)
    _keep(s, c, v)
    if rbf is not None:
        inv, sh, sc = _d(rbf[0]), _d(rbf[1]), _d(rbf[2])
        s.rbf_inv, s.rbf_shift, s.rbf_scale = _p(inv, c_dp), _p(sh, c_dp), _p(sc, c_dp)
        for k in range(3):
            s.rbf_used[k] = int(rbf[3][k])
        _keep(s, inv, sh, sc)
    return s


def make_reservoirs(facets, T, enter_prob, counter, gen=0, n_leaving=None):
    """gen: 0 'constant', 1 'fixed_rate', 2 'one_to_one' (then n_leaving[R] = particles to emit at the first step,
    Population.py:344; OracleSim keeps it up to date afterwards)."""
    r = Reservoirs()
    f = _i(facets); t = _d(T); ep = _d(enter_prob); cn = _d(counter)
    r.R = f.shape[0]
    r.facet, r.T, r.enter_prob, r.counter = _p(f, c_ip), _p(t, c_dp), _p(ep, c_dp), _p(cn, c_dp)
    r.gen = gen
    nl = np.ascontiguousarray(np.zeros(f.shape[0]) if n_leaving is None else n_leaving, dtype=np.int64)
    r.n_leaving = _p(nl, c_lp)
    _keep(r, f, t, ep, cn, nl)
    r.counter_array = cn
    r.n_leaving_array = nl
    return r


def attach_dice(res, dice):
    """'fixed_rate' test tap: the generator's dice (Population.py:410) come from `dice` [R, Q*J] instead of Philox."""
    dc = _d(dice)
    res.dice = _p(dc, c_dp)
    _keep(res, dc)
    return res


def attach_emission_taps(res, cap):
    """Allocate the optional debug outputs of nko_emit."""
    res.tap_dt_in = np.full(cap, np.nan)
    res.tap_x0 = np.full((cap, 3), np.nan)
    res.tap_level = np.zeros(cap, dtype=np.int32)
    res.tap_res = np.zeros(cap, dtype=np.int32)
    res.dbg_dt_in, res.dbg_x0 = _p(res.tap_dt_in, c_dp), _p(res.tap_x0, c_dp)
    res.dbg_level, res.dbg_res = _p(res.tap_level, c_ip), _p(res.tap_res, c_ip)
    return res


def make_rough(facets, specularity, true_spec, spec_map, roulette, degen_j2=None):
    r = Rough()
    f = _i(facets); sp = _d(specularity); ts = np.ascontiguousarray(true_spec, dtype=np.uint8)
    sm = _i(spec_map); ro = _d(roulette)
    r.Fr = f.shape[0]
    r.facet, r.specularity, r.true_spec, r.spec_map, r.roulette = (_p(f, c_ip), _p(sp, c_dp), _p(ts, c_up),
                                                                  _p(sm, c_ip), _p(ro, c_dp))
    if degen_j2 is not None:
        dj = _i(degen_j2)
        r.degen_j2 = _p(dj, c_ip)
        _keep(r, dj)
    _keep(r, f, sp, ts, sm, ro)
    return r


def make_params(dt=1.0, norm_fixed=False, particle_density=0.0, T_ref=None, seed=0, ids_from_state=False):
    p = Params()
    p.ids_from_state = int(bool(ids_from_state))
    p.dt = dt
    p.norm_fixed = int(norm_fixed)
    p.particle_density = particle_density
    p.T_ref_local = 1 if T_ref is None else 0
    p.T_ref = 0.0 if T_ref is None else float(T_ref)
    p.seed = seed
    return p


class ParticleStore(object):
    """Owns the numpy arrays behind an nko_particles struct."""

    def __init__(self, cap):
        self.cap = int(cap)
        self.pos = np.zeros((self.cap, 3))
        self.mode = np.zeros(self.cap, dtype=np.int32)
        self.occ = np.zeros(self.cap)
        self.n_ts = np.zeros(self.cap)
        self.facet = np.zeros(self.cap, dtype=np.int32)
        self.pid = np.zeros(self.cap, dtype=np.uint64)
        self.energy = np.zeros(self.cap)
        self.temp = np.zeros(self.cap)
        self.sv = np.zeros(self.cap, dtype=np.int32)
        s = Particles()
        s.N = 0
        s.cap = self.cap
        s.pos, s.mode, s.occ, s.n_ts = _p(self.pos, c_dp), _p(self.mode, c_ip), _p(self.occ, c_dp), _p(self.n_ts, c_dp)
        s.facet, s.pid, s.energy, s.temp, s.sv = (_p(self.facet, c_ip), _p(self.pid, c_u64p), _p(self.energy, c_dp),
                                                  _p(self.temp, c_dp), _p(self.sv, c_ip))
        self.s = s

    @property
    def N(self):
        return int(self.s.N)

    def load(self, pos, mode, occ, n_ts=None, facet=None, pid=None):
        n = pos.shape[0]
        assert n <= self.cap
        self.pos[:n] = pos
        self.mode[:n] = mode
        self.occ[:n] = occ
        if n_ts is not None:
            self.n_ts[:n] = n_ts
        if facet is not None:
            self.facet[:n] = facet
        self.pid[:n] = np.arange(n, dtype=np.uint64) if pid is None else pid
        self.s.N = n


class OracleSim(object):
    """Runs Population.run_timestep's stage order (Population.py:1724-1769) on the oracle."""

    def __init__(self, mat, mesh, sv, res, rough, params, store, T_sv, box=False):
        """box: False = the reference's rule (events decided on the cached, decremented n_timesteps: what the goldens pin);
        'auto' = the engine's rule: on an axis-aligned box whose sides are its facets the hit is read off the position
        (nk_oracle.h nko_params::box), unless NK_NO_BOX / NK_SPLIT / NK_LAYOUT=soa switch the engine's box store off."""
        self.L = lib()
        self.mat, self.mesh, self.sv, self.res, self.rough, self.p, self.P = mat, mesh, sv, res, rough, params, store
        self.p.box = 0
        off = os.environ.get('NK_NO_BOX') or os.environ.get('NK_SPLIT') or os.environ.get('NK_LAYOUT') == 'soa'
        if box and not off and self.L.nko_box_detect(C.byref(self.mesh), C.byref(self.p)):
            self.p.box = 1
        self.S = sv.S
        self.R = res.R
        self.T_sv = np.array(T_sv, dtype=np.float64)
        self.E_sv = np.zeros(self.S)
        self.E_raw = np.zeros(self.S)
        self.N_sv = np.zeros(self.S, dtype=np.int64)
        self.N_leaving = np.zeros(max(self.R, 1), dtype=np.int64)
        self.res_energy = np.zeros(max(self.R, 1))
        self.res_flux = np.zeros((max(self.R, 1), 3))
        self.flux = np.zeros((self.S, 3))
        self.step = 0
        self.rank, self.nranks = 0, 1

    def ref(self, x):
        return C.byref(x)

    def init_boundaries(self):
        bad = self.L.nko_init_boundaries(self.ref(self.mesh), self.ref(self.mat), self.ref(self.p), self.ref(self.P.s))
        if self.p.box and bad > 0:        # particles outside the box with a wall ahead: the engine goes back to cached hits
            self.p.box = 0

    def run_timestep_sharded(self, allreduce, emit=True, contains_every=100, halt_requests=(0, 0)):
        """run_timestep for one rank of a particle-sharded ensemble: `allreduce(vec)` sums a float64 vector over the
        ranks in place (the role RCCL plays in the HIP engine).  Tallied vector: E_raw[S] | N_sv[S] | N_leaving[R] | the two
        halt requests that ride on it in the engine (nk_kernels.h k_reduce: "a segment could overflow at the next step",
        "a segment cannot take the migrants in its inbox"); returns their sums: every rank sees a request of any rank at the
        same step."""
        L = self.L
        if contains_every and self.step % contains_every == 0:
            L.nko_contains_check(self.ref(self.mat), self.ref(self.mesh), self.ref(self.p),
                                 C.c_int64(self.step), self.ref(self.P.s))
        L.nko_drift(self.ref(self.mat), self.ref(self.p), self.ref(self.P.s))
        if emit and self.R > 0:
            n = L.nko_emit(self.ref(self.mat), self.ref(self.mesh), self.ref(self.res), self.ref(self.p),
                           C.c_int64(self.step), C.c_int32(self.rank), C.c_int32(self.nranks), self.ref(self.P.s))
            if n < 0:
                raise RuntimeError('oracle particle capacity exceeded')
        L.nko_boundary_scattering(self.ref(self.mat), self.ref(self.mesh), self.ref(self.sv), self.ref(self.res),
                                  self.ref(self.rough), self.ref(self.p), _p(self.T_sv, c_dp), C.c_int64(self.step),
                                  self.ref(self.P.s), _p(self.N_leaving, c_lp), _p(self.res_energy, c_dp),
                                  _p(self.res_flux, c_dp))
        L.nko_tally(self.ref(self.mat), self.ref(self.sv), self.ref(self.p), self.ref(self.P.s), _p(self.T_sv, c_dp),
                    _p(self.N_sv, c_lp), _p(self.E_raw, c_dp))
        vec = np.concatenate((self.E_raw, self.N_sv.astype(np.float64), self.N_leaving[:self.R].astype(np.float64),
                              np.asarray(halt_requests, dtype=np.float64)))
        allreduce(vec)
        self.E_raw[:] = vec[:self.S]
        self.N_sv[:] = np.rint(vec[self.S:2 * self.S]).astype(np.int64)
        if self.R > 0 and self.res.gen == 2:             # one_to_one: next step emits what left on ALL ranks
            self.res.n_leaving_array[:] = np.rint(vec[2 * self.S:2 * self.S + self.R]).astype(np.int64)
        halts = (float(vec[-2]), float(vec[-1]))
        L.nko_update_T(self.ref(self.mat), self.ref(self.sv), self.ref(self.p), _p(self.N_sv, c_lp), _p(self.E_raw, c_dp),
                       _p(self.T_sv, c_dp), _p(self.E_sv, c_dp))
        L.nko_assign_T(self.ref(self.sv), _p(self.T_sv, c_dp), self.ref(self.P.s))
        L.nko_lifetime_scattering(self.ref(self.mat), self.ref(self.p), self.ref(self.P.s))
        self.step += 1
        return halts

    def run_timestep(self, emit=True, contains_every=100):
        L = self.L
        if contains_every and self.step % contains_every == 0:
            L.nko_contains_check(self.ref(self.mat), self.ref(self.mesh), self.ref(self.p),
                                 C.c_int64(self.step), self.ref(self.P.s))
        L.nko_drift(self.ref(self.mat), self.ref(self.p), self.ref(self.P.s))
        if emit and self.R > 0:
            n = L.nko_emit(self.ref(self.mat), self.ref(self.mesh), self.ref(self.res), self.ref(self.p),
                           C.c_int64(self.step), C.c_int32(self.rank), C.c_int32(self.nranks), self.ref(self.P.s))
            if n < 0:
                raise RuntimeError('oracle particle capacity exceeded')
        L.nko_boundary_scattering(self.ref(self.mat), self.ref(self.mesh), self.ref(self.sv), self.ref(self.res),
                                  self.ref(self.rough), self.ref(self.p), _p(self.T_sv, c_dp), C.c_int64(self.step),
                                  self.ref(self.P.s), _p(self.N_leaving, c_lp), _p(self.res_energy, c_dp),
                                  _p(self.res_flux, c_dp))
        if self.R > 0 and self.res.gen == 2:             # one_to_one: next step emits what left now (Population.py:1749)
            self.res.n_leaving_array[:] = self.N_leaving[:self.R]
        L.nko_refresh_temperatures(self.ref(self.mat), self.ref(self.sv), self.ref(self.p), self.ref(self.P.s),
                                   _p(self.T_sv, c_dp), _p(self.E_sv, c_dp), _p(self.N_sv, c_lp), _p(self.E_raw, c_dp))
        L.nko_lifetime_scattering(self.ref(self.mat), self.ref(self.p), self.ref(self.P.s))
        self.step += 1
        if self.step % 10 == 0:
            L.nko_heat_flux(self.ref(self.mat), self.ref(self.sv), self.ref(self.p), self.ref(self.P.s),
                            _p(self.N_sv, c_lp), _p(self.flux, c_dp))
