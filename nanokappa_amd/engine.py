"""ctypes binding of libnanokappa_hip.so (include/nanokappa_hip.h) and a thin `Engine` object.

No CPU fallback: if the library is missing it must be built (`python -c "import __graft_entry__ as g;
g.build()"` or `make -C nanokappa_amd/csrc`), and `Engine()` raises when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get('NK_LIBNAME', 'libnanokappa_hip.so'))   # NK_LIBNAME: developer builds

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_int8)
c_up = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


class NkError(RuntimeError):
    pass


class nk_material(C.Structure):
    _fields_ = [('Q', C.c_int32), ('J', C.c_int32), ('NT', C.c_int32),
                ('omega', c_dp), ('group_vel', c_dp), ('T_grid', c_dp), ('lifetime', c_dp),
                ('nE', C.c_int32), ('T_fill_lo', C.c_double), ('T_fill_hi', C.c_double),
                ('T_array', c_dp), ('energy_array', c_dp),
                ('hbar', C.c_double), ('kb', C.c_double), ('QV', C.c_double), ('active_modes', C.c_int32)]


class nk_mesh(C.Structure):
    _fields_ = [('F', C.c_int32), ('normals', c_dp), ('k', c_dp), ('bounds_lo', c_dp), ('bounds_hi', c_dp),
                ('basis', c_dp), ('origins', c_dp), ('face_facet', c_ip), ('vertices', c_dp), ('face_area', c_dp),
                ('Fc', C.c_int32), ('facet_bc', c_bp), ('facet_partner', c_ip), ('facet_centroid', c_dp),
                ('facet_normal', c_dp), ('facet_face_off', c_ip), ('facet_face_idx', c_ip),
                ('tol', C.c_double), ('bbox', C.c_double * 6),
                ('nS', C.c_int32), ('simplex_pts', c_dp), ('simplex_vol', c_dp)]


class nk_subvols(C.Structure):
    _fields_ = [('S', C.c_int32), ('kind', C.c_int32), ('axis', C.c_int32), ('interp', C.c_int32),
                ('centers', c_dp), ('volumes', c_dp), ('rbf_inv', c_dp), ('rbf_shift', c_dp), ('rbf_scale', c_dp),
                ('rbf_used', C.c_int32 * 3)]


class nk_reservoirs(C.Structure):
    _fields_ = [('R', C.c_int32), ('facet', c_ip), ('T', c_dp), ('enter_prob', c_dp), ('counter', c_dp),
                ('gen', C.c_int32), ('n_leaving', C.POINTER(C.c_int64))]


class nk_rough(C.Structure):
    _fields_ = [('Fr', C.c_int32), ('facet', c_ip), ('specularity', c_dp), ('true_spec', c_up),
                ('spec_map', c_ip), ('roulette', c_dp), ('degen_j2', c_ip)]


class nk_params(C.Structure):
    _fields_ = [('dt', C.c_double), ('norm_fixed', C.c_int32), ('particle_density', C.c_double),
                ('T_ref_local', C.c_int32), ('T_ref', C.c_double), ('flux_every', C.c_int32),
                ('contains_every', C.c_int32), ('track_ids', C.c_int32)]


class nk_tally(C.Structure):
    _fields_ = [('T_sv', c_dp), ('E_sv', c_dp), ('E_raw', c_dp), ('N_sv', c_dp), ('flux_raw', c_dp),
                ('N_leaving', c_dp), ('res_energy', c_dp), ('res_flux', c_dp), ('N_emitted', c_dp)]


class nk_timing(C.Structure):
    _fields_ = [('step_kernel_ms', C.c_double), ('emit_kernel_ms', C.c_double), ('events_kernel_ms', C.c_double),
                ('total_ms', C.c_double),
                ('slots', C.c_int64), ('live', C.c_int64),
                ('regrows', C.c_int64), ('halts', C.c_int64), ('tau_rebuilds', C.c_int64), ('batches', C.c_int64),
                ('emit_fused', C.c_int64), ('place_tries', C.c_int64), ('place_gbps', C.c_double), ('place_worst_gbps', C.c_double),
                ('box_store', C.c_int64)]


class nk_comm_report(C.Structure):
    _fields_ = [('rank', C.c_int32), ('nranks', C.c_int32), ('comm_rank', C.c_int32), ('comm_nranks', C.c_int32),
                ('device', C.c_int32), ('selftest_ok', C.c_int32), ('selftest_sum', C.c_double), ('pci_bus_id', C.c_char * 32)]


EXPORTS = ['nk_device_count', 'nk_create', 'nk_destroy', 'nk_last_error', 'nk_set_material', 'nk_set_mesh', 'nk_set_subvolumes',
           'nk_set_reservoirs', 'nk_set_rough', 'nk_set_params', 'nk_reserve', 'nk_upload_particles',
           'nk_init_boundaries', 'nk_step', 'nk_download_particles', 'nk_get_subvol_temperature',
           'nk_set_subvol_temperature', 'nk_get_step', 'nk_get_timing', 'nk_comm_unique_id', 'nk_comm_init',
           'nk_find_boundary', 'nk_classify', 'nk_eval', 'nk_reflect', 'nk_uniform2', 'nk_calibrate_stream',
           'nk_specular_begin', 'nk_specular_pairs', 'nk_specular_end', 'nk_rough_begin', 'nk_rough_pairs', 'nk_rough_finish',
           'nk_rough_download', 'nk_build_enter_prob', 'nk_init_particles', 'nk_tally_state', 'nk_kspec_begin', 'nk_kspec_pairs',
           'nk_rough_finish_k', 'nk_mesh_crossings', 'nk_comm_info', 'nk_comm_allreduce']

_lib = None


def load_library():
    """Load libnanokappa_hip.so; raises NkError with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NkError('%s not found: build it with `make -C %s` (needs hipcc); there is no CPU fallback'
                      % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    L = C.CDLL(LIB_PATH)
    L.nk_last_error.restype = C.c_char_p
    L.nk_last_error.argtypes = [C.c_void_p]
    L.nk_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint64]
    L.nk_destroy.argtypes = [C.c_void_p]
    L.nk_destroy.restype = None
    for name, st in (('nk_set_material', nk_material), ('nk_set_mesh', nk_mesh), ('nk_set_reservoirs', nk_reservoirs),
                     ('nk_set_rough', nk_rough), ('nk_set_params', nk_params)):
        getattr(L, name).argtypes = [C.c_void_p, C.POINTER(st)]
    L.nk_set_subvolumes.argtypes = [C.c_void_p, C.POINTER(nk_subvols), c_dp]
    L.nk_reserve.argtypes = [C.c_void_p, C.c_int64]
    L.nk_upload_particles.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, c_ip, c_u64p, C.c_uint64]
    L.nk_init_boundaries.argtypes = [C.c_void_p]
    L.nk_init_particles.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, c_ip, C.c_int64, C.POINTER(C.c_int64)]
    L.nk_tally_state.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
    L.nk_mesh_crossings.argtypes = [C.c_int, C.c_int64, c_dp, c_dp, C.c_int64, c_dp, c_dp, c_dp, C.c_int, c_ip]
    L.nk_step.argtypes = [C.c_void_p, C.c_int32, C.POINTER(nk_tally)]
    L.nk_download_particles.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, c_ip, c_u64p,
                                        C.POINTER(C.c_int64)]
    L.nk_get_subvol_temperature.argtypes = [C.c_void_p, c_dp]
    L.nk_set_subvol_temperature.argtypes = [C.c_void_p, c_dp]
    L.nk_get_step.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.nk_get_timing.argtypes = [C.c_void_p, C.POINTER(nk_timing)]
    L.nk_comm_unique_id.argtypes = [C.c_void_p]
    L.nk_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.nk_comm_info.argtypes = [C.c_void_p, C.POINTER(nk_comm_report)]
    L.nk_comm_allreduce.argtypes = [C.c_void_p, c_dp, C.c_int64]
    L.nk_find_boundary.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_ip]
    L.nk_classify.argtypes = [C.c_void_p, C.c_int64, c_dp, c_ip]
    L.nk_eval.argtypes = [C.c_void_p, C.c_int32, C.c_int64, c_dp, c_ip, c_dp]
    L.nk_reflect.argtypes = [C.c_void_p, C.c_int64, c_ip, c_ip, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp]
    L.nk_uniform2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, c_dp, c_dp]
    L.nk_calibrate_stream.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.nk_specular_begin.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp]
    L.nk_specular_pairs.argtypes = [C.c_void_p, c_dp, C.c_double, C.c_int64, c_ip, c_ip, C.POINTER(C.c_int64)]
    L.nk_specular_end.argtypes = [C.c_void_p]
    L.nk_rough_begin.argtypes = [C.c_void_p, C.c_int32, c_ip, c_dp, c_dp, c_dp]
    L.nk_rough_pairs.argtypes = [C.c_void_p, C.c_int32, c_ip]
    L.nk_rough_finish.argtypes = [C.c_void_p]
    L.nk_rough_finish_k.argtypes = [C.c_void_p, C.c_int32, c_ip, c_ip]
    L.nk_kspec_begin.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_dp]
    L.nk_kspec_pairs.argtypes = [C.c_void_p, c_dp, C.c_int64, c_ip, c_ip, C.POINTER(C.c_int64)]
    L.nk_rough_download.argtypes = [C.c_void_p, c_dp, c_up, c_ip, c_dp]
    L.nk_build_enter_prob.argtypes = [C.c_void_p, C.c_int32, c_dp, c_dp, C.c_double, c_dp]
    _lib = L
    return L


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t=c_dp):
    return None if a is None else a.ctypes.data_as(t)


def bc_codes(bound_cond):
    """'T'/'P'/'R' strings (Geometry.bound_cond) or int8 codes -> int8 codes."""
    a = np.asarray(bound_cond)
    if a.dtype.kind in 'US':
        return np.array([ord(str(c)[0]) for c in a], dtype=np.int8)
    return np.ascontiguousarray(a, dtype=np.int8)


def device_count():
    """HIP devices this process sees."""
    return int(load_library().nk_device_count())


def mesh_crossings(origins, dirs, v0, e1, e2, skip_self=False, device=None):
    """Triangles crossed by every open ray (nk_mesh_crossings; -1 where a ray has too many distinct crossings).
    device: HIP device index; default NK_MESH_DEVICE, else this process's LOCAL_RANK (a rank's geometry set-up stays on the
    rank's own GPU), modulo the devices this process sees."""
    L = load_library()
    if device is None:
        device = int(os.environ.get('NK_MESH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        nd = L.nk_device_count()
        device = device % nd if nd > 0 else 0
    o, d = _d(origins), _d(dirs)
    a, b, c = _d(v0), _d(e1), _d(e2)
    out = np.zeros(o.shape[0], dtype=np.int32)
    rc = L.nk_mesh_crossings(int(device), o.shape[0], _p(o), _p(d), a.shape[0], _p(a), _p(b), _p(c), int(bool(skip_self)), _p(out, c_ip))
    if rc != 0:
        raise RuntimeError('nk_mesh_crossings failed (%d)' % rc)
    return out


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, every rank passes it to Engine.comm_init)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.nk_comm_unique_id(buf)
    if rc != 0:
        raise NkError('nk_comm_unique_id failed: %s' % L.nk_last_error(None).decode())
    return buf.raw


class Engine(object):
    """One nk_ctx: tables + particle store on one MI355X."""

    def __init__(self, device=0, seed=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.nk_create(C.byref(h), int(device), C.c_uint64(int(seed)))
        if rc != 0:
            raise NkError('nk_create failed (%d): %s' % (rc, self.L.nk_last_error(None).decode()))
        self.h = h
        self.S = self.R = self.J = self.M = 0

    def close(self):
        if getattr(self, 'h', None):
            self.L.nk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise NkError('%s failed (%d): %s' % (what, rc, self.L.nk_last_error(self.h).decode()))

    # ------------------------------------------------------------------ tables
    def set_material(self, t):
        """t: Phonon.tables()"""
        m = nk_material()
        om, vg, tg, lt = _d(t['omega']), _d(t['group_vel']), _d(t['T_grid']), _d(t['lifetime'])
        ta, ea = _d(t['T_array']), _d(t['energy_array'])
        m.Q, m.J = om.shape
        m.NT = tg.shape[0]
        m.omega, m.group_vel, m.T_grid, m.lifetime = _p(om), _p(vg), _p(tg), _p(lt)
        m.nE = ta.shape[0]
        m.T_fill_lo, m.T_fill_hi = float(tg.min()), float(tg.max())
        m.T_array, m.energy_array = _p(ta), _p(ea)
        m.hbar, m.kb, m.QV = float(t['hbar']), float(t['kb']), float(t['QV'])
        m.active_modes = int(t['active_modes'])
        self._ck(self.L.nk_set_material(self.h, C.byref(m)), 'nk_set_material')
        self.J = int(m.J)
        self.M = int(m.Q * m.J)

    def set_mesh(self, g):
        """g: mapping with the reference's Mesh/Geometry attribute names (face_normals, face_k, face_bounds,
        face_basis_matrix, face_origins, face_facets, vertices, faces, face_areas, bound_cond, connected_facets,
        facet_centroid, facets_normal, facets | facets_flat+facets_len, bounds, simplices*)."""
        m = nk_mesh()
        nrm, k = _d(g['face_normals']), _d(g['face_k'])
        fb = np.asarray(g['face_bounds'], dtype=np.float64)
        lo, hi = _d(fb[0]), _d(fb[1])
        basis, org, ff = _d(g['face_basis_matrix']), _d(g['face_origins']), _i(g['face_facets'])
        verts = _d(np.asarray(g['vertices'])[np.asarray(g['faces'], dtype=int)])
        area = _d(g['face_areas'])
        m.F = nrm.shape[0]
        m.normals, m.k, m.bounds_lo, m.bounds_hi = _p(nrm), _p(k), _p(lo), _p(hi)
        m.basis, m.origins, m.face_facet, m.vertices, m.face_area = _p(basis), _p(org), _p(ff, c_ip), _p(verts), _p(area)
        bc = bc_codes(g['bound_cond'])
        Fc = bc.shape[0]
        partner = -np.ones(Fc, dtype=np.int32)
        for a, b in np.asarray(g.get('connected_facets', np.zeros((0, 2))), dtype=int).reshape(-1, 2):
            partner[a], partner[b] = b, a
        cen, fn = _d(g['facet_centroid']), _d(g['facets_normal'])
        if 'facets_flat' in g:
            flat, ln = _i(g['facets_flat']), np.asarray(g['facets_len'], dtype=int)
        else:
            flat, ln = _i(np.concatenate(g['facets'])), np.array([len(f) for f in g['facets']])
        off = _i(np.concatenate(([0], np.cumsum(ln))))
        m.Fc = Fc
        m.facet_bc, m.facet_partner, m.facet_centroid, m.facet_normal = _p(bc, c_bp), _p(partner, c_ip), _p(cen), _p(fn)
        m.facet_face_off, m.facet_face_idx = _p(off, c_ip), _p(flat, c_ip)
        m.tol = float(g.get('tol', 1e-10))
        b = np.asarray(g['bounds'], dtype=np.float64)
        for d in range(3):
            m.bbox[d], m.bbox[3 + d] = b[0, d], b[1, d]
        if 'simplices' in g and len(g['simplices']):
            sp = _d(np.asarray(g['simplices_points'])[np.asarray(g['simplices'], dtype=int)])
            sv = _d(g['simplices_volumes'])
            m.nS = sv.shape[0]
            m.simplex_pts, m.simplex_vol = _p(sp), _p(sv)
        else:
            m.nS = 0
        self._ck(self.L.nk_set_mesh(self.h, C.byref(m)), 'nk_set_mesh')

    def set_subvolumes(self, centers, volumes, kind, axis, interp, T_sv, rbf=None):
        """interp 3 (cubic RBF) needs rbf = (inv, shift, scale, used) from setup_tables.rbf_system."""
        s = nk_subvols()
        c, v, t = _d(centers), _d(volumes), _d(T_sv)
        s.S = c.shape[0]
        s.kind, s.axis, s.interp = int(kind), int(axis), int(interp)
        s.centers, s.volumes = _p(c), _p(v)
        if rbf is not None:
            inv, sh, sc = _d(rbf[0]), _d(rbf[1]), _d(rbf[2])
            s.rbf_inv, s.rbf_shift, s.rbf_scale = _p(inv), _p(sh), _p(sc)
            for k in range(3):
                s.rbf_used[k] = int(rbf[3][k])
        self._ck(self.L.nk_set_subvolumes(self.h, C.byref(s), _p(t)), 'nk_set_subvolumes')
        self.S = int(s.S)

    def set_reservoirs(self, facets, T, enter_prob, counter, gen=0, n_leaving=None):
        """gen: 0 'constant', 1 'fixed_rate', 2 'one_to_one' (needs n_leaving[R], the first step's emission)."""
        r = nk_reservoirs()
        f, t, ep, cn = _i(facets), _d(T), _d(enter_prob), _d(counter)
        r.R = f.shape[0]
        r.facet, r.T, r.enter_prob, r.counter = _p(f, c_ip), _p(t), _p(ep), _p(cn)
        r.gen = int(gen)
        nl = None if n_leaving is None else np.ascontiguousarray(n_leaving, dtype=np.int64)
        r.n_leaving = None if nl is None else nl.ctypes.data_as(C.POINTER(C.c_int64))
        self._ck(self.L.nk_set_reservoirs(self.h, C.byref(r)), 'nk_set_reservoirs')
        self.R = int(r.R)

    def set_rough(self, facets, specularity, true_spec, spec_map, roulette, degen_j2=None):
        r = nk_rough()
        f, sp, ts = _i(facets), _d(specularity), np.ascontiguousarray(true_spec, dtype=np.uint8)
        sm, ro = _i(spec_map), _d(roulette)
        dj = None if degen_j2 is None else _i(degen_j2)
        r.Fr = f.shape[0]
        r.facet, r.specularity, r.true_spec, r.spec_map, r.roulette = _p(f, c_ip), _p(sp), _p(ts, c_up), _p(sm, c_ip), _p(ro)
        r.degen_j2 = _p(dj, c_ip)
        self._ck(self.L.nk_set_rough(self.h, C.byref(r)), 'nk_set_rough')

    def set_params(self, dt=1.0, norm_fixed=False, particle_density=0.0, T_ref=None, flux_every=10, contains_every=100,
                   track_ids=False):
        """track_ids: store 64-bit particle ids even when nothing draws random numbers per particle (then downloads
        return them; otherwise pid comes back as zeros unless the mesh has rough facets)."""
        p = nk_params()
        p.track_ids = int(bool(track_ids))
        p.dt, p.norm_fixed, p.particle_density = float(dt), int(bool(norm_fixed)), float(particle_density)
        p.T_ref_local = 1 if T_ref is None else 0
        p.T_ref = 0.0 if T_ref is None else float(T_ref)
        p.flux_every, p.contains_every = int(flux_every), int(contains_every)
        self._ck(self.L.nk_set_params(self.h, C.byref(p)), 'nk_set_params')

    # --------------------------------------------------------------- particles
    def reserve(self, capacity):
        self._ck(self.L.nk_reserve(self.h, int(capacity)), 'nk_reserve')

    def upload(self, positions, mode, occ, n_ts=None, facet=None, pid=None, pid_offset=0):
        pos = np.asarray(positions, dtype=np.float64)
        x, y, z = _d(pos[:, 0]), _d(pos[:, 1]), _d(pos[:, 2])
        m, o = _i(mode), _d(occ)
        nt = None if n_ts is None else _d(n_ts)
        fc = None if facet is None else _i(facet)
        pi = None if pid is None else np.ascontiguousarray(pid, dtype=np.uint64)
        self._ck(self.L.nk_upload_particles(self.h, x.shape[0], _p(x), _p(y), _p(z), _p(m, c_ip), _p(o), _p(nt),
                                            _p(fc, c_ip), _p(pi, c_u64p), C.c_uint64(int(pid_offset))),
                 'nk_upload_particles')

    def init_boundaries(self):
        self._ck(self.L.nk_init_boundaries(self.h), 'nk_init_boundaries')

    def init_particles(self, n, capacity, pid_lo, unique_modes, sv_first=None):
        """initialise_all_particles on the device (tiled modes; sv_first: first particle index of every subvolume's share)."""
        um = _i(unique_modes)
        sf = None if sv_first is None else np.ascontiguousarray(sv_first, dtype=np.int64)
        self._ck(self.L.nk_init_particles(self.h, int(n), int(capacity), int(pid_lo), um.ctypes.data_as(c_ip), um.shape[0],
                                          None if sf is None else sf.ctypes.data_as(C.POINTER(C.c_int64))), 'nk_init_particles')

    def tally_state(self):
        """E_raw[S], N_sv[S], flux_raw[S, 3] of the particles where they stand (this rank's)."""
        S = self.S
        E, N, F = np.zeros(S), np.zeros(S), np.zeros((S, 3))
        self._ck(self.L.nk_tally_state(self.h, E.ctypes.data_as(c_dp), N.ctypes.data_as(c_dp), F.ctypes.data_as(c_dp)), 'nk_tally_state')
        return E, N, F

    def step(self, nsteps=1):
        """Run nsteps timesteps; returns a dict of per-step arrays (see nk_tally in the header).  The arrays are views of ONE block
        (a driver that steps one by one calls this a thousand times a second: one allocation and one address instead of nine)."""
        S, R = self.S, max(self.R, 0)
        shapes = (('T_sv', (nsteps, S)), ('E_sv', (nsteps, S)), ('E_raw', (nsteps, S)), ('N_sv', (nsteps, S)), ('flux_raw', (nsteps, S, 3)),
                  ('N_leaving', (nsteps, R)), ('res_energy', (nsteps, R)), ('res_flux', (nsteps, R, 3)), ('N_emitted', (nsteps,)))
        total = nsteps * (7 * S + 5 * R + 1)
        block = np.zeros(total if total > 0 else 1)
        base = block.ctypes.data
        out, t, off = {}, nk_tally(), 0
        for k, shp in shapes:
            n = int(np.prod(shp))
            out[k] = block[off:off + n].reshape(shp)
            setattr(t, k, C.cast(base + 8 * off, c_dp))
            off += n
        self._ck(self.L.nk_step(self.h, int(nsteps), C.byref(t)), 'nk_step')
        return out

    def get_step(self):
        """Timesteps this engine has completed (the library's absolute step counter: flux and contains_check cadence)."""
        n = C.c_int64(0)
        self._ck(self.L.nk_get_step(self.h, C.byref(n)), 'nk_get_step')
        return int(n.value)

    def download(self):
        n = C.c_int64(0)
        self._ck(self.L.nk_download_particles(self.h, 0, None, None, None, None, None, None, None, None, C.byref(n)),
                 'nk_download_particles')
        N = int(n.value)
        x, y, z, occ, nts = (np.zeros(N) for _ in range(5))
        mode, facet = np.zeros(N, dtype=np.int32), np.zeros(N, dtype=np.int32)
        pid = np.zeros(N, dtype=np.uint64)
        if N:
            self._ck(self.L.nk_download_particles(self.h, N, _p(x), _p(y), _p(z), _p(mode, c_ip), _p(occ), _p(nts),
                                                  _p(facet, c_ip), _p(pid, c_u64p), C.byref(n)), 'nk_download_particles')
        return dict(positions=np.stack((x, y, z), axis=1), mode=mode, occupation=occ, n_timesteps=nts, facet=facet, pid=pid)

    def subvol_temperature(self):
        t = np.zeros(self.S)
        self._ck(self.L.nk_get_subvol_temperature(self.h, _p(t)), 'nk_get_subvol_temperature')
        return t

    def set_subvol_temperature(self, T):
        t = _d(T)
        self._ck(self.L.nk_set_subvol_temperature(self.h, _p(t)), 'nk_set_subvol_temperature')

    def timing(self):
        t = nk_timing()
        self._ck(self.L.nk_get_timing(self.h, C.byref(t)), 'nk_get_timing')
        return dict(step_kernel_ms=t.step_kernel_ms, emit_kernel_ms=t.emit_kernel_ms, events_kernel_ms=t.events_kernel_ms,
                    total_ms=t.total_ms,
                    slots=int(t.slots), live=int(t.live), regrows=int(t.regrows), halts=int(t.halts),
                    tau_rebuilds=int(t.tau_rebuilds), batches=int(t.batches), emit_fused=int(t.emit_fused),
                    place_tries=int(t.place_tries), place_gbps=t.place_gbps, place_worst_gbps=t.place_worst_gbps,
                    box_store=int(t.box_store))

    # ---- set-up table builder (find_specular_correspondences 'velocity', Population.py:1241-1454)
    def specular_begin(self, group_vel, omega, delta_omega):
        v, om, dl = _d(np.asarray(group_vel).reshape(-1, 3)), _d(np.ravel(omega)), _d(np.ravel(delta_omega))
        self._ck(self.L.nk_specular_begin(self.h, om.shape[0], _p(v), _p(om), _p(dl)), 'nk_specular_begin')
        self._spec_M = om.shape[0]

    def specular_pairs(self, normal, crit=1e-3, download=True):
        """(in, out) flat mode indices of every specular pair for one normal, unordered.  download=False: the pairs stay on
        the device for nk_rough_pairs and nothing is returned."""
        nrm = _d(np.asarray(normal, dtype=float))
        cap = getattr(self, '_spec_cap', 4 * self._spec_M)
        while True:
            n = C.c_int64(0)
            if download:
                pi, po = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
                self._ck(self.L.nk_specular_pairs(self.h, _p(nrm), float(crit), cap, _p(pi, c_ip), _p(po, c_ip), C.byref(n)),
                         'nk_specular_pairs')
            else:
                self._ck(self.L.nk_specular_pairs(self.h, _p(nrm), float(crit), cap, None, None, C.byref(n)), 'nk_specular_pairs')
            if n.value <= cap:
                return (pi[:n.value].astype(np.int64), po[:n.value].astype(np.int64)) if download else None
            cap = self._spec_cap = int(n.value)

    def specular_end(self):
        self._ck(self.L.nk_specular_end(self.h), 'nk_specular_end')

    # ---- rough-facet tables built on the device (between specular_begin and specular_end; 'velocity' model)
    def rough_begin(self, facets, normal_in, eta, k_norm):
        f, n, e, k = _i(facets), _d(normal_in), _d(eta), _d(k_norm)
        self._rough_Fr = f.shape[0]
        self._ck(self.L.nk_rough_begin(self.h, f.shape[0], _p(f, c_ip), _p(n), _p(e), _p(k)), 'nk_rough_begin')

    def rough_pairs(self, rough_facet_indices):
        """The pairs of the last specular_pairs call (still on the device) for the rough facets that share its normal."""
        f = _i(rough_facet_indices)
        self._ck(self.L.nk_rough_pairs(self.h, f.shape[0], _p(f, c_ip)), 'nk_rough_pairs')

    def rough_finish(self, degeneracies=None, degen_j2=None):
        """degeneracies [nd, 3] (q, j1, j2) and degen_j2 [M]: the 'k' model's degenerate branches (nk_rough_finish_k)."""
        if degeneracies is None:
            self._ck(self.L.nk_rough_finish(self.h), 'nk_rough_finish')
            return
        dg = _i(np.asarray(degeneracies, dtype=np.int32).reshape(-1, 3))
        dj = None if degen_j2 is None else _i(degen_j2)
        self._ck(self.L.nk_rough_finish_k(self.h, dg.shape[0], _p(dg, c_ip), _p(dj, c_ip)), 'nk_rough_finish_k')

    def kspec_begin(self, wavevectors, k_to_q, q_to_k, tol):
        """'k' model pair search (after specular_begin): q = k . k_to_q, k = q . q_to_k (3 x 3), tol [3]."""
        kv, a, b, t = _d(wavevectors), _d(np.asarray(k_to_q, dtype=float).reshape(3, 3)), _d(np.asarray(q_to_k, dtype=float).reshape(3, 3)), _d(tol)
        self._ck(self.L.nk_kspec_begin(self.h, kv.shape[0], _p(kv), _p(a), _p(b), _p(t)), 'nk_kspec_begin')

    def kspec_pairs(self, normal, download=True):
        """(in, out) flat mode indices of the 'k' model's specular pairs for one normal (one partner per in-mode)."""
        nrm = _d(np.asarray(normal, dtype=float))
        cap = self._spec_M
        n = C.c_int64(0)
        if download:
            pi, po = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
            self._ck(self.L.nk_kspec_pairs(self.h, _p(nrm), cap, _p(pi, c_ip), _p(po, c_ip), C.byref(n)), 'nk_kspec_pairs')
            return pi[:n.value].astype(np.int64), po[:n.value].astype(np.int64)
        self._ck(self.L.nk_kspec_pairs(self.h, _p(nrm), cap, None, None, C.byref(n)), 'nk_kspec_pairs')
        return None

    def rough_download(self):
        """(specularity, true_spec, spec_map, roulette), each [Fr, M]."""
        Fr, M = self._rough_Fr, self.M
        sp, ts = np.zeros((Fr, M)), np.zeros((Fr, M), dtype=np.uint8)
        sm, ro = np.zeros((Fr, M), dtype=np.int32), np.zeros((Fr, M))
        self._ck(self.L.nk_rough_download(self.h, _p(sp), _p(ts, c_up), _p(sm, c_ip), _p(ro)), 'nk_rough_download')
        return sp, ts, sm, ro

    def build_enter_prob(self, normal_in, thickness, dt):
        n, t = _d(normal_in), _d(thickness)
        out = np.zeros((t.shape[0], self.M))
        self._ck(self.L.nk_build_enter_prob(self.h, t.shape[0], _p(n), _p(t), float(dt), _p(out)), 'nk_build_enter_prob')
        return out

    def calibrate_stream(self, launches=3):
        """Known-traffic sweeps for counter calibration; returns (bytes_read, bytes_written) per launch."""
        r, w = C.c_int64(0), C.c_int64(0)
        self._ck(self.L.nk_calibrate_stream(self.h, int(launches), C.byref(r), C.byref(w)), 'nk_calibrate_stream')
        return int(r.value), int(w.value)

    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self.L.nk_comm_init(self.h, buf, int(rank), int(nranks)), 'nk_comm_init')

    def comm_allreduce(self, *arrays):
        """Sum of small host arrays over the ranks of the communicator; returns them unchanged when there is none."""
        flat = _d(np.concatenate([np.ravel(np.asarray(a, dtype=float)) for a in arrays]))
        self._ck(self.L.nk_comm_allreduce(self.h, _p(flat), flat.shape[0]), 'nk_comm_allreduce')
        out, off = [], 0
        for a in arrays:
            a = np.asarray(a)
            out.append(flat[off:off + a.size].reshape(a.shape).copy())
            off += a.size
        return out

    def comm_info(self):
        """What RCCL itself reports about this context's communicator (rank count, own rank, the self-test all-reduce of
        nk_comm_init), the HIP device and its PCI bus id."""
        r = nk_comm_report()
        self._ck(self.L.nk_comm_info(self.h, C.byref(r)), 'nk_comm_info')
        return dict(rank=int(r.rank), nranks=int(r.nranks), comm_rank=int(r.comm_rank), comm_nranks=int(r.comm_nranks),
                    device=int(r.device), selftest_ok=bool(r.selftest_ok), selftest_sum=float(r.selftest_sum),
                    pci_bus_id=r.pci_bus_id.decode(errors='replace'))

    # -------------------------------------------------------------------- taps
    def find_boundary(self, x, v):
        x, v = _d(x), _d(v)
        n = x.shape[0]
        xc, tc, fc = np.zeros((n, 3)), np.zeros(n), np.zeros(n, dtype=np.int32)
        self._ck(self.L.nk_find_boundary(self.h, n, _p(x), _p(v), _p(xc), _p(tc), _p(fc, c_ip)), 'nk_find_boundary')
        return xc, tc, fc

    def classify(self, x):
        x = _d(x)
        out = np.zeros(x.shape[0], dtype=np.int32)
        self._ck(self.L.nk_classify(self.h, x.shape[0], _p(x), _p(out, c_ip)), 'nk_classify')
        return out

    def eval(self, what, a, mode=None):
        code = {'occupation': 0, 'lifetime': 1, 'T_of_E': 2, 'E_of_T': 3, 'interp_T': 4, 'exp': 5}[what]
        a = _d(a)
        n = a.shape[0]
        m = None if mode is None else _i(mode)
        out = np.zeros(n)
        self._ck(self.L.nk_eval(self.h, code, n, _p(a), _p(m, c_ip), _p(out)), 'nk_eval')
        return out

    def reflect(self, facet, mode_in, col_pos, n_in, omega_in, r_spec, r_deg, r_diff):
        f, m, c = _i(facet), _i(mode_in), _d(col_pos)
        n = f.shape[0]
        a = [_d(v) for v in (n_in, omega_in, r_spec)]
        rd = None if r_deg is None else _d(r_deg)
        rf = _d(r_diff)
        mo, no, oo = np.zeros(n, dtype=np.int32), np.zeros(n), np.zeros(n)
        self._ck(self.L.nk_reflect(self.h, n, _p(f, c_ip), _p(m, c_ip), _p(c), _p(a[0]), _p(a[1]), _p(a[2]), _p(rd),
                                   _p(rf), _p(mo, c_ip), _p(no), _p(oo)), 'nk_reflect')
        return mo, no, oo

    def uniform2(self, seed, pid, step, tag):
        u0, u1 = C.c_double(), C.c_double()
        rc = self.L.nk_uniform2(C.c_uint64(seed), C.c_uint64(pid), C.c_uint32(step), C.c_uint32(tag), C.byref(u0), C.byref(u1))
        if rc != 0:
            raise NkError('nk_uniform2 failed')
        return u0.value, u1.value
