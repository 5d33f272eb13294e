"""`Population`: drop-in for the reference's class of the same name on its hot path
(reference classes/Population.py; driver contract nanokappa.py:89-107).

    pop = Population(args, geo, phonons)
    pop.run_timestep(geo, phonons)        # or pop.run(n) to batch n steps into one library call
    pop.current_timestep, pop.finish_sim, pop.write_final_state(geo), pop.f, pop.view.postprocess()

`geo` / `phonons` may be this package's Geometry / Phonon or the reference's own objects: only the attributes
listed in SURVEY.md section 8b are read.  The particle state lives in HBM inside libnanokappa_hip.so; the host
keeps the tallies (subvolume temperatures, energies, fluxes, reservoir balances), the convergence bookkeeping
and the text outputs.  There is no CPU fallback: without the library / a GPU the constructor raises.
"""
import glob
import inspect
import os
import sys
from datetime import datetime

import numpy as np

from .constants import Constants
from . import setup_tables as ST
from .engine import Engine
from .sharding import shard_range


class _Stats(object):
    """Mean / std over the last n_mean convergence rows: the numbers Visualisation.read_convergence derives by
    re-reading convergence.txt (reference classes/Visualisation.py:122-212), kept in memory instead."""

    def __init__(self, pop):
        self.pop = pop

    def update_population(self, pop, verbose=False):
        self.pop = pop

    def postprocess(self, verbose=False):
        p = self.pop
        N = p.n_mean
        rows = p.conv_rows[-N:]
        T = np.array([r['T'] for r in rows])
        phi = np.array([r['phi'].ravel() for r in rows])
        en_res = np.array([r['en_res'] for r in rows]).reshape(len(rows), -1)
        self.mean_T, self.std_T = T.mean(axis=0), T.std(axis=0)
        self.mean_sv_phi, self.std_sv_phi = phi.mean(axis=0), phi.std(axis=0)
        self.mean_en_res, self.std_en_res = en_res.mean(axis=0), en_res.std(axis=0)
        if p.subvol_type == 'slice':
            k = np.array([r['sv_k'] for r in rows])
            with np.errstate(invalid='ignore'):
                self.mean_sv_k, self.std_sv_k = np.nanmean(k, axis=0), np.nanstd(k, axis=0)
            kk = np.array([r['kappa'] for r in rows])
            self.mean_k, self.std_k = np.nanmean(kk), np.nanstd(kk)
        else:                                                   # per-connection conductivities, Visualisation.py:196-199
            ck = np.array([r['con_k'] for r in rows])
            con = p._geo.subvol_connections
            dirs = p._geo.subvol_con_vectors / np.linalg.norm(p._geo.subvol_con_vectors, axis=1, keepdims=True)
            with np.errstate(invalid='ignore'):
                self.mean_con_k, self.std_con_k = np.nanmean(ck, axis=0), np.nanstd(ck, axis=0)
                dT = T[:, con[:, 1]] - T[:, con[:, 0]]                                    # Visualisation.py:177-184
                self.mean_con_dT, self.std_con_dT = np.nanmean(dT, axis=0), np.nanstd(dT, axis=0)
                ph3 = phi.reshape(len(rows), -1, 3)
                cphi = np.sum((ph3[:, con[:, 0], :] + ph3[:, con[:, 1], :]) / 2 * dirs[None], axis=2)
                self.mean_con_phi, self.std_con_phi = np.nanmean(cphi, axis=0), np.nanstd(cphi, axis=0)
            weak = np.absolute(self.mean_con_k) < self.std_con_k                          # Visualisation.py:209-212
            self.mean_con_k = np.where(weak, np.nan, self.mean_con_k)
            self.std_con_k = np.where(weak, np.nan, self.std_con_k)


class Population(Constants):
    '''Class comprising the particles to be simulated (GPU-resident).'''

    def __init__(self, arguments, geometry, phonon, engine=None, comm=None):
        super(Population, self).__init__()
        self.args = arguments
        args = arguments
        self.results_folder_name = args.results_folder
        self.n_dt_to_conv = 10                                               # Population.py:41
        self.norm = args.energy_normal[0]
        self.n_of_subvols = geometry.n_of_subvols
        self.subvol_type = geometry.subvol_type
        self.empty_subvols = list(args.empty_subvols)
        self.n_of_empty_subvols = len(self.empty_subvols)
        self.rank, self.nranks = (0, 1) if comm is None else (int(comm[1]), int(comm[2]))

        M = phonon.number_of_active_modes
        self.particle_type = args.particles[0]                               # Population.py:50-63
        if self.particle_type == 'pmps':
            self.particles_pmps = float(args.particles[1])
            self.N_p = int(np.ceil(self.particles_pmps * M * self.n_of_subvols))
            self.particle_density = self.N_p / geometry.volume
        elif self.particle_type == 'total':
            self.N_p = int(np.ceil(float(args.particles[1])))
            self.particles_pmps = self.N_p / (M * self.n_of_subvols)
            self.particle_density = self.N_p / geometry.volume
        elif self.particle_type == 'pv':
            self.particle_density = float(args.particles[1])
            self.N_p = int(np.ceil(self.particle_density * geometry.volume))
            self.particles_pmps = self.N_p / (M * (self.n_of_subvols - self.n_of_empty_subvols))
        else:
            raise Exception('Invalid --particles keyword.')

        self.dt = float(args.timestep[0])
        self.t = 0.0
        if geometry.subvol_type == 'slice':
            self.slice_axis = geometry.slice_axis
            self.slice_length = geometry.slice_length
        self.subvol_volume = geometry.subvol_volume
        self.bound_cond = geometry.bound_cond
        self.res_gen = args.reservoir_gen[0]
        if self.res_gen not in ('constant', 'fixed_rate', 'one_to_one'):
            raise Exception('Invalid --reservoir_gen')
        self.rough_facets = np.asarray(geometry.rough_facets)
        self.rough_facets_values = np.asarray(geometry.rough_facets_values)
        self.connected_facets = geometry.connected_facets
        self.T_distribution = args.temp_dist[0]
        self.temp_interp_type = args.temp_interp[0]
        if args.reference_temp[0] != 'local':
            self.T_reference = float(args.reference_temp[0])
            self.reference_occupation = phonon.calculate_occupation(self.T_reference, phonon.omega)
            self.ref_en_density = phonon.crystal_energy_function(self.T_reference)
        else:
            self.T_reference = 'local'
        self.n_mean = int(args.n_mean[0])
        self._geo, self._ph = geometry, phonon
        self.current_timestep = 0
        self._bal_steps = 0
        self.seed = int(getattr(args, 'seed', [0])[0])
        self.rng = np.random.default_rng(self.seed)

        # ---- device engine (created first: the specular-pair search of the set-up tables already runs on it)
        self.engine = engine if engine is not None else Engine(int(getattr(args, 'device', [0])[0]), self.seed)

        print('Calculating diffuse scattering probabilities...')
        self._build_rough_tables(geometry, phonon)

        print('Initialising reservoirs...')
        self.n_of_reservoirs = int((self.bound_cond == 'T').sum() + (self.bound_cond == 'F').sum())
        if self.n_of_reservoirs > 0:
            self.initialise_reservoirs(geometry, phonon)
        else:
            self.res_facet = np.zeros(0, dtype=int)
            self.res_facet_temperature = np.zeros(0)
            self.res_energy_balance = np.zeros(0)
            self.res_heat_flux = np.zeros((0, 3))
            self.N_leaving = np.zeros(0, dtype=int)

        print('Initialising population...')
        # every rank creates only its own share of the ensemble; ids and tiled modes use global indices
        self.N_total = self.N_p
        self.pid_lo, hi = self._shard(self.N_total)
        self.N_local = hi - self.pid_lo
        self.prng = np.random.default_rng([self.seed, 7919, self.rank])
        on_device = self._init_on_device(geometry, phonon)
        if on_device:                      # the common case is generated by the engine (nk_init_particles): nothing to upload
            self.subvol_temperature = self._assign_subvol_temperatures(geometry)
        else:
            pos, modes, occ = self.initialise_all_particles(geometry, phonon)

        self._configure_engine(geometry, phonon)
        if comm is not None:               # nk_comm_init decides what one rank needs (rank / nranks; NK_FORCE_COMM)
            self.engine.comm_init(comm[0], self.rank, self.nranks)
        J = phonon.number_of_branches
        if on_device:
            um = np.vstack(np.where(~phonon.inactive_modes_mask)).T
            self.unique_modes = um
            self.engine.init_particles(self.N_local, int(1.5 * self.N_local) + 65536, self.pid_lo,
                                       (um[:, 0] * J + um[:, 1]).astype(np.int32), self._subvol_shares(geometry))
        else:
            self.engine.reserve(int(1.5 * pos.shape[0]) + 65536)
            self.engine.upload(pos, (modes[:, 0] * J + modes[:, 1]).astype(np.int32), occ, pid_offset=self.pid_lo)
        print('Getting first boundary collisions...')
        self.engine.init_boundaries()

        print('Initialising local quantities...')
        if on_device:
            self._finish_initial_tallies(geometry, phonon, *self.engine.tally_state())
        else:
            self._initial_tallies(geometry, phonon, pos, modes, occ)
            del pos, modes, occ

        self.conv_crit = float(args.conv_crit[0])
        self.conv_count_min = int(args.conv_crit[1])
        self.initialise_residue(geometry)
        self.conv_rows = []
        self.f = None
        if self.rank == 0 and self.results_folder_name:
            print('Creating convergence file...')
            self.open_convergence(geometry)
        self._record_convergence(geometry)
        self.view = _Stats(self)
        print('Initialisation done!')

    # ----------------------------------------------------------------------------------- setup
    def _shard(self, n):
        """Initial particles are split evenly by index over the ranks."""
        return shard_range(n, self.rank, self.nranks)

    def _build_rough_tables(self, geometry, phonon):
        self.scat_model = self.args.bound_scat[0]
        Q, J = phonon.omega.shape
        Fr = self.rough_facets.shape[0]
        if Fr == 0:
            print('No rough facets to calculate.')
            self.specularity = np.zeros((0, Q, J))
            self.true_specular = np.zeros((0, Q, J), dtype=bool)
            self.correspondent_modes = np.zeros((0, 7))
            self.spec_map = np.zeros((0, Q, J), dtype=np.int64)
            self.creation_roulette = np.zeros((0, Q * J))
            self.degeneracies, self.degen_index = ST.find_degeneracies(phonon)
            return
        self.k_model = self.scat_model in ('k', 'wavevector', 'wave_vector')
        self._rough_on_device = False
        if not self.k_model and self.scat_model not in ('v', 'vel', 'velocity', 'groupvel', 'group_vel'):
            raise Exception('Invalid --bound_scat')
        self.degeneracies, self.degen_index = ST.find_degeneracies(phonon)
        if hasattr(self.engine, 'rough_begin') and (not self.k_model or hasattr(self.engine, 'kspec_begin')):
            # with a device engine the tables of both reflection models are built in HBM and stay there;
            # specularity / true_specular / spec_map / creation_roulette: see rough_tables()
            self._upload_material_and_mesh(geometry, phonon)
            # the pairs only come back (and are sorted into the reference's order) when their file is going to be written
            want = bool(self.rank == 0 and self.results_folder_name)
            if self.k_model:
                self._corr = ST.rough_tables_device_k(self.engine, geometry, phonon, self.rough_facets, self.rough_facets_values,
                                                      self.degeneracies, self.degen_index, want_rows=want)
            else:
                self._corr = ST.rough_tables_device(self.engine, geometry, phonon, self.rough_facets, self.rough_facets_values,
                                                    want_rows=want)
            self._corr_src = (geometry, phonon)
            self._rough_on_device = True
            self.specularity = self.true_specular = self.spec_map = self.creation_roulette = None
        else:
            spec0 = ST.fbz_specularity(geometry, phonon, self.rough_facets, self.rough_facets_values)
            if self.k_model:
                self.correspondent_modes, self.true_specular = ST.specular_correspondences_k(geometry, phonon, self.rough_facets)
            else:
                self.correspondent_modes, self.true_specular = ST.specular_correspondences_velocity(
                    geometry, phonon, self.rough_facets,
                    engine=(self.engine if hasattr(self.engine, 'specular_begin') else None))   # pair search on the GPU
            self.specularity = self.true_specular.astype(int) * spec0                  # Population.py:1459
            self.spec_map = ST.specular_map(self.correspondent_modes, geometry, self.rough_facets, Q, J)
            self.creation_rate, self.creation_roulette = ST.diffuse_roulette(
                geometry, phonon, self.rough_facets, self.specularity, self.correspondent_modes, self.scat_model, self.degeneracies)
        if self.rank == 0 and self.results_folder_name:
            np.savetxt(os.path.join(self.results_folder_name, 'specular_correspondences.txt'), self.correspondent_modes,
                       fmt='%.3f %.3f %.3f %d %d %d %d')                            # Population.py:1461

    @property
    def correspondent_modes(self):
        """(K,7) n(3) q_in j_in q_out j_out (Population.py:1241-1454); with device-built tables fetched on first use."""
        if self._corr is None and getattr(self, '_rough_on_device', False):
            fn = ST.specular_rows_device_k if self.k_model else ST.specular_rows_device
            self._corr = fn(self.engine, self._corr_src[0], self._corr_src[1], self.rough_facets)
        return self._corr

    @correspondent_modes.setter
    def correspondent_modes(self, rows):
        self._corr = rows

    def _upload_material_and_mesh(self, geometry, phonon):
        """nk_set_material + nk_set_mesh, once (the device builders of the set-up tables need them before the rest)."""
        if not getattr(self, '_tables_uploaded', False):
            self.engine.set_material(phonon.tables() if hasattr(phonon, 'tables') else _phonon_tables(phonon))
            self.engine.set_mesh(geometry.tables() if hasattr(geometry, 'tables') else _geometry_tables(geometry))
            self._tables_uploaded = True

    def rough_tables(self):
        """(specularity, true_specular, spec_map, creation_roulette) as (Fr, Q*J) arrays: copied back from the device when
        they were built there."""
        if getattr(self, '_rough_on_device', False):
            sp, ts, sm, ro = self.engine.rough_download()
            return sp, ts.astype(bool), sm, ro
        Q, J = self._ph.omega.shape
        return (self.specularity.reshape(-1, Q * J), self.true_specular.reshape(-1, Q * J), self.spec_map.reshape(-1, Q * J),
                self.creation_roulette)

    def initialise_reservoirs(self, geometry, phonon):
        """Population.py:323-354."""
        self.res_facet = np.asarray(geometry.res_facets)
        self.res_bound_values = np.asarray(geometry.res_values, dtype=float)
        self.res_bound_cond = np.asarray(geometry.res_bound_cond)
        mask_temp = self.res_bound_cond == 'T'
        mask_flux = self.res_bound_cond == 'F'
        self.res_facet_temperature = np.full(self.n_of_reservoirs, np.nan)
        self.res_facet_temperature[mask_temp] = self.res_bound_values[mask_temp]
        if mask_flux.any():
            self.res_facet_temperature[mask_flux] = self.res_bound_values[mask_temp].mean()
        if hasattr(self.engine, 'build_enter_prob'):                # enter_probability (Population.py:146-161) on the device
            self._upload_material_and_mesh(geometry, phonon)
            thick = phonon.number_of_active_modes / (self.particle_density * geometry.facets_area[self.res_facet])
            Q, J = phonon.omega.shape
            self.enter_prob = self.engine.build_enter_prob(-geometry.facets_normal[self.res_facet, :], thick, self.dt).reshape(-1, Q, J)
        else:
            self.enter_prob = ST.enter_probability(geometry, phonon, self.res_facet, self.particle_density, self.dt)
        self.res_counter = self.rng.random(self.enter_prob.shape)                   # Population.py:343
        self.N_leaving = np.sum(self.enter_prob, axis=(1, 2)).round().astype(int)
        self.res_energy_balance = np.zeros(self.n_of_reservoirs)
        self.res_heat_flux = np.zeros((self.n_of_reservoirs, 3))

    def _init_on_device(self, geometry, phonon):
        """Can the engine create the particles itself (nk_init_particles)?  Tiled modes (Population.py:127-144 with at least
        one particle per mode and subvolume) and uniform positions in the solid or per subvolume; NK_HOST_INIT=1 keeps the
        host path (initialise_all_particles + upload), which every other case takes."""
        if os.environ.get('NK_HOST_INIT') or os.environ.get('NK_NO_PARTITION') or not hasattr(self.engine, 'init_particles'):
            return False
        if self.args.part_dist[0] not in ('random_domain', 'random_subvol') or self.particles_pmps < 1:
            return False
        if self.args.part_dist[0] == 'random_subvol' and self.n_of_subvols > 256:
            # the device samples a subvolume's points by rejection from the whole solid: about S draws per particle, and it
            # refuses the ensemble when a particle finds its subvolume in none of 4096 -- many small subvolumes take the host path
            return False
        return getattr(geometry.mesh, 'n_of_simplices', 0) > 0 and hasattr(geometry.mesh, 'simplices_points')

    def _subvol_shares(self, geometry):
        """'random_subvol' (Population.py:222-246): the first particle id of every subvolume's share of the whole ensemble,
        ceil(N vol_i / vol) particles each, cut off at N (a rank creates the ids of its shard); None for 'random_domain'."""
        if self.args.part_dist[0] != 'random_subvol':
            return None
        vol = np.asarray(geometry.subvol_volume, dtype=float)
        n = np.ceil(self.N_total * vol / (vol.sum() - vol[self.empty_subvols].sum())).astype(np.int64)
        n[self.empty_subvols] = 0
        return np.minimum(np.concatenate(([0], np.cumsum(n))), self.N_total).astype(np.int64)

    def initialise_modes(self, phonon):
        """Population.py:127-144: tiled unique modes when there is at least one particle per mode and subvolume."""
        self.unique_modes = np.vstack(np.where(~phonon.inactive_modes_mask)).T
        if self.particles_pmps >= 1:
            idx = (self.pid_lo + np.arange(self.N_local)) % self.unique_modes.shape[0]
            modes = self.unique_modes[idx, :]
        else:
            modes = self.unique_modes[self.prng.integers(0, phonon.number_of_active_modes, size=self.N_local), :]
        return modes.astype(int)

    def _sample_volume(self, mesh, n):
        """Uniform points of the volume.  This package's Mesh takes the generator (reproducible per rank); the
        reference's `Mesh.sample_volume(self, n)` (Mesh.py:890) draws from NumPy's global state."""
        try:
            takes_rng = len(inspect.signature(mesh.sample_volume).parameters) >= 2
        except (TypeError, ValueError):
            takes_rng = False
        return np.asarray(mesh.sample_volume(n, self.prng) if takes_rng else mesh.sample_volume(n), dtype=float)

    def initialise_all_particles(self, geometry, phonon):
        """Positions, modes, temperatures, occupations (Population.py:186-321)."""
        key = self.args.part_dist[0]
        S = self.n_of_subvols
        NL = self.N_local
        if key == 'random_domain':
            pos = self._sample_volume(geometry.mesh, NL)
        elif key == 'center_domain':
            pos = np.ones((NL, 3)) * geometry.mesh.center_mass
        elif key == 'random_subvol':
            vol = np.asarray(geometry.subvol_volume, dtype=float)
            n = NL * vol / (vol.sum() - vol[self.empty_subvols].sum())
            n = np.ceil(n).astype(int)
            n[self.empty_subvols] = 0
            chunks = [[] for _ in range(S)]
            have = np.zeros(S, dtype=int)
            while np.any(have < n):
                batch = int(min(max((n - have).sum() * 1.2, 1e4), 4e6))
                x_new = self._sample_volume(geometry.mesh, batch)
                sv = geometry.subvol_classifier.predict(x_new) if S > 1 else np.zeros(batch, dtype=int)
                order = np.argsort(sv, kind='stable')
                counts = np.bincount(sv, minlength=S)
                start = np.concatenate(([0], np.cumsum(counts)))
                for i in range(S):
                    need = n[i] - have[i]
                    if need > 0 and counts[i] > 0:
                        take = order[start[i]:start[i] + min(need, counts[i])]
                        chunks[i].append(x_new[take])
                        have[i] += take.shape[0]
            pos = np.vstack([np.vstack(c) if c else np.zeros((0, 3)) for c in chunks])[:NL, :]
        elif key == 'center_subvol':
            # Population.py:248-267 puts every subvolume's share at the centre of mass of `geometry.subvol_meshes[i]`, an
            # attribute the reference never creates (the option raises AttributeError there).  Here: at the subvolume's
            # centre, the point the classifier is built from.
            vol = np.asarray(geometry.subvol_volume, dtype=float)
            filled = vol.sum() - vol[self.empty_subvols].sum()
            chunks, counter = [], 0
            for i in range(S):
                if i in self.empty_subvols:
                    continue
                k = min(int(np.ceil(NL * vol[i] / filled)), NL - counter)
                counter += k
                chunks.append(np.ones((k, 3)) * np.asarray(geometry.subvol_center)[i])
            pos = np.vstack(chunks)[:NL, :]
        else:
            # resume file, Population.py:284-306; a run on several ranks leaves one file per rank beside it
            # (write_final_state), and every rank reads them all before taking its share
            files = resume_files(key)
            data = np.vstack([np.loadtxt(f, delimiter=',', comments='#', dtype=float, ndmin=2) for f in files])
            modes = data[:, [0, 1]].astype(int)
            pos = data[:, [2, 3, 4]].copy()
            occ = data[:, 5].copy()
            # subvolume temperatures: from --temp_dist, then the fixed point of refresh_temperatures on the loaded
            # occupations (:296-303), on the WHOLE ensemble -- the same on every rank
            sv = geometry.subvol_classifier.predict(pos) if S > 1 else np.zeros(pos.shape[0], dtype=int)
            self.subvol_temperature = self._resume_temperatures(phonon, sv, modes, occ, self._assign_subvol_temperatures(geometry))
            lo, hi = self._shard(pos.shape[0])
            self.N_total, self.pid_lo, self.N_local = pos.shape[0], lo, hi - lo
            modes, pos, occ = modes[lo:hi], pos[lo:hi], occ[lo:hi]
            self.subvol_id = sv[lo:hi]
            return pos, modes, occ
        modes = self.initialise_modes(phonon)
        self.subvol_id = geometry.subvol_classifier.predict(pos) if S > 1 else np.zeros(pos.shape[0], dtype=int)
        self.subvol_temperature = self._assign_subvol_temperatures(geometry)
        T = self.subvol_temperature[self.subvol_id]
        occ = phonon.calculate_occupation(T, phonon.omega[modes[:, 0], modes[:, 1]])  # Population.py:280
        return pos, modes, occ

    def _resume_temperatures(self, phonon, sv, modes, occ, T):
        """Population.py:296-303: iterate refresh_temperatures (calculate_energy :704-728 with the current subvolume
        temperatures as local reference, then T = temperature_function(E)) until no subvolume moves by more than 1e-6."""
        S = self.n_of_subvols
        om = phonon.omega[modes[:, 0], modes[:, 1]]
        N_sv = np.bincount(sv, minlength=S).astype(float)
        T = np.array(T, dtype=float)
        old_T = np.zeros(S)
        err, it = 1.0, 0
        while err > 1e-6 and it < 1000:
            if self.T_reference == 'local':
                dn = occ - phonon.calculate_occupation(T[sv], om)
                ref = phonon.crystal_energy_function(T)
            else:
                dn = occ - self.reference_occupation[modes[:, 0], modes[:, 1]]
                ref = self.ref_en_density
            E_raw = np.bincount(sv, weights=self.hbar * om * dn, minlength=S)
            T = np.asarray(phonon.temperature_function(self._normalise_energy(phonon, E_raw, N_sv) + ref), dtype=float)
            with np.errstate(divide='ignore', invalid='ignore'):
                err = np.nanmax(np.absolute((T - old_T) / T))
            old_T = T.copy()
            it += 1
        return T

    def _assign_subvol_temperatures(self, geometry):
        """assign_temperatures (Population.py:565-655), subvolume part."""
        key = self.T_distribution
        S = self.n_of_subvols
        if key == 'custom':
            return np.array(self.args.subvol_temp, dtype=float)
        if self.n_of_reservoirs > 0:
            bound_T = self.res_bound_values[self.res_bound_cond == 'T']
        else:
            bound_T = np.zeros(0)
        if len(bound_T) == 0:
            bound_T = np.array([float(self.T_reference)])
        if key == 'cold':
            return np.ones(S) * bound_T.min()
        if key == 'hot':
            return np.ones(S) * bound_T.max()
        if key == 'mean':
            return np.ones(S) * bound_T.mean()
        if key == 'random':
            return self.rng.random(S) * (bound_T.max() - bound_T.min()) + bound_T.min()     # same stream on every rank
        if key == 'linear':
            fi = self.res_facet[self.res_bound_cond == 'T']
            bp = geometry.facet_centroid[fi, :]
            if len(bound_T) == 1:
                return np.ones(S) * bound_T
            if len(bound_T) == 2:
                d = bp[1] - bp[0]
                alpha = ((geometry.subvol_center - bp[0]) * d).sum(axis=1) / (d ** 2).sum()
                return bound_T[0] + alpha * (bound_T[1] - bound_T[0])
            dd = np.sum((geometry.subvol_center - bp[:, None, :]) ** 2, axis=2).T ** 0.5
            w = 1 / dd
            w /= w.sum(axis=1, keepdims=True)
            return np.sum(bound_T * w, axis=1)
        raise Exception('Invalid --temp_dist')

    def _configure_engine(self, geometry, phonon):
        eng = self.engine
        self._upload_material_and_mesh(geometry, phonon)
        if geometry.subvol_type == 'slice' and self.temp_interp_type in ('nearest', 'linear'):
            kind, axis, interp = 0, geometry.slice_axis, (1 if self.temp_interp_type == 'linear' else 0)
        elif self.temp_interp_type == 'nearest':
            kind, axis, interp = 1, 0, 2
        elif self.temp_interp_type in ('radial', 'linear'):
            # non-slice subvolumes: cubic radial basis functions (Population.py:573-590; 'linear' falls back to them
            # there too), over the coordinates a grid actually resolves (:651-656)
            if self.temp_interp_type == 'linear':
                print('Linear T interpolation is currently valid for slice subvolumes only. Defaulting to RBF interpolation to avoid extrapolation problems.')
            kind, axis, interp = 1, 0, 3
            used = (np.asarray(geometry.grid) != 1) if (geometry.subvol_type == 'grid' and np.any(np.asarray(geometry.grid) == 1)) else (True, True, True)
            rbf = ST.rbf_system(geometry.subvol_center, used)
        else:
            raise Exception('Invalid T interpolator type.')
        eng.set_subvolumes(geometry.subvol_center, geometry.subvol_volume, kind, axis, interp, self.subvol_temperature,
                           rbf=(rbf if interp == 3 else None))
        Q, J = phonon.omega.shape
        if self.n_of_reservoirs > 0:
            eng.set_reservoirs(self.res_facet, self.res_facet_temperature, self.enter_prob.reshape(-1, Q * J),
                               self.res_counter.reshape(-1, Q * J),
                               gen={'constant': 0, 'fixed_rate': 1, 'one_to_one': 2}[self.res_gen],
                               n_leaving=(self.N_leaving if self.res_gen == 'one_to_one' else None))
        if self.rough_facets.shape[0] > 0 and not getattr(self, '_rough_on_device', False):
            degen_j2 = None
            if getattr(self, 'k_model', False):
                # 'k' model: a specular out-mode with a degenerate partner lands on the pair's second branch with
                # probability 1/2 (Population.py:963-969)
                degen_j2 = -np.ones((Q, J), dtype=np.int32)
                di = self.degen_index.astype(int)
                has = di > -1
                degen_j2[has] = self.degeneracies[di[has], 2]
            eng.set_rough(self.rough_facets, self.specularity.reshape(-1, Q * J), self.true_specular.reshape(-1, Q * J),
                          self.spec_map.reshape(-1, Q * J), self.creation_roulette, degen_j2=degen_j2)
        eng.set_params(dt=self.dt, norm_fixed=(self.norm == 'fixed'), particle_density=self.particle_density,
                       T_ref=(None if self.T_reference == 'local' else self.T_reference),
                       flux_every=self.n_dt_to_conv, contains_every=100)

    def _initial_tallies(self, geometry, phonon, pos, modes, occ):
        """calculate_energy / calculate_heat_flux / calculate_kappa on the initial state (Population.py:282, :318-321)."""
        S = self.n_of_subvols
        om = phonon.omega[modes[:, 0], modes[:, 1]]
        if self.T_reference == 'local':
            dn = occ - phonon.calculate_occupation(self.subvol_temperature[self.subvol_id], om)
        else:
            dn = occ - self.reference_occupation[modes[:, 0], modes[:, 1]]
        e = self.hbar * om * dn
        E_raw = np.bincount(self.subvol_id, weights=e, minlength=S)
        v = phonon.group_vel[modes[:, 0], modes[:, 1], :]
        flux_raw = np.stack([np.bincount(self.subvol_id, weights=v[:, d] * e, minlength=S) for d in range(3)], axis=1)
        N_sv = np.bincount(self.subvol_id, minlength=S).astype(float)
        del self.subvol_id
        self._finish_initial_tallies(geometry, phonon, E_raw, N_sv, flux_raw)

    def _finish_initial_tallies(self, geometry, phonon, E_raw, N_sv, flux_raw):
        """The t = 0 row from this rank's sums (the host's, or the engine's tally of the particles it created, nk_tally_state).
        With several ranks the sums are all-reduced over the communicator -- a rank's shard is a run of consecutive ids, which
        under 'random_subvol' covers only some of the subvolumes; only a run without communicator (NK_COMM_DRYRUN, a test
        hook) extrapolates from its own share."""
        w = 1
        if self.nranks > 1:
            info = self.engine.comm_info() if hasattr(self.engine, 'comm_info') else {'comm_nranks': 0}
            if info['comm_nranks'] == self.nranks:
                E_raw, N_sv, flux_raw = self.engine.comm_allreduce(E_raw, N_sv, flux_raw)
            else:
                w = self.nranks
        self.subvol_N_p = np.rint(N_sv).astype(np.int64) * w
        self.N_p = int(self.subvol_N_p.sum())
        ref = phonon.crystal_energy_function(self.subvol_temperature) if self.T_reference == 'local' else self.ref_en_density
        self.total_energy = float(E_raw.sum()) * w
        with np.errstate(invalid='ignore', divide='ignore'):
            self.subvol_energy = self._normalise_energy(phonon, E_raw * w, self.subvol_N_p) + ref
            self.subvol_heat_flux = self._normalise_flux(phonon, flux_raw * w, self.subvol_N_p)
        # subvolumes without particles (empty_subvols; a dry-run shard that does not reach them): nothing to say about those
        empty = self.subvol_N_p == 0
        self.subvol_energy = np.where(empty, ref, self.subvol_energy)
        self.subvol_heat_flux = np.where(empty[:, None], 0.0, self.subvol_heat_flux)
        self.calculate_kappa(geometry)

    # ----------------------------------------------------------------------------- normalisation
    def _norm(self, phonon, N_sv):
        if self.norm == 'fixed':
            return phonon.number_of_active_modes / (self.particle_density * np.asarray(self.subvol_volume, dtype=float))
        with np.errstate(divide='ignore', invalid='ignore'):
            n = phonon.number_of_active_modes / np.asarray(N_sv, dtype=float)
        return np.where(np.isnan(n), 0, n)

    def _normalise_energy(self, phonon, E_raw, N_sv):
        return phonon.normalise_to_density(E_raw * self._norm(phonon, N_sv))       # Population.py:719-726

    def _normalise_flux(self, phonon, flux_raw, N_sv):
        n = self._norm(phonon, N_sv).reshape(-1, 1)                                 # Population.py:738-747
        with np.errstate(invalid='ignore'):
            return phonon.normalise_to_density(flux_raw * n) * self.eVpsa2_in_Wm2

    def calculate_kappa(self, geometry):
        """Population.py:749-788."""
        if geometry.subvol_type == 'slice':
            S = self.n_of_subvols
            T = np.zeros(S + 2)
            T[1:-1] = self.subvol_temperature
            if self.n_of_reservoirs >= 2:
                T[[0, -1]] = self.res_facet_temperature[[0, -1]] if self.n_of_reservoirs > 2 else self.res_facet_temperature
            else:
                T[[0, -1]] = np.nan
            phi = self.subvol_heat_flux[:, geometry.slice_axis]
            ext = geometry.bounds[1, geometry.slice_axis] - geometry.bounds[0, geometry.slice_axis]
            dx = 2 * ext * self.a_in_m / S
            dT = T[2:] - T[:-2]
            DX = ext * self.a_in_m * (1 + S) / S
            DT = T[-1] - T[0]
            with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
                self.subvol_kappa = -phi * dx / dT
                self.kappa = -np.sum(phi * self.subvol_N_p) * (DX / DT) / self.N_p
            self.subvol_kappa[np.absolute(self.subvol_kappa) == np.inf] = 0
        else:
            i, j = geometry.subvol_connections[:, 0], geometry.subvol_connections[:, 1]
            dx = geometry.subvol_center[j, :] - geometry.subvol_center[i, :]
            nrm = np.linalg.norm(dx, axis=1, keepdims=True)
            dT = self.subvol_temperature[j] - self.subvol_temperature[i]
            phi = (self.subvol_heat_flux[i, :] + self.subvol_heat_flux[j, :]) / 2
            with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
                self.svcon_kappa = np.where(dT == 0, 0, -np.sum(phi * dx / nrm, axis=1) * nrm[:, 0] * self.a_in_m / dT)

    def adjust_reservoir_balance(self, geometry, phonon):
        """Population.py:1685-1693."""
        if self.n_of_reservoirs > 0:
            A = geometry.facets_area[self.res_facet].reshape(-1, 1)
            # the steps the sums cover: n_dt_to_conv (:1688), fewer only in the first window after the engine was
            # stepped behind this object's back (_sync_clock)
            nw = self._bal_steps if getattr(self, '_bal_steps', 0) > 0 else self.n_dt_to_conv
            c = phonon.number_of_active_modes / (self.particle_density * self.dt * nw)
            self.res_heat_flux = phonon.normalise_to_density(self.res_heat_flux * c / A) * self.eVpsa2_in_Wm2
            self.res_energy_balance = phonon.normalise_to_density(self.res_energy_balance * c)

    def restart_reservoir_balance(self):
        self.res_heat_flux = np.zeros((self.n_of_reservoirs, 3))                    # Population.py:1695-1699
        self.res_energy_balance = np.zeros(self.n_of_reservoirs)
        self._bal_steps = 0

    def _sync_clock(self):
        """ONE step counter: the engine's.  The library tallies the heat flux on its own absolute step ((step + 1) % flux_every,
        nk_step) and runs contains_check on it; the convergence rows (Population.py:1762-1767, n_dt_to_conv :41) and the
        100-step bookkeeping (:1729-1741) are keyed on current_timestep.  A caller that also steps the engine directly
        (nk_step through the C ABI, bench.py's timed regions) moves the former only; before every run the latter follows, so
        that every convergence row lands on a step whose flux was tallied.  The reservoir sums of the steps taken behind
        this object's back were not collected: the window in progress restarts here and is normalised by the steps it
        really covers (adjust_reservoir_balance)."""
        get = getattr(self.engine, 'get_step', None)
        if get is None:
            return
        es = int(get())
        if es != self.current_timestep:
            self.current_timestep = es
            self.t = es * self.dt
            self.restart_reservoir_balance()

    # ------------------------------------------------------------------------------- time loop
    def run_timestep(self, geometry, phonon):
        """One timestep, Population.py:1724-1769."""
        self.run(1, geometry, phonon)

    def run(self, nsteps, geometry=None, phonon=None):
        """Advance nsteps timesteps.  Library calls are cut at the 100-step bookkeeping boundaries
        (Population.py:1729-1741) so that outputs are identical to stepping one by one."""
        geometry = geometry if geometry is not None else self._geo
        phonon = phonon if phonon is not None else self._ph
        self._geo, self._ph = geometry, phonon
        self._sync_clock()
        done = 0
        while done < nsteps:
            if self.current_timestep == 0:
                print('Simulating...')
            if (self.current_timestep % 100) == 0:
                self._every_hundred(geometry)
            chunk = min(nsteps - done, 100 - (self.current_timestep % 100))
            # the reference grows its arrays as the ensemble grows; here the particle store is re-laid out with head room
            # before it can fill up (this rank's share of N_p against the engine's slots)
            # (the engine's slot count only changes when the store grows: asked for again after every 100 steps and after a reserve)
            local = self.N_p / max(self.nranks, 1)
            if getattr(self, '_slots', None) is None or (self.current_timestep % 100) == 0:
                self._slots = self.engine.timing()['slots']
            if self._slots > 0 and local > 0.8 * self._slots:
                self.engine.reserve(int(2.0 * local) + 65536)
                self._slots = None
            t = self.engine.step(chunk)
            s0 = 0
            while s0 < chunk:
                # up to the next convergence step in one go: only the reservoir tallies accumulate in between
                to_conv = self.n_dt_to_conv - (self.current_timestep % self.n_dt_to_conv)
                s1 = min(chunk, s0 + to_conv)
                s = s1 - 1
                self.current_timestep += s1 - s0
                self.t = self.current_timestep * self.dt
                self.subvol_temperature = t['T_sv'][s]
                self.subvol_energy = t['E_sv'][s]
                self.subvol_N_p = t['N_sv'][s].astype(np.int64)
                self.N_p = int(self.subvol_N_p.sum())
                self.total_energy = float(t['E_raw'][s].sum())
                if self.n_of_reservoirs > 0:
                    self.N_leaving = t['N_leaving'][s].astype(np.int64)
                    for q in range(s0, s1):                     # same summation order as stepping one by one
                        self.res_energy_balance = self.res_energy_balance + t['res_energy'][q]
                        self.res_heat_flux = self.res_heat_flux + t['res_flux'][q]
                    self._bal_steps += s1 - s0
                if (self.current_timestep % self.n_dt_to_conv) == 0:                # Population.py:1762-1767
                    self.subvol_heat_flux = self._normalise_flux(phonon, t['flux_raw'][s], self.subvol_N_p)
                    self.calculate_kappa(geometry)
                    self.adjust_reservoir_balance(geometry, phonon)
                    self._record_convergence(geometry)
                    self.restart_reservoir_balance()
                s0 = s1
            done += chunk

    def _every_hundred(self, geometry):
        if self.results_folder_name and getattr(self.args, 'checkpoint', True):     # every rank: its shard of the particles
            self.write_final_state(geometry)
        self.view.postprocess(verbose=False)
        self.update_residue(geometry)
        info = 'Timestep {:>5d} - max residue: {:>9.3e} ({:<9s}) ['.format(int(self.current_timestep), self.max_residue, self.max_residue_qt)
        for sv in range(self.n_of_subvols):
            info += ' {:>7.3f}'.format(self.subvol_temperature[sv])
        print(info + ' ]')

    # ----------------------------------------------------------------------- convergence / residue
    def initialise_residue(self, geo):
        """Population.py:1771-1795."""
        S, R = self.n_of_subvols, self.n_of_reservoirs
        if geo.subvol_type == 'slice':
            n = 3 * S + R
            ax = ['x', 'y', 'z'][self.slice_axis]
            self.residue_qts = (['T_{:d}'.format(i) for i in range(S)] + ['phi_{:s}_{:d}'.format(ax, j) for j in range(S)] +
                                ['en_res_{:d}'.format(i) for i in range(R)] + ['k_{:d}'.format(i) for i in range(S)])
        else:
            n = 4 * S + R + geo.n_of_subvol_con
            self.residue_qts = (['T_{:d}'.format(i) for i in range(S)] +
                                ['phi_{:s}_{:d}'.format(i, j) for j in range(S) for i in ['x', 'y', 'z']] +
                                ['en_res_{:d}'.format(i) for i in range(R)] + ['k_{:d}'.format(i) for i in range(geo.n_of_subvol_con)])
        self.old_mean_large = np.ones(n)
        self.old_std_large = np.ones(n)
        self.conv_count = 0
        self.finish_sim = False
        self.max_residue = 1
        self.max_residue_qt = 'none'

    def update_residue(self, geo):
        """Population.py:1797-1839."""
        v = self.view
        S = self.n_of_subvols
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            if geo.subvol_type == 'slice':
                sel = 3 * np.arange(S) + self.slice_axis
                new_mean = np.concatenate((v.mean_T, v.mean_sv_phi[sel], v.mean_en_res, v.mean_sv_k))
                new_std = np.concatenate((v.std_T, v.std_sv_phi[sel], v.std_en_res, v.std_sv_k))
            else:
                new_mean = np.concatenate((v.mean_T, v.mean_sv_phi, v.mean_en_res, v.mean_con_k))
                new_std = np.concatenate((v.std_T, v.std_sv_phi, v.std_en_res, v.std_con_k))
            residue_mean = np.absolute((new_mean - self.old_mean_large) / self.old_mean_large)
        self.residue_all = np.where(new_std > np.absolute(new_mean), 0, residue_mean)
        self.max_residue = np.nanmax(self.residue_all)
        index = np.nonzero(self.residue_all == self.max_residue)[0][0]
        self.max_residue_qt = self.residue_qts[index]
        self.conv_count = self.conv_count + 1 if self.max_residue < self.conv_crit else 0
        if self.conv_count >= self.conv_count_min:
            self.finish_sim = True
        self.old_mean_large, self.old_std_large = new_mean, new_std
        if self.rank == 0 and self.results_folder_name:
            with open(os.path.join(self.results_folder_name, 'residue.txt'), 'a+') as f:
                f.writelines(''.join('{:9.3e} '.format(i) for i in self.residue_all) + '\n')

    def open_convergence(self, geometry):
        """Header of convergence.txt, column layout of Population.py:1991-2022."""
        S, R = self.n_of_subvols, self.n_of_reservoirs
        line = '# ' + 'Real Time                  ' + 'Timest. ' + 'Simul. Time ' + 'Total Energy '
        for i in range(R):
            line += 'En Bal Res {} '.format(i)
        for i in range(R):
            line += ' Hflux x Res {} '.format(i) + ' Hflux y Res {} '.format(i) + ' Hflux z Res {} '.format(i)
        line += ' No. Part. '
        line += ''.join(' T Sv {:>3d} '.format(i) for i in range(S))
        line += ''.join(' Energ Sv {:>2d} '.format(i) for i in range(S))
        for i in range(S):
            line += ' Hflux x Sv {:>2d} '.format(i) + ' Hflux y Sv {:>2d} '.format(i) + ' Hflux z Sv {:>2d} '.format(i)
        line += ''.join(' Np Sv {:>3d} '.format(i) for i in range(S))
        if geometry.subvol_type == 'slice':
            line += ''.join(' Kappa Sv {:>2d} '.format(i) for i in range(S)) + ' Kappa total  '
        else:
            line += ''.join(' K Con {:>3d}-{:>3d} '.format(a, b) for a, b in geometry.subvol_connections)
        self.f = open(os.path.join(self.results_folder_name, 'convergence.txt'), 'a+')
        self.f.write(line + '\n')
        self.f.close()

    def _record_convergence(self, geometry):
        row = dict(step=self.current_timestep, T=np.array(self.subvol_temperature), phi=np.array(self.subvol_heat_flux),
                   en_res=np.array(self.res_energy_balance), N_p=self.N_p, sv_Np=np.array(self.subvol_N_p),
                   kappa=getattr(self, 'kappa', np.nan), sv_k=np.array(getattr(self, 'subvol_kappa', np.zeros(0))),
                   con_k=np.array(getattr(self, 'svcon_kappa', np.zeros(0))))
        self.conv_rows.append(row)
        if len(self.conv_rows) > max(self.n_mean, 1000):
            del self.conv_rows[:-max(self.n_mean, 1000)]
        if self.rank == 0 and self.results_folder_name:
            self.write_convergence(geometry)

    def write_convergence(self, geometry):
        """One row of convergence.txt, formats of Population.py:2027-2069."""
        def arr(a, fmt):
            return ' '.join(fmt.format(x) for x in np.ravel(a))
        line = datetime.now().strftime('%Y-%m-%dT%H:%M:%S.%f ')
        line += '{:>8d} '.format(int(self.current_timestep))
        line += '{:>12.5e} '.format(self.t)
        line += '{:>12.5e} '.format(self.total_energy)
        if self.n_of_reservoirs > 0:
            line += arr(self.res_energy_balance, '{:>12.5e}') + ' '
            for i in range(self.n_of_reservoirs):
                line += arr(self.res_heat_flux[i, :], '{:>14.6e}') + ' '
        line += '{:>10d} '.format(self.N_p)
        line += arr(self.subvol_temperature, '{:>9.3f}') + ' '
        line += arr(self.subvol_energy, '{:>12.5e}') + ' '
        for i in range(self.n_of_subvols):
            line += arr(self.subvol_heat_flux[i, :], '{:>14.6e}') + ' '
        line += arr(np.asarray(self.subvol_N_p).astype(int), '{:>10d}') + ' '
        if geometry.subvol_type == 'slice':
            line += arr(self.subvol_kappa, '{:>12.5e}') + ' '
            line += '{:>13.6e} '.format(self.kappa)
        else:
            line += arr(self.svcon_kappa, '{:>14.7e}') + ' '
        self.f = open(os.path.join(self.results_folder_name, 'convergence.txt'), 'a+')
        self.f.writelines(line + '\n')
        self.f.close()

    # ------------------------------------------------------------------------------ particle data
    def particles(self):
        """Download the live particles (flushes the deferred relaxation): dict with positions, modes (N,2),
        occupation, n_timesteps, collision_facets, pid."""
        p = self.engine.download()
        J = self._J
        p['modes'] = np.stack((p['mode'] // J, p['mode'] % J), axis=1)
        p['collision_facets'] = p['facet']
        return p

    @property
    def _J(self):
        return self.engine.J

    def write_final_state(self, geometry):
        """particle_data.txt and subvolumes.txt, formats of Population.py:2071-2151."""
        time = datetime.now().strftime('%Y-%m-%dT%H:%M:%S.%f')
        p = self.particles()
        # the reference's header (Population.py:2071-2086); the per-rank files of a run on several ranks (this package's own
        # extension) carry one more line, the timestep and which shard this is: a resume checks that they belong together
        header = ('Particles final state data \n' + 'Date and time: {}\n'.format(time) +
                  'hdf file = {}, POSCAR file = {}\n'.format(self.args.hdf_file, self.args.poscar_file) +
                  ('timestep = {}, rank = {} of {}\n'.format(int(self.current_timestep), self.rank, self.nranks) if self.nranks > 1 else '') +
                  'q-point, branch, pos x [angs], pos y [angs], pos z [angs], occupation')
        data = np.hstack((p['modes'], p['positions'], p['occupation'].reshape(-1, 1)))
        name = 'particle_data.txt' if self.nranks == 1 else 'particle_data.rank%dof%d.txt' % (self.rank, self.nranks)
        # written under a temporary name and renamed: a crash never leaves half a checkpoint under the real name
        final = os.path.join(self.results_folder_name, name)
        tmp = final + '.tmp%d' % os.getpid()
        np.savetxt(tmp, data, '%d, %d, %.3f, %.3f, %.3f, %.6e', delimiter=',', header=header)
        os.replace(tmp, final)
        if self.rank != 0:
            return
        if self.current_timestep > 0 and geometry.subvol_type == 'slice' and hasattr(self.view, 'mean_T'):
            v = self.view
            S = self.n_of_subvols
            header = ('subvols final state data \n' + 'Date and time: {}\n'.format(time) +
                      'hdf file = {}, POSCAR file = {}\n'.format(self.args.hdf_file, self.args.poscar_file) +
                      'subvol id, subvol x, subvol y, subvol z, subvol volume, T [K], sigma T [K], HF x [W/m^2], HF y [W/m^2], '
                      'HF z [W/m^2], sigma HF x [W/m^2], sigma HF y [W/m^2], sigma HF z [W/m^2], kappa [W/m K], sigma kappa [W/m K]')
            data = np.hstack((np.arange(S).reshape(-1, 1), geometry.subvol_center, np.asarray(self.subvol_volume).reshape(-1, 1),
                              v.mean_T.reshape(-1, 1), v.std_T.reshape(-1, 1), v.mean_sv_phi.reshape(-1, 3),
                              v.std_sv_phi.reshape(-1, 3), v.mean_sv_k.reshape(-1, 1), v.std_sv_k.reshape(-1, 1)))
            np.savetxt(os.path.join(self.results_folder_name, 'subvolumes.txt'), data,
                       '%d, %.3e, %.3e, %.3e, %.3e, %.3f, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e',
                       delimiter=',', header=header)
        elif self.current_timestep > 0 and hasattr(self.view, 'mean_con_k'):
            # non-slice subvolumes: subvolumes.txt without kappa columns + subvol_connections.txt (Population.py:2117-2151)
            v = self.view
            S = self.n_of_subvols
            header = ('subvols final state data \n' + 'Date and time: {}\n'.format(time) +
                      'hdf file = {}, POSCAR file = {}\n'.format(self.args.hdf_file, self.args.poscar_file) +
                      'subvol id, subvol position, subvol volume, T [K], sigma T [K], HF x [W/m^2], HF y [W/m^2], HF z [W/m^2], '
                      'sigma HF x [W/m^2], sigma HF y [W/m^2], sigma HF z [W/m^2]')
            data = np.hstack((np.arange(S).reshape(-1, 1), geometry.subvol_center, np.asarray(self.subvol_volume).reshape(-1, 1),
                              v.mean_T.reshape(-1, 1), v.std_T.reshape(-1, 1), v.mean_sv_phi.reshape(-1, 3),
                              v.std_sv_phi.reshape(-1, 3)))
            np.savetxt(os.path.join(self.results_folder_name, 'subvolumes.txt'), data,
                       '%d, %.3e, %.3e, %.3e, %.3e, %.3f, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e, %.3e', delimiter=',', header=header)
            C_ = geometry.n_of_subvol_con
            header = ('connections final state data \n' + 'Date and time: {}\n'.format(time) +
                      'hdf file = {}, POSCAR file = {}\n'.format(self.args.hdf_file, self.args.poscar_file) +
                      'connection id, sv 1, sv 2, con dx, con dy, con dz, dT [K], sigma dT [K], HF [W/m^2], sigma HF [W/m^2], '
                      'kappa [W/m K], sigma kappa [W/m K]')
            data = np.hstack((np.arange(C_).reshape(-1, 1), geometry.subvol_connections, geometry.subvol_con_vectors,
                              v.mean_con_dT.reshape(-1, 1), v.std_con_dT.reshape(-1, 1), v.mean_con_phi.reshape(-1, 1),
                              v.std_con_phi.reshape(-1, 1), v.mean_con_k.reshape(-1, 1), v.std_con_k.reshape(-1, 1)))
            np.savetxt(os.path.join(self.results_folder_name, 'subvol_connections.txt'), data,
                       '%d, %d, %d, %.3e, %.3e, %.3e, %.3f, %.3e, %.3e, %.3e, %.3e, %.3e', delimiter=',', header=header)

    def save_plot_real_time(self):
        """Called by the reference driver (nanokappa.py:105) but defined nowhere there; a no-op here."""
        return None


def resume_files(key):
    """The particle files a resume (--part_dist <file>, Population.py:284-306) reads: EITHER the single file `key` OR the
    complete family `<stem>.rank<r>of<N><ext>` of ONE N that a run on N ranks left beside it -- never a mixture (a folder
    that holds both, rank files of two different N, a missing rank, or rank files written at different timesteps is an
    error: stacking them would duplicate or lose particles silently)."""
    import re
    stem, ext = os.path.splitext(key)
    fam = {}
    for f in glob.glob(glob.escape(stem) + '.rank*of*' + glob.escape(ext)):
        m = re.match(re.escape(stem) + r'\.rank(\d+)of(\d+)' + re.escape(ext) + '$', f)
        if m:
            fam.setdefault(int(m.group(2)), {})[int(m.group(1))] = f
    single = os.path.exists(key)
    if single and fam:
        raise Exception('Ambiguous particle data: %s exists beside per-rank files %s; keep one checkpoint only.'
                        % (key, sorted(f for d in fam.values() for f in d.values())))
    if single:
        return [key]
    if not fam:
        raise Exception('Wrong particle data file. Change the keyword or check whether the file exists.')
    if len(fam) > 1:
        raise Exception('Ambiguous particle data: per-rank files of runs with different rank counts (%s) beside %s.'
                        % (sorted(fam), key))
    (n, files), = fam.items()
    missing = [r for r in range(n) if r not in files]
    if missing:
        raise Exception('Incomplete particle data: rank file(s) %s of %d missing beside %s.' % (missing, n, key))
    steps = set()
    for r in range(n):
        with open(files[r]) as fh:
            head = [next(fh, '') for _ in range(6)]
        ts = [re.search(r'timestep = (\d+), rank = (\d+) of (\d+)', h) for h in head]
        ts = [t for t in ts if t]
        if ts:
            if (int(ts[0].group(2)), int(ts[0].group(3))) != (r, n):
                raise Exception('%s says it is rank %s of %s.' % (files[r], ts[0].group(2), ts[0].group(3)))
            steps.add(int(ts[0].group(1)))
        else:
            steps.add(None)                  # file of an older version: no timestep line
    if len(steps) > 1:
        raise Exception('Inconsistent particle data: the rank files beside %s were written at different timesteps (%s).'
                        % (key, sorted(str(t) for t in steps)))
    return [files[r] for r in range(n)]


def _phonon_tables(ph):
    """Tables from a REFERENCE Phonon object (attribute names of classes/Phonon.py)."""
    T_min, T_max = ph.temperature_array.min(), ph.temperature_array.max()
    T_array = np.arange(T_min, T_max + 0.1, 0.1)
    return dict(omega=ph.omega, group_vel=ph.group_vel, T_grid=ph.temperature_array, lifetime=ph.lifetime,
                T_array=T_array, energy_array=ph.energy_array, hbar=ph.hbar, kb=ph.kb,
                QV=ph.number_of_qpoints * ph.volume_unitcell, active_modes=ph.number_of_active_modes)


def _geometry_tables(geo):
    """Tables from a REFERENCE Geometry object (attribute names of classes/Geometry.py, classes/Mesh.py)."""
    m = geo.mesh
    return dict(vertices=m.vertices, faces=m.faces, face_normals=m.face_normals, face_k=m.face_k, face_bounds=m.face_bounds,
                face_basis_matrix=m.face_basis_matrix, face_origins=m.face_origins, face_facets=m.face_facets,
                face_areas=m.face_areas, facets_normal=m.facets_normal, facet_centroid=m.facet_centroid, facets=m.facets,
                bounds=m.bounds, simplices_points=m.simplices_points, simplices=m.simplices,
                simplices_volumes=m.simplices_volumes, tol=m.tol, bound_cond=geo.bound_cond,
                connected_facets=geo.connected_facets)
