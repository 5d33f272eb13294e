"""Material tables behind the hot path: the attribute/method surface of the reference's
`Phonon` class that `Population` consumes (SURVEY.md section 8b "Path -> Phonon").

Reference: classes/Phonon.py.  The reference loads a phono3py HDF5 + POSCAR and expands
the irreducible wedge with phonopy symmetry operations (Phonon.py:66-116).  h5py/phonopy are
not part of this path; this class is fed FBZ-expanded tables (a dict, an .npz, or the
synthetic generator) and rebuilds the derived tables with the reference's formulas.
"""
import os

import numpy as np

from .constants import Constants
from . import synthetic


def _searchsorted_left(a, x):
    return np.searchsorted(a, x, side='left')


class _Interp1dLinear(object):
    """scipy.interpolate.interp1d(kind='linear', bounds_error=False, fill_value=(lo, hi)) evaluation
    rule (index = searchsorted clipped to [1, n-1]; slope*(x-x_lo)+y_lo), as used by
    Phonon.py:387-390."""

    def __init__(self, x, y, fill):
        self.x = np.asarray(x, dtype=float)
        self.y = np.asarray(y, dtype=float)
        self.fill = fill

    def __call__(self, xn):
        xn = np.asarray(xn, dtype=float)
        shape = xn.shape
        xf = xn.ravel()
        idx = np.clip(_searchsorted_left(self.x, xf), 1, self.x.shape[0] - 1)
        xlo, xhi, ylo, yhi = self.x[idx - 1], self.x[idx], self.y[idx - 1], self.y[idx]
        out = (yhi - ylo) / (xhi - xlo) * (xf - xlo) + ylo
        if self.fill is not None:
            out = np.where(xf < self.x[0], self.fill[0], out)
            out = np.where(xf > self.x[-1], self.fill[1], out)
        return out.reshape(shape)


def material_from_phono3py(data, poscar_path, isotope=False):
    """phono3py kappa file contents (datasets mesh, qpoint, weight, frequency, group_velocity, temperature, gamma;
    Phonon.py:158-187) + POSCAR -> the FBZ-expanded tables of Phonon.load_base_properties (Phonon.py:66-113): the
    irreducible q-points are expanded with the crystal's reciprocal point-group operations (expand_FBZ :515-564),
    negative frequencies are clipped (:163), group velocities rounded to 10 decimals (:102).  With `isotope`
    (--isotope_scat lists this material) the dataset gamma_isotope is added to gamma (load_gamma, Phonon.py:316-323)."""
    from . import crystal
    if poscar_path is None or not os.path.exists(poscar_path):
        raise IOError('material: POSCAR file %r not found (it fixes the lattice and the symmetry of the expansion)' % poscar_path)
    cell = crystal.read_poscar(poscar_path)
    lattice = cell['lattice']
    rec = np.linalg.inv(lattice) * 2 * np.pi                                     # vectors as columns, Phonon.py:72
    rot = crystal.reciprocal_operations(lattice, cell['numbers'], cell['positions'])
    w = np.array(data['weight'])
    q = np.array(data['qpoint'], dtype=float)
    freq = np.array(data['frequency'], dtype=float)
    freq = np.where(freq < 0, 0, freq)
    q_fbz, freq = crystal.expand_FBZ(w, q, freq, 0, 0, rot, rec)
    _, vg = crystal.expand_FBZ(w, q, np.array(data['group_velocity'], dtype=float), 0, 1, rot, rec)
    g_ibz = np.array(data['gamma'], dtype=float)
    if isotope:
        if 'gamma_isotope' not in data:
            raise Exception('hdf file does not contain the field "gamma_isotope".')
        g_ibz = g_ibz + np.array(data['gamma_isotope'], dtype=float)
    _, gamma = crystal.expand_FBZ(w, q, g_ibz, 1, 0, rot, rec)
    return dict(data_mesh=np.array(data['mesh']), q_points=q_fbz, frequency=freq, omega=freq * 2 * np.pi,
                group_vel=np.around(vg, decimals=10), temperature=np.array(data['temperature'], dtype=float), gamma=gamma,
                lattice=lattice, reciprocal_lattice=np.around(rec, decimals=6),
                volume_unitcell=abs(np.linalg.det(lattice)))


class Phonon(Constants):
    """Phonon(arguments, mat_index, material=None)

    `material` may be a dict in the schema of `synthetic.make_material`, or None, in which case
    `arguments.hdf_file[mat_index]` selects the source:
      * 'synthetic' | 'synthetic:<n>' | 'synthetic:<n>:<Si|Ge>'  -> analytic material,
      * '<file>.npz'                                            -> tables saved by `save_npz`,
      * '<file>.hdf5'                                           -> phono3py file (needs h5py): irreducible q-points +
        weights, expanded to the full zone with the POSCAR's point group (crystal.py; tests/test_phonon_golden.py).
    """

    def __init__(self, arguments=None, mat_index=0, material=None):
        super(Phonon, self).__init__()
        self.args = arguments
        self.mat_index = int(mat_index)
        if material is None:
            material = self._load_material()
        self._ingest(material)
        if arguments is not None and len(getattr(arguments, 'mat_rotation', [])) > 0:
            self.rotate_crystal()

    # ------------------------------------------------------------------ loading
    def _load_material(self):
        name = self.args.hdf_file[self.mat_index]
        folder = ''
        if len(getattr(self.args, 'mat_folder', [])) > 0:
            folder = self.args.mat_folder[self.mat_index]
        if name.startswith('synthetic'):
            parts = name.split(':')
            n = int(parts[1]) if len(parts) > 1 else 9
            species = parts[2] if len(parts) > 2 else 'Si'
            return synthetic.make_material(n, species)
        path = os.path.join(folder, name)
        poscar = os.path.join(folder, self.args.poscar_file[self.mat_index]) if len(getattr(self.args, 'poscar_file', [])) > self.mat_index else None
        iso = self.mat_index in [int(i) for i in getattr(self.args, 'isotope_scat', [])]     # Phonon.py:319
        if path.endswith('.npz'):
            with np.load(path) as z:
                data = {k: z[k] for k in z.files}
            if 'qpoint' in data:                     # phono3py datasets kept as .npz (irreducible wedge + weights)
                return material_from_phono3py(data, poscar, iso)
            if iso:                                  # FBZ-expanded tables: the isotope part is expanded already
                if 'gamma_isotope' not in data:
                    raise Exception('material file does not contain the field "gamma_isotope".')
                data['gamma'] = np.array(data['gamma'], dtype=float) + np.array(data['gamma_isotope'], dtype=float)
            return data                              # FBZ-expanded tables (Phonon.save_npz, synthetic.make_material)
        if path.endswith('.hdf5') or path.endswith('.h5'):
            try:
                import h5py
            except ImportError:
                raise ImportError('reading %s needs h5py, which is not installed; convert the material to .npz '
                                  '(same dataset names) or use hdf_file "synthetic"' % path)
            with h5py.File(path, 'r') as f:
                data = {k: np.array(f[k]) for k in ('mesh', 'qpoint', 'weight', 'frequency', 'group_velocity',
                                                     'temperature', 'gamma')}
                if iso and 'gamma_isotope' in f:
                    data['gamma_isotope'] = np.array(f['gamma_isotope'])
            return material_from_phono3py(data, poscar, iso)
        raise ValueError('unknown material source %r' % name)

    def _ingest(self, m):
        self.data_mesh = np.array(m['data_mesh'])
        self.q_points = np.array(m['q_points'], dtype=float)
        self.weights = np.ones(self.q_points.shape[0])
        self.omega = np.array(m['omega'], dtype=float)
        if 'frequency' in m:
            self.frequency = np.array(m['frequency'], dtype=float)
        else:
            self.frequency = self.omega / (2 * self.pi)
        self.group_vel = np.array(m['group_vel'], dtype=float)
        self.temperature_array = np.array(m['temperature'], dtype=float)
        self.gamma = np.array(m['gamma'], dtype=float)
        self.reciprocal_lattice = np.array(m['reciprocal_lattice'], dtype=float)
        self.volume_unitcell = float(m['volume_unitcell'])

        self.number_of_qpoints = self.q_points.shape[0]
        self.number_of_branches = self.omega.shape[1]
        self.number_of_modes = self.number_of_qpoints * self.number_of_branches
        self.inactive_modes_mask = np.all(self.group_vel == 0, axis=2)             # Phonon.py:123
        self.number_of_inactive_modes = int(self.inactive_modes_mask.sum())
        self.number_of_active_modes = self.number_of_modes - self.number_of_inactive_modes
        self.unique_modes = np.stack(np.meshgrid(np.arange(self.number_of_qpoints),
                                                 np.arange(self.number_of_branches)), axis=-1).reshape(-1, 2).astype(int)
        self.wavevectors = self.find_min_k(self.q_to_k(np.copy(self.q_points)))    # Phonon.py:189-193
        self.norm_group_vel = np.linalg.norm(self.group_vel, axis=2)
        self.norm_wavevectors = np.linalg.norm(self.wavevectors, axis=1)
        self.calculate_lifetime()
        self.zero_point = self.calculate_zeropoint()
        self.initialise_temperature_function()

    def save_npz(self, path):
        np.savez_compressed(path, data_mesh=self.data_mesh, q_points=self.q_points, omega=self.omega,
                            frequency=self.frequency, group_vel=self.group_vel, temperature=self.temperature_array,
                            gamma=self.gamma, reciprocal_lattice=self.reciprocal_lattice,
                            volume_unitcell=np.array(self.volume_unitcell))

    # --------------------------------------------------------------- reciprocal
    def k_to_q(self, k):
        return np.dot(k, np.linalg.inv(self.reciprocal_lattice).T)                 # Phonon.py:270-276

    def q_to_k(self, q):
        return np.dot(q, self.reciprocal_lattice.T)                                # Phonon.py:278-282

    def find_min_k(self, k, return_disp=False):
        """Shortest equivalent wavevector; same walk and same first-minimum tie rule as
        Phonon.py:209-247 (neighbour order = np.meshgrid default 'xy' order)."""
        a = np.array([-1, 0, 1])
        n = np.vstack([g.ravel() for g in np.meshgrid(a, a, a)]).T
        i0 = int(np.nonzero(np.all(n == 0, axis=1))[0][0])
        k = np.asarray(k, dtype=float)
        q = self.k_to_q(k)
        disp = np.zeros(k.shape)
        active = np.ones(q.shape[0], dtype=bool)
        while np.any(active):
            q_new = q[active, :] + n[:, None, :]
            norm = np.linalg.norm(self.q_to_k(q_new), axis=-1).T
            i_min = np.argmax(norm == norm.min(axis=1, keepdims=True), axis=1)
            if return_disp:
                disp[active, :] += n[i_min, :]
            q[active, :] = q_new[i_min, np.arange(i_min.shape[0]), :]
            active[active] = i_min != i0
        if return_disp:
            return self.q_to_k(q), self.q_to_k(disp)
        return self.q_to_k(q)

    def rotate_crystal(self):
        """Phonon.py:284-314: rotate wavevectors and group velocities (scipy Rotation)."""
        import re
        from scipy.spatial.transform import Rotation as rot
        groups, g = [], []
        for i, s in enumerate(self.args.mat_rotation):
            s = str(s)
            if re.fullmatch('[0-9.+-eE]+', s) and not re.fullmatch('[A-Za-z]+', s):
                g.append(i)
            elif re.fullmatch('[A-Z]+|[a-z]+', s):
                g.append(i)
                groups.append(g)
                g = []
        if groups:
            grp = groups[self.mat_index]
            params = [self.args.mat_rotation[i] for i in grp]
            R = rot.from_euler(params[-1], [float(i) for i in params[:-1]], degrees=True)
            self.wavevectors = R.apply(self.wavevectors)
            for j in range(self.number_of_branches):
                self.group_vel[:, j, :] = R.apply(self.group_vel[:, j, :])

    # ------------------------------------------------------------ thermodynamics
    def calculate_occupation(self, T, omega):
        """Bose-Einstein occupation, Phonon.py:338-345."""
        T = np.asarray(T, dtype=float)
        omega = np.asarray(omega, dtype=float)
        flag = (T > 0) & (omega > 0)
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            occ = np.where(~flag, 0, 1 / (np.exp(omega * self.hbar / (T * self.kb)) - 1))
        return occ

    def calculate_energy(self, T, omega):
        return self.hbar * omega * self.calculate_occupation(T, omega)             # Phonon.py:347-350

    def normalise_to_density(self, x):
        return x / (self.number_of_qpoints * self.volume_unitcell)                 # Phonon.py:392-401

    def calculate_zeropoint(self):
        return self.normalise_to_density(self.hbar * self.omega.sum() / 2)         # Phonon.py:364-370

    def calculate_crystal_energy(self, T):
        T = np.array(T, dtype=float).reshape((-1, 1, 1))                            # Phonon.py:352-362
        e = (self.calculate_energy(T, self.omega) * ~self.inactive_modes_mask).sum(axis=(1, 2))
        return self.normalise_to_density(e) + self.zero_point

    def calculate_lifetime(self):
        """Phonon.py:326-336: tau = 1/(4 pi gamma) where gamma > 0 else 0."""
        with np.errstate(divide='ignore', invalid='ignore'):
            self.lifetime = np.where(self.gamma > 0, 1 / (2 * 2 * np.pi * self.gamma), 0)

    def lifetime_function(self, Tqj):
        """RegularGridInterpolator((T, q, j), lifetime) of Phonon.py:336, evaluated at integer (q, j):
        linear in tau along T."""
        Tqj = np.asarray(Tqj, dtype=float)
        T = Tqj[:, 0]
        q = Tqj[:, 1].astype(int)
        j = Tqj[:, 2].astype(int)
        g = self.temperature_array
        if np.any(T < g[0]) or np.any(T > g[-1]):
            raise ValueError('One of the requested temperatures is out of the tabulated range')
        i = np.clip(_searchsorted_left(g, T) - 1, 0, g.shape[0] - 2)
        y = (T - g[i]) / (g[i + 1] - g[i])
        return self.lifetime[i, q, j] * (1 - y) + self.lifetime[i + 1, q, j] * y

    def initialise_temperature_function(self):
        """Phonon.py:372-390: E(T) on a 0.1 K grid, and its inverse."""
        T_min = self.temperature_array.min()
        T_max = self.temperature_array.max()
        dT = 0.1
        self.T_array = np.arange(T_min, T_max + dT, dT)
        self.energy_array = np.concatenate([self.calculate_crystal_energy(self.T_array[i:i + 256])
                                            for i in range(0, self.T_array.shape[0], 256)])
        self.temperature_function = _Interp1dLinear(self.energy_array, self.T_array, (T_min, T_max))
        self.crystal_energy_function = _Interp1dLinear(self.T_array, self.energy_array,
                                                       (self.energy_array.min(), self.energy_array.max()))

    # ------------------------------------------------------------------ export
    def tables(self):
        """Flat tables handed to the C-ABI library (nk_set_material) and to the oracle."""
        return dict(omega=self.omega, group_vel=self.group_vel, T_grid=self.temperature_array,
                    lifetime=self.lifetime, T_array=self.T_array, energy_array=self.energy_array,
                    hbar=self.hbar, kb=self.kb, QV=self.number_of_qpoints * self.volume_unitcell,
                    active_modes=self.number_of_active_modes)
