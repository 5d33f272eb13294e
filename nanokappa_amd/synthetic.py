"""Synthetic phonon material in the phono3py schema.

The reference reads `kappa-mNNN.hdf5` (datasets mesh, qpoint, weight, frequency,
group_velocity, temperature, gamma; reference classes/Phonon.py:153-187,
:316-324).  The Si/Ge HDF5 blobs are not shipped with the reference
(`.MISSING_LARGE_BLOBS`), so parity and performance work uses an analytic
material on the real Si / Ge primitive fcc lattices (test_material/*/POSCAR),
already expanded to the full Brillouin zone (the state the reference holds
after `expand_FBZ`, Phonon.py:93-116).

Pure NumPy and python-3.9 compatible on purpose: the golden-vector harness
(tests/golden/make_golden.py) imports this file under the interpreter that can
run the reference.
"""
import numpy as np

# lattice parameter "a" of the fcc primitive cells in the POSCARs shipped by the
# reference (test_material/Si/POSCAR:3-5, test_material/Ge/POSCAR:3-5) [angstrom]
LATTICE_HALF_A = {'Si': 2.7343755164098931, 'Ge': 2.8916046182323498}

# branch parameters: (kind, omega_max [rad/ps] for Si).  Ge is scaled by 0.6.
_BRANCHES = (
    ('acoustic', 2.0 * np.pi * 4.5),
    ('acoustic', 2.0 * np.pi * 5.5),
    ('acoustic', 2.0 * np.pi * 12.0),
    ('optical', 2.0 * np.pi * 13.5),
    ('optical', 2.0 * np.pi * 14.5),
    ('optical', 2.0 * np.pi * 15.5),
)


def fcc_lattice(species='Si'):
    a = LATTICE_HALF_A[species]
    return np.array([[0.0, a, a], [a, 0.0, a], [a, a, 0.0]])


def _find_min_k(q, rec):
    """Map reduced q to the shortest equivalent wavevector (first Brillouin zone).

    Same idea as Phonon.find_min_k (Phonon.py:209-247): walk to the neighbour
    image of smallest |k| until the image itself is the minimum.
    """
    a = np.array([-1, 0, 1])
    n = np.vstack([g.ravel() for g in np.meshgrid(a, a, a)]).T  # (27,3)
    i0 = int(np.nonzero(np.all(n == 0, axis=1))[0][0])
    q = np.array(q, dtype=float)
    active = np.ones(q.shape[0], dtype=bool)
    while np.any(active):
        qn = q[active][None, :, :] + n[:, None, :]        # (27,Qa,3)
        kn = qn @ rec.T
        norm = np.linalg.norm(kn, axis=-1).T               # (Qa,27)
        # deterministic tie-break: first index within 1e-12 of the minimum
        imin = np.argmax(norm <= norm.min(axis=1, keepdims=True) + 1e-12, axis=1)
        # prefer staying put when the current image is already minimal
        stay = norm[:, i0] <= norm.min(axis=1) + 1e-12
        imin[stay] = i0
        q[active] = qn[imin, np.arange(imin.shape[0])]
        active[active] = imin != i0
    return q @ rec.T


def make_material(n_mesh=9, species='Si', temperatures=None, gamma_coeff=2.0e-9):
    """Return a dict of FBZ-expanded tables.

    keys: data_mesh (3,), q_points (Q,3) reduced, frequency (Q,J) [THz],
    omega (Q,J) [rad/ps], group_vel (Q,J,3) [angstrom/ps], temperature (NT,),
    gamma (NT,Q,J) [THz], lattice (3,3 rows), reciprocal_lattice (3,3 columns,
    rounded to 6 decimals as Phonon.py:129), volume_unitcell.
    """
    if temperatures is None:
        temperatures = np.arange(0.0, 1000.0 + 1e-9, 10.0)
    temperatures = np.asarray(temperatures, dtype=float)
    lattice = fcc_lattice(species)
    rec = np.around(np.linalg.inv(lattice) * 2.0 * np.pi, decimals=6)
    vol = abs(np.linalg.det(lattice))

    g = np.arange(n_mesh) / float(n_mesh)
    q = np.vstack([m.ravel() for m in np.meshgrid(g, g, g, indexing='ij')]).T  # (Q,3)
    k = _find_min_k(q, rec)
    k = np.around(k, decimals=12)
    knorm = np.linalg.norm(k, axis=1)
    kmax = knorm.max()
    with np.errstate(divide='ignore', invalid='ignore'):
        khat = np.where(knorm[:, None] > 0, k / knorm[:, None], 0.0)

    scale = 1.0 if species == 'Si' else 0.6
    Q = q.shape[0]
    J = len(_BRANCHES)
    omega = np.zeros((Q, J))
    vg = np.zeros((Q, J, 3))
    x = knorm / kmax
    for j, (kind, wmax) in enumerate(_BRANCHES):
        wmax = wmax * scale
        if kind == 'acoustic':
            omega[:, j] = wmax * np.sin(0.46 * np.pi * x)
            dwdk = wmax * 0.46 * np.pi / kmax * np.cos(0.46 * np.pi * x)
        else:
            depth = 0.12 * wmax
            omega[:, j] = wmax - depth * x * x
            dwdk = -2.0 * depth * x / kmax
        vg[:, j, :] = dwdk[:, None] * khat
    omega = np.where(omega < 0, 0.0, omega)
    vg = np.around(vg, decimals=10)          # Phonon.py:102
    frequency = omega / (2.0 * np.pi)

    gamma = gamma_coeff * (omega ** 2)[None, :, :] * temperatures[:, None, None]
    gamma = np.where(gamma > 0, gamma, -1.0)  # Phonon.py:324

    return dict(data_mesh=np.array([n_mesh] * 3), q_points=q, frequency=frequency,
                omega=omega, group_vel=vg, temperature=temperatures, gamma=gamma,
                lattice=lattice, reciprocal_lattice=rec, volume_unitcell=vol,
                wavevectors_hint=k, species=species)
