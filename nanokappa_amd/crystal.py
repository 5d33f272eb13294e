"""Crystal structure helpers for the material loader.

The reference reads the structure with phonopy (`read_crystal_structure`, Phonon.py:70-72) and asks phonopy / spglib for
the reciprocal-space point-group operations with which it expands phono3py's irreducible q-points to the full Brillouin
zone (`expand_FBZ`, Phonon.py:515-564).  Neither library is available here, so the two pieces are written out: a POSCAR
reader and a small symmetry finder (lattice holohedry filtered by the atomic basis).  Everything is set-up code on the
host; nothing here runs per timestep.
"""
import itertools

import numpy as np


def read_poscar(path):
    """VASP POSCAR -> dict(lattice (3,3) rows = vectors [angstrom], species list, numbers (N,), positions (N,3) fractional)."""
    with open(path) as f:
        lines = [ln.strip() for ln in f.readlines()]
    scale = float(lines[1].split()[0])
    lattice = np.array([[float(x) for x in lines[2 + i].split()[:3]] for i in range(3)])
    if scale < 0:                                   # negative scale = target volume
        scale = (-scale / abs(np.linalg.det(lattice))) ** (1.0 / 3.0)
    lattice = lattice * scale
    i = 5
    tokens = lines[i].split()
    if all(t.lstrip('-').isdigit() for t in tokens):            # VASP 4: no species line
        species, counts = ['X%d' % k for k in range(len(tokens))], [int(t) for t in tokens]
    else:
        species = tokens
        i += 1
        counts = [int(t) for t in lines[i].split()]
    i += 1
    if lines[i][:1].lower() == 's':                 # selective dynamics
        i += 1
    cartesian = lines[i][:1].lower() in ('c', 'k')
    i += 1
    n = sum(counts)
    pos = np.array([[float(x) for x in lines[i + a].split()[:3]] for a in range(n)])
    if cartesian:
        pos = np.dot(pos * scale, np.linalg.inv(lattice))
    numbers = np.concatenate([np.full(c, k) for k, c in enumerate(counts)]).astype(int)
    return dict(lattice=lattice, species=species, numbers=numbers, positions=pos % 1.0)


def write_poscar(path, lattice, species, counts, positions, comment='written by nanokappa_amd'):
    """Minimal VASP-5 POSCAR (direct coordinates)."""
    with open(path, 'w') as f:
        f.write(comment + '\n   1.0\n')
        for row in np.asarray(lattice, dtype=float):
            f.write('   %22.16f %22.16f %22.16f\n' % tuple(row))
        f.write('   ' + ' '.join(species) + '\n   ' + ' '.join(str(int(c)) for c in counts) + '\nDirect\n')
        for row in np.asarray(positions, dtype=float):
            f.write('  %19.16f %19.16f %19.16f\n' % tuple(row))


def point_group(lattice, numbers, positions, tol=1e-5):
    """Rotation parts W (integer 3x3, acting on fractional coordinates x' = W x + t) of the space group of the crystal.

    Candidates are the integer matrices with entries in {-1, 0, 1} that keep the metric tensor G = L L^T; a candidate is
    kept when some translation maps every atom onto an atom of the same species."""
    L = np.asarray(lattice, dtype=float)
    G = L @ L.T
    numbers = np.asarray(numbers)
    pos = np.asarray(positions, dtype=float) % 1.0
    ops = []
    ref_species = numbers[0]
    for ent in itertools.product((-1, 0, 1), repeat=9):
        W = np.array(ent, dtype=int).reshape(3, 3)
        if abs(round(float(np.linalg.det(W)))) != 1:
            continue
        # cartesian r = L^T x, |r|^2 = x^T G x: the map x' = W x keeps lengths iff W^T G W = G
        if not np.allclose(W.T @ G @ W, G, rtol=0, atol=tol * np.abs(G).max()):
            continue
        rot = pos @ W.T
        ok = False
        for b in np.nonzero(numbers == ref_species)[0]:          # candidate translations: atom 0 -> atom b
            t = pos[b] - rot[0]
            img = (rot + t) % 1.0
            good = True
            for a in range(pos.shape[0]):
                d = img[a] - pos[numbers == numbers[a]]
                d -= np.rint(d)
                if not np.any(np.all(np.abs(d) < tol, axis=1)):
                    good = False
                    break
            if good:
                ok = True
                break
        if ok:
            ops.append(W)
    return np.array(ops, dtype=int)


def reciprocal_operations(lattice, numbers, positions, time_reversal=True):
    """Point-group operations acting on reduced reciprocal coordinates (the role of phonopy's
    `primitive_symmetry.get_reciprocal_operations()`, Phonon.py:80-81): (W^-1)^T for every real-space rotation W, plus
    the inversion that time reversal adds."""
    W = point_group(lattice, numbers, positions)
    R = np.array([np.rint(np.linalg.inv(w).T).astype(int) for w in W])
    if time_reversal:
        R = np.concatenate((R, -R))
    return np.unique(R, axis=0)


def expand_FBZ(weights, qpoints, tensor, axis, rank, rotations, reciprocal_lattice):
    """Irreducible wedge -> full Brillouin zone, Phonon.py:515-564: for every irreducible q the star {R q mod 1}
    (rounded to 6 decimals, unique, in np.unique order) with the tensor copied (rank 0) or rotated by the Cartesian image
    of R (rank 1: group velocities).  `axis` is the q axis of `tensor`.  Raises when a star's size differs from the
    file's weight."""
    rl = np.asarray(reciprocal_lattice, dtype=float)
    rl_inv = np.linalg.inv(rl)
    r_cart = np.array([rl @ (r @ rl_inv) for r in rotations])
    q_out, t_out = [], []
    for i, q in enumerate(np.asarray(qpoints, dtype=float)):
        tq = np.take(tensor, i, axis=axis)
        star = np.around(np.mod(np.mod(q, 1.0) @ np.transpose(rotations, (0, 2, 1)), 1.0), decimals=6)
        star = np.where(star == 1.0, 0.0, star)
        sq, idx = np.unique(star, return_index=True, axis=0)
        if int(weights[i]) != idx.shape[0]:
            raise ValueError('expand_FBZ: q-point %d has weight %d but its star has %d members' % (i, int(weights[i]), idx.shape[0]))
        if rank == 0:
            st = np.array([tq for _ in idx])
        elif rank == 1:
            st = np.array([np.dot(r_cart[k], tq.T).T for k in idx])
        else:
            raise ValueError('expand_FBZ: rank %d is not coded' % rank)
        q_out.append(sq)
        t_out.append(st)
    return np.concatenate(q_out, axis=0), np.swapaxes(np.concatenate(t_out, axis=0), 0, axis)


def reduce_to_IBZ(q_points, rotations):
    """Inverse bookkeeping of expand_FBZ (used by the tests and by `Phonon.save_ibz_npz`): one representative per star
    of a full Gamma-centred mesh with the star size as its weight.  Returns (indices of the representatives, weights)."""
    qx = np.mod(np.asarray(q_points, dtype=float), 1.0)
    q = np.around(qx, decimals=6)
    q = np.where(q == 1.0, 0.0, q)
    key = {tuple(v): i for i, v in enumerate(map(tuple, q))}
    seen = np.zeros(q.shape[0], dtype=bool)
    reps, weights = [], []
    for i in range(q.shape[0]):
        if seen[i]:
            continue
        star = np.around(np.mod(qx[i] @ np.transpose(rotations, (0, 2, 1)), 1.0), decimals=6)
        star = np.where(star == 1.0, 0.0, star)
        members = {key[tuple(s)] for s in map(tuple, star)}
        for m in members:
            seen[m] = True
        reps.append(i)
        weights.append(len(members))
    return np.array(reps, dtype=int), np.array(weights, dtype=int)
