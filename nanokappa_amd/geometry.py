"""Geometry: mesh + boundary conditions + subvolumes, with the attribute surface `Population` reads from the
reference's `Geometry` (classes/Geometry.py; SURVEY.md section 8b "Path -> Geometry").

Scope this round: `box`/`cuboid` and `cylinder`/`rod`/`bar` primitives, STL files, `slice` subvolumes.
`grid` / `voronoi` subvolumes and the remaining primitives are SURVEY section 8f row 4 (not built yet).
"""
import numpy as np

from .mesh import Mesh, read_stl


VORONOI_MAX_SAMPLES = 1000000      # routines/subvolumes.py: n_s_max (tests lower it)


class SubvolClassifier(object):
    """Nearest-centre classifier (Geometry.py:1198-1213).  Host copy for initialisation; the hot path classifies
    on the GPU."""

    def __init__(self, n, xc):
        self.n = n
        self.xc = np.asarray(xc, dtype=float)
        # centres in a row along one axis ('slice' subvolumes): the nearest one follows from the midpoints
        self._axis = None
        if self.xc.shape[0] > 1:
            moving = [a for a in range(3) if np.ptp(self.xc[:, a]) > 0]
            if len(moving) == 1 and np.all(np.diff(self.xc[:, moving[0]]) > 0):
                self._axis = moving[0]
                c = self.xc[:, self._axis]
                self._mid = 0.5 * (c[1:] + c[:-1])

    def predict(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        if self._axis is not None:
            return np.searchsorted(self._mid, x[:, self._axis], side='left').astype(int)   # an exact tie goes to the lower index
        out = np.empty(x.shape[0], dtype=int)
        for s in range(0, x.shape[0], 1 << 18):
            d = ((x[s:s + (1 << 18), None, :] - self.xc[None]) ** 2).sum(axis=2)
            out[s:s + (1 << 18)] = np.argmin(d, axis=1)
        return out


def box_primitive(dims):
    """Vertices / faces of Geometry.generate_primitives('box') (Geometry.py:87-108)."""
    v = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]], dtype=float)
    v = v * np.array(dims[:3], dtype=float)
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [0, 4, 5], [0, 5, 1],
                  [3, 7, 6], [3, 6, 2], [0, 4, 7], [0, 7, 3], [1, 5, 6], [1, 6, 2]], dtype=int)
    return v, f


def cylinder_primitive(dims):
    """Geometry.generate_primitives('cylinder') (Geometry.py:110-142): dims = L, R, N_sides; axis along z."""
    L, R, N = float(dims[0]), float(dims[1]), int(dims[2])
    ang = np.arange(N) * 2 * np.pi / N
    ring = np.vstack((np.cos(ang), np.sin(ang), np.zeros(N))).T * R
    v = np.vstack((np.zeros((1, 3)), ring, np.array([[0, 0, L]]), ring + np.array([0, 0, L])))
    # face order of the reference primitive: lower cap fan, side quads (two triangles each), upper cap fan
    nxt = lambda i: 1 if i == N else i + 1
    f = [[0, i, nxt(i)] for i in range(1, N + 1)]
    for i in range(1, N + 1):
        f.append([i, i + N + 1, nxt(i) + N + 1])
        f.append([i, nxt(i), nxt(i) + N + 1])
    f += [[N + 1, i + N + 1, nxt(i) + N + 1] for i in range(1, N + 1)]
    return v, np.array(f, dtype=int)


def ring_stack(rings, bottom=None, top=None):
    """Closed surface from a stack of polygonal rings (each (N,3), same N, same angular order): a fan cap on the first
    and the last ring, two triangles per quad between consecutive rings.  Two consecutive rings in the same plane give a
    flat annulus (the lids of the 'castle').  All the wire-like primitives of Geometry.generate_primitives are such
    stacks; the triangulation is this builder's own (coplanar triangles are merged into facets by Mesh anyway)."""
    N = rings[0].shape[0]
    bottom = rings[0].mean(axis=0) if bottom is None else np.asarray(bottom, dtype=float)
    top = rings[-1].mean(axis=0) if top is None else np.asarray(top, dtype=float)
    v = np.vstack([bottom[None, :]] + list(rings) + [top[None, :]])
    nxt = (np.arange(N) + 1) % N
    f = [[0, 1 + i, 1 + nxt[i]] for i in range(N)]
    for k in range(len(rings) - 1):
        a, b = 1 + k * N, 1 + (k + 1) * N
        for i in range(N):
            f.append([a + i, a + nxt[i], b + nxt[i]])
            f.append([a + i, b + nxt[i], b + i])
    last, tip = 1 + (len(rings) - 1) * N, v.shape[0] - 1
    f += [[tip, last + i, last + nxt[i]] for i in range(N)]
    return v, np.array(f, dtype=int)


def _circle(R, N, z=0.0, dx=0.0, dy=0.0, phase=0.0):
    ang = (np.arange(N) + phase) * 2 * np.pi / N
    return np.vstack((np.cos(ang) * R + dx, np.sin(ang) * R + dy, np.full(N, float(z)))).T


def zigzag_primitive(dims):
    """'zigzag' (Geometry.py:143-178): L, R, dx, dy, N_sides, N_sections -- rings of radius R every L along z, every odd
    one displaced by (dx, dy)."""
    L, R, dx, dy, Ns, Nc = float(dims[0]), float(dims[1]), float(dims[2]), float(dims[3]), int(dims[4]), int(dims[5])
    rings = [_circle(R, Ns, i * L, (i % 2) * dx, (i % 2) * dy) for i in range(Nc + 1)]
    return ring_stack(rings, bottom=[0, 0, 0], top=[(Nc % 2) * dx, (Nc % 2) * dy, Nc * L])


def corrugated_primitive(dims):
    """'corrugated' (Geometry.py:180-215): L, R, r, N_sides, N_sections -- rings every L, radius R at even and r at odd
    stations."""
    L, R, r, Ns, Nc = float(dims[0]), float(dims[1]), float(dims[2]), int(dims[3]), int(dims[4])
    return ring_stack([_circle(R if i % 2 == 0 else r, Ns, i * L) for i in range(Nc + 1)])


def castle_primitive(dims):
    """'castle' (Geometry.py:217-316): L, l, R, r, N_sides, N_sections, start_large -- alternating cylinders of radius R
    (length L) and r (length l) joined by flat annular lids."""
    L, l, R, r, Ns, Nc, s = float(dims[0]), float(dims[1]), float(dims[2]), float(dims[3]), int(dims[4]), int(dims[5]), bool(float(dims[6]))
    if R <= r:
        raise Exception('Outer radius smaller or equal to the inner radius. Check parameters.')
    rings, z, large = [_circle(r, Ns, 0.0)], 0.0, s
    for _ in range(Nc):
        if large:
            rings += [_circle(R, Ns, z), _circle(R, Ns, z + L), _circle(r, Ns, z + L)]
            z += L
        else:
            rings.append(_circle(r, Ns, z + l))
            z += l
        large = not large
    return ring_stack(rings)


def star_primitive(dims):
    """'star' (Geometry.py:318-368): H, R, r, N_points -- prism of height H over a 2N-gon alternating between r (at
    half-step angles) and R."""
    H, R, r, N = float(dims[0]), float(dims[1]), float(dims[2]), int(dims[3])
    if R <= r:
        raise Exception('Outer radius smaller or equal to the inner radius. Check parameters.')
    inner, outer = _circle(r, N, 0.0, phase=-0.5), _circle(R, N, 0.0)
    poly = np.empty((2 * N, 3))
    poly[0::2], poly[1::2] = inner, outer
    return ring_stack([poly, poly + np.array([0, 0, H])], bottom=[0, 0, 0], top=[0, 0, H])


def freewire_primitive(dims):
    """'freewire' (Geometry.py:370-400): R0, L0, R1, L1, ..., R(n), N_sides -- rings of the given radii at the running
    sum of the section lengths."""
    R = np.array([dims[i] for i in range(0, len(dims) - 1, 2)], dtype=float)
    L = np.array([dims[i] for i in range(1, len(dims) - 1, 2)], dtype=float)
    N = int(float(dims[-1]))
    z = np.concatenate(([0.0], np.cumsum(L)))[:R.shape[0]]
    return ring_stack([_circle(R[i], N, z[i]) for i in range(R.shape[0])])


class Geometry(object):
    def __init__(self, args):
        self.args = args
        self.scale = np.array(args.scale, dtype=float)
        self.shape = args.geometry[0]
        self.dimensions = args.dimensions
        self.folder = getattr(args, 'results_folder', '.')
        self.subvol_type = args.subvolumes[0]
        gr = list(getattr(args, 'geo_rotation', []))
        if len(gr) > 0:
            self.rotation = np.array(gr[:-1]).astype(float)
            self.rot_order = gr[-1]
        else:
            self.rotation, self.rot_order = None, None

        self.load_geo_file(self.shape)
        self.transform_mesh()
        self.get_mesh_properties()
        self.get_bound_facets(args)
        self.check_facet_connections(args)
        self.set_subvolumes()

    # ---------------------------------------------------------------------- mesh
    def load_geo_file(self, shape):
        if shape in ('cuboid', 'box'):
            v, f = box_primitive(self.dimensions)
        elif shape in ('cylinder', 'rod', 'bar'):
            v, f = cylinder_primitive(self.dimensions)
        elif shape == 'zigzag':
            v, f = zigzag_primitive(self.dimensions)
        elif shape == 'corrugated':
            v, f = corrugated_primitive(self.dimensions)
        elif shape == 'castle':
            v, f = castle_primitive(self.dimensions)
        elif shape == 'star':
            v, f = star_primitive(self.dimensions)
        elif shape == 'freewire':
            v, f = freewire_primitive(self.dimensions)
        elif str(shape).lower().endswith('.stl'):
            v, f = read_stl(shape)
        else:
            raise Exception('geometry %r: not a standard shape and not an STL file' % shape)
        self._raw = (v, f)

    def transform_mesh(self):
        """rezero -> scale -> rotate -> rezero (Geometry.py:414-433)."""
        v, f = self._raw
        v = (v - v.min(axis=0)) * self.scale
        if self.rotation is not None and np.any(self.rotation != 0):
            from scipy.spatial.transform import Rotation as rot
            v = rot.from_euler(self.rot_order, self.rotation, degrees=True).apply(v)
        v = v - v.min(axis=0)
        self.mesh = Mesh(v, f)

    def get_mesh_properties(self):
        m = self.mesh
        self.faces, self.facets = m.faces, m.facets
        self.n_of_faces, self.n_of_facets = m.n_of_faces, m.n_of_facets
        self.bounds, self.facet_centroid, self.volume = m.bounds, m.facet_centroid, m.volume
        self.facets_normal, self.facets_area = m.facets_normal, m.facets_area

    def scale_positions(self, x, inv=False):
        ext = self.bounds[1] - self.bounds[0]                                          # Geometry.py:946-959
        return x * ext + self.bounds[0] if inv else (x - self.bounds[0]) / ext

    # ------------------------------------------------------- boundary conditions
    def get_bound_facets(self, args):
        """Geometry.get_bound_facets (Geometry.py:652-709): the last --bound_cond entry is the default; the
        others are attached to the facet closest to each --bound_pos point."""
        self.bound_cond = np.array([args.bound_cond[-1] for _ in range(self.n_of_facets)])
        try:
            self.bound_pos = np.array(args.bound_pos[1:]).reshape(-1, 3).astype(float)
        except Exception:
            raise Exception('Boundary positions ill defined. Check input parameters.')
        if args.bound_pos[0] == 'relative':
            self.bound_pos = self.scale_positions(self.bound_pos, True)
        elif args.bound_pos[0] != 'absolute':
            raise Exception('Please specify the type of position for BC with the keyword "absolute" or "relative".')
        self.bound_facets, _, _ = self.mesh.closest_facet(self.bound_pos)
        for j, i in enumerate(self.bound_facets):
            self.bound_cond[i] = args.bound_cond[j]
        is_res = (self.bound_cond == 'T') | (self.bound_cond == 'F')
        self.res_facets = np.arange(self.n_of_facets, dtype=int)[is_res]
        self.res_bound_cond = self.bound_cond[is_res]
        self.rough_facets = np.arange(self.n_of_facets, dtype=int)[self.bound_cond == 'R']
        self.n_of_reservoirs = self.res_facets.shape[0]
        self.n_of_rough_facets = self.rough_facets.shape[0]
        self.res_values = np.ones(self.n_of_reservoirs) * np.nan
        self.rough_facets_values = np.ones(self.n_of_rough_facets) * np.nan
        if args.bound_cond[-1] in ('T', 'F'):
            self.res_values[:] = args.bound_values[-1]
        elif args.bound_cond[-1] == 'R':
            self.rough_facets_values[:] = args.bound_values[-1]
        bound_indices = [-1] * len(self.bound_cond)
        i = 0
        for f, facet in enumerate(self.bound_facets):
            if self.bound_cond[facet] != 'P':
                bound_indices[f] = i
                i += 1
        for i, bf in enumerate(self.bound_facets):
            if bf in self.res_facets:
                self.res_values[self.res_facets == bf] = args.bound_values[bound_indices[i]]
            elif bf in self.rough_facets:
                self.rough_facets_values[self.rough_facets == bf] = args.bound_values[bound_indices[i]]

    def check_facet_connections(self, args):
        """Pairs of periodic facets (Geometry.py:711-726).  The reference's polygon-equality sanity check
        (:728-766) only prints; here opposite normals and equal areas are enforced."""
        self.connected_facets = np.zeros((0, 2), dtype=int)
        cp = list(getattr(args, 'connect_pos', []))
        if len(cp) > 0:
            pts = np.array(cp[1:], dtype=float).reshape(-1, 3)
            if cp[0] == 'relative':
                pts = self.scale_positions(pts, True)
            elif cp[0] != 'absolute':
                raise Exception("Wrong option in --connect_pos. Choose between 'relative' or 'absolute'.")
            self.connected_facets = self.mesh.closest_facet(pts)[0].reshape(-1, 2)
        for a, b in self.connected_facets:
            if not np.all(np.abs(self.facets_normal[a] + self.facets_normal[b]) < 0.1):
                raise Exception('Connected facets normals do not agree!!')
            if abs(self.facets_area[a] - self.facets_area[b]) > 1e-6 * self.facets_area[a]:
                raise Exception('Connected facets have different areas. Check --connect_pos.')
        for f in np.nonzero(self.bound_cond == 'P')[0]:
            if f not in self.connected_facets:
                raise Exception('Facet %d has periodic BC but no partner in --connect_pos.' % f)

    # ---------------------------------------------------------------- subvolumes
    def set_subvolumes(self):
        if self.subvol_type == 'grid':
            grid = np.array(self.args.subvolumes[1:4]).astype(int)
            if (grid == 1).sum() == 2:                                                  # Geometry.py:497-506
                ax = int(np.nonzero(grid != 1)[0][0])
                self.args.subvolumes = ['slice', int(grid[ax]), ax]
                self.subvol_type = 'slice'
        if self.subvol_type == 'grid':
            self._set_grid_subvolumes(grid)
            return
        if self.subvol_type == 'voronoi':
            self._set_voronoi_subvolumes(int(self.args.subvolumes[1]))
            return
        if self.subvol_type != 'slice':
            raise Exception('Invalid subvolume type!')
        self.n_of_subvols = int(self.args.subvolumes[1])                                # Geometry.py:449-471
        self.slice_axis = int(self.args.subvolumes[2])
        ext = self.bounds[1] - self.bounds[0]
        c = np.zeros((self.n_of_subvols, 3)) + np.mean(self.bounds, axis=0)
        arr = (np.arange(self.n_of_subvols) + 0.5) / self.n_of_subvols
        arr *= ext[self.slice_axis]
        arr += self.bounds[0, self.slice_axis]
        c[:, self.slice_axis] = arr
        self.subvol_center = c[np.lexsort((c[:, 2], c[:, 1], c[:, 0]))]
        self.slice_length = ext[self.slice_axis] / self.n_of_subvols
        self.subvol_classifier = SubvolClassifier(self.n_of_subvols, self.subvol_center)
        self.subvol_volume = self.calculate_subvol_volume()
        con = np.vstack((np.arange(self.n_of_subvols - 1), np.arange(self.n_of_subvols - 1) + 1)).T
        self.subvol_connections = con                                                   # Geometry.py:968-975
        self.n_of_subvol_con = con.shape[0]
        self.subvol_con_vectors = self.subvol_center[con[:, 1]] - self.subvol_center[con[:, 0]]

    def _set_grid_subvolumes(self, grid):
        """nx x ny x nz cell centres over the bounding box (Geometry.py:508-538), those on the surface dropped, sorted
        lexicographically; neighbours from get_subvol_connections; 3-D nearest-centre classifier."""
        self.grid = grid
        nx, ny, nz = (int(g) for g in grid)
        xx = np.linspace(0.5 / nx, 1 - 0.5 / nx, nx)
        yy = np.linspace(0.5 / ny, 1 - 0.5 / ny, ny)
        zz = np.linspace(0.5 / nz, 1 - 0.5 / nz, nz)
        g = np.meshgrid(xx, yy, zz)
        ext = self.bounds[1] - self.bounds[0]
        c = np.vstack(list(map(np.ravel, g))).T * ext + self.bounds[0, :]
        _, dist, _ = self.mesh.closest_face(c)
        c = c[dist > 0, :]                                                             # closest_point(...)[1] > 0
        self.subvol_center = c[np.lexsort((c[:, 2], c[:, 1], c[:, 0]))]
        self.n_of_subvols = self.subvol_center.shape[0]
        self.get_subvol_connections()
        self.subvol_classifier = SubvolClassifier(self.n_of_subvols, self.subvol_center)
        self.subvol_volume = self.calculate_subvol_volume()

    def _set_voronoi_subvolumes(self, n, seed=20231003):
        """n reference points spread by Lloyd relaxation on Monte-Carlo samples of the solid (routines/subvolumes.py:
        centres move to the centroid of the samples nearest to them; the sample set doubles every time the centres
        stop moving, up to 1e6), then as Geometry.py:474-491.  The reference draws from numpy's global generator;
        here the generator is seeded, so the same arguments always give the same subvolumes."""
        from scipy.spatial import cKDTree
        rng = np.random.default_rng(seed)
        x_r = self.mesh.sample_volume(n, rng)
        n_s, n_s_max, crit = 1000, VORONOI_MAX_SAMPLES, 1e-8
        x_s = self.mesh.sample_volume(n_s, rng)
        for _ in range(10000):
            r = cKDTree(x_r).query(x_s)[1]
            cnt = np.bincount(r, minlength=n)
            new = np.vstack([np.bincount(r, weights=x_s[:, k], minlength=n) for k in range(3)]).T
            new = np.where(cnt[:, None] > 0, new / np.maximum(cnt, 1)[:, None], x_r)
            move = np.linalg.norm(new - x_r, axis=1).max()
            x_r = new
            if move < crit:
                if n_s >= n_s_max:
                    break
                n_s = min(2 * n_s, n_s_max)
                x_s = self.mesh.sample_volume(n_s, rng)
        c = x_r[self.mesh.contains(x_r), :]
        self.subvol_center = c[np.lexsort((c[:, 2], c[:, 1], c[:, 0]))]
        self.n_of_subvols = self.subvol_center.shape[0]
        self.get_subvol_connections()
        self.subvol_classifier = SubvolClassifier(self.n_of_subvols, self.subvol_center)
        self.subvol_volume = self.calculate_subvol_volume()

    def get_subvol_connections(self):
        """Which subvolumes are neighbours (Geometry.py:961-1052): candidate pairs whose midpoint is inside the solid and
        whose connecting segment does not leave it, confirmed nearest first; a pair is dropped when its midpoint lies
        beyond the interface plane of an already confirmed neighbour of either end."""
        sc = self.subvol_center
        S = self.n_of_subvols
        o = (sc + sc[:, None, :]) / 2                                     # o[i, j] midpoint
        n = sc - sc[:, None, :]                                           # n[i, j] = c_j - c_i
        c_d = np.linalg.norm(n, axis=-1)
        iu = np.triu_indices(S, k=1)
        sv_con = np.vstack(iu).T                                          # sorted unique pairs (i < j), lexicographic
        sv_con = sv_con[self.mesh.contains(o[sv_con[:, 0], sv_con[:, 1], :])]
        _, d, _ = self.mesh.find_boundary(sc[sv_con[:, 0], :], n[sv_con[:, 0], sv_con[:, 1], :])
        sv_con = sv_con[d > 1, :]
        confirmed = np.zeros(sv_con.shape[0], dtype=bool)
        remove = np.zeros(sv_con.shape[0], dtype=bool)
        order = np.argsort(c_d[sv_con[:, 0], sv_con[:, 1]])
        for index, con in enumerate(sv_con[order, :]):
            k = order[index]
            if confirmed[k]:
                continue
            i_sv, j_sv = con
            for end in (i_sv, j_sv):
                if remove[k]:
                    break
                e_con = np.nonzero(np.any(sv_con == end, axis=1))[0]
                e_conf = e_con[confirmed[e_con]]
                for k_sv in sv_con[e_conf, :][sv_con[e_conf, :] != end]:
                    if np.sum((o[i_sv, j_sv, :] - o[end, k_sv, :]) * n[end, k_sv, :]) >= 0:
                        remove[k] = True
            if not remove[k]:
                confirmed[k] = True
        sv_con = sv_con[~remove, :]
        u_sv = np.unique(sv_con)                                          # only connected subvolumes survive
        self.subvol_center = sc[u_sv, :]
        new = np.zeros(sv_con.shape, dtype=int)
        for i, sv in enumerate(u_sv):
            new = np.where(sv_con == sv, i, new)
        self.subvol_connections = new
        self.n_of_subvols = self.subvol_center.shape[0]
        self.n_of_subvol_con = new.shape[0]
        self.subvol_con_vectors = self.subvol_center[new[:, 1], :] - self.subvol_center[new[:, 0], :]

    def calculate_subvol_volume(self, tol=1e-4, seed=20231003):
        """Exact V/S for boxes (Geometry.py:551-552); Monte-Carlo cover otherwise (:605-639), seeded."""
        if self.subvol_type in ('slice', 'grid') and self.shape in ('cuboid', 'box'):
            return self.volume * np.ones(self.n_of_subvols) / self.n_of_subvols
        rng = np.random.default_rng(seed)
        cover = np.zeros(self.n_of_subvols)
        nt, ns, err = 0, 1 << 14, 1.0
        while err > tol and nt < (1 << 24):
            r = self.subvol_classifier.predict(self.mesh.sample_volume(ns, rng))
            nr = np.bincount(r, minlength=self.n_of_subvols)
            new = (cover * nt + nr) / (nt + ns)
            nt += ns
            with np.errstate(divide='ignore', invalid='ignore'):
                e = np.abs((new - cover) / cover)
            e[np.isnan(e)] = 1
            err = e.max()
            cover = new
        return cover * self.volume

    def tables(self):
        """Mesh + BC arrays handed to nk_set_mesh."""
        t = self.mesh.tables()
        t.update(bound_cond=self.bound_cond, connected_facets=self.connected_facets)
        return t
