"""Triangle mesh with the attribute surface of the reference's `Mesh` that the hot path reads
(reference classes/Mesh.py; SURVEY.md section 8b "Path -> Geometry").

Built once on the host (NumPy); the tables go to HBM through nk_set_mesh.  Own implementation: orientation
by signed volume / ray parity, facets by union-find over coplanar neighbours, tetrahedra from a Delaunay
triangulation filtered by an inside test.
"""
import numpy as np


_DEVICE_HELPER = None


def _device_helper():
    """Is there a GPU (and the library) for the set-up helper nk_mesh_crossings?  NK_HOST_MESH=1 keeps everything in NumPy."""
    global _DEVICE_HELPER
    if _DEVICE_HELPER is None:
        import os
        ok = False
        if not os.environ.get('NK_HOST_MESH'):
            try:
                from . import engine
                ok = engine.device_count() > 0
            except Exception:
                ok = False
        _DEVICE_HELPER = ok
    return _DEVICE_HELPER


class Mesh(object):
    def __init__(self, vertices, faces, tol=1e-10):
        vertices = np.asarray(vertices, dtype=float)
        if vertices.shape[1] == 2:
            vertices = np.hstack((vertices, np.zeros((vertices.shape[0], 1))))
        self.vertices = vertices
        self.faces = np.asarray(faces, dtype=int)
        self.tol = tol                                       # Mesh.py:24
        self.update_mesh_properties()

    # ------------------------------------------------------------------ building
    def update_mesh_properties(self):
        self._drop_unreferenced()
        self._orient_outward()
        self._face_tables()
        self._facets()
        self._volume_tables()

    def _drop_unreferenced(self):
        used = np.unique(self.faces)
        remap = -np.ones(self.vertices.shape[0], dtype=int)
        remap[used] = np.arange(used.shape[0])
        self.vertices = self.vertices[used]
        self.faces = remap[self.faces]
        self.n_of_vertices = self.vertices.shape[0]

    def _raw_normals(self):
        v = self.vertices[self.faces]
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    def _orient_outward(self):
        """Flip faces (swap the first two vertices, as Mesh.check_winding does, Mesh.py:155-157) so that every
        normal points out of the solid.  Parity of ray crossings from the face centroid along its normal."""
        v = self.vertices[self.faces]
        cen = v.mean(axis=1)
        n = self._raw_normals()
        hits = self._count_crossings(cen, n, skip_self=True)
        flip = (hits % 2) == 1
        self.faces[flip] = self.faces[flip][:, [1, 0, 2]]

    def _count_crossings(self, origins, dirs, skip_self=False):
        """Number of triangles crossed by the open rays origin + t*dir, t > 0 (Moller-Trumbore, all pairs, written out
        by components: (rays, faces) planes instead of (rays, faces, 3) temporaries)."""
        v = self.vertices[self.faces]
        v0, e1, e2 = v[:, 0], v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        own = np.arange(origins.shape[0]) if skip_self else None            # ray i ignores face i
        if origins.shape[0] * v.shape[0] >= 4e6 and _device_helper():
            # large meshes: the same all-pairs test as a kernel (nk_mesh_crossings, same order of operations); the few rays
            # it gives up on (more distinct crossings than it keeps) take the NumPy form
            from .engine import mesh_crossings
            counts = mesh_crossings(origins, dirs, v0, e1, e2, skip_self).astype(int)
            redo = np.nonzero(counts < 0)[0]
            if redo.size:
                counts[redo] = self._count_crossings_host(origins[redo], dirs[redo], v0, e1, e2, None if own is None else own[redo])
            return counts
        return self._count_crossings_host(origins, dirs, v0, e1, e2, own)

    @staticmethod
    def _count_crossings_host(origins, dirs, v0, e1, e2, own=None):
        """The NumPy form; own[i]: the face ray i ignores (None: none)."""
        counts = np.zeros(origins.shape[0], dtype=int)
        step = max(1, int(2e6 // max(1, v0.shape[0])))
        e1x, e1y, e1z = e1[:, 0][None], e1[:, 1][None], e1[:, 2][None]
        e2x, e2y, e2z = e2[:, 0][None], e2[:, 1][None], e2[:, 2][None]
        for s in range(0, origins.shape[0], step):
            o = origins[s:s + step]
            d = dirs[s:s + step]
            dx, dy, dz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
            px, py, pz = dy * e2z - dz * e2y, dz * e2x - dx * e2z, dx * e2y - dy * e2x          # p = d x e2
            det = e1x * px + e1y * py + e1z * pz
            tx, ty, tz = o[:, 0:1] - v0[:, 0][None], o[:, 1:2] - v0[:, 1][None], o[:, 2:3] - v0[:, 2][None]
            with np.errstate(divide='ignore', invalid='ignore'):
                inv = 1.0 / det
                u = (tx * px + ty * py + tz * pz) * inv
                qx, qy, qz = ty * e1z - tz * e1y, tz * e1x - tx * e1z, tx * e1y - ty * e1x      # q = tv x e1
                w = (dx * qx + dy * qy + dz * qz) * inv
                t = (e2x * qx + e2y * qy + e2z * qz) * inv
            eps = 1e-9
            with np.errstate(invalid='ignore'):
                ok = (np.abs(det) > 1e-14) & (u >= -eps) & (w >= -eps) & (u + w <= 1 + eps) & (t > 1e-9)
            if own is not None:
                idx = own[s:s + step]
                ok[np.arange(idx.shape[0]), idx] = False
            # a ray through a shared edge/vertex would be counted once per triangle: merge equal distances
            rows, cols = np.nonzero(ok)
            if rows.size:
                tv = np.round(t[rows, cols], 8)
                order = np.lexsort((tv, rows))
                r_s, t_s = rows[order], tv[order]
                first = np.ones(r_s.shape[0], dtype=bool)
                first[1:] = (r_s[1:] != r_s[:-1]) | (t_s[1:] != t_s[:-1])
                counts[s:s + ok.shape[0]] = np.bincount(r_s[first], minlength=ok.shape[0])
        return counts

    def _face_tables(self):
        f = self.faces
        v = self.vertices[f]
        self.n_of_faces = f.shape[0]
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        cr = np.cross(e1, e2)
        self.face_areas = np.linalg.norm(cr, axis=1) / 2                               # Mesh.py:220
        self.face_centroid = v.mean(axis=1)
        self.area = self.face_areas.sum()
        self.face_normals = cr / np.linalg.norm(cr, axis=1, keepdims=True)             # Mesh.py:228-229
        self.face_basis = np.stack((e1, e2), axis=0)
        self.face_basis_matrix = np.stack((e1, e2, self.face_normals), axis=2)        # (F,3,3), columns e1 e2 n
        self.face_origins = v[:, 0].copy()                                             # Mesh.py:234
        self.face_k = -np.sum(self.face_normals * self.face_origins, axis=1)           # Mesh.py:323-324
        self.face_bounds = np.stack((v.min(axis=1), v.max(axis=1)), axis=0)            # (2,F,3) Mesh.py:238-242
        self.bounds = np.vstack((self.vertices.min(axis=0), self.vertices.max(axis=0)))
        self.extents = self.bounds[1] - self.bounds[0]
        # edge -> faces
        e = np.sort(np.concatenate((f[:, [0, 1]], f[:, [0, 2]], f[:, [1, 2]])), axis=1)
        owner = np.tile(np.arange(f.shape[0]), 3)
        self.edges, inv = np.unique(e, axis=0, return_inverse=True)
        inv = inv.ravel()
        self.n_of_edges = self.edges.shape[0]
        order = np.argsort(inv, kind='stable')
        splits = np.cumsum(np.bincount(inv, minlength=self.n_of_edges))[:-1]
        self.edges_faces = [np.sort(g) for g in np.split(owner[order], splits)]
        adj = [(g[a], g[b]) for g in self.edges_faces for a in range(len(g)) for b in range(a + 1, len(g))]
        self.face_adjacency = np.array(sorted(set(adj)), dtype=int).reshape(-1, 2)

    def _facets(self):
        """Facets = connected groups of coplanar adjacent faces (Mesh.get_facets_properties, Mesh.py:244-308),
        numbered by their lowest face index."""
        n, k = self.face_normals, self.face_k
        parent = np.arange(self.n_of_faces)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for a, b in self.face_adjacency:
            same_n = abs(np.dot(n[a], n[b])) > (1 - self.tol)
            same_k = abs(k[a]) - abs(k[b]) < self.tol                                   # Mesh.py:263-264 (one-sided)
            if same_n and same_k:
                ra, rb = find(a), find(b)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
        roots = np.array([find(i) for i in range(self.n_of_faces)])
        uniq = np.unique(roots)
        self.facets = [np.nonzero(roots == r)[0] for r in uniq]
        self.n_of_facets = len(self.facets)
        self.face_facets = np.searchsorted(uniq, roots)                                 # Mesh.py:314-321
        self.facets_normal = np.array([n[fct[0]] for fct in self.facets])
        self.facets_area = np.array([self.face_areas[fct].sum() for fct in self.facets])
        self.facet_centroid = np.array([np.sum(self.face_centroid[fct] * self.face_areas[fct][:, None], axis=0)
                                        / self.facets_area[i] for i, fct in enumerate(self.facets)])
        self.facets_origin = np.array([self.vertices[self.faces[fct[0], 0]] for fct in self.facets])
        self.facets_k = -np.sum(self.facets_normal * self.facets_origin, axis=1)

    def _volume_tables(self):
        """Volume, centre of mass, and a tetrahedralisation for uniform volume sampling
        (roles of Mesh.get_volume_properties / triangulate_volume, Mesh.py:354-568)."""
        v = self.vertices[self.faces]
        sv = np.einsum('ij,ij->i', v[:, 0], np.cross(v[:, 1], v[:, 2])) / 6.0
        self.volume = float(abs(sv.sum()))
        if self.n_of_facets < 2 or self.volume <= 0:
            self.n_of_simplices = 0
            self.simplices = np.zeros((0, 4), dtype=int)
            self.simplices_points = np.zeros((0, 3))
            self.simplices_volumes = np.zeros(0)
            self.center_mass = self.facet_centroid[0]
            return
        cen = (v[:, 0] + v[:, 1] + v[:, 2]) / 4.0
        self.center_mass = (cen * sv[:, None]).sum(axis=0) / sv.sum()
        from scipy.spatial import Delaunay
        pts = self.vertices
        tri = Delaunay(pts, qhull_options='Qbb Qc Qz Q12')
        simp = tri.simplices
        p = pts[simp]
        vol = np.abs(np.einsum('ij,ij->i', p[:, 1] - p[:, 0], np.cross(p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]))) / 6.0
        keep = vol > 1e-6
        keep[keep] = self.contains(p[keep].mean(axis=1))
        self.simplices = simp[keep]
        self.simplices_points = pts
        self.simplices_volumes = vol[keep]
        self.n_of_simplices = int(keep.sum())

    # ------------------------------------------------------------------ transforms
    def rezero(self):
        dx = self.vertices.min(axis=0)                                                  # Mesh.py:40-56
        self.vertices = self.vertices - dx
        self.update_mesh_properties()

    # ------------------------------------------------------------------ queries
    def contains(self, x):
        """Inside test by ray parity (three directions, majority vote)."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        dirs = np.array([[0.5377, 0.2821, 0.7946], [-0.3199, 0.8847, 0.3390], [0.7071, -0.6124, 0.3536]])
        votes = np.zeros(x.shape[0], dtype=int)
        for d in dirs:
            d = d / np.linalg.norm(d)
            votes += self._count_crossings(x, np.tile(d, (x.shape[0], 1))) % 2
        return votes >= 2

    contains_naive = contains

    def find_boundary(self, x, v):
        """Mesh.find_boundary (Mesh.py:806-856) on the host -- setup use only; the hot path runs it on the GPU."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        v = np.atleast_2d(np.asarray(v, dtype=float))
        n, k, tol = self.face_normals, self.face_k, self.tol
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            # products summed in the order of the reference's np.sum(..., axis=2) and without fused multiply-adds (a BLAS
            # product may fuse): x.n + k cancels near a face, so the rounding shows in t
            xn = (x[:, 0:1] * n[:, 0] + x[:, 1:2] * n[:, 1]) + x[:, 2:3] * n[:, 2]
            vn = (v[:, 0:1] * n[:, 0] + v[:, 1:2] * n[:, 1]) + v[:, 2:3] * n[:, 2]
            t = -(xn + k) / vn
        ok = (t >= tol) & np.isfinite(t)
        ip, jf = np.nonzero(ok)
        c = x[ip] + t[ip, jf][:, None] * v[ip]
        inb = np.all(c >= self.face_bounds[0, jf] - tol, axis=1) & np.all(c <= self.face_bounds[1, jf] + tol, axis=1)
        ok[ip, jf] = inb
        ip, jf, c = ip[inb], jf[inb], c[inb]
        bar = np.linalg.solve(self.face_basis_matrix[jf], (c - self.face_origins[jf])[:, :, None])[:, :2, 0]
        bar = np.concatenate((bar, 1 - bar.sum(axis=1, keepdims=True)), axis=1)
        ok[ip, jf] = np.all((bar >= -tol) & (bar <= 1 + tol), axis=1)
        t = np.where(ok, t, np.inf)
        tc = t.min(axis=1)
        fc = self.face_facets[np.argmax(t == tc[:, None], axis=1)].astype(int)
        fc[np.isinf(tc)] = -1
        with np.errstate(invalid='ignore'):
            xc = x + tc[:, None] * v
        return xc, tc, fc

    def closest_face(self, x):
        """Nearest face among those whose plane projection falls inside the triangle (Mesh.py:686-720)."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        n, o, tol = self.face_normals, self.face_origins, self.tol
        dist = np.einsum('pfd,fd->pf', x[:, None, :] - o[None], n)
        pj = x[:, None, :] - n[None] * dist[:, :, None]
        valid = np.all(pj >= self.face_bounds[0][None] - tol, axis=2) & np.all(pj <= self.face_bounds[1][None] + tol, axis=2)
        ip, jf = np.nonzero(valid)
        bar = np.linalg.solve(self.face_basis_matrix[jf], (pj[ip, jf] - o[jf])[:, :, None])[:, :2, 0]
        bar = np.concatenate((bar, 1 - bar.sum(axis=1, keepdims=True)), axis=1)
        valid[ip, jf] = np.all((bar >= -tol) & (bar <= 1 + tol), axis=1)
        d = np.where(valid, np.abs(dist), np.inf)
        f = np.argmin(d, axis=1)
        dmin = d[np.arange(x.shape[0]), f]
        f = np.where(np.isinf(dmin), -1, f)
        return f.astype(int), dmin, pj[np.arange(x.shape[0]), f]

    def closest_facet(self, x):
        f, d, xc = self.closest_face(x)                                                 # Mesh.py:722-729
        out = np.where(f >= 0, self.face_facets[np.clip(f, 0, None)], -1)
        return out.astype(int), d, xc

    # ------------------------------------------------------------------ sampling (host, initialisation only)
    def sample_volume(self, n, rng=np.random):
        """Uniform points in the solid: tetrahedron ~ volume, Dirichlet(1,1,1,1) weights (Mesh.py:890-904)."""
        if self.n_of_simplices == 0:
            raise Exception('Number of simplices is zero. The mesh may be a plane and has no volume to sample from.')
        p = self.simplices_volumes / self.simplices_volumes.sum()
        s = rng.choice(self.n_of_simplices, size=n, p=p)
        v = self.simplices_points[self.simplices[s]]
        a = -np.log(rng.random((n, 4, 1)))
        a /= a.sum(axis=1, keepdims=True)
        return np.sum(a * v, axis=1)

    def sample_surface(self, n, facets=None, rng=np.random):
        """Uniform points on the given facets (Mesh.py:923-951)."""
        if facets is None:
            faces = np.arange(self.n_of_faces)
        else:
            faces = np.concatenate([self.facets[int(f)] for f in np.atleast_1d(facets)])
        p = self.face_areas[faces] / self.face_areas[faces].sum()
        f = rng.choice(faces, size=n, p=p)
        v = self.vertices[self.faces[f]]
        s = rng.random((n, 1)) ** 0.5
        r = rng.random((n, 1))
        return (1 - s) * v[:, 0] + (1 - r) * s * v[:, 1] + r * s * v[:, 2]

    def export_stl(self, name, path='.'):
        """ASCII STL in the layout of Mesh.export_stl (Mesh.py:953-975)."""
        import os
        name = name.replace('.stl', '')
        lines = ['solid %s' % name]
        for f in range(self.n_of_faces):
            lines.append('facet normal {:.6e} {:.6e} {:.6e}'.format(*self.face_normals[f]))
            lines.append('    outer loop')
            for k in range(3):
                lines.append('        vertex {:.6e} {:.6e} {:.6e}'.format(*self.vertices[self.faces[f, k]]))
            lines.append('    endloop')
            lines.append('endfacet')
        lines.append('endsolid %s' % name)
        with open(os.path.join(path, name + '.stl'), 'w') as fh:
            fh.write('\n'.join(lines))

    def tables(self):
        """Arrays handed to nk_set_mesh (reference attribute names)."""
        return dict(vertices=self.vertices, faces=self.faces, face_normals=self.face_normals, face_k=self.face_k,
                    face_bounds=self.face_bounds, face_basis_matrix=self.face_basis_matrix,
                    face_origins=self.face_origins, face_facets=self.face_facets, face_areas=self.face_areas,
                    facets_normal=self.facets_normal, facet_centroid=self.facet_centroid, facets=self.facets,
                    bounds=self.bounds, simplices_points=self.simplices_points, simplices=self.simplices,
                    simplices_volumes=self.simplices_volumes, tol=self.tol)


def read_stl(path):
    """ASCII or binary STL -> (vertices, faces) with coincident vertices merged (role of trimesh.load at
    Geometry.py:82-84; vertices rounded to 10 decimals as there)."""
    with open(path, 'rb') as fh:
        raw = fh.read()
    tri = None
    head = raw[:512].lstrip().lower()
    if head.startswith(b'solid') and b'facet' in raw[:4096].lower():
        vals = []
        for line in raw.decode('ascii', errors='ignore').splitlines():
            parts = line.split()
            if len(parts) == 4 and parts[0] == 'vertex':
                vals.append([float(parts[1]), float(parts[2]), float(parts[3])])
        tri = np.array(vals).reshape(-1, 3, 3)
    else:
        n = int(np.frombuffer(raw[80:84], dtype='<u4')[0])
        rec = np.frombuffer(raw[84:84 + 50 * n], dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]))
        tri = rec['v'].astype(float)
    pts = np.around(tri.reshape(-1, 3), decimals=10)
    verts, inv = np.unique(pts, axis=0, return_inverse=True)
    return verts, inv.reshape(-1, 3)
