"""Per-facet x per-mode tables consumed by the hot loop (SURVEY.md section 8 row a2/a7 inputs, 8f row 1).

Host-side NumPy restatements of the reference's one-off builders in classes/Population.py:
  enter_probability            :146-161
  calculate_fbz_specularity    :852-877
  find_specular_correspondences ('velocity' model) :1241-1454, specular_function :1457
  diffuse_scat_probability     :879-939
  find_degeneracies            :1017-1040
  find_specular_correspondences ('k' model)        :1058-1239
With an engine the 'velocity' tables are built on the device (`rough_tables_device`): pair search, specularity, specular map,
creation rates and roulette never leave HBM; the NumPy functions below are their CPU statements, pinned to the goldens.
"""
import numpy as np


def enter_probability(geometry, phonon, res_facet, particle_density, dt):
    """p[r,q,j] = max(0, v.n_in) * dt / bound_thickness, bound_thickness = M / (rho * A_facet)."""
    thick = phonon.number_of_active_modes / (particle_density * geometry.facets_area[res_facet])
    normals = -geometry.facets_normal[res_facet, :]                      # inward
    vpar = np.einsum('rd,qjd->rqj', normals, phonon.group_vel)
    p = vpar * dt / thick.reshape(-1, 1, 1)
    return np.where(p < 0, 0, p)


def fbz_specularity(geometry, phonon, rough_facets, eta):
    """exp(-(2 eta cos(theta))^2 k^2) per (facet, q, j)."""
    n = -geometry.facets_normal[rough_facets, :]
    k_norm = np.sum(phonon.wavevectors ** 2, axis=1) ** 0.5
    v = phonon.group_vel
    v_norm = np.sum(v ** 2, axis=-1) ** 0.5
    dot = np.einsum('fd,qjd->fqj', n, v)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        cos = dot / v_norm[None]
    const = -(2 * np.asarray(eta, dtype=float).reshape(-1, 1, 1) * cos) ** 2
    spec = np.exp(const * (k_norm.reshape(1, -1, 1) ** 2))
    spec[np.isnan(spec)] = 0
    return spec


def find_degeneracies(phonon):
    """[q, j1, j2] rows with equal omega at the same q (j1 < j2), and the (Q,J) index table."""
    om = phonon.omega
    J = om.shape[1]
    rows = []
    for j1 in range(J):
        for j2 in range(j1 + 1, J):
            q = np.nonzero(np.abs(om[:, j1] - om[:, j2]) < 1e-10)[0]
            rows += [(int(a), j1, j2) for a in q]
    deg = np.array(sorted(rows), dtype=int).reshape(-1, 3)
    idx = -np.ones(om.shape)
    if deg.shape[0]:
        idx[deg[:, 0], deg[:, 1]] = np.arange(deg.shape[0])
        idx[deg[:, 0], deg[:, 2]] = np.arange(deg.shape[0])
    return deg, idx


def specular_correspondences_velocity(geometry, phonon, rough_facets, crit=1e-3, keep_nan_pairs=False, engine=None):
    """Pairs (in-mode -> out-mode) whose mirrored group velocity and frequency agree within the grid tolerance.

    Returns (correspondent_modes (K,7): n(3) q_in j_in q_out j_out, true_spec (Fr,Q,J) bool).

    With `engine` (a nanokappa_amd.engine.Engine) the pair search of every normal runs on the GPU
    (nk_specular_pairs: same arithmetic, same pair set); the NumPy path below is its CPU statement and the one the
    reference goldens pin.
    """
    normals = -np.round(geometry.facets_normal[rough_facets, :], decimals=10)
    normals, inv_normals = np.unique(normals, axis=0, return_inverse=True)
    inv_normals = np.asarray(inv_normals).ravel()
    v = phonon.group_vel
    Q, J = phonon.omega.shape
    true_spec = np.zeros((len(rough_facets), Q, J), dtype=bool)
    k_grid = phonon.q_to_k(np.absolute(1 / (2 * phonon.data_mesh)))
    delta_omega = np.sum((v * k_grid) ** 2, axis=2) ** 0.5
    rows = []
    if engine is not None and not keep_nan_pairs:
        engine.specular_begin(v.reshape(-1, 3), phonon.omega.ravel(), delta_omega.ravel())
    for i_n, n in enumerate(normals):
        vdn = np.sum(v * n, axis=2)
        in_modes = np.vstack(np.nonzero(vdn < 0)).T
        out_modes = np.vstack(np.nonzero(vdn > 0)).T
        v_in = v[in_modes[:, 0], in_modes[:, 1], :]
        v_ref = v_in - 2 * n * np.sum(v_in * n, axis=1, keepdims=True)
        if engine is not None and not keep_nan_pairs:
            gi, go = engine.specular_pairs(n, crit)
            pi = np.searchsorted(in_modes[:, 0] * J + in_modes[:, 1], gi)        # positions in the in / out lists
            po = np.searchsorted(out_modes[:, 0] * J + out_modes[:, 1], go)
            rows.append(_corr_rows(n, pi, po, in_modes, out_modes, v_ref, true_spec, inv_normals, i_n))
            continue
        v_out = v[out_modes[:, 0], out_modes[:, 1], :]
        nrm_in = np.linalg.norm(v_ref, axis=1)
        nrm_out = np.linalg.norm(v_out, axis=1)
        om_in = phonon.omega[in_modes[:, 0], in_modes[:, 1]]
        om_out = phonon.omega[out_modes[:, 0], out_modes[:, 1]]
        d_in = delta_omega[in_modes[:, 0], in_modes[:, 1]]
        d_out = delta_omega[out_modes[:, 0], out_modes[:, 1]]
        # candidate window in the sorted x-velocities, then the exact per-pair criteria
        order = np.argsort(v_out[:, 0], kind='stable')
        sx = v_out[order, 0]
        vmax = max(nrm_in.max(), nrm_out.max())
        lo = np.searchsorted(sx, v_ref[:, 0] - crit * vmax, side='left')
        hi = np.searchsorted(sx, v_ref[:, 0] + crit * vmax, side='right')
        pi, po = [], []
        for a in range(in_modes.shape[0]):
            if hi[a] > lo[a]:
                cand = order[lo[a]:hi[a]]
                ref = np.fmax(nrm_in[a], nrm_out[cand])
                ok = np.all(np.abs(v_ref[a] - v_out[cand]) / ref[:, None] < crit, axis=1)
                ok &= np.abs(om_in[a] - om_out[cand]) < d_in[a] + d_out[cand]
                cand = cand[ok]
                if cand.size:
                    # angle test exactly as the reference evaluates it (Population.py:1357-1369): normalise the
                    # un-mirrored in-velocity, mirror it, dot with the normalised out-velocity.  When rounding
                    # pushes the dot product above 1 arccos is NaN and the reference REJECTS the pair (angle := pi);
                    # that drops ~1/4 of the perfectly aligned pairs, and is reproduced for table parity.
                    u_in = v_in[a] / np.sqrt(v_in[a, 0] ** 2 + v_in[a, 1] ** 2 + v_in[a, 2] ** 2)
                    vo = v_out[cand]
                    u_out = vo / np.sqrt(vo[:, 0] ** 2 + vo[:, 1] ** 2 + vo[:, 2] ** 2)[:, None]
                    u_try = u_in - 2 * n * np.sum(u_in * n)
                    with np.errstate(invalid='ignore'):
                        ang = np.arccos(np.sum(u_try * u_out, axis=1))
                    if keep_nan_pairs:
                        ang = np.where(np.isnan(ang), 0.0, ang)
                    else:
                        ang[np.isnan(ang)] = np.pi
                    cand = cand[ang < crit]
                    pi += [a] * cand.size
                    po += list(np.sort(cand))
        rows.append(_corr_rows(n, np.array(pi, dtype=int), np.array(po, dtype=int), in_modes, out_modes, v_ref, true_spec,
                               inv_normals, i_n))
    if engine is not None and not keep_nan_pairs:
        engine.specular_end()
    corr = np.vstack(rows) if rows else np.zeros((0, 7))
    return corr, true_spec


def _corr_rows(n, pi, po, in_modes, out_modes, v_ref, true_spec, inv_normals, i_n):
    """Rows of correspondent_modes for one normal in the reference's order (by sorted reflected-vx of the in-mode, then
    ascending out index within it) and the true_specular mask of the facets that share the normal (None: not wanted)."""
    key = np.lexsort((po, np.argsort(np.argsort(v_ref[:, 0], kind='stable'))[pi])) if pi.size else np.zeros(0, dtype=int)
    pi, po = pi[key], po[key]
    im, om_ = in_modes[pi], out_modes[po]
    if true_spec is not None:
        for fct in np.nonzero(inv_normals == i_n)[0]:
            true_spec[fct, im[:, 0], im[:, 1]] = True
    return np.hstack((np.tile(n, (im.shape[0], 1)), im, om_))


def _distinct_normals(geometry, rough_facets):
    normals = -np.round(geometry.facets_normal[np.asarray(rough_facets), :], decimals=10)
    normals, inv_normals = np.unique(normals, axis=0, return_inverse=True)
    return normals, np.asarray(inv_normals).ravel()


def _pair_rows(engine, phonon, normals, inv_normals, i_n, crit, after_pairs=None):
    """correspondent_modes rows of one normal from the device's pair search (after_pairs: called between the search and the
    download of the next one, while the pairs are still on the device)."""
    v = phonon.group_vel
    J = phonon.omega.shape[1]
    n = normals[i_n]
    vdn = np.sum(v * n, axis=2)
    in_modes = np.vstack(np.nonzero(vdn < 0)).T
    out_modes = np.vstack(np.nonzero(vdn > 0)).T
    v_in = v[in_modes[:, 0], in_modes[:, 1], :]
    v_ref = v_in - 2 * n * np.sum(v_in * n, axis=1, keepdims=True)
    gi, go = engine.specular_pairs(n, crit)
    if after_pairs is not None:
        after_pairs()
    pi = np.searchsorted(in_modes[:, 0] * J + in_modes[:, 1], gi)
    po = np.searchsorted(out_modes[:, 0] * J + out_modes[:, 1], go)
    return _corr_rows(n, pi, po, in_modes, out_modes, v_ref, None, inv_normals, i_n)


def _spec_begin(engine, phonon):
    v = phonon.group_vel
    k_grid = phonon.q_to_k(np.absolute(1 / (2 * phonon.data_mesh)))
    delta_omega = np.sum((v * k_grid) ** 2, axis=2) ** 0.5
    engine.specular_begin(v.reshape(-1, 3), phonon.omega.ravel(), delta_omega.ravel())


def rough_tables_device(engine, geometry, phonon, rough_facets, eta, crit=1e-3, want_rows=True):
    """The 'velocity' reflection tables built and kept on the device (nk_rough_begin / nk_rough_pairs / nk_rough_finish):
    for every distinct normal the pair search (nk_specular_pairs) and, straight from the device-resident pairs, the
    truly-specular mask, the specular map and what the pairs take out of the diffuse creation rates; then specularity,
    rates, roulette and its bucket index for all facets at once.  Only the pairs come back to the host, and only when
    wanted (they are the rows of `correspondent_modes`, which the reference writes to specular_correspondences.txt; on a
    mesh with a thousand distinct normals sorting them into the reference's order is most of the set-up time):
    returns the rows, or None (specular_rows_device builds them later)."""
    rough_facets = np.asarray(rough_facets)
    normals, inv_normals = _distinct_normals(geometry, rough_facets)
    k_norm = np.sum(phonon.wavevectors ** 2, axis=1) ** 0.5
    _spec_begin(engine, phonon)
    engine.rough_begin(rough_facets, -geometry.facets_normal[rough_facets, :], np.asarray(eta, dtype=float).ravel(), k_norm)
    order = np.argsort(inv_normals, kind='stable')                   # the rough facets of every normal, in one pass
    first = np.searchsorted(inv_normals[order], np.arange(normals.shape[0] + 1))
    rows = []
    for i_n in range(normals.shape[0]):
        share = order[first[i_n]:first[i_n + 1]]
        if want_rows:
            rows.append(_pair_rows(engine, phonon, normals, inv_normals, i_n, crit, after_pairs=lambda: engine.rough_pairs(share)))
        else:
            engine.specular_pairs(normals[i_n], crit, download=False)
            engine.rough_pairs(share)
    engine.rough_finish()
    engine.specular_end()
    if not want_rows:
        return None
    return np.vstack(rows) if rows else np.zeros((0, 7))


def degen_partner(phonon, degeneracies, degen_index):
    """(Q, J) partner branch of the 'k' model's coin flip (Population.py:963-969): the third column of the degeneracy row
    a mode belongs to, -1 for the others."""
    Q, J = phonon.omega.shape
    j2 = -np.ones((Q, J), dtype=np.int32)
    di = np.asarray(degen_index).astype(int)
    has = di > -1
    j2[has] = np.asarray(degeneracies)[di[has], 2]
    return j2


def _kspec_begin(engine, phonon):
    _spec_begin(engine, phonon)
    R = np.asarray(phonon.reciprocal_lattice, dtype=float)
    tol = phonon.q_to_k(np.absolute(1 / (2 * phonon.data_mesh)))
    engine.kspec_begin(phonon.wavevectors, np.linalg.inv(R).T, R.T, tol)          # Phonon.k_to_q / q_to_k as q = k . A, k = q . B


def _k_rows(n, gi, go, J):
    """correspondent_modes rows of one normal, 'k' model: in (q, j) order, as np.nonzero walks the reference's table."""
    o = np.argsort(gi, kind='stable')
    gi, go = gi[o], go[o]
    return np.vstack((np.ones(gi.shape[0]) * n.reshape(-1, 1), gi // J, gi % J, go // J, go % J)).T


def rough_tables_device_k(engine, geometry, phonon, rough_facets, eta, degeneracies, degen_index, want_rows=True):
    """rough_tables_device for the 'k' / wavevector model (Population.py:1056-1240): the pair search of every distinct normal
    as a kernel (nk_kspec_pairs), the tables from the device-resident pairs as for the 'velocity' model, creation rates of
    degenerate branches averaged before the roulette (:926-930)."""
    rough_facets = np.asarray(rough_facets)
    normals, inv_normals = _distinct_normals(geometry, rough_facets)
    J = phonon.omega.shape[1]
    k_norm = np.sum(phonon.wavevectors ** 2, axis=1) ** 0.5
    _kspec_begin(engine, phonon)
    engine.rough_begin(rough_facets, -geometry.facets_normal[rough_facets, :], np.asarray(eta, dtype=float).ravel(), k_norm)
    order = np.argsort(inv_normals, kind='stable')
    first = np.searchsorted(inv_normals[order], np.arange(normals.shape[0] + 1))
    rows = []
    for i_n in range(normals.shape[0]):
        pairs = engine.kspec_pairs(normals[i_n], download=want_rows)
        engine.rough_pairs(order[first[i_n]:first[i_n + 1]])
        if want_rows:
            rows.append(_k_rows(normals[i_n], pairs[0], pairs[1], J))
    engine.rough_finish(degeneracies, degen_partner(phonon, degeneracies, degen_index).ravel())
    engine.specular_end()
    if not want_rows:
        return None
    return np.vstack(rows) if rows else np.zeros((0, 7))


def specular_rows_device_k(engine, geometry, phonon, rough_facets):
    normals, _ = _distinct_normals(geometry, rough_facets)
    J = phonon.omega.shape[1]
    _kspec_begin(engine, phonon)
    rows = [_k_rows(n, *engine.kspec_pairs(n), J) for n in normals]
    engine.specular_end()
    return np.vstack(rows) if rows else np.zeros((0, 7))


def specular_rows_device(engine, geometry, phonon, rough_facets, crit=1e-3):
    """`correspondent_modes` (K,7) alone, from the device's pair search (for rough_tables_device(want_rows=False) callers)."""
    normals, inv_normals = _distinct_normals(geometry, rough_facets)
    _spec_begin(engine, phonon)
    rows = [_pair_rows(engine, phonon, normals, inv_normals, i_n, crit) for i_n in range(normals.shape[0])]
    engine.specular_end()
    return np.vstack(rows) if rows else np.zeros((0, 7))


def specular_correspondences_k(geometry, phonon, rough_facets):
    """'k' / 'wavevector' reflection model (Population.py:1056-1240): an in-mode (q, j) is specular when the mirrored
    wavevector k - 2 n (k.n) stays in the first Brillouin zone (a normal process), lands on a grid point within half
    a grid step, and that q-point has an outgoing branch whose frequency interval overlaps; of the overlapping
    branches the one with the smallest relative frequency difference is taken.

    Returns (correspondent_modes (K,7): n(3) q_in j_in q_out j_out, true_spec (Fr,Q,J) bool).
    """
    from scipy.interpolate import NearestNDInterpolator
    normals = -np.round(geometry.facets_normal[rough_facets, :], decimals=10)
    normals, inv_normals = np.unique(normals, axis=0, return_inverse=True)
    inv_normals = np.asarray(inv_normals).ravel()
    k = phonon.wavevectors
    v = phonon.group_vel
    Q, J = phonon.omega.shape
    true_spec = np.zeros((len(rough_facets), Q, J), dtype=bool)
    tol = phonon.q_to_k(np.absolute(1 / (2 * phonon.data_mesh)))
    near_k = NearestNDInterpolator(k, np.arange(Q))
    rows = []

    def mirror(active):
        return k[active, :] - 2 * n * np.sum(k[active, :] * n, axis=1, keepdims=True)

    for i_n, n in enumerate(normals):
        vdn = np.sum(v * n, axis=2)
        s_in, s_out = vdn < 0, vdn > 0
        active = np.any(s_in, axis=1)
        _, disp = phonon.find_min_k(mirror(active), return_disp=True)
        active[active] = np.all(disp == 0, axis=1)                       # normal processes only
        k_try = mirror(active)
        q_near = near_k(k_try).astype(int)
        k_dist = np.absolute(k_try - k[q_near, :])
        active[active] = np.logical_and(np.any(s_out[q_near, :], axis=1), np.all(k_dist < tol, axis=1))
        out_q = near_k(mirror(active)).astype(int)
        in_q = np.arange(Q)[active]
        valid_v = np.logical_and(s_in[in_q, :], np.transpose(s_out[out_q, :][None], (2, 1, 0)))     # (Jout, Qa, Jin)
        keep = np.any(valid_v, axis=(0, 2))
        active[active] = keep
        valid_v = valid_v[:, keep, :]
        out_q = near_k(mirror(active)).astype(int)
        in_q = np.arange(Q)[active]
        in_delta = np.sum(np.absolute(v[in_q, :, :]) * tol[None, None, :], axis=2)                  # (Qa, J)
        out_delta = np.sum(np.absolute(v[out_q, :, :]) * tol[None, None, :], axis=2)
        in_om, out_om = phonon.omega[in_q, :], phonon.omega[out_q, :]
        in_up, in_dn = in_om + in_delta, in_om - in_delta
        out_up = np.transpose((out_om + out_delta)[None], (2, 1, 0))                                # (J, Qa, 1)
        out_dn = np.transpose((out_om - out_delta)[None], (2, 1, 0))
        overlap = (np.where(in_up < out_up, in_up, out_up) - np.where(in_dn > out_dn, in_dn, out_dn)) > 0
        with np.errstate(divide='ignore', invalid='ignore'):
            om_diff = np.absolute((in_om - np.transpose(out_om[None], (2, 1, 0))) / in_om)          # (Jout, Qa, Jin)
        om_diff = np.where(overlap, om_diff, np.inf)
        valid = np.logical_and(overlap, valid_v)
        vk = np.any(valid, axis=(0, 2))
        active[active] = vk
        valid = valid[:, vk, :]
        om_diff = np.where(valid, om_diff[:, vk, :], np.inf)
        min_diff = np.amin(om_diff, axis=0)                                                         # (Qa, Jin)
        branch = np.where(np.any(valid, axis=0), np.argmax(om_diff == min_diff, axis=0), -1).astype(int)
        in_q = np.arange(Q)[active]
        out_q = near_k(mirror(active)).astype(int)
        for f in np.nonzero(inv_normals == i_n)[0]:
            true_spec[f][in_q, :] = branch != -1
        iq, j_in = np.nonzero(branch != -1)
        rows.append(np.vstack((np.ones(iq.shape[0]) * n.reshape(-1, 1), in_q[iq], j_in, out_q[iq], branch[iq, j_in])).T)
    corr = np.vstack(rows) if rows else np.zeros((0, 7))
    return corr, true_spec


def _corr_blocks(corr):
    """{normal (tuple): slice of corr} -- the builders append one contiguous block of rows per distinct normal."""
    if corr.shape[0] == 0:
        return {}
    chg = np.nonzero(np.any(corr[1:, :3] != corr[:-1, :3], axis=1))[0] + 1
    starts = np.concatenate(([0], chg))
    ends = np.concatenate((chg, [corr.shape[0]]))
    blocks = {}
    for a, b in zip(starts, ends):
        key = tuple(corr[a, :3] + 0.0)          # + 0.0: -0.0 and 0.0 are the same key
        assert key not in blocks, 'rows of one normal must be contiguous'
        blocks[key] = slice(int(a), int(b))
    return blocks


def specular_map(corr, geometry, rough_facets, Q, J):
    """Flat out-mode per (rough facet, q, j), -1 where the mode has no specular partner.  When an in-mode has
    several partners the reference's nearest-neighbour lookup (Population.py:1457) returns one of them; here the
    one with the smallest (q_out, j_out) is taken."""
    out = -np.ones((len(rough_facets), Q, J), dtype=np.int64)
    if corr.shape[0] == 0:
        return out
    normals = -np.round(geometry.facets_normal[rough_facets, :], decimals=10)
    blocks = _corr_blocks(corr)
    for i, n in enumerate(normals):
        sl = blocks.get(tuple(n + 0.0))
        if sl is None:
            continue
        c = corr[sl].astype(np.int64)
        order = np.lexsort((c[:, 6], c[:, 5]))[::-1]        # descending, so the smallest is written last
        c = c[order]
        out[i, c[:, 3], c[:, 4]] = c[:, 5] * J + c[:, 6]
    return out


def diffuse_roulette(geometry, phonon, rough_facets, specularity, corr, scat_model='velocity', degeneracies=None):
    """creation_rate and its cumulative 'roulette' per rough facet."""
    n = -geometry.facets_normal[rough_facets, :]
    vdn = np.einsum('fd,qjd->fqj', n, phonon.group_vel)
    C_total = np.where(vdn > 0, vdn, 0)
    D_total = np.where(vdn < 0, -vdn, 0)
    specular_D = D_total * specularity
    rate = np.where(np.isnan(C_total), 0, C_total).copy()
    if corr.shape[0]:
        in_q, in_j = corr[:, 3].astype(int), corr[:, 4].astype(int)
        out_q, out_j = corr[:, 5].astype(int), corr[:, 6].astype(int)
        un, inv_n = np.unique(n, axis=0, return_inverse=True)
        inv_n = np.asarray(inv_n).ravel()
        blocks = _corr_blocks(corr)
        for i_n, u in enumerate(un):
            sel = blocks.get(tuple(np.round(u, decimals=10) + 0.0))
            if sel is None:
                continue
            for f in np.nonzero(inv_n == i_n)[0]:
                np.subtract.at(rate[f], (out_q[sel], out_j[sel]), specular_D[f, in_q[sel], in_j[sel]])
    if scat_model in ('k', 'wavevector', 'wave_vector') and degeneracies is not None:
        for q, j1, j2 in degeneracies:
            rate[:, q, [j1, j2]] = rate[:, q, [j1, j2]].mean(axis=-1, keepdims=True)
    rate = np.around(rate, decimals=10)
    Fr = len(rough_facets)
    roul = np.zeros((Fr, phonon.number_of_qpoints * phonon.number_of_branches))
    for f in range(Fr):
        c = np.cumsum(rate[f])
        roul[f] = c / c.max()
    return rate, roul


def rbf_system(centers, used=(True, True, True)):
    """Inverse of the linear system behind scipy's RBFInterpolator(centers, T, kernel='cubic') -- the temperature
    interpolator of non-slice subvolumes with --temp_interp radial / linear (Population.py:573-590, :651-656).

    T(x) = sum_i w_i |x - c_i|^3 + p_0 + sum_k p_k (x_k - shift_k) / scale_k over the used coordinates, with
    [w; p] = inv @ [T_sv; 0].  The centres do not move, so the inverse is computed once and every new T_sv costs one
    matrix-vector product.  Returns (inv (P,P), shift (3,), scale (3,), used (3,) int32) with P = S + n_used + 1.
    """
    c = np.asarray(centers, dtype=float)
    used = np.asarray(used, dtype=bool)
    y = c[:, used]
    S, nd = y.shape
    mins, maxs = y.min(axis=0), y.max(axis=0)
    shift = (maxs + mins) / 2
    scale = (maxs - mins) / 2
    scale[scale == 0.0] = 1.0
    r = np.sqrt(((y[:, None, :] - y[None, :, :]) ** 2).sum(axis=2))
    P = S + nd + 1
    lhs = np.zeros((P, P))
    lhs[:S, :S] = r ** 3
    poly = np.hstack((np.ones((S, 1)), (y - shift) / scale))
    lhs[:S, S:] = poly
    lhs[S:, :S] = poly.T
    sh, sc = np.zeros(3), np.ones(3)
    sh[used], sc[used] = shift, scale
    return np.linalg.inv(lhs), sh, sc, used.astype(np.int32)


def rbf_evaluate(inv, shift, scale, used, centers, T_sv, x):
    """NumPy statement of the device / oracle evaluation (used by the tests)."""
    used = np.asarray(used, dtype=bool)
    S = centers.shape[0]
    coef = inv[:, :S] @ np.asarray(T_sv, dtype=float)
    d = np.asarray(x, dtype=float)[:, None, used] - centers[None, :, used]
    r2 = (d ** 2).sum(axis=2)
    out = (r2 * np.sqrt(r2)) @ coef[:S] + coef[S]
    xh = (np.asarray(x, dtype=float)[:, used] - shift[used]) / scale[used]
    return out + xh @ coef[S + 1:]
