/* nk_oracle.c -- CPU restatement (plain C, scalar, one particle at a time) of the
 * reference's Population timestep loop.  TEST INFRASTRUCTURE ONLY -- see nk_oracle.h.
 *
 * The reference is whole-population NumPy; this file states the same arithmetic per
 * particle.  Citations are file:line under /root/reference.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile) so that no FMA
 * contraction changes the reference's rounding.
 */
#include "nk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RNG ---- */
/* Philox4x32-10 (Salmon et al., SC'11).  The reference uses NumPy's global MT19937
 * in data-dependent vector order (Population.py:949, :967, :1003; Mesh.py:894-898,
 * :937-942), which no parallel code can replay; parity on random decisions is
 * statistical.  Oracle and HIP kernels share THIS generator
 * and keying so that they agree decision for decision. */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}
void nko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u53(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0);   /* [0,1), as np.random.rand */
}
void nko_uniform2(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *u0, double *u1) {
    uint32_t ctr[4] = {(uint32_t)pid, (uint32_t)(pid >> 32), step, tag};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    nko_philox4x32_10(ctr, key, o);
    *u0 = u53(o[0], o[1]);
    *u1 = u53(o[2], o[3]);
}
#define TAG_REFLECT 0x00000u   /* + event index within the step */
#define TAG_EMIT    0x10000u   /* +0: (face, s)  +1: (r, r_dt) */
#define TAG_RESAMP  0x20000u   /* +0: (simplex, a0) +1: (a1,a2) +2: (a3, -) */
#define TAG_DICE    0x30000u

/* ------------------------------------------------------- find_boundary ---- */
/* Mesh.find_boundary, Mesh.py:806-856 */
static void find_boundary_one(const nko_mesh *m, const double x[3], const double v[3],
                              double xc[3], double *tc, int32_t *fc) {
    const double tol = m->tol;
    double tbest = INFINITY;
    int32_t fbest = -1;
    for (int32_t f = 0; f < m->F; ++f) {
        const double *n = m->normals + 3 * f;
        double num = (x[0] * n[0] + x[1] * n[1]) + x[2] * n[2];
        double den = (v[0] * n[0] + v[1] * n[1]) + v[2] * n[2];
        double t = -(num + m->k[f]) / den;                               /* :818 */
        if (!(t >= tol) || isnan(t) || isinf(t)) continue;               /* :820-822 */
        double c[3] = {x[0] + t * v[0], x[1] + t * v[1], x[2] + t * v[2]}; /* :826 */
        const double *lo = m->bounds_lo + 3 * f, *hi = m->bounds_hi + 3 * f;
        int inb = 1;
        for (int d = 0; d < 3; ++d)
            if (!(c[d] >= lo[d] - tol) || !(c[d] <= hi[d] + tol)) inb = 0; /* :828-829 */
        if (!inb) continue;
        /* barycentric: solve A w = c - o, A = face_basis_matrix[f] (3x3), :837-843 */
        const double *A = m->basis + 9 * f;
        const double *o = m->origins + 3 * f;
        double b[3] = {c[0] - o[0], c[1] - o[1], c[2] - o[2]};
        double a00 = A[0], a01 = A[1], a02 = A[2], a10 = A[3], a11 = A[4], a12 = A[5],
               a20 = A[6], a21 = A[7], a22 = A[8];
        double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
        double u = (b[0] * (a11 * a22 - a12 * a21) - a01 * (b[1] * a22 - a12 * b[2]) + a02 * (b[1] * a21 - a11 * b[2])) / det;
        double w = (a00 * (b[1] * a22 - a12 * b[2]) - b[0] * (a10 * a22 - a12 * a20) + a02 * (a10 * b[2] - b[1] * a20)) / det;
        double z = 1.0 - (u + w);
        if (!(u >= -tol && u <= 1 + tol && w >= -tol && w <= 1 + tol && z >= -tol && z <= 1 + tol)) continue;
        if (t < tbest) { tbest = t; fbest = f; }                          /* :847-849 first minimum */
    }
    *tc = tbest;
    *fc = fbest < 0 ? -1 : m->face_facet[fbest];                          /* :849-852 */
    for (int d = 0; d < 3; ++d) xc[d] = x[d] + tbest * v[d];              /* :854 */
}
void nko_find_boundary(const nko_mesh *m, int64_t n, const double *x, const double *v,
                       double *xc, double *tc, int32_t *fc) {
    for (int64_t i = 0; i < n; ++i) find_boundary_one(m, x + 3 * i, v + 3 * i, xc + 3 * i, tc + i, fc + i);
}

/* ---------------------------------------------------------- classifier ---- */
/* SubvolClassifier.predict, Geometry.py:1198-1213 (nearest centre) */
static int32_t classify_one(const nko_subvols *sv, const double x[3]) {
    int32_t best = 0;
    double dbest = INFINITY;
    for (int32_t s = 0; s < sv->S; ++s) {
        const double *c = sv->centers + 3 * s;
        double d = (x[0] - c[0]) * (x[0] - c[0]) + (x[1] - c[1]) * (x[1] - c[1]) + (x[2] - c[2]) * (x[2] - c[2]);
        if (d < dbest) { dbest = d; best = s; }
    }
    return best;
}
void nko_classify(const nko_subvols *sv, int64_t n, const double *x, int32_t *id) {
    for (int64_t i = 0; i < n; ++i) id[i] = classify_one(sv, x + 3 * i);
}

/* ------------------------------------------------------ material tables ---- */
/* Phonon.calculate_occupation, Phonon.py:338-345 */
static inline double occupation(const nko_material *mat, double T, double omega) {
    if (!(T > 0) || !(omega > 0)) return 0.0;
    return 1.0 / (exp(omega * mat->hbar / (T * mat->kb)) - 1.0);
}
void nko_occupation(const nko_material *mat, int64_t n, const double *T, const double *omega, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = occupation(mat, T[i], omega[i]);
}
/* np.searchsorted(a, x, side='left') */
static inline int32_t ss_left(const double *a, int32_t n, double x) {
    int32_t lo = 0, hi = n;
    while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
static inline int32_t ss_right(const double *a, int32_t n, double x) {
    int32_t lo = 0, hi = n;
    while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
/* lifetime_function = RegularGridInterpolator((T,q,j), tau), Phonon.py:326-336;
 * q and j are exact grid points so only the T axis interpolates (linear in tau). */
static inline double lifetime(const nko_material *mat, double T, int32_t mode) {
    int32_t NT = mat->NT;
    if (!(T >= mat->T_grid[0]) || !(T <= mat->T_grid[NT - 1])) return NAN;  /* reference raises ValueError */
    int32_t i = ss_left(mat->T_grid, NT, T) - 1;
    if (i < 0) i = 0;
    if (i > NT - 2) i = NT - 2;
    double y = (T - mat->T_grid[i]) / (mat->T_grid[i + 1] - mat->T_grid[i]);
    int64_t M = (int64_t)mat->Q * mat->J;
    double t0 = mat->lifetime[(int64_t)i * M + mode], t1 = mat->lifetime[(int64_t)(i + 1) * M + mode];
    return t0 * (1.0 - y) + t1 * y;
}
void nko_lifetime(const nko_material *mat, int64_t n, const double *T, const int32_t *mode, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = lifetime(mat, T[i], mode[i]);
}
/* scipy interp1d(kind='linear') evaluation rule (scipy/interpolate/interpolate.py _call_linear) */
static inline double interp_lin(const double *xs, const double *ys, int32_t n, double x) {
    int32_t idx = ss_left(xs, n, x);
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    double xlo = xs[idx - 1], xhi = xs[idx], ylo = ys[idx - 1], yhi = ys[idx];
    double slope = (yhi - ylo) / (xhi - xlo);
    return slope * (x - xlo) + ylo;
}
/* temperature_function, Phonon.py:387 (fill = (T_min, T_max)) */
static inline double T_of_E(const nko_material *mat, double E) {
    int32_t n = mat->nE;
    if (E < mat->energy_array[0]) return mat->T_fill_lo;
    if (E > mat->energy_array[n - 1]) return mat->T_fill_hi;
    return interp_lin(mat->energy_array, mat->T_array, n, E);
}
/* crystal_energy_function, Phonon.py:390 (fill = (E_min, E_max)) */
static inline double E_of_T(const nko_material *mat, double T) {
    int32_t n = mat->nE;
    if (T < mat->T_array[0]) return mat->energy_array[0];
    if (T > mat->T_array[n - 1]) return mat->energy_array[n - 1];
    return interp_lin(mat->T_array, mat->energy_array, n, T);
}
void nko_T_of_E(const nko_material *mat, int64_t n, const double *E, double *T) {
    for (int64_t i = 0; i < n; ++i) T[i] = T_of_E(mat, E[i]);
}
void nko_E_of_T(const nko_material *mat, int64_t n, const double *T, double *E) {
    for (int64_t i = 0; i < n; ++i) E[i] = E_of_T(mat, T[i]);
}

/* per-particle temperature: Population.py:570-571, :694-702 */
/* RBFInterpolator(kernel='cubic') evaluation: coefficients [w; p] = inv[:, :S] @ T_sv, cached until T_sv changes
 * (the oracle is single-threaded test infrastructure) */
static double rbf_T_one(const nko_subvols *sv, const double *T_sv, const double x[3]) {
    static double cache_T[1024], coef[1032];
    static const nko_subvols *cache_sv = NULL;
    const int32_t S = sv->S;
    int32_t nd = sv->rbf_used[0] + sv->rbf_used[1] + sv->rbf_used[2];
    const int32_t P = S + nd + 1;
    if (cache_sv != sv || memcmp(cache_T, T_sv, sizeof(double) * (size_t)S) != 0) {
        for (int32_t j = 0; j < P; ++j) {
            double acc = 0.0;
            for (int32_t i = 0; i < S; ++i) acc += sv->rbf_inv[(int64_t)j * P + i] * T_sv[i];
            coef[j] = acc;
        }
        memcpy(cache_T, T_sv, sizeof(double) * (size_t)S);
        cache_sv = sv;
    }
    double out = 0.0;
    for (int32_t i = 0; i < S; ++i) {
        double r2 = 0.0;
        for (int k = 0; k < 3; ++k)
            if (sv->rbf_used[k]) { double dd = x[k] - sv->centers[3 * i + k]; r2 += dd * dd; }
        out += coef[i] * (r2 * sqrt(r2));
    }
    out += coef[S];
    int32_t q = S + 1;
    for (int k = 0; k < 3; ++k)
        if (sv->rbf_used[k]) out += coef[q++] * ((x[k] - sv->rbf_shift[k]) / sv->rbf_scale[k]);
    return out;
}
static double interp_T_one(const nko_subvols *sv, const double *T_sv, const double x[3], int32_t svid) {
    int32_t S = sv->S;
    if (sv->interp == 3) return rbf_T_one(sv, T_sv, x);
    if (sv->interp == 2 || S == 1) return T_sv[svid >= 0 ? svid : classify_one(sv, x)];
    double xa = x[sv->axis];
    /* centres along the axis are ascending (Geometry.py:456-463) */
    double cbuf[4096];
    double *c = cbuf;
    for (int32_t s = 0; s < S; ++s) c[s] = sv->centers[3 * s + sv->axis];
    if (sv->interp == 1) return interp_lin(c, T_sv, S, xa);           /* linear + extrapolate */
    /* interp1d kind='nearest': x_bds = (x[1:]+x[:-1])/2 via halves, side='left' */
    int32_t lo = 0, hi = S - 1;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        double b = c[mid + 1] / 2.0 + c[mid] / 2.0;
        if (b < xa) lo = mid + 1; else hi = mid;
    }
    return T_sv[lo];
}
void nko_interp_T(const nko_subvols *sv, const double *T_sv, int64_t n, const double *x,
                  const int32_t *svid, double *T) {
    for (int64_t i = 0; i < n; ++i) T[i] = interp_T_one(sv, T_sv, x + 3 * i, svid ? svid[i] : -1);
}

/* ------------------------------------------------------------- reflect ---- */
static inline int32_t rough_index(const nko_rough *rg, int32_t facet) {
    for (int32_t i = 0; i < rg->Fr; ++i) if (rg->facet[i] == facet) return i;
    return -1;
}
/* select_reflected_modes + pick_diffuse_modes, Population.py:941-1015 */
static void reflect_one(const nko_material *mat, const nko_subvols *sv, const nko_rough *rg,
                        const double *T_sv, int32_t facet, int32_t mode_in, const double col[3],
                        double n_in, double omega_in, double r_spec, double r_deg, double r_diff,
                        int32_t *mode_out, double *n_out, double *omega_out) {
    int64_t M = (int64_t)mat->Q * mat->J;
    int32_t ir = rough_index(rg, facet);
    int64_t idx = (int64_t)ir * M + mode_in;
    int spec = rg->true_spec[idx] && (r_spec <= rg->specularity[idx]);      /* :951 */
    if (spec) {
        int32_t out = rg->spec_map[idx];                                    /* :961 */
        if (rg->degen_j2) {                                                 /* :963-969 */
            int32_t j2 = rg->degen_j2[out];
            if (j2 > -1 && r_deg >= 0.5) out = (out / mat->J) * mat->J + j2;
        }
        *mode_out = out; *n_out = n_in; *omega_out = omega_in;              /* :955-956 */
    } else {
        const double *roul = rg->roulette + (int64_t)ir * M;
        double r = r_diff * roul[M - 1];                                    /* :1003 */
        int32_t flat = ss_left(roul, (int32_t)M, r);                        /* :1005 */
        if (flat > M - 1) flat = (int32_t)M - 1;
        *mode_out = flat;                                                   /* :1007-1008 */
        *omega_out = mat->omega[flat];                                      /* :976 */
        double T = interp_T_one(sv, T_sv, col, -1);                         /* :978-984 */
        *n_out = occupation(mat, T, *omega_out);                            /* :986 */
    }
}
void nko_reflect(const nko_material *mat, const nko_mesh *mesh, const nko_subvols *sv, const nko_rough *rg,
                 const double *T_sv, int64_t n, const int32_t *facet, const int32_t *mode_in,
                 const double *col_pos, const double *n_in, const double *omega_in,
                 const double *r_spec, const double *r_deg, const double *r_diff,
                 int32_t *mode_out, double *n_out, double *omega_out) {
    (void)mesh;
    for (int64_t i = 0; i < n; ++i)
        reflect_one(mat, sv, rg, T_sv, facet[i], mode_in[i], col_pos + 3 * i, n_in[i], omega_in[i],
                    r_spec[i], r_deg ? r_deg[i] : 0.0, r_diff[i], mode_out + i, n_out + i, omega_out + i);
}

/* --------------------------------------------------------------- stages ---- */
/* timesteps_to_boundary, Population.py:797-830 (first call :310-314) */
/* ---- box rule (nk_oracle.h nko_params::box; engine: nk_device.h nk_box_out / nk_box_first_hit) ---- */
int32_t nko_box_detect(const nko_mesh *m, nko_params *p) {
    if (m->F != 12 || m->Fc != 6) return 0;
    int seen = 0, fm = 0;
    double area[6] = {0, 0, 0, 0, 0, 0};
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 6; ++w) p->box_face0[w] = 0x7fffffff;
    for (int32_t f = 0; f < m->F; ++f) {
        const double *n = m->normals + 3 * f;
        int a = -1, sgn = 0;
        for (int k = 0; k < 3; ++k) {
            if (n[k] == 1.0 || n[k] == -1.0) { if (a >= 0) return 0; a = k; sgn = n[k] > 0 ? 1 : 0; }
            else if (n[k] != 0.0) return 0;
        }
        if (a < 0) return 0;
        const int w = 2 * a + sgn;
        const double wall = sgn ? -m->k[f] : m->k[f];
        if (!(fabs(wall - m->bbox[(sgn ? 3 : 0) + a]) <= 1e-9 * (1.0 + fabs(wall)))) return 0;
        if (seen & (1 << w)) { if (p->box_k[w] != m->k[f] || p->box_facet[w] != m->face_facet[f]) return 0; }
        else { seen |= 1 << w; p->box_k[w] = m->k[f]; p->box_facet[w] = m->face_facet[f]; }
        if (f < p->box_face0[w]) p->box_face0[w] = f;
        area[w] += m->face_area[f];
        cnt[w] += 1;
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) {
                const double v = m->vertices[9 * (int64_t)f + 3 * c + k], lo = m->bbox[k], hi = m->bbox[3 + k];
                const double e = 1e-9 * (1.0 + fabs(lo) + fabs(hi));
                if (k == a) { if (fabs(v - wall) > e) return 0; }
                else if (fabs(v - lo) > e && fabs(v - hi) > e) return 0;
            }
    }
    if (seen != 63) return 0;
    for (int w = 0; w < 6; ++w) {
        const int a = w / 2, b = (a + 1) % 3, c = (a + 2) % 3;
        const double side = (m->bbox[3 + b] - m->bbox[b]) * (m->bbox[3 + c] - m->bbox[c]);
        if (cnt[w] != 2 || !(fabs(area[w] - side) <= 1e-9 * side)) return 0;
        if (p->box_facet[w] < 0 || p->box_facet[w] >= 6) return 0;
        fm |= 1 << p->box_facet[w];
    }
    return fm == 63;
}
/* does the particle lie beyond a wall it flies towards?  (nk_box_out) */
static int box_out(const nko_params *p, const double x[3], const double v[3]) {
    for (int a = 0; a < 3; ++a) {
        if (v[a] > 0.0 && x[a] > -p->box_k[2 * a + 1]) return 1;
        if (v[a] < 0.0 && x[a] < p->box_k[2 * a]) return 1;
    }
    return 0;
}
/* the wall it crossed first, Mesh.py:816-818 on the walls it lies beyond; timesteps from the end of the step.  num = x.n + k and
 * den = v.n are both positive for such a wall; the earliest crossing = the largest num / den is picked on the cross products
 * (the engine's nk_box_first_hit selects the same way: no division per wall), the lowest face index among equals (:846-848) */
static void box_first_hit(const nko_params *p, const double x[3], const double v[3], double *nts, int32_t *facet) {
    double nb = -1.0, db = 1.0;
    int32_t fb = -1, f0b = 0x7fffffff;
    for (int a = 0; a < 3; ++a) {
        const int fwd = v[a] > 0.0;
        const double num = fwd ? x[a] - (-p->box_k[2 * a + 1]) : p->box_k[2 * a] - x[a];   /* x.n + k of the wall it flies towards */
        const double den = fabs(v[a]);                                                     /* v.n of that wall */
        const int w = 2 * a + fwd;
        const double l = num * db, r = nb * den;
        if (num > 0.0 && den > 0.0 && (l > r || (l == r && p->box_face0[w] < f0b))) { nb = num; db = den; fb = p->box_facet[w]; f0b = p->box_face0[w]; }
    }
    *nts = -(nb / db) / p->dt;                                            /* t = -num / den (:818), n_timesteps = t / dt */
    *facet = fb;
}

int64_t nko_init_boundaries(const nko_mesh *mesh, const nko_material *mat, const nko_params *p, nko_particles *P) {
    int64_t bad = 0;
    for (int64_t i = 0; i < P->N; ++i) {
        double xc[3], tc; int32_t fc;
        find_boundary_one(mesh, P->pos + 3 * i, mat->group_vel + 3 * (int64_t)P->mode[i], xc, &tc, &fc);
        P->n_ts[i] = tc / p->dt;
        P->facet[i] = fc;
        if (p->box && fc >= 0) {
            const double *x = P->pos + 3 * i;
            for (int a = 0; a < 3; ++a) if (!(x[a] >= p->box_k[2 * a] && x[a] <= -p->box_k[2 * a + 1])) { ++bad; break; }
        }
    }
    return bad;
}
/* Population.drift, Population.py:790-795 */
void nko_drift(const nko_material *mat, const nko_params *p, nko_particles *P) {
    for (int64_t i = 0; i < P->N; ++i) {
        const double *v = mat->group_vel + 3 * (int64_t)P->mode[i];
        for (int d = 0; d < 3; ++d) P->pos[3 * i + d] += v[d] * p->dt;
        P->n_ts[i] -= 1.0;
    }
}

static inline int32_t res_index(const nko_reservoirs *res, int32_t facet) {
    for (int32_t i = 0; i < res->R; ++i) if (res->facet[i] == facet) return i;
    return -1;
}

/* fill_reservoirs ('constant' :358-406, 'fixed_rate' :408-455) + add_reservoir_particles (:525-552)
 * + Mesh.sample_surface (Mesh.py:923-951).  Emission ids: pid = (step+1)<<40 | (r*M+m)<<12 | level. */
int64_t nko_emit(const nko_material *mat, const nko_mesh *mesh, nko_reservoirs *res, const nko_params *p,
                 int64_t step, int32_t rank, int32_t nranks, nko_particles *P) {
    int64_t M = (int64_t)mat->Q * mat->J;
    int64_t emitted = 0;
    for (int32_t r = 0; r < res->R; ++r) {
        int32_t facet = res->facet[r];
        int32_t f0 = mesh->facet_face_off[facet], f1 = mesh->facet_face_off[facet + 1];
        int32_t nf = f1 - f0;
        double cdf[1024];
        double *cd = nf <= 1024 ? cdf : (double *)malloc(sizeof(double) * nf);
        double acc = 0, tot = 0;
        for (int32_t a = 0; a < nf; ++a) tot += mesh->face_area[mesh->facet_face_idx[f0 + a]];
        for (int32_t a = 0; a < nf; ++a) { acc += mesh->face_area[mesh->facet_face_idx[f0 + a]] / tot; cd[a] = acc; }
        for (int32_t a = 0; a < nf; ++a) cd[a] /= cd[nf - 1];
        if (res->gen == 2) {                                       /* one_to_one :457-489 */
            /* one particle in for every particle that left through this facet at the previous step: mode drawn
             * from the cumulative enter_prob (:467-472), entry time uniform in the step (:482) */
            double *roul = (double *)malloc(sizeof(double) * (size_t)M);
            double run = 0.0, mx = 0.0;
            for (int64_t m = 0; m < M; ++m) { run += res->enter_prob[(int64_t)r * M + m]; roul[m] = run; if (run > mx) mx = run; }
            for (int64_t m = 0; m < M; ++m) roul[m] /= mx;
            int64_t n = res->n_leaving ? res->n_leaving[r] : 0;
            for (int64_t i = 0; i < n; ++i) {
                if (((i + step) % nranks) != rank) continue;
                uint64_t pid = ((uint64_t)((step + 1) & 0xFFFFFF) << 40) | ((uint64_t)r << 32) | (uint64_t)i;
                double um, u_unused, uf, us, ur, ut;
                nko_uniform2(p->seed, pid, (uint32_t)step, TAG_DICE, &um, &u_unused);
                nko_uniform2(p->seed, pid, (uint32_t)step, TAG_EMIT, &uf, &us);
                nko_uniform2(p->seed, pid, (uint32_t)step, TAG_EMIT + 1, &ur, &ut);
                int64_t m = ss_left(roul, (int32_t)M, um);                                /* np.searchsorted :472 */
                if (m > M - 1) m = M - 1;
                double dt_in = p->dt * ut;                                               /* :482 */
                int32_t a = ss_right(cd, nf, uf);
                if (a > nf - 1) a = nf - 1;
                const double *vx = mesh->vertices + 9 * (int64_t)mesh->facet_face_idx[f0 + a];
                double sq = sqrt(us);
                double a0 = 1.0 - sq, a1 = (1.0 - ur) * sq, a2 = ur * sq;
                double x[3];
                for (int d = 0; d < 3; ++d) x[d] = (a0 * vx[d] + a1 * vx[3 + d]) + a2 * vx[6 + d];
                if (P->N >= P->cap) { free(roul); if (cd != cdf) free(cd); return -1; }
                int64_t k = P->N++;
                const double *v = mat->group_vel + 3 * m;
                double xc[3], tc; int32_t fc;
                find_boundary_one(mesh, x, v, xc, &tc, &fc);
                P->n_ts[k] = tc / p->dt - dt_in / p->dt;
                for (int d = 0; d < 3; ++d) P->pos[3 * k + d] = x[d] + v[d] * dt_in;
                if (res->dbg_dt_in) res->dbg_dt_in[k] = dt_in;
                if (res->dbg_x0) { res->dbg_x0[3 * k] = x[0]; res->dbg_x0[3 * k + 1] = x[1]; res->dbg_x0[3 * k + 2] = x[2]; }
                if (res->dbg_level) res->dbg_level[k] = 0;
                if (res->dbg_res) res->dbg_res[k] = r;
                P->facet[k] = fc;
                P->mode[k] = (int32_t)m;
                P->occ[k] = occupation(mat, res->T[r], mat->omega[m]);
                P->pid[k] = pid;
                P->energy[k] = 0.0;
                P->temp[k] = res->T[r];
                P->sv[k] = -1;
                ++emitted;
            }
            free(roul);
            if (cd != cdf) free(cd);
            continue;
        }
        for (int64_t m = 0; m < M; ++m) {
            int64_t rm = (int64_t)r * M + m;
            double prob = res->enter_prob[rm];
            double fixed = floor(prob);
            int mask; double cnt;
            if (res->gen == 0) {                                   /* constant :359-367 */
                res->counter[rm] += prob - fixed;
                mask = res->counter[rm] >= 1.0;
                res->counter[rm] -= mask;
                cnt = res->counter[rm];
            } else {                                               /* fixed_rate :410-417 */
                double d0, d1;
                nko_uniform2(p->seed, (uint64_t)rm | 0xFFFFFFFF00000000ull, (uint32_t)step, TAG_DICE, &d0, &d1);
                if (res->dice) d0 = res->dice[rm];                 /* test tap: the reference's own dice */
                mask = d0 <= (prob - fixed);
                cnt = d0;
            }
            int32_t c = (int32_t)fixed + mask;
            for (int32_t level = c; level >= 1; --level) {
                if (((rm + level + step) % nranks) != rank) continue;
                uint64_t pid = ((uint64_t)((step + 1) & 0xFFFFFF) << 40) | ((uint64_t)rm << 12) | (uint64_t)level;
                double uf, us, ur, ut;
                nko_uniform2(p->seed, pid, (uint32_t)step, TAG_EMIT, &uf, &us);
                nko_uniform2(p->seed, pid, (uint32_t)step, TAG_EMIT + 1, &ur, &ut);
                double dt_in = (level == 1) ? p->dt * (1.0 - (cnt / prob))              /* :391 / :440 */
                                            : p->dt * (1.0 - (level - 1 + ut) / prob);  /* :394 */
                int32_t a = ss_right(cd, nf, uf);
                if (a > nf - 1) a = nf - 1;
                const double *vx = mesh->vertices + 9 * (int64_t)mesh->facet_face_idx[f0 + a];
                double sq = sqrt(us);
                double a0 = 1.0 - sq, a1 = (1.0 - ur) * sq, a2 = ur * sq;                /* Mesh.py:945-947 */
                double x[3];
                for (int d = 0; d < 3; ++d) x[d] = (a0 * vx[d] + a1 * vx[3 + d]) + a2 * vx[6 + d];
                if (P->N >= P->cap) { if (cd != cdf) free(cd); return -1; }
                int64_t i = P->N++;
                const double *v = mat->group_vel + 3 * m;
                double xc[3], tc; int32_t fc;
                find_boundary_one(mesh, x, v, xc, &tc, &fc);
                P->n_ts[i] = tc / p->dt - dt_in / p->dt;                                 /* :535 */
                for (int d = 0; d < 3; ++d) P->pos[3 * i + d] = x[d] + v[d] * dt_in;     /* :536 */
                if (res->dbg_dt_in) res->dbg_dt_in[i] = dt_in;
                if (res->dbg_x0) { res->dbg_x0[3 * i] = x[0]; res->dbg_x0[3 * i + 1] = x[1]; res->dbg_x0[3 * i + 2] = x[2]; }
                if (res->dbg_level) res->dbg_level[i] = level;
                if (res->dbg_res) res->dbg_res[i] = r;
                P->facet[i] = fc;
                P->mode[i] = (int32_t)m;
                P->occ[i] = occupation(mat, res->T[r], mat->omega[m]);                   /* :506 */
                P->pid[i] = pid;
                P->energy[i] = 0.0;
                P->temp[i] = res->T[r];
                P->sv[i] = -1;
                ++emitted;
            }
        }
        if (cd != cdf) free(cd);
    }
    return emitted;
}

/* boundary_scattering, Population.py:1546-1683, one particle at a time.  The vectorised
 * passes I-IV of the reference act on disjoint per-particle states, so the per-particle
 * event loop below visits the same sequence of events. */
void nko_boundary_scattering(const nko_material *mat, const nko_mesh *mesh, const nko_subvols *sv,
                             const nko_reservoirs *res, const nko_rough *rg, const nko_params *p,
                             const double *T_sv, int64_t step, nko_particles *P,
                             int64_t *N_leaving, double *res_energy, double *res_flux) {
    const double dt = p->dt;
    int64_t w = 0;
    for (int32_t r = 0; r < res->R; ++r) N_leaving[r] = 0;                    /* :1560 */
    for (int64_t i = 0; i < P->N; ++i) {
        double x[3] = {P->pos[3 * i], P->pos[3 * i + 1], P->pos[3 * i + 2]};
        int32_t mode = P->mode[i];
        double occ = P->occ[i], nts = P->n_ts[i];
        int32_t fct = P->facet[i];
        int alive = 1;
        int has_event = nts < 0;                                              /* :1551-1556 */
        if (p->box) {
            /* box rule: read off the position; a particle whose last cast missed (the reference's n_timesteps = inf) never has one */
            const double *v0 = mat->group_vel + 3 * (int64_t)mode;
            has_event = fct >= 0 && box_out(p, x, v0);
            if (has_event) box_first_hit(p, x, v0, &nts, &fct);
        }
        if (has_event) {
            double cts = 0.0;
            double omega = mat->omega[mode];
            uint32_t ev = 0;
            while (cts < 1.0) {                                               /* :1563 */
                const double *v = mat->group_vel + 3 * (int64_t)mode;
                double rem = 1.0 - cts;
                if (rem > nts) {
                    /* bound_cond[facet]; a miss (facet -1) indexes the LAST facet */
                    int32_t fi = fct < 0 ? mesh->Fc - 1 : fct;
                    int8_t bc = mesh->facet_bc[fi];
                    if (bc == 'T' || bc == 'F') {                             /* I. :1568-1608 */
                        int32_t r = res_index(res, fct);
                        if (r >= 0) {
                            N_leaving[r] += 1;
                            double Tr = p->T_ref_local ? res->T[r] : p->T_ref;
                            double dn = occ - occupation(mat, Tr, omega);     /* :1590-1592 */
                            double e = mat->hbar * omega * dn;                /* :1594 */
                            res_energy[r] -= e;                               /* :1595 */
                            const double *nf = mesh->facet_normal + 3 * fct;
                            double vn = (v[0] * nf[0] + v[1] * nf[1]) + v[2] * nf[2];
                            for (int d = 0; d < 3; ++d) res_flux[3 * r + d] += e * v[d] / vn;  /* :1601-1602 */
                        }
                        alive = 0;
                        break;
                    }
                    /* collision position = where the stored next hit lies: current x + v*(nts*dt) */
                    double col[3], prev[3];
                    for (int d = 0; d < 3; ++d) col[d] = x[d] + v[d] * (nts * dt);
                    for (int d = 0; d < 3; ++d) prev[d] = (cts == 0.0) ? x[d] - v[d] * dt : x[d];  /* :1472-1474, :1504-1508 */
                    double dist = sqrt(((col[0] - prev[0]) * (col[0] - prev[0]) + (col[1] - prev[1]) * (col[1] - prev[1])) + (col[2] - prev[2]) * (col[2] - prev[2]));
                    double vnorm = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
                    if (bc == 'P') {                                          /* II. :1463-1489 */
                        int32_t pf = mesh->facet_partner[fct];
                        const double *c1 = mesh->facet_centroid + 3 * pf, *c0 = mesh->facet_centroid + 3 * fct;
                        for (int d = 0; d < 3; ++d) x[d] = col[d] + (c1[d] - c0[d]);  /* :1476-1477 */
                        cts += dist / (vnorm * dt);                          /* :1482 */
                    } else {                                                  /* III. 'R' :1491-1544 */
                        double r0, r1;
                        nko_uniform2(p->seed, P->pid[i], (uint32_t)step, TAG_REFLECT + ev, &r0, &r1);
                        cts += dist / (vnorm * dt);                          /* :1514 */
                        for (int d = 0; d < 3; ++d) x[d] = col[d];            /* :1517 */
                        int32_t mo; double no, oo;
                        reflect_one(mat, sv, rg, T_sv, fct, mode, col, occ, omega, r0, r1, r1, &mo, &no, &oo);
                        mode = mo; occ = no; omega = oo;
                    }
                    double xc[3], tc; int32_t fc;
                    find_boundary_one(mesh, x, mat->group_vel + 3 * (int64_t)mode, xc, &tc, &fc);
                    nts = tc / dt;
                    fct = fc;
                    ++ev;
                    if (ev > 4096) { cts = 1.0; }                             /* safety: the reference would spin */
                } else {                                                      /* IV. :1673-1681 */
                    for (int d = 0; d < 3; ++d) x[d] += v[d] * dt * rem;
                    nts -= rem;
                    cts = 1.0;
                }
            }
        }
        if (alive) {                                                          /* np.delete keeps order, :832-850 */
            P->pos[3 * w] = x[0]; P->pos[3 * w + 1] = x[1]; P->pos[3 * w + 2] = x[2];
            P->mode[w] = mode; P->occ[w] = occ; P->n_ts[w] = nts; P->facet[w] = fct;
            P->pid[w] = P->pid[i];
            ++w;
        }
    }
    P->N = w;
}

/* The three parts of refresh_temperatures are exposed separately so that a particle-sharded run can sum the tallies
 * of all ranks between them; nko_refresh_temperatures chains them for the single-rank case. */
void nko_tally(const nko_material *mat, const nko_subvols *sv, const nko_params *p, nko_particles *P,
               const double *T_sv, int64_t *N_sv, double *E_raw) {
    int32_t S = sv->S;
    for (int32_t s = 0; s < S; ++s) { N_sv[s] = 0; E_raw[s] = 0.0; }
    for (int64_t i = 0; i < P->N; ++i) {
        int32_t s = classify_one(sv, P->pos + 3 * i);                         /* :688 */
        P->sv[i] = s;
        N_sv[s] += 1;                                                         /* :679 */
        double omega = mat->omega[P->mode[i]];
        double Tr = p->T_ref_local ? T_sv[s] : p->T_ref;
        double dn = P->occ[i] - occupation(mat, Tr, omega);                   /* :707 / :710 */
        double e = mat->hbar * omega * dn;                                    /* :713 */
        P->energy[i] = e;
        E_raw[s] += e;                                                        /* :715-717 */
    }
}
void nko_update_T(const nko_material *mat, const nko_subvols *sv, const nko_params *p, const int64_t *N_sv,
                  const double *E_raw, double *T_sv, double *E_sv) {
    int32_t S = sv->S;
    for (int32_t s = 0; s < S; ++s) {
        double norm;
        if (p->norm_fixed) norm = mat->active_modes / (p->particle_density * sv->volumes[s]);   /* :720 */
        else { norm = (double)mat->active_modes / (double)N_sv[s]; if (isnan(norm)) norm = 0; } /* :722-723 */
        double E = E_raw[s] * norm;
        E = E / mat->QV;                                                      /* :726, Phonon.py:401 */
        double ref = p->T_ref_local ? E_of_T(mat, T_sv[s]) : E_of_T(mat, p->T_ref);  /* :708 / :93 */
        E_sv[s] = E + ref;                                                    /* :728 */
    }
    for (int32_t s = 0; s < S; ++s) T_sv[s] = T_of_E(mat, E_sv[s]);           /* :692 */
}
void nko_assign_T(const nko_subvols *sv, const double *T_sv, nko_particles *P) {
    for (int64_t i = 0; i < P->N; ++i) P->temp[i] = interp_T_one(sv, T_sv, P->pos + 3 * i, P->sv[i]);  /* :694-702 */
}
/* refresh_temperatures + calculate_energy, Population.py:685-728 */
void nko_refresh_temperatures(const nko_material *mat, const nko_subvols *sv, const nko_params *p,
                              nko_particles *P, double *T_sv, double *E_sv, int64_t *N_sv, double *E_raw) {
    nko_tally(mat, sv, p, P, T_sv, N_sv, E_raw);
    nko_update_T(mat, sv, p, N_sv, E_raw, T_sv, E_sv);
    nko_assign_T(sv, T_sv, P);
}

/* lifetime_scattering, Population.py:1701-1710 */
void nko_lifetime_scattering(const nko_material *mat, const nko_params *p, nko_particles *P) {
    for (int64_t i = 0; i < P->N; ++i) {
        double T = P->temp[i];
        double tau = lifetime(mat, T, P->mode[i]);
        double n0 = occupation(mat, T, mat->omega[P->mode[i]]);
        P->occ[i] = (tau > 0) ? n0 + (P->occ[i] - n0) * exp(-p->dt / tau) : n0;
    }
}

/* calculate_heat_flux, Population.py:730-747 (uses energies and subvol ids of the last tally) */
void nko_heat_flux(const nko_material *mat, const nko_subvols *sv, const nko_params *p,
                   const nko_particles *P, const int64_t *N_sv, double *flux) {
    int32_t S = sv->S;
    for (int32_t s = 0; s < 3 * S; ++s) flux[s] = 0.0;
    for (int64_t i = 0; i < P->N; ++i) {
        const double *v = mat->group_vel + 3 * (int64_t)P->mode[i];
        for (int d = 0; d < 3; ++d) flux[3 * P->sv[i] + d] += v[d] * P->energy[i];
    }
    const double eVpsa2_in_Wm2 = 1.602176634e-19 / (1e-12 * (1e-10 * 1e-10));   /* Constants.py:9-12 */
    for (int32_t s = 0; s < S; ++s) {
        double norm = p->norm_fixed ? mat->active_modes / (p->particle_density * sv->volumes[s])
                                    : (double)mat->active_modes / (double)N_sv[s];
        for (int d = 0; d < 3; ++d) flux[3 * s + d] = flux[3 * s + d] * norm / mat->QV * eVpsa2_in_Wm2;
    }
}

/* Key of a particle's random draws when ids are not tracked (test infrastructure mirror of the engine's nk_state_key). */
static uint64_t state_key(int32_t mode, const double *x) {
    uint64_t k = (uint64_t)(uint32_t)mode, b;
    for (int d = 0; d < 3; ++d) { memcpy(&b, x + d, 8); k = k * 0x9E3779B97F4A7C15ull + b; }
    return k;
}

/* contains_check, Population.py:1712-1722 + Mesh.sample_volume, Mesh.py:890-904 */
int64_t nko_contains_check(const nko_material *mat, const nko_mesh *mesh, const nko_params *p,
                           int64_t step, nko_particles *P) {
    int64_t moved = 0;
    double tot = 0;
    for (int32_t s = 0; s < mesh->nS; ++s) tot += mesh->simplex_vol[s];
    for (int64_t i = 0; i < P->N; ++i) {
        double *x = P->pos + 3 * i;
        int out = 0;
        for (int d = 0; d < 3; ++d)
            if (x[d] < mesh->bbox[d] - 1e-10 || x[d] > mesh->bbox[3 + d] + 1e-10) out = 1;
        if (!out) continue;
        double u[6];
        const uint64_t key = p->ids_from_state ? state_key(P->mode[i], x) : P->pid[i];
        nko_uniform2(p->seed, key, (uint32_t)step, TAG_RESAMP + 0, &u[0], &u[1]);
        nko_uniform2(p->seed, key, (uint32_t)step, TAG_RESAMP + 1, &u[2], &u[3]);
        nko_uniform2(p->seed, key, (uint32_t)step, TAG_RESAMP + 2, &u[4], &u[5]);
        double acc = 0; int32_t s = mesh->nS - 1;
        for (int32_t k = 0; k < mesh->nS; ++k) { acc += mesh->simplex_vol[k] / tot; if (u[0] < acc) { s = k; break; } }
        double a[4], asum = 0;
        for (int k = 0; k < 4; ++k) { a[k] = -log(u[1 + k]); asum += a[k]; }
        const double *sp = mesh->simplex_pts + 12 * (int64_t)s;
        for (int d = 0; d < 3; ++d) {
            double acc2 = 0;
            for (int k = 0; k < 4; ++k) acc2 += (a[k] / asum) * sp[3 * k + d];
            x[d] = acc2;
        }
        double xc[3], tc; int32_t fc;
        find_boundary_one(mesh, x, mat->group_vel + 3 * (int64_t)P->mode[i], xc, &tc, &fc);
        P->n_ts[i] = tc / p->dt;
        P->facet[i] = fc;
        ++moved;
    }
    return moved;
}
