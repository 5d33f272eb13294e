/* nk_oracle.h -- CPU restatement of Nano-kappa's Population timestep loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nanokappa_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors
 * produced by running the reference itself (tests/golden/make_golden.py, outputs
 * in tests/golden/ *.npz) -- see tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line (under /root/reference) it restates.
 * All arithmetic is IEEE double, indices int32/int64, as in the reference (NumPy
 * float64/int64).
 */
#ifndef NK_ORACLE_H
#define NK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t Q, J, NT;
    const double *omega;      /* [Q*J]   rad/ps            Phonon.py:165-167 */
    const double *group_vel;  /* [Q*J*3] angstrom/ps       Phonon.py:181-183 */
    const double *T_grid;     /* [NT]    K                 Phonon.py:177-179 */
    const double *lifetime;   /* [NT*Q*J] ps, 0 = no decay Phonon.py:326-336 */
    int32_t nE;               /* size of the E(T) table    Phonon.py:372-390 */
    double T_fill_lo, T_fill_hi; /* temperature_function fill values = temperature_array min/max, Phonon.py:376-387 */
    const double *T_array;    /* [nE] */
    const double *energy_array; /* [nE] eV/angstrom^3 */
    double hbar, kb;          /* Constants.py:7-8 */
    double QV;                /* number_of_qpoints*volume_unitcell, Phonon.py:401 */
    int32_t active_modes;     /* Phonon.py:126 */
} nko_material;

typedef struct {
    int32_t F;                 /* triangles */
    const double *normals;     /* [F*3]  Mesh.py:228-229 */
    const double *k;           /* [F]    Mesh.py:323-324 */
    const double *bounds_lo;   /* [F*3]  Mesh.py:238-242 */
    const double *bounds_hi;   /* [F*3] */
    const double *basis;       /* [F*9]  face_basis_matrix (F,3,3), Mesh.py:231-232 */
    const double *origins;     /* [F*3]  Mesh.py:234 */
    const int32_t *face_facet; /* [F]    Mesh.py:314-321 */
    const double *vertices;    /* [F*9]  the three corners of each face (sampling, Mesh.py:939) */
    const double *face_area;   /* [F] */
    int32_t Fc;                /* facets */
    const int8_t *facet_bc;    /* [Fc] 'T','F','P','R'  Geometry.py:652-677 */
    const int32_t *facet_partner; /* [Fc] periodic partner or -1, Population.py:1468-1470 */
    const double *facet_centroid; /* [Fc*3] */
    const double *facet_normal;   /* [Fc*3] */
    const int32_t *facet_face_off; /* [Fc+1] CSR of faces per facet, Mesh.py:271-287 */
    const int32_t *facet_face_idx;
    double tol;                /* Mesh.py:24 */
    double bbox[6];            /* lo xyz, hi xyz, Geometry bounds */
    int32_t nS;                /* volume simplices (Mesh.py:354-486) */
    const double *simplex_pts; /* [nS*12] */
    const double *simplex_vol; /* [nS] */
} nko_mesh;

typedef struct {
    int32_t S;
    int32_t kind;            /* 0 = slice (nearest centre along axis), 1 = general nearest centre */
    int32_t axis;
    int32_t interp;          /* 0 nearest (interp1d kind='nearest' on slices), 1 linear slice, 2 nearest-ND,
                              * 3 cubic RBF (RBFInterpolator, Population.py:573-590) */
    const double *centers;   /* [S*3] Geometry.py:453-463 */
    const double *volumes;   /* [S] */
    /* interp 3 only: inverse of scipy's RBF system for these centres, P = S + n_used + 1 (row-major), the shift /
     * scale of its polynomial part and which coordinates take part (Population.py:651-656) */
    const double *rbf_inv;   /* [P*P] */
    const double *rbf_shift; /* [3] */
    const double *rbf_scale; /* [3] */
    int32_t rbf_used[3];
} nko_subvols;

typedef struct {
    int32_t R;
    const int32_t *facet;    /* [R] */
    const double *T;         /* [R] */
    const double *enter_prob;/* [R*Q*J] Population.py:146-161 */
    double *counter;         /* [R*Q*J] state, Population.py:343, :361-365 */
    int32_t gen;             /* 0 constant, 1 fixed_rate, 2 one_to_one (Population.py:457-489) */
    /* optional test taps (NULL = off), indexed by the slot the new particle is appended to */
    double *dbg_dt_in;       /* [cap]   entry-time offset, Population.py:391-394 */
    double *dbg_x0;          /* [cap*3] sampled position on the facet, Mesh.py:949 */
    int32_t *dbg_level;      /* [cap]   which of the mode's particles (1 = deterministic time) */
    int32_t *dbg_res;        /* [cap]   reservoir index */
    const int64_t *n_leaving;/* [R] one_to_one: particles to emit at this step = those that left through the facet at
                              * the previous step, all ranks together (Population.py:344, :466, :1585) */
    const double *dice;      /* optional test tap (NULL = off): 'fixed_rate' takes its dice (Population.py:410) from here
                              * instead of the counter-based generator, [R*Q*J] -- replays the uniforms the reference drew
                              * (tests/golden/emission_fixed.npz) */
} nko_reservoirs;

typedef struct {
    int32_t Fr;
    const int32_t *facet;       /* [Fr] rough facet ids */
    const double *specularity;  /* [Fr*Q*J] Population.py:852-877, :1459 */
    const uint8_t *true_spec;   /* [Fr*Q*J] Population.py:1458 */
    const int32_t *spec_map;    /* [Fr*Q*J] flat out mode q*J+j or -1, Population.py:1457 */
    const double *roulette;     /* [Fr*Q*J] Population.py:938-939 */
    const int32_t *degen_j2;    /* [Q*J] partner branch or -1 ('k' model only, :963-969); may be NULL */
} nko_rough;

typedef struct {
    double dt;
    int32_t norm_fixed;      /* 0 'mean', 1 'fixed'  Population.py:719-723 */
    double particle_density;
    int32_t T_ref_local;     /* 1 = 'local' */
    double T_ref;
    uint64_t seed;
    int32_t ids_from_state;  /* 1: the resampling draws of contains_check are keyed on the particle's state (mode and the bits
                              * of its position) instead of its id -- the engine's rule when it does not track ids */
    /* Box rule (round 4; the engine's "box store", nk_device.h NkDev::box): on an axis-aligned box whose six sides are its six
     * facets the engine keeps no cached next hit.  Whether a particle meets a wall inside the step is read off its end-of-step
     * position (beyond a wall it flies towards), and that hit -- time and facet -- is evaluated then, with Mesh.find_boundary's
     * expression t = -(x.n + k) / (v.n) (Mesh.py:818).  The reference decides on the cached n_timesteps it decremented
     * (Population.py:795, :1551): the same hit up to the rounding of the drift.  box = 1 makes the oracle decide the engine's
     * way (nko_box_detect fills the rest); n_ts / facet are still carried the reference's way, for comparison. */
    int32_t box;
    double box_k[6];         /* plane constants of the walls: [2a] normal -e_a, [2a+1] normal +e_a */
    int32_t box_facet[6];
    int32_t box_face0[6];    /* lowest face index of each wall (ties, Mesh.py:846-848) */
} nko_params;

/* Particle arrays in the reference's own layout (Population.py:199-321). */
typedef struct {
    int64_t N, cap;
    double *pos;        /* [cap*3] */
    int32_t *mode;      /* [cap] flat q*J+j */
    double *occ;        /* [cap] */
    double *n_ts;       /* [cap] time to boundary / dt */
    int32_t *facet;     /* [cap] next collision facet, -1 = none */
    uint64_t *pid;      /* [cap] */
    double *energy;     /* [cap] scratch: hbar*omega*dn of the last tally */
    double *temp;       /* [cap] per-particle T of the last refresh */
    int32_t *sv;        /* [cap] */
} nko_particles;

/* ---- deterministic primitives ---- */
void nko_find_boundary(const nko_mesh *m, int64_t n, const double *x, const double *v,
                       double *xc, double *tc, int32_t *fc);
void nko_classify(const nko_subvols *sv, int64_t n, const double *x, int32_t *id);
void nko_occupation(const nko_material *mat, int64_t n, const double *T, const double *omega, double *out);
void nko_lifetime(const nko_material *mat, int64_t n, const double *T, const int32_t *mode, double *out);
void nko_T_of_E(const nko_material *mat, int64_t n, const double *E, double *T);
void nko_E_of_T(const nko_material *mat, int64_t n, const double *T, double *E);
void nko_interp_T(const nko_subvols *sv, const double *T_sv, int64_t n, const double *x,
                  const int32_t *svid, double *T);
void nko_reflect(const nko_material *mat, const nko_mesh *mesh, const nko_subvols *sv, const nko_rough *rg,
                 const double *T_sv, int64_t n, const int32_t *facet, const int32_t *mode_in,
                 const double *col_pos, const double *n_in, const double *omega_in,
                 const double *r_spec, const double *r_deg, const double *r_diff,
                 int32_t *mode_out, double *n_out, double *omega_out);

/* ---- stages of run_timestep (Population.py:1724-1769) ---- */
/* returns the number of particles the box rule cannot express: outside the box with a wall ahead of them (the caller then
 * clears p->box, as the engine goes back to its cached layout) */
int64_t nko_init_boundaries(const nko_mesh *mesh, const nko_material *mat, const nko_params *p, nko_particles *P);
/* 1 and p->box_* filled if the mesh is such a box (the engine's rule, nk_engine.hip nk_set_mesh), else 0; p->box is left alone */
int32_t nko_box_detect(const nko_mesh *mesh, nko_params *p);
void nko_drift(const nko_material *mat, const nko_params *p, nko_particles *P);
/* emission: appends to P; returns number emitted or -1 on capacity overflow */
int64_t nko_emit(const nko_material *mat, const nko_mesh *mesh, nko_reservoirs *res, const nko_params *p,
                 int64_t step, int32_t rank, int32_t nranks, nko_particles *P);
/* boundary events; res_tally = [R*(1+1+3)]: N_leaving, energy balance, heat flux (accumulated) */
void nko_boundary_scattering(const nko_material *mat, const nko_mesh *mesh, const nko_subvols *sv,
                             const nko_reservoirs *res, const nko_rough *rg, const nko_params *p,
                             const double *T_sv, int64_t step, nko_particles *P,
                             int64_t *N_leaving, double *res_energy, double *res_flux);
/* tally + T update; E_raw[S] = plain sums, E_sv normalised, T_sv in/out */
void nko_refresh_temperatures(const nko_material *mat, const nko_subvols *sv, const nko_params *p,
                              nko_particles *P, double *T_sv, double *E_sv, int64_t *N_sv, double *E_raw);
void nko_tally(const nko_material *mat, const nko_subvols *sv, const nko_params *p, nko_particles *P,
               const double *T_sv, int64_t *N_sv, double *E_raw);
void nko_update_T(const nko_material *mat, const nko_subvols *sv, const nko_params *p, const int64_t *N_sv,
                  const double *E_raw, double *T_sv, double *E_sv);
void nko_assign_T(const nko_subvols *sv, const double *T_sv, nko_particles *P);
void nko_lifetime_scattering(const nko_material *mat, const nko_params *p, nko_particles *P);
void nko_heat_flux(const nko_material *mat, const nko_subvols *sv, const nko_params *p,
                   const nko_particles *P, const int64_t *N_sv, double *flux /* [S*3] W/m^2 */);
int64_t nko_contains_check(const nko_material *mat, const nko_mesh *mesh, const nko_params *p,
                           int64_t step, nko_particles *P);

/* ---- RNG (counter based, shared spec with the HIP kernels; DESIGN.md "RNG") ---- */
void nko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void nko_uniform2(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *u0, double *u1);

#ifdef __cplusplus
}
#endif
#endif
