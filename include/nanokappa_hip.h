/* nanokappa_hip.h -- C ABI of libnanokappa_hip.so: the MI355X (gfx950) engine for Nano-kappa's
 * Population timestep loop.
 *
 * The reference (brunohs1993/Nanokappa) is pure Python with no FFI layer; the boundary this
 * library replaces is the body of `Population.run_timestep` (classes/Population.py:1724-1769)
 * and the helpers it calls.  Each entry point names the reference interface it stands for.
 * Python binds it with ctypes (nanokappa_amd/engine.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: plain pointers + sizes, caller owns every host buffer (the library copies, never
 * frees caller memory), the library owns device memory.  Every function returns 0 on success or a
 * negative nk_status; nk_last_error() gives the text.  A context is used from one host thread.
 * Units are the reference's: angstrom, ps, K, eV, rad/ps.  All real data is IEEE double.
 */
#ifndef NANOKAPPA_HIP_H
#define NANOKAPPA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nk_ctx nk_ctx;

enum nk_status {
    NK_OK = 0,
    NK_ERR_HIP = -1,        /* a HIP runtime call failed */
    NK_ERR_ARG = -2,        /* invalid argument / call order */
    NK_ERR_CAPACITY = -3,   /* particle capacity exceeded (raise it with nk_reserve) */
    NK_ERR_COMM = -4,       /* RCCL failure */
    NK_ERR_NODEVICE = -5    /* no usable gfx950 device */
};

/* Phonon tables, reference classes/Phonon.py (attributes read by Population: omega :165-167,
 * group_vel :181-183, lifetime :326-336, temperature_function / crystal_energy_function :372-390,
 * normalise_to_density :392-401, number_of_active_modes :126) */
typedef struct {
    int32_t Q, J, NT;
    const double *omega;         /* [Q*J]    rad/ps */
    const double *group_vel;     /* [Q*J*3]  angstrom/ps */
    const double *T_grid;        /* [NT]     K, ascending */
    const double *lifetime;      /* [NT*Q*J] ps; 0 means "relax fully" */
    int32_t nE;                  /* length of the E(T) table */
    double T_fill_lo, T_fill_hi; /* clamp values of T(E) outside the table */
    const double *T_array;       /* [nE] */
    const double *energy_array;  /* [nE] eV/angstrom^3, ascending */
    double hbar, kb;             /* classes/Constants.py:7-8 */
    double QV;                   /* number_of_qpoints * volume_unitcell */
    int32_t active_modes;
} nk_material;

/* Triangle mesh + boundary conditions, reference classes/Mesh.py:205-242, :314-327 and
 * classes/Geometry.py:652-709 (bound_cond), :711-726 (connected_facets) */
typedef struct {
    int32_t F;                     /* triangular faces */
    const double *normals;         /* [F*3] face_normals */
    const double *k;               /* [F]   face_k */
    const double *bounds_lo;       /* [F*3] face_bounds[0] */
    const double *bounds_hi;       /* [F*3] face_bounds[1] */
    const double *basis;           /* [F*9] face_basis_matrix (F,3,3) */
    const double *origins;         /* [F*3] face_origins */
    const int32_t *face_facet;     /* [F]   face_facets */
    const double *vertices;        /* [F*9] corners of each face (surface sampling, Mesh.py:939) */
    const double *face_area;       /* [F] */
    int32_t Fc;                    /* facets (groups of coplanar adjacent faces) */
    const int8_t *facet_bc;        /* [Fc] 'T', 'F', 'P' or 'R' */
    const int32_t *facet_partner;  /* [Fc] periodic partner facet, -1 if none */
    const double *facet_centroid;  /* [Fc*3] */
    const double *facet_normal;    /* [Fc*3] */
    const int32_t *facet_face_off; /* [Fc+1] CSR: faces of each facet */
    const int32_t *facet_face_idx;
    double tol;                    /* Mesh.tol (1e-10) */
    double bbox[6];                /* Geometry.bounds: lo xyz, hi xyz */
    int32_t nS;                    /* volume simplices for Mesh.sample_volume (Mesh.py:890-904) */
    const double *simplex_pts;     /* [nS*12] */
    const double *simplex_vol;     /* [nS] */
} nk_mesh;

/* Subvolumes, reference classes/Geometry.py:446-544 and SubvolClassifier :1198-1213 */
typedef struct {
    int32_t S;
    int32_t kind;            /* 0 slice (centres ascending along `axis`), 1 general nearest centre */
    int32_t axis;
    int32_t interp;          /* per-particle T: 0 interp1d 'nearest' on slices, 1 'linear' on slices,
                                2 nearest centre, 3 cubic radial basis functions = scipy RBFInterpolator
                                (Population.py:570-590, :694-702) */
    const double *centers;   /* [S*3] */
    const double *volumes;   /* [S] */
    /* interp 3 only (else NULL / 0): inverse of the RBF system of these centres, row-major (P x P), P = S + n_used
     * + 1; shift and scale of its polynomial part; which coordinates take part (Population.py:651-656) */
    const double *rbf_inv;
    const double *rbf_shift; /* [3] */
    const double *rbf_scale; /* [3] */
    int32_t rbf_used[3];
} nk_subvols;

/* Reservoirs, reference Population.py:323-354 (initialise_reservoirs), :146-161 (enter_probability) */
typedef struct {
    int32_t R;
    const int32_t *facet;      /* [R] facet index of each reservoir */
    const double *T;           /* [R] imposed temperature */
    const double *enter_prob;  /* [R*Q*J] */
    const double *counter;     /* [R*Q*J] initial res_counter (Population.py:343) */
    int32_t gen;               /* 0 'constant', 1 'fixed_rate', 2 'one_to_one' (Population.py:358-489) */
    const int64_t *n_leaving;  /* [R] 'one_to_one' only: particles to emit at the first step (Population.py:344:
                                * round(sum of enter_prob)), all ranks together; afterwards the engine emits what
                                * left through each reservoir at the previous step (:466, :1585).  NULL otherwise */
} nk_reservoirs;

/* Rough facets, reference Population.py:852-877 (specularity), :1042-1461 (specular map),
 * :879-939 (creation_roulette), :1017-1040 (degeneracies) */
typedef struct {
    int32_t Fr;
    const int32_t *facet;       /* [Fr] */
    const double *specularity;  /* [Fr*Q*J] */
    const uint8_t *true_spec;   /* [Fr*Q*J] */
    const int32_t *spec_map;    /* [Fr*Q*J] flat out-mode (q*J+j), -1 where not specular */
    const double *roulette;     /* [Fr*Q*J] cumulative, last = 1 */
    const int32_t *degen_j2;    /* [Q*J] partner branch for the 'k' model, or NULL */
} nk_rough;

/* Scalars of Population.__init__ (Population.py:41-95) */
typedef struct {
    double dt;                /* --timestep */
    int32_t norm_fixed;       /* --energy_normal: 0 'mean', 1 'fixed' */
    double particle_density;
    int32_t T_ref_local;      /* --reference_temp local */
    double T_ref;
    int32_t flux_every;       /* tally the subvolume heat flux every this many steps (n_dt_to_conv = 10) */
    int32_t contains_every;   /* contains_check period (100, Population.py:1729-1734); 0 = never */
    int32_t track_ids;        /* 0: 64-bit particle ids are stored only when the configuration draws random numbers per
                               *    particle (rough facets) -- 44 instead of 52 bytes of state per particle; downloads then
                               *    return pid = 0.  1: always stored (ids as uploaded / as nk_upload_particles numbers them).
                               *    The reference has no particle ids; they only key the counter-based RNG. */
} nk_params;

/* Per-step results of nk_step; every pointer may be NULL.  Row r describes the r-th step of the call.
 * Raw sums are un-normalised (the host applies Population.py:719-728 / :738-747 scalings where the
 * library has not already done so).  All sums are over ALL ranks once nk_comm_init was called. */
typedef struct {
    double *T_sv;        /* [nsteps*S]   subvol_temperature after the step (Population.py:692) */
    double *E_sv;        /* [nsteps*S]   subvol_energy, normalised + reference (Population.py:724-728) */
    double *E_raw;       /* [nsteps*S]   sum_i hbar*omega_i*dn_i per subvolume (Population.py:715-717) */
    double *N_sv;        /* [nsteps*S]   subvol_N_p (Population.py:679) */
    double *flux_raw;    /* [nsteps*S*3] sum_i v_i*e_i (Population.py:736); NaN on steps without flux tally */
    double *N_leaving;   /* [nsteps*R]   particles absorbed by each reservoir (Population.py:1585) */
    double *res_energy;  /* [nsteps*R]   this step's increment of res_energy_balance (Population.py:1595) */
    double *res_flux;    /* [nsteps*R*3] this step's increment of res_heat_flux (Population.py:1602) */
    double *N_emitted;   /* [nsteps]     particles that entered from reservoirs (Population.py:370) */
} nk_tally;

/* Kernel timing of the last nk_step call, from HIP events on the library's stream. */
typedef struct {
    double step_kernel_ms;   /* mean duration of k_sweep: relax + drift + boundary events + emission + tally */
    double emit_kernel_ms;   /* mean duration of the reservoir emission (k_emit; + k_emit_one_to_one for that generator) */
    double events_kernel_ms; /* mean duration of the step's tail: k_reduce (+ all-reduce) + k_update */
    double total_ms;         /* wall time of the whole call on the stream */
    int64_t slots;           /* particle capacity (nseg * segcap) */
    int64_t live;            /* live particles after the call (this rank) */
    /* housekeeping that can land inside a timed region, counted since nk_create: a benchmark reads them per region */
    int64_t regrows;         /* times the particle store was grown on the device after a halt request */
    int64_t halts;           /* batches that ended early on a halt request (every rank at the same step) */
    int64_t tau_rebuilds;    /* rebuilds of the packed mode records (lifetime window / E0 reference moved, or new segmentation) */
    int64_t batches;         /* nk_step_batch calls = stream drains + history copies */
    int64_t emit_fused;      /* 1: the next step's emission ran inside the tail launch (k_tail) in the last call: emit_kernel_ms then
                              *    covers only a batch's first step, events_kernel_ms the reduce / update WITH the emission beside it */
    int64_t place_tries;     /* allocations of the particle store that were timed when it was last (re)allocated (0: small store, not
                              * timed): where a store lies in memory decides between two speeds of the sweep, 15 % apart */
    double place_gbps;       /* GB/s of an in-place copy pass over the store that was kept ... */
    double place_worst_gbps; /* ... and over the slowest candidate */
    int64_t box_store;       /* 1: the particle store holds no cached next hit (axis-aligned box meshes: the hit is read off the
                                position, the reference's expression of Mesh.py:818 at event time); 36 B per particle, else 44 */
} nk_timing;

/* lifetime: `Population.__init__` / end of run */
int nk_device_count(void);                            /* HIP devices this process sees (0 if none) */
/* Set-up helper of the host geometry (no context): for every ray o + t d, t > 0 the number of triangles (v0, e1 = v1 - v0,
 * e2 = v2 - v0; arrays [n*3]) it crosses, crossings at the same distance (to 1e-8) counted once; skip_self: ray i ignores
 * triangle i.  The inside tests of nanokappa_amd.mesh (role of trimesh's ray queries in the reference's Mesh.py) run this on
 * large meshes; counts[i] = -1 where a ray has more than 24 distinct crossings (the caller counts that ray itself). */
int nk_mesh_crossings(int device, int64_t n_rays, const double *origins, const double *dirs, int64_t n_faces, const double *v0,
                      const double *e1, const double *e2, int skip_self, int32_t *counts);
int nk_create(nk_ctx **out, int device_id, uint64_t seed);
void nk_destroy(nk_ctx *ctx);
const char *nk_last_error(const nk_ctx *ctx);        /* ctx may be NULL: error of a failed nk_create */

/* setup tables (copied to HBM) */
int nk_set_material(nk_ctx *ctx, const nk_material *m);
int nk_set_mesh(nk_ctx *ctx, const nk_mesh *m);
int nk_set_subvolumes(nk_ctx *ctx, const nk_subvols *s, const double *T_sv_init /* [S] */);
int nk_set_reservoirs(nk_ctx *ctx, const nk_reservoirs *r);
int nk_set_rough(nk_ctx *ctx, const nk_rough *r);
int nk_set_params(nk_ctx *ctx, const nk_params *p);

/* particle state: Population.initialise_all_particles (Population.py:186-321) hands over positions, modes
 * and occupations; n_ts/facet/pid may be NULL (then nk_init_boundaries computes the first two, and pid = index
 * + pid_offset) */
int nk_reserve(nk_ctx *ctx, int64_t capacity);
int nk_upload_particles(nk_ctx *ctx, int64_t N, const double *x, const double *y, const double *z,
                        const int32_t *mode, const double *occ, const double *n_ts, const int32_t *facet,
                        const uint64_t *pid, uint64_t pid_offset);
/* Population.timesteps_to_boundary for the whole population (Population.py:310-314) */
int nk_init_boundaries(nk_ctx *ctx);
/* Population.initialise_all_particles on the device (Population.py:186-321), instead of nk_reserve + nk_upload_particles, for
 * the common case: modes tiled over the particle ids (initialise_modes, :127-144: particle p has mode unique_modes[p %
 * n_unique], flat indices q * J + j of the active modes), ids pid_lo .. pid_lo + N - 1, positions uniform in the solid
 * (Mesh.sample_volume, Mesh.py:890-904; sv_first NULL = 'random_domain') or uniform in the subvolume the particle's id
 * belongs to (sv_first[S + 1] ascending from 0: id p in [sv_first[s], sv_first[s + 1]) lies in subvolume s =
 * 'random_subvol', :222-246; ids, not local indices, so that the shards of several ranks make up the single-rank ensemble), occupations Bose-Einstein at the subvolume's temperature (:280).  Needs nk_set_material,
 * nk_set_mesh (with the volume tables), nk_set_subvolumes and the boundary-condition tables; nk_init_boundaries follows. */
int nk_init_particles(nk_ctx *ctx, int64_t N, int64_t capacity, uint64_t pid_lo, const int32_t *unique_modes, int64_t n_unique,
                      const int64_t *sv_first);
/* calculate_energy and the heat-flux sums of the particles where they stand, before normalisation (Population.py:704-717,
 * :734-736; the reference's t = 0 row): E_raw[S], N_sv[S], flux_raw[S * 3].  This rank's particles only. */
int nk_tally_state(nk_ctx *ctx, double *E_raw, double *N_sv, double *flux_raw);
/* Population.run_timestep x nsteps (Population.py:1724-1769) without the file output.  The particle store grows by
 * itself (on the device, nothing is dropped) when the ensemble outgrows it, like the reference's arrays do; NK_ERR_CAPACITY
 * only if that growth fails (out of memory), with the state of the last completed step intact. */
int nk_step(nk_ctx *ctx, int32_t nsteps, nk_tally *out);
/* live particles, in slot order; arrays may be NULL; *N_out receives the count (call with capacity 0 to query) */
int nk_download_particles(nk_ctx *ctx, int64_t capacity, double *x, double *y, double *z, int32_t *mode,
                          double *occ, double *n_ts, int32_t *facet, uint64_t *pid, int64_t *N_out);
int nk_get_subvol_temperature(nk_ctx *ctx, double *T_sv /* [S] */);
int nk_set_subvol_temperature(nk_ctx *ctx, const double *T_sv /* [S] */);
int nk_get_step(nk_ctx *ctx, int64_t *step);
int nk_get_timing(nk_ctx *ctx, nk_timing *t);

/* multi-GPU: one context per rank; tallies are all-reduced (sum, f64) over RCCL every step */
int nk_comm_unique_id(void *id128 /* 128 bytes out */);
/* Joins the communicator and PROVES it before returning: the rank count and this rank's index are read back from RCCL
 * (ncclCommCount / ncclCommUserRank) and a vector {1, rank + 1} is all-reduced on the context's stream; NK_ERR_COMM unless
 * the sums are nranks and nranks (nranks + 1) / 2, i.e. unless exactly the expected ranks took part. */
int nk_comm_init(nk_ctx *ctx, const void *id128, int rank, int nranks);
/* What the communicator itself says (not what the caller passed): for logs and the bench line. */
typedef struct {
    int32_t rank, nranks;            /* as given to nk_comm_init (1 rank before it is called) */
    int32_t comm_rank, comm_nranks;  /* ncclCommUserRank / ncclCommCount; -1 / 0 when there is no communicator (one rank, dry run) */
    int32_t device;                  /* HIP device of this context */
    int32_t selftest_ok;             /* 1: the all-reduce of ones at nk_comm_init returned nranks */
    double selftest_sum;             /* what that all-reduce returned (0 without communicator) */
    char pci_bus_id[32];             /* hipDeviceGetPCIBusId of that device */
} nk_comm_report;
int nk_comm_info(nk_ctx *ctx, nk_comm_report *out);
/* Sum of a small host vector over the ranks of the communicator (in place; unchanged when there is none): the t = 0 tallies
 * of the shards (Population.py:282, :318-321 on an ensemble that is spread over the ranks). */
int nk_comm_allreduce(nk_ctx *ctx, double *inout, int64_t n);

/* device versions of the reference's primitives, for parity tests (tests/ -m gpu) */
int nk_find_boundary(nk_ctx *ctx, int64_t n, const double *x /* [n*3] */, const double *v /* [n*3] */,
                     double *xc, double *tc, int32_t *fc);                       /* Mesh.py:806-856 */
int nk_classify(nk_ctx *ctx, int64_t n, const double *x, int32_t *id);           /* Geometry.py:1212 */
int nk_eval(nk_ctx *ctx, int32_t what, int64_t n, const double *a, const int32_t *mode, double *out);
/* what: 0 occupation(T=a[i], omega[mode[i]]) Phonon.py:338; 1 lifetime(T=a[i], mode[i]) Phonon.py:336;
 *       2 T(E=a[i]) Phonon.py:387; 3 E(T=a[i]) Phonon.py:390; 4 per-particle T at x=a[3i..] Population.py:696;
 *       5 the kernels' own exp(a[i]) */
int nk_reflect(nk_ctx *ctx, int64_t n, const int32_t *facet, const int32_t *mode_in, const double *col_pos,
               const double *n_in, const double *omega_in, const double *r_spec, const double *r_deg,
               const double *r_diff, int32_t *mode_out, double *n_out, double *omega_out); /* Population.py:941-1015 */
int nk_uniform2(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *u0, double *u1);

/* measurement helper: `launches` sweeps of the particle arrays with a known byte count per launch (returned), in
 * the access shape of the step kernel, so that rocprofv3 FETCH_SIZE / WRITE_SIZE readings can be calibrated */
int nk_calibrate_stream(nk_ctx *ctx, int32_t launches, int64_t *bytes_read, int64_t *bytes_written);

/* Set-up table builder: find_specular_correspondences, 'velocity' model
 * (Population.py:1241-1454) for one surface normal.  group_vel [M*3], omega [M], delta_omega [M] (the grid tolerance
 * of :1245-1247) are uploaded by nk_specular_begin; nk_specular_pairs returns every (in-mode, out-mode) pair of flat
 * mode indices, unordered, for the (rounded, inward) normal; *n_pairs receives the count (call again with a larger
 * `cap` when it exceeds it).  pair_in = pair_out = NULL: the pairs stay on the device (for nk_rough_pairs), none returned. */
int nk_specular_begin(nk_ctx *ctx, int64_t M, const double *group_vel, const double *omega, const double *delta_omega);
int nk_specular_pairs(nk_ctx *ctx, const double *normal /* [3] */, double crit, int64_t cap, int32_t *pair_in,
                      int32_t *pair_out, int64_t *n_pairs);
int nk_specular_end(nk_ctx *ctx);


/* The rough-facet tables built on the device and installed in place of nk_set_rough ('velocity' reflection model):
 * calculate_fbz_specularity (Population.py:852-877), true_specular and the specular map of find_specular_correspondences
 * (:1457-1459), diffuse_scat_probability (:879-939).  Protocol, inside nk_specular_begin .. nk_specular_end:
 *   nk_rough_begin(Fr, facet[Fr], inward normals [Fr*3] (= -facets_normal), eta [Fr], |k| per q-point [Q]);
 *   for every distinct (rounded) normal: nk_specular_pairs(normal, ...), then nk_rough_pairs(the rough-facet indices that
 *   share it) -- the pairs are taken from the device, where the search left them;
 *   nk_rough_finish(): specularity, creation rates, their cumulative roulette and its bucket index; the tables stay on the
 *   device.  nk_rough_download copies them back (tests, host attributes); any pointer may be NULL. */
int nk_rough_begin(nk_ctx *ctx, int32_t Fr, const int32_t *facet, const double *normal_in, const double *eta, const double *k_norm);
int nk_rough_pairs(nk_ctx *ctx, int32_t nf, const int32_t *fidx);
int nk_rough_finish(nk_ctx *ctx);
/* The 'k' / wavevector reflection model (Population.py:1056-1240) through the same calls: after nk_specular_begin,
 * nk_kspec_begin uploads the wavevectors [Q*3] with Phonon.k_to_q / q_to_k as row-major 3 x 3 matrices (q = k . k_to_q,
 * k = q . q_to_k) and tol[3] = q_to_k(|1 / (2 mesh)|); nk_kspec_pairs is nk_specular_pairs for that model (one partner per
 * in-mode); nk_rough_finish_k also averages the creation rates of the degenerate branches (find_degeneracies :1017-1040:
 * nd rows q, j1, j2, applied in order, :926-930) and installs degen_j2 [M] (as in nk_rough) for the reflection's coin flip. */
int nk_kspec_begin(nk_ctx *ctx, int64_t Q, const double *wavevectors, const double *k_to_q, const double *q_to_k, const double *tol);
int nk_kspec_pairs(nk_ctx *ctx, const double *normal /* [3] */, int64_t cap, int32_t *pair_in, int32_t *pair_out, int64_t *n_pairs);
int nk_rough_finish_k(nk_ctx *ctx, int32_t nd, const int32_t *degen /* [nd*3] */, const int32_t *degen_j2 /* [M] or NULL */);
int nk_rough_download(nk_ctx *ctx, double *specularity, uint8_t *true_spec, int32_t *spec_map, double *roulette);
/* enter_probability (Population.py:146-161) on the device: out[r*M + m] = max(0, v_m . n_in_r) * dt / thickness_r */
int nk_build_enter_prob(nk_ctx *ctx, int32_t R, const double *normal_in, const double *thickness, double dt, double *out);

#ifdef __cplusplus
}
#endif
#endif
