// nk_device.h -- device-side data model and per-particle physics of the MI355X engine.
//
// One particle = one lane.  Particle state is SoA in HBM: x, y, z, occupation, time-to-boundary (doubles), ONE packed
// 32-bit word {next facet + 1 | mode index} and -- only when something draws random numbers per particle (rough facets)
// or the caller asks for it -- a 64-bit particle id that keys the counter-based RNG: 44 or 52 bytes per particle.
// The store is split into `nseg` equal segments; a segment holds its live particles contiguously from its start and is
// owned by one wave at a time.
// Modes are PARTITIONED over the segments: every mode has a slot = local index * nseg + segment (NkDev::m2s / s2m) and is
// stored as its local index (a few bits).  The slots are dealt on the host (nk_build_mode_map) in the order of the modes'
// boundary-event rate, each mode to the segment that is furthest behind its share of the work (particles + events), so that
// every segment gets its share of fast and slow modes: a wave owns a segment for a whole sweep, and with "segment = m % nseg"
// a segment count that is a multiple of the branch count gave every segment ONE branch -- events per segment between 4 and
// 22 per step around a mean of 8.6 in the 200 A box, the slowest wave 1.45 x the mean (profiles/r03_notes.txt).  The shares
// are not equal: the SIMD's arbiter issues the oldest wave first, so of the workgroups resident on a CU the first-dispatched
// runs ahead (tile loops of 164 / 183 / 209 us for equal work); a segment's share follows its workgroup's place in that order.  A segment therefore only ever touches its own few dozen 64-byte mode records (they sit contiguously in a
// permuted copy of the mode table and stay in L1/L2), and it emits the reservoir particles of its own modes itself: no
// spawn list, no global atomics, no gathers across an 11 MB table.  (With rough facets a reflection changes the mode: such a
// particle finishes its step where it is and then migrates to the owning segment through that segment's inbox, k_deliver.)
// Small read-only tables (planes, faces, facets, per-slice {centre, T, slope, 1/T}) and the tally bins live in LDS.
//
// Reference semantics cited as file:line under the reference checkout (classes/Population.py etc.).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NK_WG 256            // threads per workgroup (4 waves of 64)
#ifndef NK_NREP
#define NK_NREP 8            // LDS replicas of the tally bins (lane & 7) to thin same-address atomics
#endif
#define NK_PLANE_DOUBLES 6   // nx ny nz k {face_begin, face_end} pad  (48 B: three aligned 16-byte reads)
#define NK_FACE_DOUBLES 16   // lo(3) hi(3) o(3) iu(3) iw(3) {orig_face, facet}
#define NK_LDS_FACES 256     // meshes up to this many faces keep their plane/face tables in LDS
#define NK_LDS_RESFACES 64   // reservoir facets with up to this many faces in total keep their sampling tables in LDS
#define NK_TAU_ROWS 3        // lifetime rows packed into each mode record (two intervals of the temperature grid)
#define NK_MAX_SEGMENTS 16384 // upper bound of nseg
#define NK_EMIT_CHUNK 128    // (reservoir, mode) entries a wave evaluates at a time (two per lane)
#define NK_LREC 128          // most mode records a wave of the sweep stages in LDS (segments that own more read them from L2)
#define NK_NEWBORN 0x80000000u
#define NK_LOST 0x40000000u  // box store (NkDev::box): the particle's last ray cast was a miss -- the reference's n_timesteps = inf, facet -1

// RNG stream tags (shared spec with the oracle; DESIGN.md "RNG")
#define NK_TAG_REFLECT 0x00000u
#define NK_TAG_EMIT 0x10000u
#define NK_TAG_RESAMP 0x20000u
#define NK_TAG_DICE 0x30000u
#define NK_TAG_INIT 0x40000u      // k_init_particles: 3 tags per draw of a position

// A particle field in HBM.  The store is cut into BLOCKS of 64 slots (one tile of the sweep); a block holds its 64 x, then
// its 64 y, z, occupations, times to the boundary, (ids,) and packed words one after the other -- 2816 bytes (3328 with
// ids) that a wave reads as ONE contiguous piece.  (Round 2 kept six separate arrays: six DRAM streams per wave whose
// relative placement -- the arrays are 2 MB-aligned allocations, so slot i of every array shares its low address bits --
// decided the memory system's efficiency: the same binary ran 10-25 % apart from one process to the next.)
// blk = distance between two blocks in units of T; slot i lives at p[(i / 64) * blk + i % 64].  blk = 64 with separate
// base pointers gives back the plain arrays (NK_LAYOUT=soa, developer comparison).
template <class T>
struct NkField {
    T *p;
    int32_t blk;
    __host__ __device__ __forceinline__ int64_t off(int64_t i) const { return (i >> 6) * (int64_t)blk + (i & 63); }
    __host__ __device__ __forceinline__ T &operator[](int64_t i) const { return p[off(i)]; }
    __host__ __device__ __forceinline__ T *operator+(int64_t i) const { return p + off(i); }
    // slot i0 + lane of the tile that starts at slot i0 (a multiple of 64, the same for the whole wave: scalar arithmetic)
    __host__ __device__ __forceinline__ T *tile(int64_t i0, int lane) const { return p + ((i0 >> 6) * (int64_t)blk + lane); }
    __host__ __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};

// The same geometry for code that knows at compile time which fields the store has (the sweep): every field of slot i is a
// fixed distance from the slot's x, so a tile's loads and stores share ONE address and differ by immediate offsets -- one
// 64-bit address computation per tile instead of one per field, and one base pointer in scalar registers instead of six.
template <bool NTS, bool PID>
struct NkBlock {
    static constexpr int NF = 4 + (NTS ? 1 : 0) + (PID ? 1 : 0);     // 8-byte fields per particle
    static constexpr int BD = NF * 64 + 32;                           // doubles per block (nk_engine.hip nk_store_bytes)
    static constexpr int O_Y = 64, O_Z = 128, O_OCC = 192, O_NTS = 256, O_PID = 64 * (NTS ? 5 : 4), O_W0 = 64 * NF;   // in doubles
    // x of slot i (any slot) / of lane `lane` in the tile that starts at slot i0 (a multiple of 64)
    static __device__ __forceinline__ double *slot(double *s0, int64_t i) { return s0 + (i >> 6) * (int64_t)BD + (i & 63); }
    static __device__ __forceinline__ double *tile(double *s0, int64_t i0, int lane) { return s0 + (i0 >> 6) * (int64_t)BD + lane; }
    // the packed word that belongs to the slot whose x is at q
    static __device__ __forceinline__ uint32_t *word(double *s0, int64_t i) { return reinterpret_cast<uint32_t *>(s0 + (i >> 6) * (int64_t)BD + O_W0) + (i & 63); }
};

struct __attribute__((aligned(16))) NkFacet {   // 96 bytes
    double cx, cy, cz;    // centroid
    double nx, ny, nz;    // outward normal
    double tx, ty, tz;    // periodic facets: centroid(partner) - centroid(this), the translation of Population.py:1467
    int32_t bc;           // 'T','F','P','R'
    int32_t partner;      // periodic partner facet or -1
    int32_t res;          // reservoir index or -1
    int32_t rough;        // rough-facet index or -1
    int32_t pad[2];
};

struct __attribute__((aligned(64))) NkMode {   // 64 bytes, read as two 32-byte halves
    double omega, vx, vy, vz;
    double E0;                                 // exp(hbar omega / (kB T0)) at the window's reference temperature T0
    double tau[NK_TAU_ROWS];                   // lifetime at T_grid[tau_row0 .. tau_row0+2]
};

// Everything a kernel needs, passed by value (pointers are device pointers).
struct NkDev {
    // ---- material
    int32_t Q, J, NT, M;              // M = Q*J
    const NkMode *modetab;            // [M] by mode index
    const NkMode *modetab_p;          // [nseg * nlmax] permuted copy: record of the mode with slot l * nseg + s at s * nlmax + l
    const int32_t *m2s;               // [M] slot of a mode (slot % nseg = its segment, slot / nseg = its local index there)
    const int32_t *s2m;               // [nseg * nlmax] mode of (segment s, local index l) at s * nlmax + l; -1 where there is none
    const int32_t *seg_nl;            // [nseg] modes a segment owns (its local indices 0 .. seg_nl - 1)
    int32_t tau_row0;                 // first T_grid row held in NkMode::tau
    double tau_g[NK_TAU_ROWS];        // T_grid[tau_row0 .. tau_row0+2] by value (scalar registers, no loads)
    double tau_ig[NK_TAU_ROWS - 1];   // 1 / (g[k+1] - g[k]) of the two packed intervals
    double T0, invT0;                 // reference temperature of NkMode::E0 (middle of the live temperature range)
    double c_hk;                      // hbar / kB
    const double *tau;                // [NT*M] full table (fallback outside the packed window)
    const double *Tgrid;              // [NT]
    int32_t nE;
    const double *Tarr, *Earr;        // [nE]
    double Tfill_lo, Tfill_hi;
    double hbar, kb, QV;
    double active_modes;
    // ---- mesh
    int32_t F, Fc, NP;                // faces, facets, distinct planes
    const float *tree_boxes;          // large meshes: node boxes (6 floats, rounded outwards) of the face tree, level by level; or NG = 0
    double tree_bound;                // largest |coordinate| of a box (error bound of the single-precision slab test)
    const int32_t *tree_tags;         // [nodes] facet of a node whose faces all belong to one facet, else -1
    const double *facet_skip;         // [2*Fc] per facet: largest |n_face - n_facet|, largest |plane_face(centroid)| (nk_tree_skip)
    const double *tree_faces;         // [tree_leaves * NK_TREE_LEAF_DOUBLES] leaf records: the four faces' boxes, then the faces (padded with null faces)
    int32_t tree_base[8];             // first node of each level in tree_boxes
    int32_t tree_top, tree_leaves;
    // k_events: the families from index tree_lds_fam0 up (whole levels from the top of the tree; the levels are stored bottom-up, so these
    // are the array's tail) are copied to LDS at byte offset tree_lds_off of the workgroup's dynamic area (set per launch, nk_step_batch)
    int32_t tree_nfam, tree_lds_fam0, tree_lds_off;
    int32_t NG;                       // 1: walk the face tree, 0: sweep all planes
    const double *planes;             // [NP*NK_PLANE_DOUBLES]
    const double *faces;              // [F*NK_FACE_DOUBLES], grouped by plane
    const NkFacet *facets;            // [Fc]
    double tol;
    double bbox[6];
    // ---- box store (round 4): the mesh is an axis-aligned box whose six sides are its six facets (nk_set_mesh decides).  The
    // particles then carry NO cached next hit (no nts field, no facet bits): whether a particle meets a wall inside the step is
    // read off its end-of-step position (outside a wall and flying outwards), and the hit itself -- time and facet -- is
    // evaluated when the event is run, with the reference's own expression t = -(x.n + k) / (v.n) (Mesh.py:818; for unit axis
    // normals the two dot products are exact, x.n + k is one rounding).  36 B per particle instead of 44.
    int32_t box;                      // 1: box store
    double box_k[6];                  // plane constants k (n.x + k = 0) of the walls: [2 a] the wall with normal -e_a, [2 a + 1] +e_a
    uint64_t box_ids;                 // per wall w, bits [8 w, 8 w + 4): its facet, bits [8 w + 4, 8 w + 8): its lowest face index
                                      // (Mesh.find_boundary: the lowest face index wins a tie); one scalar pair instead of twelve
    double inv_dt;                    // 1 / dt
    const double *face_verts;         // [F*9] original face order
    const int32_t *facet_face_off;    // [Fc+1]
    const int32_t *facet_face_idx;
    const double *facet_face_cdf;     // per facet, cumulative area fraction of its faces (same CSR)
    int32_t nS;
    const double *simplex_pts;        // [nS*12]
    const double *simplex_cdf;        // [nS]
    // ---- subvolumes
    int32_t S, sv_kind, sv_axis, sv_interp;
    const double *centers;            // [S*3]
    const double *sv_volume;          // [S]
    double sv_lo, sv_invL;            // slice fast path: first edge and 1/slice_length
    double *T_sv;                     // [S + rbf_P] current subvolume temperatures (updated by k_update), followed by the
                                      // RBF coefficients [w (S); p (n_used + 1)] when sv_interp == 3
    const double *rbf_inv;            // [rbf_P * rbf_P] inverse of the RBF system (sv_interp 3)
    int32_t rbf_P, rbf_used[3];
    double rbf_shift[3], rbf_scale[3];
    // ---- reservoirs
    int32_t R, res_gen;
    const int32_t *res_facet;         // [R]
    const double *res_T;              // [R]
    const int32_t *res_face_off;      // [R+1] faces of each reservoir facet (CSR), in Mesh.sample_surface order
    const double *res_face_cdf;       // cumulative area fractions (np.random.choice), same CSR
    const double *res_face_verts;     // 9 doubles per face
    int32_t res_nf;                   // total faces of all reservoir facets
    int32_t res_lds;                  // 1: the three tables above are staged in LDS by the sweep
    const double *enter_prob;         // [R*M]
    double *res_counter;              // [R*M] as uploaded; the live counters are rc_p (nk_engine.hip keeps the two in step)
    const double *ep_p;               // [nseg * R * nlmax] enter_prob in the segments' order: entry (r, l) of segment s at
    double *rc_p;                     //   (s * R + r) * nlmax + l -- k_emit reads its entries coalesced; rc_p: the counters,
    int64_t rc_len;                   //   TWO copies of rc_len entries: step k reads copy k & 1 and writes copy (k + 1) & 1, so the
                                      //   emission that k_tail runs ahead of a halt can simply be run again (nk_step_batch)
    const double *res_roulette;       // [R*M] 'one_to_one': cumulative enter_prob per reservoir, last = 1 (Population.py:467-468)
    int32_t *nleave_prev;             // [R] 'one_to_one': particles that left at the previous step, all ranks (Population.py:466)
    uint64_t *sp_inbox;               // 'one_to_one': [nseg * sp_icap] spawn records (i << 40 | rm << 12) routed to the owner
    int32_t *sp_inbox_n;              //   segment of their mode by k_emit_one_to_one; [nseg] counts (reset by the sweep)
    int32_t sp_icap;
    // ---- rough facets
    int32_t Fr;
    const double *specularity;        // [Fr*M]
    const uint8_t *true_spec;         // [Fr*M]
    const int32_t *spec_map;          // [Fr*M]
    const double *roulette;           // [Fr*M]
    const int32_t *roul_lut;          // [Fr*(roul_nlut+1)] roul_lut[f][k] = searchsorted(roulette[f], k/roul_nlut * last)
    int32_t roul_nlut;                // buckets of that index: a power of two near M / 4 (1024 .. 65536)
    const int32_t *degen_j2;          // [M] or null
    // ---- parameters
    double dt;
    int32_t norm_fixed, T_ref_local;
    double particle_density, T_ref;
    uint64_t seed;
    int32_t rank, nranks;
    // ---- particles: nseg * segcap slots in blocks of 64 (NkField); segment s = slots [s*segcap, s*segcap + seg_count[s])
    int64_t cap;
    int32_t nseg, segcap;
    int32_t *seg_count;               // [nseg] live particles per segment (contiguous from the segment start)
    int32_t *seg_new;                 // [nseg] particles k_emit appended behind them at this step (the sweep takes them in)
    // Alternating walk (fused sweeps of small meshes, box store or cached store).  The store is larger than the 256 MB memory-side cache, and a sweep that walks every segment
    // front to back begins with the blocks the sweep before wrote FIRST -- the ones that cache has dropped again.  So the sweeps
    // take turns: an UP sweep reads its segment's particles from the lowest slot upwards and packs the survivors upwards from
    // there; a DOWN sweep (`down`, set per launch) reads from the highest slot downwards and packs the survivors downwards from
    // there.  Either way a sweep starts with what the sweep before wrote last, and writes right behind its read cursor.  The
    // live particles of segment s then occupy slots [seg_lo[s], seg_lo[s] + seg_count[s]) of its range: a DOWN sweep moves the
    // lower end up by the particles that died, an UP sweep starts its output at 0 again once the lower end has used up half of the
    // segment's head room.  The emission appends above the live particles as ever.  Every other kernel expects seg_lo = 0: the
    // host moves the segments down first (k_anchor).  The split sweep and the fused sweep over a face tree always walk upwards from slot 0.
    int32_t *seg_lo;
    int32_t down;
    int32_t *seg_bound;               // [nseg] upper bound of the particles that can enter the segment in one step
    NkField<double> x, y, z, occ, nts;   // nts: null in a box store
    NkField<uint32_t> w0;             // newborn << 31 | (facet + 1) << lb | idx;  idx = the mode's local index in its segment (part) or the mode itself;
                                      // newborn: appended by k_emit at this step (no relaxation, no drift yet)
                                      // box store: newborn << 31 | lost << 30 | idx
    NkField<uint64_t> pid;            // null: particle ids are not tracked (no per-particle random draws in this configuration)
    int32_t part;                     // 1: idx is the local index of a mode of the owning segment; 0: the global mode index
    int32_t lb;                       // bits of idx in w0
    int32_t nlmax;                    // modes per segment, rounded up: ceil(M / nseg)
    int32_t nlrec;                    // mode records a sweep wave keeps in LDS (= nlmax when the sweep reads its segments' records
                                      // from LDS, else 0)
    // ---- bookkeeping words in device memory
    int32_t *ticket;                  // arrival counter of k_reduce's workgroups (the last one runs the update)
    int32_t *overflow;                // set when a particle had to be dropped for lack of capacity
    int32_t *halt;                    // [0] set (by the update, at the end of a step) when a sweep saw a segment that could
                                      // overflow at the NEXT step: later steps of the same nk_step call do nothing, so the host can
                                      // grow the store with the state intact; [1] the sweep's request; [2] k_deliver could not place
                                      // a segment's migrants (they wait in its inbox); [3] an inbox is more than half full
    // ---- event queues (split sweep, large meshes): the particles that meet a boundary inside the step, per segment, in
    // arrays of their own with the segments' geometry; k_events works them off at a residency the fused sweep cannot reach
    double *qx, *qy, *qz, *qocc, *qnts;
    uint32_t *qw0;
    uint64_t *qpid;
    int32_t *seg_evq;                 // [2 nseg + 1] entries in each segment's queue, then their exclusive prefix sums (k_events_begin)
    int32_t *ev_ticket;               // next entry of the concatenated queues to hand out (k_events)
    // ---- migration (rough facets: a reflection changes a particle's mode, hence the segment that owns it)
    double2 *mig_buf;                 // [nseg * mig_cap * 4] 64-byte records {x, y} {z, occ} {nts, pid} {w0, -, -, -}
    int32_t *mig_n;                   // [nseg] records waiting in each segment's inbox (k_deliver empties them every step)
    int32_t mig_cap;
    double *partials;                 // [rows][NB] per-workgroup tally rows
    int32_t NB;                       // bins per row = 5*S + 5*R + 1
    unsigned long long *stamps;       // developer build NK_STAMPS (make stamps): per-wave cycle sums of the sweep's sections
};

// Offsets of slot i in the double-sized fields (x y z occ nts pid share their block stride) and in the packed words, computed
// once for a whole particle (hot paths; the fields' operator[] recomputes them per access).
struct NkSlot { int64_t od, ow; };
__device__ __forceinline__ NkSlot nk_slot(const NkDev &d, int64_t i) {
    const int64_t b = i >> 6, l = i & 63;
    NkSlot s;
    s.od = b * (int64_t)d.x.blk + l;
    s.ow = b * (int64_t)d.w0.blk + l;
    return s;
}

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-10, counter = {pid_lo, pid_hi, step, tag}, key = seed.  Stateless: nothing is stored per particle.
__device__ __forceinline__ void nk_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                          uint32_t o[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ double nk_u53(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void nk_uniform2_dev(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double &u0,
                                                double &u1) {
    uint32_t o[4];
    nk_philox((uint32_t)pid, (uint32_t)(pid >> 32), step, tag, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    u0 = nk_u53(o[0], o[1]);
    u1 = nk_u53(o[2], o[3]);
}

// ------------------------------------------------------------------------------------- small helpers
// np.searchsorted(a, x, side='left') / 'right'
__device__ __forceinline__ int nk_ss_left(const double *a, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int nk_ss_right(const double *a, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
// searchsorted-left with a starting guess: gallop away from `hint`, then bisect the bracket.  Same result as
// nk_ss_left; a good guess costs 2-3 dependent loads instead of log2(n).
__device__ __forceinline__ int nk_ss_left_hint(const double *a, int n, double x, int hint) {
    int lo, hi;
    hint = hint < 0 ? 0 : (hint > n - 1 ? n - 1 : hint);
    if (a[hint] < x) {                       // answer in (hint, n]
        int stepw = 1;
        lo = hint + 1; hi = lo;
        while (hi < n && a[hi] < x) { lo = hi + 1; hi += stepw; stepw <<= 1; }
        if (hi > n) hi = n;
    } else {                                 // answer in [0, hint]
        int stepw = 1;
        hi = hint; lo = hint;
        while (lo > 0 && !(a[lo - 1] < x)) { hi = lo - 1; lo -= stepw; stepw <<= 1; if (lo < 0) lo = 0; }
        // invariant: a[hi] >= x (or hi == hint), everything below lo is < x or lo == 0
        if (lo > 0 && !(a[lo - 1] < x)) lo = 0;
    }
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ double nk_interp_lin_hint(const double *xs, const double *ys, int n, double x, int hint, int &idx_out) {
    int idx = nk_ss_left_hint(xs, n, x, hint);
    idx_out = idx;
    idx = idx < 1 ? 1 : (idx > n - 1 ? n - 1 : idx);
    double xlo = xs[idx - 1], xhi = xs[idx], ylo = ys[idx - 1], yhi = ys[idx];
    return (yhi - ylo) / (xhi - xlo) * (x - xlo) + ylo;
}
// ---------------------------------------------------------------------------- lean FP64 arithmetic
// The hot loop is bound by FP64 issue as much as by HBM (three exponentials and seven divides per phonon-step in the
// reference's formulas), so the per-particle functions avoid the library exp and the IEEE division sequence:
//   * 1/x by v_rcp_f64 and two Newton steps (<= 1 ulp; never used where the reference's rounding decides a hit);
//   * exp(x) by Cody-Waite reduction and a degree-13 Taylor polynomial (|r| <= ln2/2: truncation 4e-18);
//   * Bose-Einstein occupations from the per-mode E0 = exp(a / T0) of the mode record: exp(a / T) = E0 exp(a (1/T - 1/T0)),
//     the second factor a degree-10 polynomial while |a (1/T - 1/T0)| < 1/8 (temperatures within ~5 % of T0 for every
//     mode of Si/Ge), the general exponential otherwise.
__device__ __forceinline__ double nk_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
// One Horner step p * r + c with a coefficient that is not an inline constant, as ONE instruction.  Left to itself the
// compiler builds these steps as v_fmac_f64 (accumulate into the addend's register), and since the coefficient must survive
// -- it sits in a register across the tile loop -- it copies it first: v_mov_b64 + v_fmac_f64, two vector instructions per
// step, ~30 per particle (profiles/r03_notes.txt (23)).  v_fma_f64 takes the coefficient as a third source.  Same operation,
// same rounding.
__device__ __forceinline__ double nk_fma_c(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "v"(c));
    return o;
}
__device__ __forceinline__ double nk_exp(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = nk_fma_c(p, r, 2.08767569878681e-09);
    p = nk_fma_c(p, r, 2.505210838544172e-08);
    p = nk_fma_c(p, r, 2.755731922398589e-07);
    p = nk_fma_c(p, r, 2.7557319223985893e-06);
    p = nk_fma_c(p, r, 2.48015873015873e-05);
    p = nk_fma_c(p, r, 1.984126984126984e-04);
    p = nk_fma_c(p, r, 0.001388888888888889);
    p = nk_fma_c(p, r, 0.008333333333333333);
    p = nk_fma_c(p, r, 0.041666666666666664);
    p = nk_fma_c(p, r, 0.16666666666666666);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double kc = k < -2000.0 ? -2000.0 : (k > 2000.0 ? 2000.0 : k);
    return __builtin_ldexp(p, (int)kc);
}
__device__ __forceinline__ double nk_exp_small(double dl) {      // exp(dl), |dl| < 1/8: dl^11 / 11! < 3e-18
    double p = 2.755731922398589e-07;
    p = nk_fma_c(p, dl, 2.7557319223985893e-06);
    p = nk_fma_c(p, dl, 2.48015873015873e-05);
    p = nk_fma_c(p, dl, 1.984126984126984e-04);
    p = nk_fma_c(p, dl, 0.001388888888888889);
    p = nk_fma_c(p, dl, 0.008333333333333333);
    p = nk_fma_c(p, dl, 0.041666666666666664);
    p = nk_fma_c(p, dl, 0.16666666666666666);
    p = __builtin_fma(p, dl, 0.5);
    p = __builtin_fma(p, dl, 1.0);
    p = __builtin_fma(p, dl, 1.0);
    return p;
}
// Bose-Einstein occupation, Phonon.py:338-345: n0 = 1 / (exp(hbar omega / (kB T)) - 1), 0 where T <= 0 or omega <= 0.
// a = hbar omega / kB, E0 = exp(a / T0) from the mode record, invT = 1 / T.
__device__ __forceinline__ double nk_be(double a, double E0, double invT, double invT0) {
    if (!(a > 0.0) || !(invT > 0.0)) return 0.0;
    const double dl = a * (invT - invT0);
    double em1;
    if (fabs(dl) < 0.125) em1 = __builtin_fma(E0, nk_exp_small(dl), -1.0);
    else {
        const double xx = a * invT;
        if (!(xx < 700.0)) return 0.0;               // T -> 0 (also invT = inf)
        em1 = nk_exp(xx) - 1.0;
    }
    return nk_rcp(em1);
}
// the same from a temperature (taps, set-up paths)
__device__ __forceinline__ double nk_occupation(const NkDev &d, double T, double omega, double E0) {
    if (!(T > 0.0) || !(omega > 0.0)) return 0.0;
    return nk_be(omega * d.c_hk, E0, 1.0 / T, d.invT0);
}
// scipy interp1d(kind='linear') evaluation rule on a sorted table
__device__ __forceinline__ double nk_interp_lin(const double *xs, const double *ys, int n, double x) {
    int idx = nk_ss_left(xs, n, x);
    idx = idx < 1 ? 1 : (idx > n - 1 ? n - 1 : idx);
    double xlo = xs[idx - 1], xhi = xs[idx], ylo = ys[idx - 1], yhi = ys[idx];
    return (yhi - ylo) / (xhi - xlo) * (x - xlo) + ylo;
}
// temperature_function (Phonon.py:387) and crystal_energy_function (Phonon.py:390)
__device__ __forceinline__ double nk_T_of_E(const NkDev &d, double E) {
    if (E < d.Earr[0]) return d.Tfill_lo;
    if (E > d.Earr[d.nE - 1]) return d.Tfill_hi;
    return nk_interp_lin(d.Earr, d.Tarr, d.nE, E);
}
__device__ __forceinline__ double nk_E_of_T(const NkDev &d, double T) {
    if (T < d.Tarr[0]) return d.Earr[0];
    if (T > d.Tarr[d.nE - 1]) return d.Earr[d.nE - 1];
    return nk_interp_lin(d.Tarr, d.Earr, d.nE, T);
}
// lifetime_function = RegularGridInterpolator((T,q,j), tau) at integer (q,j): linear in tau along T (Phonon.py:336).
// Out-of-table T gives NaN (the reference raises ValueError there).  ta, tb, tc are the three rows packed into the
// particle's mode record (two grid intervals around the live temperature range); anything else reads the full table.
// The packed window's grid values and reciprocal widths.  park = true (the sweep) keeps them in VECTOR registers: as kernel
// arguments they are scalars, the sweep has more scalars than scalar registers, and every use of a parked one costs a
// v_readlane per dword and basic block (see NkBoxWalls).
struct NkTauWin {
    double g0, g1, g2, i0, i1;
    __device__ __forceinline__ void load(const NkDev &d, bool park) {
        g0 = d.tau_g[0]; g1 = d.tau_g[1]; g2 = d.tau_g[2]; i0 = d.tau_ig[0]; i1 = d.tau_ig[1];
        if (park) asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(i0), "+v"(i1));
    }
};
template <class SegModes>
__device__ __forceinline__ double nk_lifetime(const NkDev &d, const NkTauWin &w, double ta, double tb, double tc, double T, const SegModes &sm, int idx) {
    // Fast path: T inside the packed window (g0, g2].  searchsorted-left - 1 puts T in (g_k, g_k+1] at interval k; grid
    // values and reciprocal widths come from registers, lifetimes from the mode record.
    if (T > w.g0 && T <= w.g2) {
        const bool b = T > w.g1;
        const double glo = b ? w.g1 : w.g0;
        const double ig = b ? w.i1 : w.i0;
        const double t0 = b ? tb : ta, t1 = b ? tc : tb;
        const double y = (T - glo) * ig;
        return t0 * (1.0 - y) + t1 * y;
    }
    const int NT = d.NT;
    const double *g = d.Tgrid;
    if (!(T >= g[0]) || !(T <= g[NT - 1])) return __builtin_nan("");
    int i = (int)((T - g[0]) / (g[1] - g[0]));       // uniform-grid guess, then the exact searchsorted fix-up
    i = i < 0 ? 0 : (i > NT - 1 ? NT - 1 : i);
    while (i < NT && g[i] < T) ++i;                  // i = number of grid points < T  (searchsorted left)
    while (i > 0 && g[i - 1] >= T) --i;
    i -= 1;
    i = i < 0 ? 0 : (i > NT - 2 ? NT - 2 : i);
    const double y = (T - g[i]) / (g[i + 1] - g[i]);
    const int mode = sm.mode(idx);                   // (the global mode index is only read here, outside the packed window)
    const double t0 = d.tau[(int64_t)i * d.M + mode];
    const double t1 = d.tau[(int64_t)(i + 1) * d.M + mode];
    return t0 * (1.0 - y) + t1 * y;
}

// --------------------------------------------------------------------------------- subvolume lookups
// Per-subvolume record staged in LDS by the kernels (one 32-byte read pair per lookup): centre along the slice axis,
// temperature, slope of the piecewise-linear profile towards the next centre, 1 / T.
struct __attribute__((aligned(16))) NkSv { double c, T, slope, invT; };
struct NkSvTab {
    const double *cen;     // [3 S] centres
    const double *Tsv;     // [S + rbf_P] temperatures, then the RBF coefficients
    const NkSv *sv;        // [S]
};
// SubvolClassifier.predict, Geometry.py:1198-1213: nearest centre.
__device__ __forceinline__ int nk_classify(const NkDev &d, const NkSvTab &tb, double x, double y, double z) {
    const int S = d.S;
    const double *cen = tb.cen;
    if (d.sv_kind == 0) {
        const int a = d.sv_axis;
        const double xa = a == 0 ? x : (a == 1 ? y : z);
        const double t = (xa - d.sv_lo) * d.sv_invL, fl = floor(t);
        int s = (int)fl;
        s = s < 0 ? 0 : (s > S - 1 ? S - 1 : s);
        // well inside a slice the nearest centre is the slice's own (the margin is far above any rounding of t); only a
        // point within 1e-9 of a slice edge takes the exact comparison below, where ties go to the lower index
        const double f = t - fl;
        if (f > 1e-9 && f < 1.0 - 1e-9) return s;
        const double cm = cen[3 * (s > 0 ? s - 1 : 0) + a], c0 = cen[3 * s + a], cp = cen[3 * (s + 1 < S ? s + 1 : S - 1) + a];
        const double d0 = fabs(xa - c0);
        if (s + 1 < S && fabs(xa - cp) < d0) {
            ++s;
            while (s + 1 < S && fabs(xa - cen[3 * (s + 1) + a]) < fabs(xa - cen[3 * s + a])) ++s;
        } else if (s > 0 && fabs(xa - cm) <= d0) {
            --s;
            while (s > 0 && fabs(xa - cen[3 * (s - 1) + a]) <= fabs(xa - cen[3 * s + a])) --s;
        }
        return s;
    }
    int best = 0;
    double dbest = __builtin_inf();
    for (int s = 0; s < S; ++s) {
        double dx = x - cen[3 * s], dy = y - cen[3 * s + 1], dz = z - cen[3 * s + 2];
        double dd = dx * dx + dy * dy + dz * dz;
        if (dd < dbest) { dbest = dd; best = s; }
    }
    return best;
}
// per-particle temperature, Population.py:570-571, :694-702; also returns 1 / T (from the table where T is a subvolume's)
// RBF = false compiles the radial-basis branch out (the sweep is instantiated with and without it: inlined it costs
// registers in every configuration, and it is a rarely used mode).
template <bool RBF = true>
__device__ __forceinline__ double nk_interp_T(const NkDev &d, const NkSvTab &tb, double x, double y, double z, double &invT) {
    const int S = d.S;
    const double *cen = tb.cen;
    if (RBF && d.sv_interp == 3) {
        // RBFInterpolator(kernel='cubic'): sum_i w_i |x - c_i|^3 + p_0 + sum_k p_k (x_k - shift_k) / scale_k; the
        // coefficients follow the temperatures in the same array (k_update refreshes them with every new T_sv)
        const double *w = tb.Tsv + S;
        double out = 0.0;
        for (int i = 0; i < S; ++i) {
            const double dx = d.rbf_used[0] ? x - cen[3 * i] : 0.0, dy = d.rbf_used[1] ? y - cen[3 * i + 1] : 0.0,
                         dz = d.rbf_used[2] ? z - cen[3 * i + 2] : 0.0;
            const double r2 = dx * dx + dy * dy + dz * dz;
            out += w[i] * (r2 * sqrt(r2));
        }
        out += w[S];
        int q = S + 1;
        if (d.rbf_used[0]) out += w[q++] * ((x - d.rbf_shift[0]) / d.rbf_scale[0]);
        if (d.rbf_used[1]) out += w[q++] * ((y - d.rbf_shift[1]) / d.rbf_scale[1]);
        if (d.rbf_used[2]) out += w[q++] * ((z - d.rbf_shift[2]) / d.rbf_scale[2]);
        invT = nk_rcp(out);
        return out;
    }
    if (d.sv_interp == 2 || S == 1) { const NkSv q = tb.sv[S == 1 ? 0 : nk_classify(d, tb, x, y, z)]; invT = q.invT; return q.T; }
    const int a = d.sv_axis;
    const double xa = a == 0 ? x : (a == 1 ? y : z);
    const double t = (xa - d.sv_lo) * d.sv_invL, fl = floor(t);
    int g = (int)fl;
    g = g < 0 ? 0 : (g > S - 1 ? S - 1 : g);
    if (d.sv_interp == 1) {
        // interp1d 'linear' with extrapolation: idx = searchsorted(centres, xa, 'left') clipped to [1, S-1], bracket
        // (idx-1, idx).  The slice guess g is off by at most one (rounding of t at a slice edge), so xa lies between the
        // centres g-1 and g+1 and one comparison with centre g decides: idx = g + (c_g < xa).
        int idx = g + (tb.sv[g].c < xa ? 1 : 0);
        idx = idx < 1 ? 1 : (idx > S - 1 ? S - 1 : idx);
        const NkSv q = tb.sv[idx - 1];                // slope = (T[idx] - T[idx-1]) / (c[idx] - c[idx-1])
        const double T = q.slope * (xa - q.c) + q.T;
        invT = nk_rcp(T);
        return T;
    }
    // interp1d 'nearest': bounds b[i] = c[i+1]/2 + c[i]/2, idx = #(b < xa); away from a slice edge that is the slice itself
    int idx = g;
    const double f = t - fl;
    if (!(f > 1e-9 && f < 1.0 - 1e-9)) {
        while (idx < S - 1 && (cen[3 * (idx + 1) + a] / 2.0 + cen[3 * idx + a] / 2.0) < xa) ++idx;
        while (idx > 0 && !((cen[3 * idx + a] / 2.0 + cen[3 * (idx - 1) + a] / 2.0) < xa)) --idx;
    }
    const NkSv q = tb.sv[idx];
    invT = q.invT;
    return q.T;
}

// -------------------------------------------------------------------------------------- ray casting
// Mesh.find_boundary, Mesh.py:806-856: nearest valid triangle hit; the lowest face index wins ties; miss -> inf, -1.
// Coplanar triangles share one plane record, so t = -(x.n + k)/(v.n) (Mesh.py:818) is evaluated once per distinct
// plane and only when x.n + k and v.n have opposite signs (t >= tol > 0 is impossible otherwise, Mesh.py:820); the
// per-triangle AABB (:828-829) and barycentric tests (:837-843) then run on that plane's faces.  Face record rows
// iu/iw are the first two rows of the inverse of face_basis_matrix: (u, w) = rows . (c - o) is the solve() of :840.
struct NkHit { double t; int face, facet; };
// The planes [pl0, pl1) against one ray; `h` carries the best hit so far.
__device__ __forceinline__ void nk_fb_planes(const double *planes, const double *faces, int pl0, int pl1, double tol, double x,
                                             double y, double z, double vx, double vy, double vz, NkHit &h) {
    // numpy evaluates these dot products without fused multiply-adds; num cancels near a plane, so a fused evaluation
    // moves t by many ulps.  Keep the reference's rounding here.
#pragma clang fp contract(off)
    // The tables are the same for all lanes, so every read is a latency, not a bandwidth, cost: records are fetched with
    // 16-byte reads issued together, and the next plane is requested while the current one is evaluated.
    const double2 *P = reinterpret_cast<const double2 *>(planes);
    double2 n0 = P[3 * pl0], n1 = P[3 * pl0 + 1], n2 = P[3 * pl0 + 2];
    for (int pl = pl0; pl < pl1; ++pl) {
        const double2 p01 = n0, p23 = n1, pr = n2;
        if (pl + 1 < pl1) { n0 = P[3 * pl + 3]; n1 = P[3 * pl + 4]; n2 = P[3 * pl + 5]; }
        const double num = x * p01.x + y * p01.y + z * p23.x + p23.y;
        const double den = vx * p01.x + vy * p01.y + vz * p23.x;
        if (!((num < 0.0 && den > 0.0) || (num > 0.0 && den < 0.0))) continue;
        const double t = -num / den;
        if (!(t >= tol) || isinf(t) || t > h.t) continue;
        const double cx = x + t * vx, cy = y + t * vy, cz = z + t * vz;
        const int f_lo = __double2loint(pr.x), f_hi = __double2hiint(pr.x);
        for (int f = f_lo; f < f_hi; ++f) {
            const double2 *q = reinterpret_cast<const double2 *>(faces + f * NK_FACE_DOUBLES);
            const double2 q0 = q[0], q1 = q[1], q2 = q[2];              // lo.x lo.y | lo.z hi.x | hi.y hi.z
            const bool inside = (cx >= q0.x - tol) & (cy >= q0.y - tol) & (cz >= q1.x - tol) & (cx <= q1.y + tol) &
                                (cy <= q2.x + tol) & (cz <= q2.y + tol);
            if (!inside) continue;
            const double2 q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];   // o(3) iu(3) iw(3) {face, facet}
            const double bx = cx - q3.x, by = cy - q3.y, bz = cz - q4.x;
            const double u = q4.y * bx + q5.x * by + q5.y * bz;
            const double w = q6.x * bx + q6.y * by + q7.x * bz;
            const double r = 1.0 - (u + w);
            if (!(u >= -tol && u <= 1.0 + tol && w >= -tol && w <= 1.0 + tol && r >= -tol && r <= 1.0 + tol)) continue;
            const int idf = __double2loint(q7.y), idc = __double2hiint(q7.y);
            if (t < h.t || idf < h.face) { h.t = t; h.face = idf; h.facet = idc; }
        }
    }
}
// One plane's faces against the hit point of a ray at distance t on that plane (AABB :828-829, barycentric :837-843).
__device__ __forceinline__ void nk_fb_faces(const double *faces, int f_lo, int f_hi, double tol, double t, double cx, double cy,
                                            double cz, NkHit &h) {
    for (int f = f_lo; f < f_hi; ++f) {
        const double2 *q = reinterpret_cast<const double2 *>(faces + f * NK_FACE_DOUBLES);
        const double2 q0 = q[0], q1 = q[1], q2 = q[2];              // lo.x lo.y | lo.z hi.x | hi.y hi.z
        const bool inside = (cx >= q0.x - tol) & (cy >= q0.y - tol) & (cz >= q1.x - tol) & (cx <= q1.y + tol) &
                            (cy <= q2.x + tol) & (cz <= q2.y + tol);
        if (!inside) continue;
        const double2 q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];   // o(3) iu(3) iw(3) {face, facet}
        const double bx = cx - q3.x, by = cy - q3.y, bz = cz - q4.x;
        const double u = q4.y * bx + q5.x * by + q5.y * bz;
        const double w = q6.x * bx + q6.y * by + q7.x * bz;
        const double r = 1.0 - (u + w);
        if (!(u >= -tol && u <= 1.0 + tol && w >= -tol && w <= 1.0 + tol && r >= -tol && r <= 1.0 + tol)) continue;
        const int idf = __double2loint(q7.y), idc = __double2hiint(q7.y);
        if (t < h.t || idf < h.face) { h.t = t; h.face = idf; h.facet = idc; }
    }
}
// The reference takes the smallest t among the (plane, face) pairs that pass the face tests.  First every plane's t (cheap),
// remembering the smallest admissible one; if that plane has a face that passes, it IS the answer -- no other pair can have
// a smaller t -- and only its faces were tested (from inside a convex body that is always the case: a third of the work of
// testing faces plane by plane).  If it has none, or two planes tie for the smallest t, the plain sweep runs.  Same result
// in every case, same arithmetic (no fused multiply-adds: numpy does not fuse, and num cancels near a plane).
__device__ __forceinline__ void nk_find_boundary(const double *planes, const double *faces, int NP, double tol, double x,
                                                 double y, double z, double vx, double vy, double vz, double &tc, int &fc) {
    NkHit h = {__builtin_inf(), 0x7fffffff, -1};
    {
#pragma clang fp contract(off)
        const double2 *P = reinterpret_cast<const double2 *>(planes);
        double tmin = __builtin_inf();
        int pmin = -1, ties = 0;
        double2 n0 = P[0], n1 = P[1];
        for (int pl = 0; pl < NP; ++pl) {
            const double2 p01 = n0, p23 = n1;
            if (pl + 1 < NP) { n0 = P[3 * pl + 3]; n1 = P[3 * pl + 4]; }
            const double num = x * p01.x + y * p01.y + z * p23.x + p23.y;
            const double den = vx * p01.x + vy * p01.y + vz * p23.x;
            if (!((num < 0.0 && den > 0.0) || (num > 0.0 && den < 0.0))) continue;
            const double t = -num / den;
            if (!(t >= tol) || isinf(t)) continue;
            if (t < tmin) { tmin = t; pmin = pl; ties = 0; }
            else if (t == tmin) ties = 1;
        }
        if (pmin < 0) { tc = h.t; fc = h.facet; return; }            // no admissible plane at all: a miss
        if (!ties) {
            const double cx = x + tmin * vx, cy = y + tmin * vy, cz = z + tmin * vz;
            const double pr = planes[NK_PLANE_DOUBLES * pmin + 4];
            nk_fb_faces(faces, __double2loint(pr), __double2hiint(pr), tol, tmin, cx, cy, cz, h);
            if (h.facet >= 0 || h.face != 0x7fffffff) { tc = h.t; fc = h.facet; return; }
        }
    }
    nk_fb_planes(planes, faces, 0, NP, tol, x, y, z, vx, vy, vz, h);
    tc = h.t;
    fc = h.facet;
}
// ---- box store (NkDev::box): the next hit is not cached, it is read off the position.
// Does the particle at (x, y, z) -- its end-of-step position -- lie beyond a wall it is flying towards?  Then it crossed that
// wall inside this step (it was inside the solid when the step began: every particle is, after its last event, emission or
// resampling; a particle behind its reservoir face with a negative entry time flies INWARDS and is not caught here, like in
// the reference, whose cached next hit is the far wall).  x_a > -k exactly when the reference's numerator x.n + k rounds above
// zero, so this is the sign of the very t the event pass then computes.
// The six walls, held in VECTOR registers for the whole kernel (NkBoxWalls::load launders them through an empty asm): as kernel
// arguments they are scalar values, of which the sweep has more than scalar registers -- the compiler parks the surplus in
// vector lanes and fetches every use back with v_readlane, per basic block (+134 of those in the tile loop for the twelve
// dwords of the walls; a vector instruction each, as dear as an FMA).  Twelve vector registers are free (144 of 168 in use).
struct NkBoxWalls {
    double lx, hx, ly, hy, lz, hz;      // lo = k of the wall with normal -e_a, hi = -k of the wall with normal +e_a
    uint32_t ids_lo, ids_hi;            // NkDev::box_ids
    __device__ __forceinline__ void load(const NkDev &d) {
        lx = d.box_k[0]; hx = -d.box_k[1]; ly = d.box_k[2]; hy = -d.box_k[3]; lz = d.box_k[4]; hz = -d.box_k[5];
        ids_lo = (uint32_t)d.box_ids; ids_hi = (uint32_t)(d.box_ids >> 32);
        asm volatile("" : "+v"(lx), "+v"(hx), "+v"(ly), "+v"(hy), "+v"(lz), "+v"(hz), "+v"(ids_lo), "+v"(ids_hi));
    }
};
__device__ __forceinline__ bool nk_box_out(const NkBoxWalls &b, double x, double y, double z, double vx, double vy, double vz) {
    const bool ox = ((x > b.hx) & (vx > 0.0)) | ((x < b.lx) & (vx < 0.0));
    const bool oy = ((y > b.hy) & (vy > 0.0)) | ((y < b.ly) & (vy < 0.0));
    const bool oz = ((z > b.hz) & (vz > 0.0)) | ((z < b.lz) & (vz < 0.0));
    return ox | oy | oz;
}
// The wall such a particle crossed FIRST and when, in timesteps counted from the end of the step (negative): for every wall it
// lies beyond, t = -(x.n + k) / (v.n) as Mesh.find_boundary evaluates it (Mesh.py:816-818; unit axis normal: x.n = +-x_a and
// v.n = +-v_a exactly, so num = x.n + k is one rounding -- x_a - hi resp. lo - x_a -- and den = |v_a| none), the earliest wins,
// the lowest face index among equals (:846-848).  What the reference holds in n_timesteps / collision_facets at this point is
// the same hit, cast from where the particle's free flight began and decremented once per step (Population.py:795): equal up
// to the rounding of the drift.
// num and den are positive for a wall the particle lies beyond (num <= 0 for the others), so "earliest" = largest num / den
// is decided on the cross products -- no division per wall -- and only the winner is divided (v_rcp_f64 + Newton: the quotient
// places the hit point, it decides nothing).  The tests' CPU checker selects the same way.
// one axis: the wall with normal -e_a (w = 2 a) or +e_a (w = 2 a + 1) against the best so far (nb / db, wall wb)
__device__ __forceinline__ void nk_box_axis(double xa, double va, double lo, double hi, int wlo, uint64_t ids,
                                            double &nb, double &db, int &wb) {
#pragma clang fp contract(off)
    const bool fwd = va > 0.0;
    const double num = fwd ? xa - hi : lo - xa;          // x.n + k of the wall the particle flies towards
    const double den = fabs(va);                          // v.n of that wall
    const int w = wlo + (fwd ? 1 : 0);
    const double l = num * db, r = nb * den;
    // (a tie between two walls: the lower face index wins; wb < 0: nothing chosen yet)
    const unsigned f0w = (unsigned)(ids >> (8 * w + 4)) & 15u, f0b = wb < 0 ? 16u : ((unsigned)(ids >> (8 * wb + 4)) & 15u);
    if ((num > 0.0) & (den > 0.0) & ((l > r) | ((l == r) & (f0w < f0b)))) { nb = num; db = den; wb = w; }
}
__device__ __forceinline__ void nk_box_first_hit(const NkBoxWalls &b, double inv_dt, double x, double y, double z, double vx, double vy,
                                                 double vz, double &nts, int &facet) {
    const uint64_t ids = ((uint64_t)b.ids_hi << 32) | b.ids_lo;
    double nb = -1.0, db = 1.0;
    int wb = -1;
    nk_box_axis(x, vx, b.lx, b.hx, 0, ids, nb, db, wb);
    nk_box_axis(y, vy, b.ly, b.hy, 2, ids, nb, db, wb);
    nk_box_axis(z, vz, b.lz, b.hz, 4, ids, nb, db, wb);
    nts = -(nb * nk_rcp(db)) * inv_dt;
    facet = wb < 0 ? -1 : (int)((ids >> (8 * wb)) & 15u);
}
// The NEXT wall of a particle inside the box (after a boundary event: it stands on a wall, or in the body after a periodic
// crossing, and flies inwards): what Mesh.find_boundary returns for it -- per wall it flies towards t = -(x.n + k) / (v.n) with
// the same x.n + k and v.n as above (one rounding / none), admissible when x.n + k < 0 < v.n and t >= tol (Mesh.py:820), the
// smallest wins, the lowest face index among equals; its hit point lies on the wall's two triangles (the nearest wall of a
// convex body always passes the face tests, Mesh.py:828-843).  The smallest is chosen on cross products, the winner divided
// exactly: tc is the very quotient the general search (nk_find_boundary over the LDS planes: six planes, up to three
// divisions, the winner's two triangles through box and barycentric tests -- a chain of ~30 dependent LDS reads) returns,
// unless two walls' quotients differ by less than a rounding (a ray through an edge).
__device__ __forceinline__ void nk_box_next_axis(double xa, double va, double lo, double hi, int wlo, uint64_t ids, double tol,
                                                 double &mb, double &db, int &wb) {
#pragma clang fp contract(off)
    const bool fwd = va > 0.0;
    const double num = fwd ? xa - hi : lo - xa;          // x.n + k of the wall the particle flies towards: negative inside
    const double den = fabs(va), m = -num;
    const int w = wlo + (fwd ? 1 : 0);
    const double l = m * db, r = mb * den;               // t < t_best  <=>  m / den < mb / db
    const unsigned f0w = (unsigned)(ids >> (8 * w + 4)) & 15u, f0b = wb < 0 ? 16u : ((unsigned)(ids >> (8 * wb + 4)) & 15u);
    if ((num < 0.0) & (den > 0.0) & (m >= tol * den) & ((wb < 0) | (l < r) | ((l == r) & (f0w < f0b)))) { mb = m; db = den; wb = w; }
}
__device__ __forceinline__ void nk_box_next_hit(const NkBoxWalls &b, double tol, double x, double y, double z, double vx, double vy,
                                                double vz, double &tc, int &facet) {
    const uint64_t ids = ((uint64_t)b.ids_hi << 32) | b.ids_lo;
    double mb = 0.0, db = 1.0;
    int wb = -1;
    nk_box_next_axis(x, vx, b.lx, b.hx, 0, ids, tol, mb, db, wb);
    nk_box_next_axis(y, vy, b.ly, b.hy, 2, ids, tol, mb, db, wb);
    nk_box_next_axis(z, vz, b.lz, b.hz, 4, ids, tol, mb, db, wb);
    const double t = mb / db;
    const bool hit = wb >= 0 && !isinf(t);
    tc = hit ? t : __builtin_inf();
    facet = hit ? (int)((ids >> (8 * wb)) & 15u) : -1;
}
// Large meshes (tables in global memory): a 4-ary tree of bounding boxes over the FACES, walked by every lane on its own.
// The faces are sorted along a space-filling curve; a leaf is 4 consecutive faces, node i of level l + 1 the union of
// nodes 4i .. 4i + 3 of level l, so the tree is implicit (no child pointers) and the walk needs no stack: the only state
// is the level, the index of the current family of four siblings, and four "still to visit" bits per level in one
// register.  Boxes are slightly inflated; a ray only enters the boxes it crosses before the best hit so far, and a hit
// point lies inside its face's box, so nothing is lost and the result equals the plain sweep (ties still go to the
// lowest face index, whatever the visiting order).  The previous structure (two levels of wave-uniform groups of 16
// planes) made a wave visit the union of its 64 rays' groups -- on a 5000-face wire nearly the whole mesh per batch.
#define NK_TREE_LEVELS 8            // 4^8 leaves x 4 faces: meshes up to 262 144 faces
#define NK_TREE_FACE_DOUBLES 20     // per face: n(3) k | lo(3) hi(3) | o(3) iu(3) iw(3) {face, facet}; a leaf's record: NK_TREE_LEAF_DOUBLES below
#define NK_TREE_FAMILY_FLOATS 24    // four sibling boxes (6 floats each); their facet tags sit in tree_tags (int4 per family)
// The boxes are single precision, rounded outwards on the host: a family of four is 96 bytes instead of 192 and the slab
// test runs at twice the FP64 rate.  The test stays conservative: a slab's entry and exit distances are widened by a bound
// of their rounding error, e = (|x| + B) 2^-20 |1 / v| per axis (B = largest box coordinate; the error proper is below
// (|x| + B) 2^-21 |1 / v|: conversion of x and 1 / v, one subtraction, one product).  A box that is entered although the
// exact ray misses it costs time, never a hit: the faces themselves are tested in FP64 as before.
// A ray for the slab test: 1 / v per axis and, per axis, the two constants that turn a slab distance into ONE fused multiply-add,
//   (w - x) / v -+ e  =  fma(w, 1 / v, -x / v -+ e)      (n.: towards the entry, f.: towards the exit)
// e = the widening above.  The roundings (conversion of x and of 1 / v, the product x / v, the constant's sum, the FMA's result;
// the product inside the FMA is exact) add up to less than 5 x 2^-24 (|x| + B) |1 / v| -- a third of e.
typedef float nk_f2 __attribute__((ext_vector_type(2)));
struct NkRayF { float ix, iy, iz, nx, ny, nz, fx, fy, fz; };
// A ray parallel to an axis (v_a == 0 exactly) gets a huge FINITE 1 / v_a instead of +-inf: the slab's entry / exit distances
// (lo - x) 1e30 and (hi - x) 1e30 are then -huge / +huge when x lies strictly inside the slab, both > tmax when it lies outside by
// more than the rounding margin c, and the widening e = c |1 / v| = c 1e30 makes every position in between pass -- the same
// answers as an explicit "parallel and outside?" test, with no special case.  |lo - x| < 1e5 keeps every product finite.
// Conservative like before: a box that is entered although the exact ray misses it costs time, never a hit.
#define NK_RAY_BIG 1.0e30f
__device__ __forceinline__ NkRayF nk_ray_f32(double x, double y, double z, double vx, double vy, double vz, double B) {
    NkRayF r;
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    r.ix = vx != 0.0 ? (float)(1.0 / vx) : NK_RAY_BIG; r.iy = vy != 0.0 ? (float)(1.0 / vy) : NK_RAY_BIG; r.iz = vz != 0.0 ? (float)(1.0 / vz) : NK_RAY_BIG;
    // (a component so small that 1 / v overflows a float is as good as parallel)
    r.ix = fminf(fmaxf(r.ix, -NK_RAY_BIG), NK_RAY_BIG); r.iy = fminf(fmaxf(r.iy, -NK_RAY_BIG), NK_RAY_BIG); r.iz = fminf(fmaxf(r.iz, -NK_RAY_BIG), NK_RAY_BIG);
    const float k = 9.5367431640625e-07f;                                               // 2^-20
    const float ex = (fabsf(fx) + (float)B) * k * fabsf(r.ix), ey = (fabsf(fy) + (float)B) * k * fabsf(r.iy), ez = (fabsf(fz) + (float)B) * k * fabsf(r.iz);
    const float px = fx * r.ix, py = fy * r.iy, pz = fz * r.iz;
    r.nx = -px - ex; r.ny = -py - ey; r.nz = -pz - ez;
    r.fx = ex - px; r.fy = ey - py; r.fz = ez - pz;
    return r;
}
// Does the ray cross the box lo = (lx, ly, lz), hi = (hx, hy, hz) before tmax ?  No branches, no special cases: per axis the wall
// the ray meets first is the low one when it flies upwards (a select on the sign of 1 / v, the same for every box of the walk),
// one FMA per wall, three-operand max / min over the axes.  A padding box (lo = +3e38, hi = -3e38) fails for every ray: its entry
// distance is +huge or +inf, its exit distance -huge or -inf (never NaN: the constants are finite).
__device__ __forceinline__ bool nk_ray_box(float lx, float ly, float lz, float hx, float hy, float hz, const NkRayF &r, float tmax) {
    const bool ux = r.ix >= 0.0f, uy = r.iy >= 0.0f, uz = r.iz >= 0.0f;
    // (entry, exit) of an axis as one packed FMA (v_pk_fma_f32: two single-precision FMAs per lane and instruction)
    const nk_f2 tx = __builtin_elementwise_fma(nk_f2{ux ? lx : hx, ux ? hx : lx}, nk_f2{r.ix, r.ix}, nk_f2{r.nx, r.fx});
    const nk_f2 ty = __builtin_elementwise_fma(nk_f2{uy ? ly : hy, uy ? hy : ly}, nk_f2{r.iy, r.iy}, nk_f2{r.ny, r.fy});
    const nk_f2 tz = __builtin_elementwise_fma(nk_f2{uz ? lz : hz, uz ? hz : lz}, nk_f2{r.iz, r.iz}, nk_f2{r.nz, r.fz});
    const float ax = tx.x, bx = tx.y, ay = ty.x, by = ty.y, az = tz.x, bz = tz.y;
    const float t0 = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.0f)), t1 = fminf(fminf(bx, by), fminf(bz, tmax));
    return t0 <= t1;
}
// The four faces of one leaf against one ray (same arithmetic and the same rounding as nk_fb_planes).
// A leaf's record (NK_TREE_LEAF_DOUBLES = 96 doubles) by cache line: first the BOXES of its four faces (single precision, rounded
// outwards like the tree's: 4 x 6 floats, one line), then per face 20 doubles: the plane (n, k) and the face's box, origin,
// barycentric rows and ids.  Two rounds of loads: the boxes; then -- while any lane of the wave has a face left whose box its ray
// pierces in front of its best hit -- every lane ITS next such face, plane and rest at once.  The leaf records of a large mesh
// do not fit the L2: every dependent round is a trip to the memory side.  (Round 3: face by face, plane -> box -> barycentric
// rows, up to nine rounds per leaf: 17 000 cycles per faces pass of k_events on the 5000-triangle wire.  Round 4 first: the four
// planes in one line, then the faces whose PLANE is met in front of the best hit -- most of a leaf's four, their planes being
// nearly the same: 13 600.)  The boxes reject what the planes cannot: a face the ray passes beside.  Conservative like the tree's
// boxes (a face whose hit point lies inside its box is never rejected), so the result is the plain search's: earliest hit, lowest
// face index among equals, whatever the order.
#define NK_TREE_LEAF_DOUBLES 96
__device__ __forceinline__ void nk_tree_leaf(const double *tree_faces, int leaf, double tol, double x, double y, double z,
                                             double vx, double vy, double vz, const NkRayF &rf, NkHit &h) {
#pragma clang fp contract(off)
    const double *L = tree_faces + (size_t)leaf * NK_TREE_LEAF_DOUBLES;
    const float4 *B = reinterpret_cast<const float4 *>(L);
    float4 b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = B[k];
    float tmax = (float)h.t;                             // the best hit so far as a float that is not below it (inf stays inf)
    tmax += tmax * 1.1920929e-07f;
    unsigned mask = ((unsigned)nk_ray_box(b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, rf, tmax)) |
                    ((unsigned)nk_ray_box(b[1].z, b[1].w, b[2].x, b[2].y, b[2].z, b[2].w, rf, tmax) << 1) |
                    ((unsigned)nk_ray_box(b[3].x, b[3].y, b[3].z, b[3].w, b[4].x, b[4].y, rf, tmax) << 2) |
                    ((unsigned)nk_ray_box(b[4].z, b[4].w, b[5].x, b[5].y, b[5].z, b[5].w, rf, tmax) << 3);
    while (__ballot(mask != 0u) != 0ull) {
        if (mask != 0u) {
            const int c = __builtin_ctz(mask);
            mask &= mask - 1u;
            const double2 *q = reinterpret_cast<const double2 *>(L + 16 + 20 * c);
            const double2 p01 = q[0], p23 = q[1];                       // n.x n.y | n.z k     (padding faces have n = k = 0)
            const double2 q0 = q[2], q1 = q[3], q2 = q[4];              // lo.x lo.y | lo.z hi.x | hi.y hi.z
            const double2 q3 = q[5], q4 = q[6], q5 = q[7], q6 = q[8], q7 = q[9];   // o(3) iu(3) iw(3) {face, facet}
            const double num = x * p01.x + y * p01.y + z * p23.x + p23.y;
            const double den = vx * p01.x + vy * p01.y + vz * p23.x;
            if ((num < 0.0 && den > 0.0) || (num > 0.0 && den < 0.0)) {
                const double t = -num / den;
                if ((t >= tol) && !isinf(t) && !(t > h.t)) {
                    const double cx = x + t * vx, cy = y + t * vy, cz = z + t * vz;
                    const bool inside = (cx >= q0.x - tol) & (cy >= q0.y - tol) & (cz >= q1.x - tol) & (cx <= q1.y + tol) &
                                        (cy <= q2.x + tol) & (cz <= q2.y + tol);
                    const double bx = cx - q3.x, by = cy - q3.y, bz = cz - q4.x;
                    const double u = q4.y * bx + q5.x * by + q5.y * bz;
                    const double w = q6.x * bx + q6.y * by + q7.x * bz;
                    const double r = 1.0 - (u + w);
                    const bool in_tri = u >= -tol && u <= 1.0 + tol && w >= -tol && w <= 1.0 + tol && r >= -tol && r <= 1.0 + tol;
                    const int idf = __double2loint(q7.y), idc = __double2hiint(q7.y);
                    if (inside && in_tri && (t < h.t || idf < h.face)) { h.t = t; h.face = idf; h.facet = idc; }
                }
            }
        }
    }
}
// A ray that starts ON a planar facet (an entering particle on its reservoir, a reflected one on the wall it just met)
// cannot hit that facet again: for each of its faces t = -(x.n + k) / (v.n) is either of the wrong sign or below tol.
// When that is certain -- the bound below holds for every face of the facet, whose planes differ from the facet's by at
// most (dn, dk) -- the walk skips every node that holds faces of that facet only (tags, set on the host).  Otherwise the
// function returns NK_TREE_NO_SKIP and nothing is skipped.  On a cap of 1250 fan triangles this is the difference
// between a walk like any other and one through hundreds of leaves for a ray that starts near the fan's centre.
#define NK_TREE_NO_SKIP (-2)
__device__ __forceinline__ int nk_tree_skip(const NkDev &d, int facet, double cx, double cy, double cz, double nx, double ny,
                                            double nz, double x, double y, double z, double vx, double vy, double vz) {
    if (facet < 0) return NK_TREE_NO_SKIP;
    const double dn = d.facet_skip[2 * facet], dk = d.facet_skip[2 * facet + 1];
    const double rx = x - cx, ry = y - cy, rz = z - cz;
    const double dist = fabs(rx * nx + ry * ny + rz * nz);
    const double r = fabs(rx) + fabs(ry) + fabs(rz), vv = fabs(vx) + fabs(vy) + fabs(vz);
    const double vn = fabs(vx * nx + vy * ny + vz * nz);
    // every face: |x.n_g + k_g| <= dist + dn r + dk and |v.n_g| >= vn - dn |v|  =>  |t_g| < tol / 2
    return ((dist + dn * r + dk) * 1.000001 + 1e-300 < 0.5 * d.tol * (vn - dn * vv)) ? facet : NK_TREE_NO_SKIP;
}
// The walk as a state machine, one visit per step: k_events keeps a wave's lanes at different points of different walks.
// A visit is either a family's four boxes (nk_walk_boxes: ends with the choice of the next node, which is a leaf when the
// family is on level 0 -- the walk then holds it in `leaf` and waits) or that leaf's four faces (nk_walk_leaf).  The two
// are separate so that a wave can run the boxes of the lanes that have boxes to do and the faces of the lanes that wait at
// a leaf as two passes of their own, each with many lanes, instead of both with few.
struct NkWalk {
    NkRayF rf;
    NkHit h;
    int l, fam;
    uint32_t todo;                      // bits 4l .. 4l + 3: siblings of the current family of level l still to visit
    int leaf;                           // >= 0: the leaf the walk stands at (its faces are due), -1: none
    bool enter;
#ifdef NK_TREE_STATS
    int n_enter, n_leaf;
#endif
};
__device__ __forceinline__ void nk_walk_begin(const NkDev &d, NkWalk &w, double x, double y, double z, double vx, double vy, double vz) {
    w.h.t = __builtin_inf(); w.h.face = 0x7fffffff; w.h.facet = -1;
    w.rf = nk_ray_f32(x, y, z, vx, vy, vz, d.tree_bound);
    w.l = d.tree_top; w.fam = 0; w.todo = 0u; w.enter = true; w.leaf = -1;
#ifdef NK_TREE_STATS
    w.n_enter = w.n_leaf = 0;
#endif
}
// the boxes of the family the walk stands at (if it has just entered it), then the next node; true when the walk is over
// (w.h holds the hit).  Not to be called while w.leaf >= 0.
// lds / lds_fam0: the families from lds_fam0 up in LDS (k_events), the others -- all of them for the other callers -- in global memory
__device__ __forceinline__ bool nk_walk_boxes(const NkDev &d, int skip, NkWalk &w, const float4 *lds = nullptr, int lds_fam0 = 0x7fffffff) {
    int l = w.l;
    if (w.enter) {                      // the four boxes of family `fam` of level l, requested together
#ifdef NK_TREE_STATS
        ++w.n_enter;
#endif
        int base = d.tree_base[0];
#pragma unroll
        for (int k = 1; k < NK_TREE_LEVELS; ++k) base = (l == k) ? d.tree_base[k] : base;
        const int fi = (base >> 2) + w.fam;
        float4 b[6];
        if (fi >= lds_fam0) {
            const float4 *B = lds + (size_t)(fi - lds_fam0) * (NK_TREE_FAMILY_FLOATS / 4);
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] = B[k];
        } else {
            const float4 *B = reinterpret_cast<const float4 *>(d.tree_boxes + (size_t)fi * NK_TREE_FAMILY_FLOATS);
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] = B[k];
        }
        int4 tg = make_int4(-1, -1, -1, -1);             // facet tags, read only by a ray that starts on a large facet
        if (skip != NK_TREE_NO_SKIP) tg = reinterpret_cast<const int4 *>(d.tree_tags)[(base >> 2) + w.fam];
        // the best hit so far as a float that is not below it
        float tmax = (float)w.h.t;                       // inf stays inf
        tmax += tmax * 1.1920929e-07f;
        // (nodes beyond a level's count exist only as padding boxes, which no ray enters: no index test)
        const uint32_t m = ((uint32_t)((tg.x != skip) & nk_ray_box(b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, w.rf, tmax))) |
                           ((uint32_t)((tg.y != skip) & nk_ray_box(b[1].z, b[1].w, b[2].x, b[2].y, b[2].z, b[2].w, w.rf, tmax)) << 1) |
                           ((uint32_t)((tg.z != skip) & nk_ray_box(b[3].x, b[3].y, b[3].z, b[3].w, b[4].x, b[4].y, w.rf, tmax)) << 2) |
                           ((uint32_t)((tg.w != skip) & nk_ray_box(b[4].z, b[4].w, b[5].x, b[5].y, b[5].z, b[5].w, w.rf, tmax)) << 3);
        w.todo = (w.todo & ~(0xFu << (4 * l))) | (m << (4 * l));
        w.enter = false;
    }
    uint32_t m = (w.todo >> (4 * l)) & 0xFu;
    while (m == 0) {                    // family done: back to the nearest ancestor's family that has a sibling left
        if (l == d.tree_top) return true;
        ++l;
        w.fam >>= 2;
        m = (w.todo >> (4 * l)) & 0xFu;
    }
    w.l = l;
    const int c = __builtin_ctz(m);
    w.todo &= ~(1u << (4 * l + c));
    const int node = 4 * w.fam + c;
    if (l > 0) { w.l = l - 1; w.fam = node; w.enter = true; }
    else w.leaf = node;
    return false;
}
// the faces of the leaf the walk stands at
__device__ __forceinline__ void nk_walk_leaf(const NkDev &d, NkWalk &w, double x, double y, double z, double vx, double vy, double vz) {
    nk_tree_leaf(d.tree_faces, w.leaf, d.tol, x, y, z, vx, vy, vz, w.rf, w.h);
    w.leaf = -1;
#ifdef NK_TREE_STATS
    ++w.n_leaf;
#endif
}
// one visit of either kind
__device__ __forceinline__ bool nk_walk_step(const NkDev &d, int skip, NkWalk &w, double x, double y, double z, double vx, double vy, double vz) {
    if (w.leaf >= 0) { nk_walk_leaf(d, w, x, y, z, vx, vy, vz); return false; }
    return nk_walk_boxes(d, skip, w);
}
__device__ __forceinline__ void nk_find_boundary_tree(const NkDev &d, int skip, double x, double y, double z, double vx,
                                                      double vy, double vz, double &tc, int &fc) {
    NkWalk w;
    nk_walk_begin(d, w, x, y, z, vx, vy, vz);
    while (!nk_walk_step(d, skip, w, x, y, z, vx, vy, vz)) { }
    tc = w.h.t;
    fc = w.h.facet;
#ifdef NK_TREE_STATS
    tc = (double)w.n_enter;     // developer probe (make stats): visits instead of the hit
    fc = w.n_leaf;
#endif
}

// ---------------------------------------------------------------------------------- rough reflection
// select_reflected_modes + pick_diffuse_modes, Population.py:941-1015.  E0 travels with omega (a specular reflection keeps
// both of the incoming mode).
template <bool RBF = true>
__device__ __forceinline__ void nk_reflect(const NkDev &d, const NkSvTab &tb, int rough_idx, int mode_in,
                                           double cx, double cy, double cz, double n_in, double omega_in, double E0_in,
                                           double r_spec, double r_deg, double r_diff, int &mode_out, double &n_out,
                                           double &omega_out, double &E0_out) {
    // Every table read that does not depend on another one is issued up front, for both outcomes: the event pass is a chain
    // of memory round trips, and a load inside the branch it feeds only starts once the branch is resolved.
    const int64_t idx = (int64_t)rough_idx * d.M + mode_in;
    const int nlut = d.roul_nlut;
    int kb = (int)(r_diff * (double)nlut);
    kb = kb < 0 ? 0 : (kb > nlut - 1 ? nlut - 1 : kb);
    const int32_t *lut = d.roul_lut + (int64_t)rough_idx * (nlut + 1);
    const double *roul = d.roulette + (int64_t)rough_idx * d.M;
    const uint8_t ts = d.true_spec[idx];
    const double sp = d.specularity[idx];
#ifndef NK_REFLECT_LAZY
    const int smap = d.spec_map[idx];
    const int lo = lut[kb], hi = lut[kb + 1];
    const double rlast = roul[d.M - 1];
#endif
    const bool spec = ts && (r_spec <= sp);
    if (spec) {
#ifdef NK_REFLECT_LAZY
        const int smap = d.spec_map[idx];
#endif
        int out = smap;
        if (d.degen_j2) {
            int j2 = d.degen_j2[out];
            if (j2 > -1 && r_deg >= 0.5) out = (out / d.J) * d.J + j2;
        }
        mode_out = out; n_out = n_in; omega_out = omega_in; E0_out = E0_in;
    } else {
#ifdef NK_REFLECT_LAZY
        const int lo = lut[kb], hi = lut[kb + 1];
        const double rlast = roul[d.M - 1];
#endif
        const double r = r_diff * rlast;
        // np.searchsorted(roulette, r) (Population.py:1005) through a bucket index: r_diff in [k, k+1) / roul_nlut brackets
        // the answer between two precomputed positions a few entries apart; up to four of them are read in one go (same
        // result as the bisection over the whole table, one memory round trip instead of ~18)
        int flat;
        if (hi - lo <= 4) {
            const double a0 = lo + 0 < hi ? roul[lo + 0] : __builtin_inf(), a1 = lo + 1 < hi ? roul[lo + 1] : __builtin_inf(),
                         a2 = lo + 2 < hi ? roul[lo + 2] : __builtin_inf(), a3 = lo + 3 < hi ? roul[lo + 3] : __builtin_inf();
            flat = lo + (a0 < r ? 1 : 0) + (a1 < r ? 1 : 0) + (a2 < r ? 1 : 0) + (a3 < r ? 1 : 0);     // sorted: count of entries < r
            if (flat > hi) flat = hi;
        } else flat = lo + nk_ss_left(roul + lo, hi - lo, r);
        if (flat > d.M - 1) flat = d.M - 1;
        mode_out = flat;
        omega_out = d.modetab[flat].omega;
        E0_out = d.modetab[flat].E0;
        double invT;
        nk_interp_T<RBF>(d, tb, cx, cy, cz, invT);
        n_out = nk_be(omega_out * d.c_hk, E0_out, invT, d.invT0);
    }
}

// --------------------------------------------------------------------------------- tally bins in LDS
// Layout (doubles unless noted): E[NREP][S], flux[NREP][3S], resb[4R] (energy, fx, fy, fz), then uint32 N[NREP][S],
// nleave[R], emitted[1].
struct NkBins {
    double *E, *flux, *resb;
    unsigned int *N, *nleave, *misc;
};

struct NkParticle {
    double x, y, z, occ, nts, omega, E0, vx, vy, vz;
    int mode, facet;          // mode: the GLOBAL mode index
    int slot;                 // rough facets, partitioned modes: the mode's slot (segment, local index); a reflection reads the
                              // new one from m2s together with the new mode's record (no memory round trip of its own)
};

// Boundary events inside a timestep: Population.boundary_scattering (Population.py:1546-1683) restated per particle.
// On entry (x,y,z) is the end-of-step position of the free drift and nts < 0 (first call: cts = 0, ev = 0).
// ONE event per call: the reference loops "while any particle still has time left"; here a particle that needs another
// event stays in the wave's carry registers and joins the next batch, so every pass of the wave runs up to 64 particles
// with exactly one event each (a while-loop per lane would idle most lanes: few particles cross more than one wall).
// Returns NK_EV_DONE (remainder drifted, particle final), NK_EV_DEAD (absorbed) or NK_EV_MORE (another event pending;
// cts / ev carry the consumed fraction of the step and the event count, which also numbers the RNG draws).
// ROUGH = false compiles the rough-facet branch out (meshes without 'R' facets): fewer registers in the sweep.
#define NK_EV_DONE 0
#define NK_EV_DEAD 1
#define NK_EV_MORE 2
// The event in two halves around the ray cast (k_events runs the casts of a wave's lanes as interleaved walks):
// nk_event_pre: absorption (returns NK_EV_DEAD) or the particle moved to the wall and sent off again (returns NK_EV_MORE:
// a ray cast from (p.x, p.y, p.z) along (p.vx, p.vy, p.vz) is due); nk_event_post: the cast's result (tc, fcn) -> NK_EV_MORE
// (the next wall is inside this step as well) or NK_EV_DONE (remainder drifted, particle final).
template <bool ROUGH, bool RBF = true>
__device__ __forceinline__ int nk_event_pre(const NkDev &d, const NkFacet *facets, const NkSvTab &tb, const double *resT, NkBins &b,
                                            NkParticle &p, double &cts, uint32_t ev, uint64_t pid, uint32_t step) {
    const double dt = d.dt;
    int fi = p.facet < 0 ? d.Fc - 1 : p.facet;          // a miss indexes the last facet
    const NkFacet fc = facets[fi];
    if (fc.bc == 'T' || fc.bc == 'F') {                 // I. absorbed by a reservoir, Population.py:1568-1608
        int r = p.facet < 0 ? -1 : fc.res;
        if (r >= 0) {
            const double n0 = d.T_ref_local ? nk_be(p.omega * d.c_hk, p.E0, resT[2 * r + 1], d.invT0)
                                            : nk_occupation(d, d.T_ref, p.omega, p.E0);
            double e = d.hbar * p.omega * (p.occ - n0);
            const double evn = e * nk_rcp(p.vx * fc.nx + p.vy * fc.ny + p.vz * fc.nz);      // e / (v . n), :1602
            atomicAdd(&b.nleave[r], 1u);
            atomicAdd(&b.resb[4 * r + 0], -e);
            atomicAdd(&b.resb[4 * r + 1], evn * p.vx);
            atomicAdd(&b.resb[4 * r + 2], evn * p.vy);
            atomicAdd(&b.resb[4 * r + 3], evn * p.vz);
        }
        return NK_EV_DEAD;
    }
    double tcol = p.nts * dt;
    double cx = p.x + p.vx * tcol, cy = p.y + p.vy * tcol, cz = p.z + p.vz * tcol;
    // consumed fraction of the step, Population.py:1482 / :1514: |x_col - x_prev| / |v dt| with x_prev = the
    // start-of-step position for a first event (:1472-1474), else the current position.  Both points lie on
    // the ray x + s v, so the quotient is |nts + 1| resp. |nts| exactly; evaluated in that closed form (it
    // differs from the reference's sqrt/sqrt/divide by rounding only, and saves two square roots and a divide).
    cts += (cts == 0.0) ? fabs(p.nts + 1.0) : fabs(p.nts);
    if (fc.bc == 'P') {                                                      // II. periodic, :1463-1489
        p.x = cx + fc.tx; p.y = cy + fc.ty; p.z = cz + fc.tz;   // t = centroid(partner) - centroid(this)
    } else if (ROUGH) {                                                      // III. rough, :1491-1544
        double r0, r1;
        nk_uniform2_dev(d.seed, pid, step, NK_TAG_REFLECT + ev, r0, r1);
        p.x = cx; p.y = cy; p.z = cz;
        int mo; double no, oo, eo;
        nk_reflect<RBF>(d, tb, fc.rough, p.mode, cx, cy, cz, p.occ, p.omega, p.E0, r0, r1, r1, mo, no, oo, eo);
        p.mode = mo; p.occ = no; p.omega = oo; p.E0 = eo;
        const NkMode *rec = d.modetab + mo;
        if (d.part) p.slot = d.m2s[mo];
        p.vx = rec->vx; p.vy = rec->vy; p.vz = rec->vz;
    }
    return NK_EV_MORE;
}
__device__ __forceinline__ int nk_event_post(const NkDev &d, NkParticle &p, double &cts, uint32_t &ev, double tc, int fcn) {
    const double dt = d.dt;
    p.nts = tc / dt;
    p.facet = fcn;
    if (++ev > 4096u) cts = 1.0;                         // the reference would spin
    if (cts < 1.0) {
        const double rem = 1.0 - cts;
        if (rem > p.nts) return NK_EV_MORE;                  // the next wall is inside this step as well
        p.x += p.vx * dt * rem; p.y += p.vy * dt * rem; p.z += p.vz * dt * rem;      // IV. drift the remainder, :1673-1681
        p.nts -= rem;
    }
    return NK_EV_DONE;
}
template <bool ROUGH, bool RBF = true>
__device__ __forceinline__ int nk_event_one(const NkDev &d, int NG, const double *planes, const double *faces,
                                            const NkFacet *facets, const NkSvTab &tb, const double *resT, NkBins &b,
                                            NkParticle &p, double &cts, uint32_t &ev, uint64_t pid, uint32_t step,
                                            const NkBoxWalls *bw = nullptr) {
    if (nk_event_pre<ROUGH, RBF>(d, facets, tb, resT, b, p, cts, ev, pid, step) == NK_EV_DEAD) return NK_EV_DEAD;
    double tc; int fcn;
    if (bw) nk_box_next_hit(*bw, d.tol, p.x, p.y, p.z, p.vx, p.vy, p.vz, tc, fcn);                      // box store: the six walls, in registers
    else if (NG > 0) nk_find_boundary_tree(d, NK_TREE_NO_SKIP, p.x, p.y, p.z, p.vx, p.vy, p.vz, tc, fcn);   // (walls are small facets)
    else nk_find_boundary(planes, faces, d.NP, d.tol, p.x, p.y, p.z, p.vx, p.vy, p.vz, tc, fcn);
    return nk_event_post(d, p, cts, ev, tc, fcn);
}

// Population.calculate_energy's per-particle part (Population.py:704-717) + the heat-flux sum (:734-736).
__device__ __forceinline__ void nk_tally_one(const NkDev &d, const NkSvTab &tb, NkBins &b, double x, double y, double z,
                                             double occ, double omega, double E0, double vx, double vy, double vz,
                                             bool do_flux, int rep) {
    const int s = nk_classify(d, tb, x, y, z);
    const double n0 = d.T_ref_local ? nk_be(omega * d.c_hk, E0, tb.sv[s].invT, d.invT0) : nk_occupation(d, d.T_ref, omega, E0);
    const double e = d.hbar * omega * (occ - n0);
    atomicAdd(&b.E[rep * d.S + s], e);
    atomicAdd(&b.N[rep * d.S + s], 1u);
    if (do_flux) {
        double *fl = b.flux + (rep * d.S + s) * 3;
        atomicAdd(fl + 0, vx * e);
        atomicAdd(fl + 1, vy * e);
        atomicAdd(fl + 2, vz * e);
    }
}
