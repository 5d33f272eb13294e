// nk_engine.hip -- host side (C ABI, include/nanokappa_hip.h) of libnanokappa_hip.so; kernels in nk_kernels.h.
// gfx950 / MI355X only.  HIP runtime + (lazily, multi-GPU only) RCCL; no PyTorch, no CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <queue>
#include <string>
#include <type_traits>
#include <chrono>
#include <time.h>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/nanokappa_hip.h"
#include "nk_kernels.h"
// instantiated in nk_sweep_plain.hip (compiled with the machine LICM on; see there; -DNK_PLAIN_IN_ENGINE: here, for comparisons)
#ifndef NK_PLAIN_IN_ENGINE
extern template __global__ void k_sweep<1, false, false, false, false, true, 1>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, true, 2>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, false, 1>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, false, 2>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, true, 1, true>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, true, 2, true>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, false, 1, true>(NkDev, uint32_t, int, int);
extern template __global__ void k_sweep<1, false, false, false, false, false, 2, true>(NkDev, uint32_t, int, int);
#endif

struct NkRccl {      // symbols resolved lazily with dlopen: a single-GPU run never loads librccl
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: ends a communicator whose peers never arrived
};

static thread_local std::string g_create_error;

struct nk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    NkDev d;                       // device view (pointers into `allocs` / `pallocs`)
    std::vector<void *> allocs;    // everything hipMalloc'ed except the particle arrays
    std::vector<void *> pallocs;   // particle arrays (re-allocated by nk_reserve)
    void *store_buf = nullptr;     // the one among them that holds the particle fields (nk_alloc_fields)
    size_t store_pad = 0;
    bool store_pid = false;
    bool store_nts = true;         // false: box store (no cached next hit, NkDev::box)
    bool mesh_box = false;         // nk_set_mesh: the mesh is an axis-aligned box whose six sides are its six facets
    bool box_forbidden = false;    // this context met particles a box store cannot express (nk_init_boundaries): cached layout from then on
    int32_t *anomalies = nullptr;  // device word of k_init_boundaries
    bool have_material = false, have_mesh = false, have_sv = false, have_params = false;
    int64_t step = 0;
    bool pending_relax = false;
    int g_sweep = 0;               // persistent grid of k_sweep = rows of `partials`
    std::vector<int32_t> h_seg_count;
    std::vector<hipEvent_t> evpool;
    int g_sweep_key = -1;              // which k_sweep instantiation g_sweep was sized for
    size_t g_sweep_lds = 0;            //   and with how much LDS per workgroup
    int layout_key = -1;               // how the particle store was laid out: 1 modes partitioned | 2 ids tracked
    std::vector<double> h_enter_prob;  // host copy of enter_prob (sizes the segments' head room)
    // set-up table builder state (nk_specular_*)
    int64_t spec_M = 0;
    double *spec_v = nullptr, *spec_om = nullptr, *spec_dl = nullptr, *spec_svx = nullptr;
    int32_t *spec_rank = nullptr;
    double spec_vmax = 0.0;
    NkSpecMode *spec_modes = nullptr;
    int32_t *spec_in = nullptr, *spec_out = nullptr;
    unsigned long long *spec_count = nullptr;
    double *ks_kv = nullptr, *ks_mat = nullptr;   // 'k' model pair search: wavevectors [Q*3]; k_to_q, q_to_k, tol (21 doubles)
    double ks_tol[3] = {0, 0, 0};
    int64_t ks_Q = 0;
    int64_t spec_cap = 0;
    int64_t spec_last = 0;            // pairs of the last nk_specular_pairs call (still on the device)
    // rough tables under construction on the device (nk_rough_begin .. nk_rough_finish)
    int rb_Fr = 0;
    double *rb_k2 = nullptr, *rb_nin = nullptr, *rb_eta = nullptr, *rb_sub = nullptr, *rb_spec = nullptr, *rb_roul = nullptr;
    uint8_t *rb_ts = nullptr;
    unsigned int *rb_map = nullptr;
    std::vector<int32_t> rb_facet;
    int64_t o2o_first = 0;            // 'one_to_one': particles entering at the first step (sizes the spawn inboxes)
    bool stepped = false;             // a timestep has run (the emission ownership of a rank is fixed from then on)
    int64_t emitted_for = -1;         // step whose emission already ran in the tail launch of the step before it (k_tail), also across
                                      // nk_step calls: a driver that steps one by one (Population.run_timestep) then never launches k_emit;
                                      // -1 after anything that changes the store or the tables (the emission is simply run again: its
                                      // counters are double-buffered)
    std::vector<double> h_vg;         // host copy of the group velocities (the mode map deals the modes by their event rate)
    std::vector<int32_t> h_m2s, h_s2m;  // host copies of the mode map (NkDev::m2s / s2m), built with the segmentation
    int32_t *m2s_dev = nullptr, *s2m_dev = nullptr, *nl_dev = nullptr;
    unsigned int *rbar = nullptr;     // k_resident: step word, halt word, one flag per workgroup
    bool walked = false;              // box store: sweeps have alternated since the segments were last moved down (NkDev::seg_lo may be > 0)
    int64_t ev_lds_set = -1;          // dynamic LDS k_events was last allowed (hipFuncSetAttribute)
    int map_nseg = 0;                 // segmentation the map was dealt for
    NkMode *modetab_p = nullptr;      // permuted mode table (own allocation: its size follows nseg)
    int64_t modetab_p_len = 0;
    void *inbox = nullptr, *inbox_n = nullptr;   // 'one_to_one' spawn inboxes (sized with nseg)
    void *mig_buf = nullptr, *mig_n = nullptr;   // migration inboxes (rough facets; sized with nseg and segcap)
    double *ep_p = nullptr, *rc_p = nullptr;     // (reservoir, mode) tables in the segments' order (sized with nseg)
    int rm_nseg = 0, rm_nlmax = 0;               // segmentation rc_p was built for (0: the counters live in res_counter)
    void *pin = nullptr;           // pinned host staging of the history rows + the halt words of a batch
    size_t pin_bytes = 0;
    int32_t halt_words[4] = {0, 0, 0, 0};
    double *acc = nullptr;         // [NB + 2]: tally columns, then the two halt requests that travel with them
    double *hist = nullptr;        // [hist_cap][HROW]
    int hist_cap = 0;
    nk_params params;
    nk_timing timing;
    NkRccl rccl;
    ncclComm_t comm = nullptr;
    int comm_rank = -1, comm_nranks = 0;   // as RCCL reports them (nk_comm_info)
    double comm_selftest = 0.0;
    int num_cu = 256;
    std::vector<int32_t> h_ffo, h_ffi;   // host copies of the mesh's facet -> faces CSR, its area cdf and the vertices
    std::vector<double> h_fcdf, h_verts;
    std::vector<double> h_rbf_inv;       // host copy of the RBF system inverse (sv_interp 3)
    std::vector<NkFacet> host_facets;   // host mirror of d.facets (patched by nk_set_reservoirs / nk_set_rough)
    const double *d_omega = nullptr, *d_vg = nullptr;   // kept to rebuild the packed mode records
    std::vector<double> h_Tgrid;
    double T_lo = 0.0, T_hi = 0.0;      // range of the subvolume temperatures last seen by the host
};

#define NK_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return NK_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)
#define NK_ARG(cond, msg)                                                                              \
    do {                                                                                               \
        if (!(cond)) { ctx->err = msg; return NK_ERR_ARG; }                                            \
    } while (0)

template <class T>
static int nk_upload(nk_ctx *ctx, const T *src, size_t n, const T **dst, bool particle = false) {
    void *p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    NK_HIP(hipMalloc(&p, bytes));
    (particle ? ctx->pallocs : ctx->allocs).push_back(p);
    if (src && n) NK_HIP(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    else NK_HIP(hipMemset(p, 0, bytes));
    *dst = (const T *)p;
    return NK_OK;
}
#define NK_UP(src, n, dst)                                                                             \
    do { int rc_ = nk_upload(ctx, src, n, dst); if (rc_) return rc_; } while (0)

// The particle fields of a store of `cap` slots (a multiple of 64): ONE allocation cut into blocks of 64 slots, every block
// x | y | z | occ | [nts] | [pid] | w0 (NkField, nk_device.h; the sweep addresses a block's fields from ONE base pointer at
// compile-time distances, NkBlock).  (Round 3 also had one plain array per field as a developer layout, NK_LAYOUT=soa: measured
// equal, removed with the single-base addressing.)  The allocation is registered in ctx->pallocs.
static inline bool nk_layout_soa() { return false; }
// bytes of one particle in the store: x y z occ (+ nts unless it is a box store) (+ pid) + the packed word
static inline int nk_particle_bytes(bool with_pid, bool with_nts) { return 36 + (with_nts ? 8 : 0) + (with_pid ? 8 : 0); }
static inline size_t nk_store_bytes(int64_t cap, bool with_pid, bool with_nts) {
    const int64_t nblk = (cap + 63) / 64;
    const int nf = 4 + (with_nts ? 1 : 0) + (with_pid ? 1 : 0);   // 8-byte fields per particle
    const int bd = nf * 64 + 32;                                  // doubles per block (the 64 packed words take 32)
    const size_t bytes = nk_layout_soa() ? (size_t)nblk * 64 * (nf * 8 + 4) : (size_t)nblk * bd * 8;
    return bytes ? bytes : 64;
}
// the fields of a store of `cap` slots that starts at b
static void nk_point_fields(NkDev &d, double *b, int64_t cap, bool with_pid, bool with_nts) {
    const int64_t nblk = (cap + 63) / 64;
    const int nf = 4 + (with_nts ? 1 : 0) + (with_pid ? 1 : 0);
    const int bd = nf * 64 + 32;
    if (nk_layout_soa()) {
        const int64_t n = nblk * 64;
        int k = 4;
        d.x = {b, 64}; d.y = {b + n, 64}; d.z = {b + 2 * n, 64}; d.occ = {b + 3 * n, 64};
        d.nts = {with_nts ? b + (k++) * n : nullptr, 64};
        d.pid = {with_pid ? (uint64_t *)(b + (k++) * n) : nullptr, 64};
        d.w0 = {(uint32_t *)(b + k * n), 64};
    } else {
        int k = 4;
        d.x = {b, bd}; d.y = {b + 64, bd}; d.z = {b + 128, bd}; d.occ = {b + 192, bd};
        d.nts = {with_nts ? b + 64 * (k++) : nullptr, bd};
        d.pid = {with_pid ? (uint64_t *)(b + 64 * (k++)) : nullptr, bd};
        d.w0 = {(uint32_t *)(b + 64 * k), 2 * bd};
    }
}
static int nk_alloc_fields(nk_ctx *ctx, NkDev &d, int64_t cap, bool with_pid, bool with_nts) {
    const size_t bytes = nk_store_bytes(cap, with_pid, with_nts);
    void *buf = nullptr;
    // developer probe (scripts/placement_probe.py): NK_STORE_PAD_KB shifts the store inside a larger allocation
    const size_t pad = getenv("NK_STORE_PAD_KB") ? (size_t)atol(getenv("NK_STORE_PAD_KB")) * 1024 : 0;
    NK_HIP(hipMalloc(&buf, bytes + pad));
    ctx->pallocs.push_back(buf);
    NK_HIP(hipMemsetAsync(buf, 0, bytes + pad, ctx->stream));
    ctx->store_buf = buf; ctx->store_pad = pad; ctx->store_pid = with_pid; ctx->store_nts = with_nts;
    nk_point_fields(d, (double *)((char *)buf + pad), cap, with_pid, with_nts);
    return NK_OK;
}
// Host copy of a whole field (slot order) and back.
template <class T>
static int nk_field_download(nk_ctx *ctx, const NkDev &d, const NkField<T> &f, std::vector<T> &out) {
    // the field's elements are strided by blocks: copy block-wise with hipMemcpy2D (rows of 64 elements)
    out.resize((size_t)d.cap);
    const int64_t nblk = d.cap / 64;
    if (nblk == 0) return NK_OK;
    NK_HIP(hipMemcpy2D(out.data(), 64 * sizeof(T), f.p, (size_t)f.blk * sizeof(T), 64 * sizeof(T), (size_t)nblk, hipMemcpyDeviceToHost));
    return NK_OK;
}
template <class T>
static int nk_field_upload(nk_ctx *ctx, const NkDev &d, const NkField<T> &f, const T *src) {
    const int64_t nblk = d.cap / 64;
    if (nblk == 0) return NK_OK;
    NK_HIP(hipMemcpy2D(f.p, (size_t)f.blk * sizeof(T), src, 64 * sizeof(T), 64 * sizeof(T), (size_t)nblk, hipMemcpyHostToDevice));
    return NK_OK;
}

// 1 = ray-casting tables fit LDS, 2 = they stay in global memory
static inline int nk_geom_mode(const nk_ctx *ctx) { return (ctx->d.F <= NK_LDS_FACES && ctx->d.Fc <= NK_LDS_FACES) ? 1 : 2; }
// kind: 0 plain, 1 k_emit (emission scratch), 2 / 3 k_sweep (mode records, output ring without / with ids)
static inline bool nk_want_split(const nk_ctx *ctx);
static inline size_t nk_lds(const nk_ctx *ctx, bool geom, int kind = 0) {
    const NkDev &d = ctx->d;
    const int gm = geom ? nk_geom_mode(ctx) : 0;
    return nk_lds_bytes(d.S, d.R, d.F, d.NP, d.Fc, gm, kind, (gm == 1 && d.res_lds) ? d.res_nf : 0, d.rbf_P, d.nlrec, nk_want_split(ctx) ? 0 : 1);
}
#define NK_GEOM_LAUNCH(kernel, grid, lds, ...)                                                        \
    do {                                                                                               \
        if (nk_geom_mode(ctx) == 1) kernel<1><<<grid, NK_WG, lds, ctx->stream>>>(__VA_ARGS__);         \
        else kernel<2><<<grid, NK_WG, lds, ctx->stream>>>(__VA_ARGS__);                                \
    } while (0)
// k_emit / k_tail: the same, plus the box store's variant (no first ray cast)
#define NK_EMIT_LAUNCH(kernel, grid, lds, ...)                                                        \
    do {                                                                                               \
        if (ctx->d.box) kernel<1, true><<<grid, NK_WG, lds, ctx->stream>>>(__VA_ARGS__);               \
        else if (nk_geom_mode(ctx) == 1) kernel<1, false><<<grid, NK_WG, lds, ctx->stream>>>(__VA_ARGS__);   \
        else kernel<2, false><<<grid, NK_WG, lds, ctx->stream>>>(__VA_ARGS__);                         \
    } while (0)
static inline int nk_sweep_grid(const nk_ctx *ctx) { return ctx->num_cu * 8; }
// The sweep is instantiated per (table placement, rough facets, RBF temperatures, particle ids, split): run STMT with
// KERNEL bound to the one that matches.  Rough facets draw random numbers per particle, so they imply ids.  The last
// parameter says where the mode records are read from (LDS copies of the segment's share, or the table in HBM).
#define NK_SWEEP_CASE(G, R, B, P, S, lrec, STMT) { if (lrec) { auto KERNEL = k_sweep<G, R, B, P, S, true>; STMT; } else { auto KERNEL = k_sweep<G, R, B, P, S, false>; STMT; } }
// the plain sweep of small meshes also exists with the commonest switches compiled in (nk_sweep_fast)
#define NK_SWEEP_CASE_FAST(lrec, fast, STMT)                                                                                   \
    { if (lrec) { if ((fast) == 1) { auto KERNEL = k_sweep<1, false, false, false, false, true, 1>; STMT; } else { auto KERNEL = k_sweep<1, false, false, false, false, true, 2>; STMT; } } \
      else { if ((fast) == 1) { auto KERNEL = k_sweep<1, false, false, false, false, false, 1>; STMT; } else { auto KERNEL = k_sweep<1, false, false, false, false, false, 2>; STMT; } } }
#define NK_SWEEP_CASE_S(G, R, B, P, split, lrec, STMT) { if (split) NK_SWEEP_CASE(G, R, B, P, true, lrec, STMT) else NK_SWEEP_CASE(G, R, B, P, false, lrec, STMT) }
#define NK_SWEEP_CASE_RP(G, B, rough, pid, split, lrec, STMT)                                         \
    { if (rough) NK_SWEEP_CASE_S(G, true, B, true, split, lrec, STMT) else if (pid) NK_SWEEP_CASE_S(G, false, B, true, split, lrec, STMT) else NK_SWEEP_CASE_S(G, false, B, false, split, lrec, STMT) }
// ... and every sweep of a small mesh exists for the box store (k_sweep's BOX; ctx->d.box decides)
#define NK_SWEEP_CASE_BOX(R, B, P, lrec, STMT) { if (lrec) { auto KERNEL = k_sweep<1, R, B, P, false, true, 0, true>; STMT; } else { auto KERNEL = k_sweep<1, R, B, P, false, false, 0, true>; STMT; } }
#define NK_SWEEP_CASE_BOX_RP(B, rough, pid, lrec, STMT)                                               \
    { if (rough) NK_SWEEP_CASE_BOX(true, B, true, lrec, STMT) else if (pid) NK_SWEEP_CASE_BOX(false, B, true, lrec, STMT) else NK_SWEEP_CASE_BOX(false, B, false, lrec, STMT) }
#define NK_SWEEP_CASE_FAST_BOX(lrec, fast, STMT)                                                                               \
    { if (lrec) { if ((fast) == 1) { auto KERNEL = k_sweep<1, false, false, false, false, true, 1, true>; STMT; } else { auto KERNEL = k_sweep<1, false, false, false, false, true, 2, true>; STMT; } } \
      else { if ((fast) == 1) { auto KERNEL = k_sweep<1, false, false, false, false, false, 1, true>; STMT; } else { auto KERNEL = k_sweep<1, false, false, false, false, false, 2, true>; STMT; } } }
#define NK_SWEEP_DISPATCH(gm, rough, rbf, pid, split, lrec, STMT)                                     \
    do {                                                                                               \
        const int fast_ = ((gm) == 1 && !(rough) && !(rbf) && !(pid) && !(split)) ? nk_sweep_fast(ctx) : 0;   \
        if (ctx->d.box && (gm) == 1 && !(split)) {                                                     \
            if (fast_) { NK_SWEEP_CASE_FAST_BOX(lrec, fast_, STMT) break; }                            \
            if (rbf) NK_SWEEP_CASE_BOX_RP(true, rough, pid, lrec, STMT) else NK_SWEEP_CASE_BOX_RP(false, rough, pid, lrec, STMT)   \
            break;                                                                                     \
        }                                                                                              \
        if (fast_) { NK_SWEEP_CASE_FAST(lrec, fast_, STMT) break; }                                    \
        if ((gm) == 1) { if (rbf) NK_SWEEP_CASE_RP(1, true, rough, pid, split, lrec, STMT) else NK_SWEEP_CASE_RP(1, false, rough, pid, split, lrec, STMT) }   \
        else { if (rbf) NK_SWEEP_CASE_RP(2, true, rough, pid, split, lrec, STMT) else NK_SWEEP_CASE_RP(2, false, rough, pid, split, lrec, STMT) }             \
    } while (0)
// 1 / 2: slice subvolumes, 'nearest' / 'linear' particle temperatures and the local reference temperature (k_sweep's FAST)
static inline int nk_sweep_fast(const nk_ctx *ctx) {
    const NkDev &d = ctx->d;
    if (getenv("NK_NO_FAST")) return 0;              // developer probe
    return (d.sv_kind == 0 && d.T_ref_local && (d.sv_interp == 0 || d.sv_interp == 1)) ? 1 + d.sv_interp : 0;
}
// the sweep keeps its segments' mode records in LDS when the modes are partitioned over the segments and a segment's share fits
static inline bool nk_want_lrec(const nk_ctx *ctx) { return ctx->d.nlrec > 0; }      // decided with the segmentation (nk_alloc_particles)
// k_events, the same way
#define NK_EVENTS_CASE(G, R, B, P, STMT) { auto KERNEL = k_events<G, R, B, P>; STMT; }
#define NK_EVENTS_CASE_RP(G, B, rough, pid, STMT)                                                     \
    { if (rough) NK_EVENTS_CASE(G, true, B, true, STMT) else if (pid) NK_EVENTS_CASE(G, false, B, true, STMT) else NK_EVENTS_CASE(G, false, B, false, STMT) }
#define NK_EVENTS_DISPATCH(gm, rough, rbf, pid, STMT)                                                 \
    do {                                                                                               \
        if ((gm) == 1) { if (rbf) NK_EVENTS_CASE_RP(1, true, rough, pid, STMT) else NK_EVENTS_CASE_RP(1, false, rough, pid, STMT) }   \
        else { if (rbf) NK_EVENTS_CASE_RP(2, true, rough, pid, STMT) else NK_EVENTS_CASE_RP(2, false, rough, pid, STMT) }             \
    } while (0)

// The mode records hold three lifetime rows (two grid intervals) around the live temperature range [T_lo, T_hi] and
// E0 = exp(hbar omega / (kB T0)) at its middle T0 (nk_device.h "lean FP64 arithmetic"); rebuilt when the range leaves the
// window or drifts away from T0, and whenever the segmentation (hence the permuted copy) changes.
static int nk_update_tau_window(nk_ctx *ctx, bool force) {
    NkDev &d = ctx->d;
    if (!ctx->have_material) return NK_OK;
    const std::vector<double> &g = ctx->h_Tgrid;
    const double Tm = 0.5 * (ctx->T_lo + ctx->T_hi);
    int i = (int)(std::lower_bound(g.begin(), g.end(), Tm) - g.begin()) - 1;          // interval (g[i], g[i+1]] holds T_mid
    if (i < 0) i = 0;
    // the second interval on the side where the range reaches further
    int row0 = (i + 1 < d.NT && ctx->T_hi - g[std::min(i + 1, d.NT - 1)] > g[i] - ctx->T_lo) ? i : i - 1;
    if (row0 > d.NT - NK_TAU_ROWS) row0 = d.NT - NK_TAU_ROWS;
    if (row0 < 0) row0 = 0;
    const bool window_ok = d.tau_row0 >= 0 && ctx->T_lo > g[d.tau_row0] &&
                           ctx->T_hi <= g[std::min(d.tau_row0 + NK_TAU_ROWS - 1, d.NT - 1)];
    const bool window_best = d.tau_row0 == row0;                   // a range wider than two intervals: keep the best window
    const bool t0_ok = d.T0 > 0.0 && fabs(ctx->T_lo - d.T0) <= 0.02 * d.T0 && fabs(ctx->T_hi - d.T0) <= 0.02 * d.T0;
    const bool t0_best = d.T0 > 0.0 && fabs(Tm - d.T0) <= 0.002 * d.T0;
    if (!force && (window_ok || window_best) && (t0_ok || t0_best)) return NK_OK;
    d.T0 = Tm > 0.0 ? Tm : 300.0;
    d.invT0 = 1.0 / d.T0;
    d.c_hk = d.hbar / d.kb;
    NkMode *tp = nullptr;
    if (d.part && d.nseg > 0 && d.m2s) {
        const int64_t need = (int64_t)d.nseg * d.nlmax;
        if (need > ctx->modetab_p_len) {
            if (ctx->modetab_p) hipFree(ctx->modetab_p);
            ctx->modetab_p = nullptr; ctx->modetab_p_len = 0;
            NK_HIP(hipMalloc((void **)&ctx->modetab_p, (size_t)need * sizeof(NkMode)));
            NK_HIP(hipMemset(ctx->modetab_p, 0, (size_t)need * sizeof(NkMode)));
            ctx->modetab_p_len = need;
        }
        tp = ctx->modetab_p;
    }
    d.modetab_p = tp;
    k_build_modetab<<<(d.M + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_omega, ctx->d_vg, d.tau, d.M, d.NT, row0, d.c_hk, d.invT0,
                                                                  d.nseg > 0 ? d.nseg : 1, d.nlmax, d.m2s, (NkMode *)d.modetab, tp);
    NK_HIP(hipGetLastError());
    d.tau_row0 = row0;
    ctx->timing.tau_rebuilds += 1;
    for (int k = 0; k < NK_TAU_ROWS; ++k) d.tau_g[k] = (row0 + k < d.NT) ? g[row0 + k] : INFINITY;
    for (int k = 0; k + 1 < NK_TAU_ROWS; ++k) d.tau_ig[k] = 1.0 / (d.tau_g[k + 1] - d.tau_g[k]);
    if (d.NT < NK_TAU_ROWS) d.tau_g[0] = INFINITY;     // no packed window: every lookup takes the full-table path
    return NK_OK;
}

static void nk_track_T(nk_ctx *ctx, const double *T, int S) {
    double lo = T[0], hi = T[0];
    for (int i = 1; i < S; ++i) { lo = std::min(lo, T[i]); hi = std::max(hi, T[i]); }
    ctx->T_lo = lo; ctx->T_hi = hi;
}

extern "C" {

const char *nk_last_error(const nk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int nk_device_count(void) {
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess ? ndev : 0;
}

int nk_create(nk_ctx **out, int device_id, uint64_t seed) {
    if (!out) { g_create_error = "nk_create: out is NULL"; return NK_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("nk_create: no HIP device available (") + hipGetErrorString(e) +
                         "); this library has no CPU fallback";
        return NK_ERR_NODEVICE;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_error = "nk_create: device_id out of range"; return NK_ERR_ARG; }
    nk_ctx *ctx = new nk_ctx();
    memset(&ctx->d, 0, sizeof(NkDev));
    memset(&ctx->params, 0, sizeof(nk_params));
    memset(&ctx->timing, 0, sizeof(nk_timing));
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess) {
        g_create_error = std::string("nk_create: ") + hipGetErrorString(e);
        delete ctx;
        return NK_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_create_error = std::string("nk_create: device is ") + prop.gcnArchName + ", this build targets gfx950 only";
            hipStreamDestroy(ctx->stream);
            delete ctx;
            return NK_ERR_NODEVICE;
        }
    }
    ctx->d.seed = seed;
    ctx->d.rank = 0;
    ctx->d.nranks = 1;
    ctx->d.tau_row0 = -1;
    ctx->d.stamps = nullptr;
    ctx->params.dt = 1.0; ctx->params.T_ref_local = 1; ctx->params.flux_every = 10; ctx->params.contains_every = 100;
    ctx->d.dt = 1.0; ctx->d.inv_dt = 1.0; ctx->d.T_ref_local = 1;
    // bookkeeping words in device memory: halt[4], overflow, ticket, ev_ticket
    const int32_t *p32 = nullptr;
    int rc;
    if ((rc = nk_upload<int32_t>(ctx, nullptr, 16, &p32))) {
        g_create_error = ctx->err;
        delete ctx;
        return rc;
    }
    ctx->d.halt = (int32_t *)p32;
    ctx->d.overflow = (int32_t *)p32 + 4;
    ctx->d.ticket = (int32_t *)p32 + 5;
    ctx->d.ev_ticket = (int32_t *)p32 + 6;
    ctx->anomalies = (int32_t *)p32 + 7;
    *out = ctx;
    return NK_OK;
}

static void nk_specular_free(nk_ctx *ctx);
static inline void nk_specular_free_fwd(nk_ctx *ctx) { nk_specular_free(ctx); }

void nk_destroy(nk_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
    for (auto &e : ctx->evpool) hipEventDestroy(e);
    nk_specular_free_fwd(ctx);
    for (void *p : ctx->allocs) hipFree(p);
    for (void *p : ctx->pallocs) hipFree(p);
    if (ctx->acc) hipFree(ctx->acc);
    if (ctx->rbar) hipFree(ctx->rbar);
    if (ctx->pin) hipHostFree(ctx->pin);
    if (ctx->hist) hipHostFree(ctx->hist);
    if (ctx->modetab_p) hipFree(ctx->modetab_p);
    if (ctx->m2s_dev) hipFree(ctx->m2s_dev);
    if (ctx->s2m_dev) hipFree(ctx->s2m_dev);
    if (ctx->nl_dev) hipFree(ctx->nl_dev);
    if (ctx->inbox) hipFree(ctx->inbox);
    if (ctx->inbox_n) hipFree(ctx->inbox_n);
    if (ctx->mig_buf) hipFree(ctx->mig_buf);
    if (ctx->mig_n) hipFree(ctx->mig_n);
    if (ctx->ep_p) hipFree(ctx->ep_p);
    if (ctx->rc_p) hipFree(ctx->rc_p);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

int nk_set_material(nk_ctx *ctx, const nk_material *m) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && m, "nk_set_material: NULL argument");
    NK_ARG(m->Q > 0 && m->J > 0 && m->NT >= 2 && m->nE >= 2, "nk_set_material: bad sizes");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.Q = m->Q; d.J = m->J; d.NT = m->NT; d.M = m->Q * m->J;
    for (int i = 1; i < m->NT; ++i) NK_ARG(m->T_grid[i] > m->T_grid[i - 1], "nk_set_material: T_grid must ascend");
    NK_UP(m->omega, (size_t)d.M, &ctx->d_omega);
    NK_UP(m->group_vel, (size_t)d.M * 3, &ctx->d_vg);
    ctx->h_vg.assign(m->group_vel, m->group_vel + (size_t)d.M * 3);
    NK_UP((const NkMode *)nullptr, (size_t)d.M, &d.modetab);
    NK_UP(m->lifetime, (size_t)d.NT * d.M, &d.tau);
    NK_UP(m->T_grid, (size_t)d.NT, &d.Tgrid);
    ctx->h_Tgrid.assign(m->T_grid, m->T_grid + m->NT);
    d.nE = m->nE;
    NK_UP(m->T_array, (size_t)d.nE, &d.Tarr);
    NK_UP(m->energy_array, (size_t)d.nE, &d.Earr);
    d.Tfill_lo = m->T_fill_lo; d.Tfill_hi = m->T_fill_hi;
    d.hbar = m->hbar; d.kb = m->kb; d.QV = m->QV; d.active_modes = (double)m->active_modes;
    ctx->have_material = true;
    ctx->T_lo = ctx->T_hi = m->T_grid[0];
    int rc = nk_update_tau_window(ctx, true);
    if (rc) return rc;
    NK_HIP(hipStreamSynchronize(ctx->stream));
    return NK_OK;
}

// One reference of the face tree: a face and the box of the piece of it this reference stands for.
struct NkFaceRef { int face; double lo[3], hi[3]; };
// Cuts the (convex, planar) polygon P at the plane x_axis = c; keeps the side `upper`.
static int nk_clip_polygon(const double (*P)[3], int n, int axis, double c, bool upper, double (*out)[3]) {
    int no = 0;
    for (int i = 0; i < n; ++i) {
        const double *A = P[i], *B = P[(i + 1) % n];
        const bool ina = upper ? A[axis] >= c : A[axis] <= c, inb = upper ? B[axis] >= c : B[axis] <= c;
        if (ina) { for (int k = 0; k < 3; ++k) out[no][k] = A[k]; ++no; }
        if (ina != inb) {
            const double t = (c - A[axis]) / (B[axis] - A[axis]);
            for (int k = 0; k < 3; ++k) out[no][k] = A[k] + t * (B[k] - A[k]);
            out[no][axis] = c;
            ++no;
        }
    }
    return no;
}
static void nk_split_polygon(const double (*P)[3], int n, int face, int depth, std::vector<NkFaceRef> &refs) {
    NkFaceRef r;
    r.face = face;
    for (int k = 0; k < 3; ++k) { r.lo[k] = 1e300; r.hi[k] = -1e300; }
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { r.lo[k] = std::min(r.lo[k], P[i][k]); r.hi[k] = std::max(r.hi[k], P[i][k]); }
    double area2[3] = {0, 0, 0};                        // twice the polygon's vector area
    for (int i = 1; i + 1 < n; ++i) {
        double a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = P[i][k] - P[0][k]; b[k] = P[i + 1][k] - P[0][k]; }
        area2[0] += a[1] * b[2] - a[2] * b[1]; area2[1] += a[2] * b[0] - a[0] * b[2]; area2[2] += a[0] * b[1] - a[1] * b[0];
    }
    const double area = 0.5 * sqrt(area2[0] * area2[0] + area2[1] * area2[1] + area2[2] * area2[2]);
    const double e[3] = {r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]};
    const double box_area = e[0] * e[1] + e[1] * e[2] + e[0] * e[2];      // half the box's surface
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (e[k] > e[axis]) axis = k;
    if (depth <= 0 || n < 3 || n > 12 || !(box_area > 3.0 * area) || !(e[axis] > 0.0)) { refs.push_back(r); return; }
    const double c = 0.5 * (r.lo[axis] + r.hi[axis]);
    double lo[16][3], hi[16][3];
    const int nl = nk_clip_polygon(P, n, axis, c, false, lo), nh = nk_clip_polygon(P, n, axis, c, true, hi);
    if (nl < 3 || nh < 3) { refs.push_back(r); return; }
    nk_split_polygon(lo, nl, face, depth - 1, refs);
    nk_split_polygon(hi, nh, face, depth - 1, refs);
}
static void nk_split_face(const double *V, int face, int max_depth, std::vector<NkFaceRef> &refs) {
    double P[16][3];
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) P[a][k] = V[3 * a + k];
    nk_split_polygon(P, 3, face, max_depth, refs);
}

int nk_set_mesh(nk_ctx *ctx, const nk_mesh *m) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && m, "nk_set_mesh: NULL argument");
    NK_ARG(m->F > 0 && m->Fc > 0, "nk_set_mesh: empty mesh");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.F = m->F; d.Fc = m->Fc; d.tol = m->tol;
    for (int i = 0; i < 6; ++i) d.bbox[i] = m->bbox[i];
    // distinct planes (bitwise-equal normal and k), in order of their first face; faces grouped by plane
    std::vector<std::vector<int>> members;
    for (int f = 0; f < m->F; ++f) {
        int found = -1;
        for (size_t pl = 0; pl < members.size() && found < 0; ++pl) {
            int g = members[pl][0];
            if (m->normals[3 * g] == m->normals[3 * f] && m->normals[3 * g + 1] == m->normals[3 * f + 1] &&
                m->normals[3 * g + 2] == m->normals[3 * f + 2] && m->k[g] == m->k[f])
                found = (int)pl;
        }
        if (found < 0) { members.push_back(std::vector<int>()); found = (int)members.size() - 1; }
        members[found].push_back(f);
    }
    d.NP = (int)members.size();
    std::vector<double> planes((size_t)d.NP * NK_PLANE_DOUBLES, 0.0), faces((size_t)m->F * NK_FACE_DOUBLES, 0.0);
    std::vector<int> face_pos((size_t)m->F, 0);     // where face f went
    int pos = 0;
    for (int pl = 0; pl < d.NP; ++pl) {
        double *p = &planes[(size_t)pl * NK_PLANE_DOUBLES];
        int g = members[pl][0];
        p[0] = m->normals[3 * g]; p[1] = m->normals[3 * g + 1]; p[2] = m->normals[3 * g + 2]; p[3] = m->k[g];
        int32_t rng[2] = {pos, pos + (int)members[pl].size()};
        memcpy(p + 4, rng, 8);
        for (int f : members[pl]) {
            double *q = &faces[(size_t)pos * NK_FACE_DOUBLES];
            for (int k = 0; k < 3; ++k) {
                q[k] = m->bounds_lo[3 * f + k]; q[3 + k] = m->bounds_hi[3 * f + k]; q[6 + k] = m->origins[3 * f + k];
            }
            const double *A = m->basis + 9 * f;      // A[d][b]: columns are (e1, e2, n)
            double a00 = A[0], a01 = A[1], a02 = A[2], a10 = A[3], a11 = A[4], a12 = A[5], a20 = A[6], a21 = A[7], a22 = A[8];
            double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
            NK_ARG(det != 0.0 && isfinite(det), "nk_set_mesh: degenerate face");
            q[9] = (a11 * a22 - a12 * a21) / det; q[10] = (a02 * a21 - a01 * a22) / det; q[11] = (a01 * a12 - a02 * a11) / det;
            q[12] = (a12 * a20 - a10 * a22) / det; q[13] = (a00 * a22 - a02 * a20) / det; q[14] = (a02 * a10 - a00 * a12) / det;
            int32_t id[2] = {f, m->face_facet[f]};
            memcpy(q + 15, id, 8);
            face_pos[f] = pos;
            ++pos;
        }
    }
    // Large meshes (tables in global memory): the face tree of nk_find_boundary_tree.  Faces in Morton order of their box
    // centres, leaves of 4, every level the unions of 4 consecutive nodes of the level below.
    d.NG = 0;
    d.tree_top = 0; d.tree_leaves = 0;
    d.tree_nfam = 0; d.tree_lds_fam0 = 0; d.tree_lds_off = 0;
    for (int k = 0; k < NK_TREE_LEVELS; ++k) d.tree_base[k] = 0;
    bool use_tree = !(m->F <= NK_LDS_FACES && m->Fc <= NK_LDS_FACES) && !getenv("NK_NO_TREE");   // env: developer probe
    std::vector<NkFaceRef> refs;
    if (use_tree) {
        // A sliver lying diagonally has a box far larger than itself (the 1250 fan triangles of a wire's cap each cover an
        // eighth of the cap), so such faces enter the tree as several references, each with the box of one piece of the
        // triangle.  The face test itself is unchanged: a face met twice gives the same answer twice.
        const int max_depth = getenv("NK_TREE_SPLIT") ? atoi(getenv("NK_TREE_SPLIT")) : 4;
        for (int f = 0; f < m->F; ++f) nk_split_face(m->vertices + 9 * (size_t)f, f, max_depth, refs);
        if (refs.size() > (size_t)4 * (1 << (2 * NK_TREE_LEVELS))) {      // too many for the implicit tree: whole faces
            refs.clear();
            for (int f = 0; f < m->F; ++f) nk_split_face(m->vertices + 9 * (size_t)f, f, 0, refs);
        }
        use_tree = refs.size() <= (size_t)4 * (1 << (2 * NK_TREE_LEVELS));
    }
    if (use_tree) {
        const int F = (int)refs.size();
        // Shape: `top` levels of 4 below a top family of c = 2..4 nodes, every node present, so that the walk's index
        // arithmetic needs no child pointers; leaves hold 2 to 4 references.  Contents: top-down, every node's references
        // dealt evenly to its children after ordering them along the longest axis of their centres (median cuts) --
        // consecutive Morton codes cut at fixed counts gave boxes that a ray entered 35-45 times per cast.
        int top = 0;
        while ((int64_t)4 << (2 * top) < (int64_t)(F + 3) / 4) ++top;
        const int per_top = 1 << (2 * top);                                    // leaves below one node of the top family
        const int c_top = std::max(1, (int)(((int64_t)(F + 3) / 4 + per_top - 1) / per_top));
        const int NL = c_top * per_top;
        std::vector<int> order((size_t)F), slot((size_t)NL * 4, -1);
        for (int i = 0; i < F; ++i) order[i] = i;
        std::vector<double> cen((size_t)F * 3);
        for (int i = 0; i < F; ++i) for (int k = 0; k < 3; ++k) cen[3 * (size_t)i + k] = 0.5 * (refs[i].lo[k] + refs[i].hi[k]);
        // cut [a, b) into parts of the given sizes along the longest axis of the centres, two groups at a time
        std::function<void(int, int, const int *, int)> cut = [&](int a, int b, const int *sizes, int n) {
            if (n <= 1 || b - a <= 1) return;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int i = a; i < b; ++i)
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], cen[3 * (size_t)order[i] + k]); hi[k] = std::max(hi[k], cen[3 * (size_t)order[i] + k]); }
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
            const int nl = n / 2;
            int left = 0;
            for (int i = 0; i < nl; ++i) left += sizes[i];
            if (left > 0 && left < b - a)
                std::nth_element(order.begin() + a, order.begin() + a + left, order.begin() + b, [&](int p, int q) {
                    const double cp = cen[3 * (size_t)p + axis], cq = cen[3 * (size_t)q + axis];
                    return cp < cq || (cp == cq && p < q);
                });
            cut(a, a + left, sizes, nl);
            cut(a + left, b, sizes + nl, n - nl);
        };
        std::function<void(int, int, int, int)> deal = [&](int a, int b, int level, int node) {   // node of `level` gets [a, b)
            if (level == 0) {
                for (int i = a; i < b; ++i) slot[(size_t)4 * node + (i - a)] = order[i];
                return;
            }
            int sizes[4];
            for (int k = 0; k < 4; ++k) sizes[k] = (b - a) / 4 + (k < (b - a) % 4 ? 1 : 0);
            cut(a, b, sizes, 4);
            int pos = a;
            for (int k = 0; k < 4; ++k) { deal(pos, pos + sizes[k], level - 1, 4 * node + k); pos += sizes[k]; }
        };
        {
            int sizes[4] = {0, 0, 0, 0};
            for (int k = 0; k < c_top; ++k) sizes[k] = F / c_top + (k < F % c_top ? 1 : 0);
            cut(0, F, sizes, c_top);
            int pos = 0;
            for (int k = 0; k < c_top; ++k) { deal(pos, pos + sizes[k], top, k); pos += sizes[k]; }
        }
        // leaf records (NK_TREE_LEAF_DOUBLES, nk_device.h): the four faces' boxes in single precision (filled in below, once the
        // slack is known), then per face plane, box, barycentric rows; the padding faces keep n = k = 0 and can never be hit
        std::vector<double> tf((size_t)NL * NK_TREE_LEAF_DOUBLES, 0.0);
        for (size_t i = 0; i < slot.size(); ++i) {
            if (slot[i] < 0) continue;
            const int f = refs[slot[i]].face;
            double *q = &tf[(i / 4) * NK_TREE_LEAF_DOUBLES + 16 + 20 * (i % 4)];
            const double *src = &faces[(size_t)face_pos[f] * NK_FACE_DOUBLES];
            q[0] = m->normals[3 * f]; q[1] = m->normals[3 * f + 1]; q[2] = m->normals[3 * f + 2]; q[3] = m->k[f];
            memcpy(q + 4, src, NK_FACE_DOUBLES * sizeof(double));
        }
        // boxes: union of the member faces' boxes, inflated by far more than any rounding in the slab test
        double big = 0.0;
        for (int k = 0; k < 6; ++k) big = std::max(big, fabs(m->bbox[k]));
        // ... and, for a piece of a face, by the slack of the barycentric test (a point accepted with u = -tol lies
        // tol * |edge| outside the triangle)
        const double margin = m->tol + 1e-6 * (1.0 + big);
        std::vector<double> slack((size_t)m->F);
        for (int f = 0; f < m->F; ++f) {
            const double *V = m->vertices + 9 * (size_t)f;
            double e = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < 3; ++k) e = std::max(e, fabs(V[3 * a + k] - V[3 * ((a + 1) % 3) + k]));
            slack[f] = margin + 8.0 * m->tol * e;
        }
        for (size_t i = 0; i < slot.size(); ++i) {           // the faces' own boxes, rounded outwards (padding: a box no ray enters)
            float *fb = reinterpret_cast<float *>(&tf[(i / 4) * NK_TREE_LEAF_DOUBLES]) + 6 * (i % 4);
            for (int k = 0; k < 3; ++k) {
                if (slot[i] < 0) { fb[k] = 3.0e38f; fb[3 + k] = -3.0e38f; continue; }
                const NkFaceRef &r = refs[slot[i]];
                fb[k] = nextafterf((float)(r.lo[k] - slack[r.face]), -INFINITY);
                fb[3 + k] = nextafterf((float)(r.hi[k] + slack[r.face]), INFINITY);
            }
        }
        std::vector<double> boxes;
        int count = NL, level = 0, prev_base = 0;
        for (;;) {
            const int padded = (count + 3) / 4 * 4, base = (int)(boxes.size() / 6);
            d.tree_base[level] = base;
            boxes.resize(boxes.size() + (size_t)padded * 6, 0.0);
            for (int i = 0; i < count; ++i) {
                double *B = &boxes[(size_t)(base + i) * 6];
                for (int k = 0; k < 3; ++k) { B[k] = 1e300; B[3 + k] = -1e300; }
                if (level == 0) {
                    for (int c = 4 * i; c < 4 * i + 4; ++c) {
                        if (slot[c] < 0) continue;
                        const NkFaceRef &r = refs[slot[c]];
                        for (int k = 0; k < 3; ++k) {
                            B[k] = std::min(B[k], r.lo[k] - slack[r.face]);
                            B[3 + k] = std::max(B[3 + k], r.hi[k] + slack[r.face]);
                        }
                    }
                } else {
                    const int below = (int)((size_t)(base - prev_base));   // nodes stored for the level below (padded)
                    for (int c = 4 * i; c < 4 * i + 4 && c < below; ++c) {
                        const double *C = &boxes[(size_t)(prev_base + c) * 6];
                        if (C[0] > C[3]) continue;                          // padding node
                        for (int k = 0; k < 3; ++k) { B[k] = std::min(B[k], C[k]); B[3 + k] = std::max(B[3 + k], C[3 + k]); }
                    }
                }
            }
            for (int i = count; i < padded; ++i) {                          // padding nodes: never entered (index >= count)
                double *B = &boxes[(size_t)(base + i) * 6];
                for (int k = 0; k < 3; ++k) { B[k] = 1e300; B[3 + k] = -1e300; }
            }
            if (count <= 4) break;
            prev_base = base;
            count = (count + 3) / 4;
            ++level;
        }
        d.tree_top = level;
        d.tree_nfam = (int32_t)(boxes.size() / 24);
        d.tree_lds_fam0 = d.tree_nfam; d.tree_lds_off = 0;
        auto tree_base_of = [&](int lv) { return (int)d.tree_base[lv]; };
        d.tree_leaves = NL;
        d.NG = 1;
        // facet tags: the facet of a node whose faces all belong to one facet, else -1 (nk_tree_skip)
        std::vector<int32_t> tags(boxes.size() / 6, -1);
        {
            const int nlev = level + 1;
            int cnt = NL;
            for (int lv = 0; lv < nlev; ++lv) {
                const int b0 = tree_base_of(lv);
                for (int i = 0; i < cnt; ++i) {
                    int tag = -3;                               // -3: nothing seen yet
                    for (int c = 4 * i; c < 4 * i + 4; ++c) {
                        int t;
                        if (lv == 0) { if (slot[c] < 0) continue; t = m->face_facet[refs[slot[c]].face]; }
                        else { const int below = tree_base_of(lv) - tree_base_of(lv - 1); if (c >= below) continue; t = tags[(size_t)tree_base_of(lv - 1) + c]; if (t == -3) continue; }
                        tag = (tag == -3) ? t : (tag == t ? tag : -1);
                    }
                    tags[(size_t)b0 + i] = tag == -3 ? -1 : tag;
                }
                cnt = (cnt + 3) / 4;
            }
        }
        // families of four: single-precision boxes rounded outwards (one ulp further than the nearest float: cheap and safe)
        const size_t nfam = boxes.size() / 24;
        std::vector<float> fboxes(nfam * NK_TREE_FAMILY_FLOATS);
        double bound = 0.0;
        for (size_t f = 0; f < nfam; ++f) {
            for (int j = 0; j < 24; ++j) {
                const bool is_lo = (j % 6) < 3;
                double v = boxes[f * 24 + j];
                if (v > 3.0e38) v = 3.0e38;                     // padding nodes (never entered)
                if (v < -3.0e38) v = -3.0e38;
                fboxes[f * NK_TREE_FAMILY_FLOATS + j] = nextafterf((float)v, is_lo ? -INFINITY : INFINITY);
                if (fabs(v) < 1e37) bound = std::max(bound, fabs(v));
            }
        }
        d.tree_bound = bound;
        NK_UP(fboxes.data(), fboxes.size(), &d.tree_boxes);
        NK_UP(tags.data(), tags.size(), &d.tree_tags);
        NK_UP(tf.data(), tf.size(), &d.tree_faces);
    }
    NK_UP(planes.data(), planes.size(), &d.planes);
    NK_UP(faces.data(), faces.size(), &d.faces);
    NK_UP(m->vertices, (size_t)m->F * 9, &d.face_verts);
    NK_UP(m->facet_face_off, (size_t)m->Fc + 1, &d.facet_face_off);
    const int nidx = m->facet_face_off[m->Fc];
    NK_UP(m->facet_face_idx, (size_t)nidx, &d.facet_face_idx);
    std::vector<double> cdf((size_t)nidx, 1.0);
    for (int fc = 0; fc < m->Fc; ++fc) {          // np.random.choice(p=areas/sum): cdf = cumsum(p); cdf /= cdf[-1]
        int f0 = m->facet_face_off[fc], f1 = m->facet_face_off[fc + 1];
        double tot = 0.0, acc = 0.0;
        for (int a = f0; a < f1; ++a) tot += m->face_area[m->facet_face_idx[a]];
        for (int a = f0; a < f1; ++a) { acc += m->face_area[m->facet_face_idx[a]] / tot; cdf[a] = acc; }
        for (int a = f0; a < f1; ++a) cdf[a] /= cdf[f1 - 1];
    }
    NK_UP(cdf.data(), cdf.size(), &d.facet_face_cdf);
    ctx->h_ffo.assign(m->facet_face_off, m->facet_face_off + m->Fc + 1);
    ctx->h_ffi.assign(m->facet_face_idx, m->facet_face_idx + nidx);
    ctx->h_fcdf = cdf;
    ctx->h_verts.assign(m->vertices, m->vertices + (size_t)m->F * 9);
    std::vector<NkFacet> fct((size_t)m->Fc);
    for (int fc = 0; fc < m->Fc; ++fc) {
        NkFacet &q = fct[fc];
        memset(&q, 0, sizeof(q));
        q.cx = m->facet_centroid[3 * fc]; q.cy = m->facet_centroid[3 * fc + 1]; q.cz = m->facet_centroid[3 * fc + 2];
        q.nx = m->facet_normal[3 * fc]; q.ny = m->facet_normal[3 * fc + 1]; q.nz = m->facet_normal[3 * fc + 2];
        q.bc = m->facet_bc[fc]; q.partner = m->facet_partner[fc]; q.res = -1; q.rough = -1;
        NK_ARG(q.bc == 'T' || q.bc == 'F' || q.bc == 'P' || q.bc == 'R', "nk_set_mesh: unknown boundary condition");
        NK_ARG(q.bc != 'P' || (q.partner >= 0 && q.partner < m->Fc), "nk_set_mesh: periodic facet without partner");
        if (q.bc == 'P') {                          // translation of a periodic crossing, Population.py:1467
            const int pf = q.partner;
            q.tx = m->facet_centroid[3 * pf] - q.cx; q.ty = m->facet_centroid[3 * pf + 1] - q.cy; q.tz = m->facet_centroid[3 * pf + 2] - q.cz;
        }
    }
    ctx->host_facets = fct;
    NK_UP(fct.data(), fct.size(), &d.facets);
    {   // how far a facet's faces are from its own plane (nk_tree_skip): largest component of n_face - n_facet, and the
        // largest |n_face . centroid + k_face|
        std::vector<double> fs((size_t)m->Fc * 2, 0.0);
        for (int f = 0; f < m->F; ++f) {
            const int fc = m->face_facet[f];
            if (fc < 0 || fc >= m->Fc) continue;
            const NkFacet &q = fct[fc];
            const double nx = m->normals[3 * f], ny = m->normals[3 * f + 1], nz = m->normals[3 * f + 2];
            const double dn = std::max(fabs(nx - q.nx), std::max(fabs(ny - q.ny), fabs(nz - q.nz)));
            const double dk = fabs(nx * q.cx + ny * q.cy + nz * q.cz + m->k[f]);
            fs[2 * (size_t)fc] = std::max(fs[2 * (size_t)fc], dn);
            fs[2 * (size_t)fc + 1] = std::max(fs[2 * (size_t)fc + 1], dk);
        }
        // only facets of many faces are worth it (a cap, a flat side of an imported mesh): the others say "never"
        std::vector<int> nfaces((size_t)m->Fc, 0);
        for (int f = 0; f < m->F; ++f) if (m->face_facet[f] >= 0 && m->face_facet[f] < m->Fc) ++nfaces[m->face_facet[f]];
        for (int fc = 0; fc < m->Fc; ++fc) if (nfaces[fc] < 16) fs[2 * (size_t)fc] = 1e300;
        NK_UP(fs.data(), fs.size(), &d.facet_skip);
    }
    d.nS = m->nS;
    if (m->nS > 0) {
        NK_UP(m->simplex_pts, (size_t)m->nS * 12, &d.simplex_pts);
        std::vector<double> sc((size_t)m->nS);
        double tot = 0.0, acc = 0.0;
        for (int s = 0; s < m->nS; ++s) tot += m->simplex_vol[s];
        for (int s = 0; s < m->nS; ++s) { acc += m->simplex_vol[s] / tot; sc[s] = acc; }
        for (int s = 0; s < m->nS; ++s) sc[s] /= sc[m->nS - 1];
        NK_UP(sc.data(), sc.size(), &d.simplex_cdf);
    }
    // Box store (nk_device.h, NkDev::box): six planes with exact unit axis normals, one facet each, whose faces are the whole
    // side of the bounding box (two triangles over its four corners).  Then "which wall does a particle meet next, and when"
    // follows from its position alone, with the reference's own expression (nk_box_first_hit), and need not be stored.
    ctx->mesh_box = false;
    if (d.cap == 0) d.box = 0;                       // (an existing store keeps its layout until nk_check_ready re-deals it)
    if (m->F == 12 && m->Fc == 6 && d.NP == 6) {
        bool ok = true;
        int seen = 0;
        int box_facet[6] = {0, 0, 0, 0, 0, 0};
        d.box_ids = 0;
        for (int pl = 0; pl < 6 && ok; ++pl) {
            const double *pn = &planes[(size_t)pl * NK_PLANE_DOUBLES];
            int a = -1, sgn = 0;
            for (int k = 0; k < 3; ++k) {
                if (pn[k] == 1.0 || pn[k] == -1.0) { if (a >= 0) ok = false; a = k; sgn = pn[k] > 0 ? 1 : 0; }
                else if (pn[k] != 0.0) ok = false;
            }
            if (!ok || a < 0) { ok = false; break; }
            const int w = 2 * a + sgn;
            if (seen & (1 << w)) { ok = false; break; }
            seen |= 1 << w;
            const double wall = sgn ? -pn[3] : pn[3];              // n.x + k = 0  ->  x_a = -k (normal +e_a) or k (-e_a)
            if (!(fabs(wall - m->bbox[(sgn ? 3 : 0) + a]) <= 1e-9 * (1.0 + fabs(wall)))) { ok = false; break; }
            if (members[pl].size() != 2) { ok = false; break; }
            int fct0 = -1, face0 = 0x7fffffff;
            double area = 0.0;
            for (int f : members[pl]) {
                if (fct0 < 0) fct0 = m->face_facet[f]; else if (m->face_facet[f] != fct0) ok = false;
                face0 = std::min(face0, f);
                area += m->face_area[f];
                const double *V = m->vertices + 9 * (size_t)f;
                for (int c = 0; c < 3 && ok; ++c)
                    for (int k = 0; k < 3; ++k) {
                        const double v = V[3 * c + k], lo = m->bbox[k], hi = m->bbox[3 + k];
                        const double e = 1e-9 * (1.0 + fabs(lo) + fabs(hi));
                        if (k == a) { if (fabs(v - wall) > e) ok = false; }
                        else if (fabs(v - lo) > e && fabs(v - hi) > e) ok = false;      // a corner of the side
                    }
            }
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            const double side = (m->bbox[3 + b] - m->bbox[b]) * (m->bbox[3 + c] - m->bbox[c]);
            if (!(fabs(area - side) <= 1e-9 * side) || fct0 < 0 || fct0 >= m->Fc) ok = false;
            if (!ok) break;
            d.box_k[w] = pn[3];
            box_facet[w] = fct0;
            d.box_ids = (d.box_ids & ~(0xFFull << (8 * w))) | ((uint64_t)(fct0 & 15) << (8 * w)) | ((uint64_t)(face0 & 15) << (8 * w + 4));
        }
        if (ok && seen == 63) {
            // one facet per wall
            int fm = 0;
            for (int w = 0; w < 6; ++w) fm |= 1 << box_facet[w];
            ctx->mesh_box = fm == 63;
        }
    }
    ctx->have_mesh = true;
    return NK_OK;
}

// The box store is used when the mesh allows it, the tables sit in LDS, and nothing in this context spoke against it
// (NK_NO_BOX=1: developer switch / tests of the cached layout on boxes).
static inline bool nk_want_box(const nk_ctx *ctx) {
    return ctx->have_mesh && ctx->mesh_box && !ctx->box_forbidden && !getenv("NK_NO_BOX") && !getenv("NK_SPLIT") && !nk_layout_soa() &&
           ctx->d.F <= NK_LDS_FACES && ctx->d.Fc <= NK_LDS_FACES;
}

static int nk_patch_facets(nk_ctx *ctx) {
    NK_HIP(hipMemcpy((void *)ctx->d.facets, ctx->host_facets.data(), (size_t)ctx->d.Fc * sizeof(NkFacet), hipMemcpyHostToDevice));
    return NK_OK;
}

// The reservoir counters live in the segments' order (rc_p) while a store exists; before the segmentation changes (or
// the reservoirs are replaced) they go back to the caller's order.
static int nk_entry_tables_drop(nk_ctx *ctx, bool keep_counters) {
    NkDev &d = ctx->d;
    if (ctx->rm_nseg > 0 && keep_counters && d.R > 0 && d.res_counter) {
        const int64_t n = (int64_t)d.R * d.M;
        k_perm_rm<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(0, d.R, d.M, ctx->rm_nseg, ctx->rm_nlmax,
                                                                   d.m2s, d.res_counter, ctx->rc_p + (ctx->step & 1) * d.rc_len);      // (the map the tables were built with; the copy the next step reads)
        NK_HIP(hipGetLastError());
        NK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (ctx->ep_p) { hipFree(ctx->ep_p); ctx->ep_p = nullptr; }
    if (ctx->rc_p) { hipFree(ctx->rc_p); ctx->rc_p = nullptr; }
    d.ep_p = nullptr; d.rc_p = nullptr;
    ctx->rm_nseg = 0;
    return NK_OK;
}
static int nk_entry_tables_build(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    if (!(d.R > 0 && d.nseg > 0 && d.m2s && d.enter_prob && d.res_counter) || ctx->rm_nseg == d.nseg) return NK_OK;
    int rc = nk_entry_tables_drop(ctx, true);
    if (rc) return rc;
    const size_t len = (size_t)d.nseg * d.R * d.nlmax;
    NK_HIP(hipMalloc((void **)&ctx->ep_p, len * 8));
    NK_HIP(hipMalloc((void **)&ctx->rc_p, 2 * len * 8));
    NK_HIP(hipMemsetAsync(ctx->ep_p, 0, len * 8, ctx->stream));
    NK_HIP(hipMemsetAsync(ctx->rc_p, 0, 2 * len * 8, ctx->stream));
    d.rc_len = (int64_t)len;
    const int64_t n = (int64_t)d.R * d.M;
    k_perm_rm<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(1, d.R, d.M, d.nseg, d.nlmax, d.m2s, (double *)d.enter_prob, ctx->ep_p);
    k_perm_rm<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(1, d.R, d.M, d.nseg, d.nlmax, d.m2s, d.res_counter, ctx->rc_p + (ctx->step & 1) * d.rc_len);
    NK_HIP(hipGetLastError());
    d.ep_p = ctx->ep_p; d.rc_p = ctx->rc_p;
    ctx->rm_nseg = d.nseg;
    ctx->rm_nlmax = d.nlmax;
    return NK_OK;
}

static int nk_alloc_tally(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    d.NB = 5 * d.S + 5 * d.R + 1;
    if (ctx->acc) { hipFree(ctx->acc); ctx->acc = nullptr; }
    if (ctx->hist) { hipHostFree(ctx->hist); ctx->hist = nullptr; }      // the history rows follow NB
    ctx->hist_cap = 0;
    NK_HIP(hipMalloc((void **)&ctx->acc, (size_t)(d.NB + 2) * sizeof(double)));
    NK_HIP(hipMemset(ctx->acc, 0, (size_t)(d.NB + 2) * sizeof(double)));
    const double *p;
    NK_UP((const double *)nullptr, (size_t)(ctx->num_cu * 16) * d.NB, &p);     // >= the sweep's and k_events' persistent grids
    d.partials = (double *)p;
    return NK_OK;
}

// Append [w; p] = inv[:, :S] @ T_sv to a host copy of the temperatures (what k_update does on the device).
static void nk_rbf_coefficients(nk_ctx *ctx, std::vector<double> &T) {
    const int S = ctx->d.S, P = ctx->d.rbf_P;
    T.resize((size_t)S + P);
    for (int j = 0; j < P; ++j) {
        double a = 0.0;
        for (int i = 0; i < S; ++i) a += ctx->h_rbf_inv[(size_t)j * P + i] * T[i];
        T[(size_t)S + j] = a;
    }
}

int nk_set_subvolumes(nk_ctx *ctx, const nk_subvols *s, const double *T_sv_init) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && s && T_sv_init, "nk_set_subvolumes: NULL argument");
    NK_ARG(s->S > 0 && s->S <= 512, "nk_set_subvolumes: S must be in [1, 512]");
    NK_ARG(ctx->have_material, "nk_set_subvolumes: call nk_set_material first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.S = s->S; d.sv_kind = s->kind; d.sv_axis = s->axis; d.sv_interp = s->interp;
    NK_ARG(s->axis >= 0 && s->axis < 3, "nk_set_subvolumes: axis");
    NK_ARG(!(s->kind != 0 && s->interp != 2 && s->interp != 3), "nk_set_subvolumes: slice interpolation needs slice subvolumes");
    NK_ARG(s->interp >= 0 && s->interp <= 3, "nk_set_subvolumes: interp must be 0..3");
    NK_UP(s->centers, (size_t)s->S * 3, &d.centers);
    NK_UP(s->volumes, (size_t)s->S, &d.sv_volume);
    d.rbf_P = 0;
    std::vector<double> t0(T_sv_init, T_sv_init + s->S);
    if (s->interp == 3) {
        NK_ARG(s->rbf_inv && s->rbf_shift && s->rbf_scale, "nk_set_subvolumes: interp 3 needs the RBF tables");
        const int nd = (s->rbf_used[0] != 0) + (s->rbf_used[1] != 0) + (s->rbf_used[2] != 0);
        NK_ARG(nd >= 1, "nk_set_subvolumes: RBF interpolation over no coordinate");
        d.rbf_P = s->S + nd + 1;
        for (int k = 0; k < 3; ++k) {
            d.rbf_used[k] = s->rbf_used[k] != 0; d.rbf_shift[k] = s->rbf_shift[k]; d.rbf_scale[k] = s->rbf_scale[k];
            NK_ARG(!d.rbf_used[k] || d.rbf_scale[k] != 0.0, "nk_set_subvolumes: zero RBF scale");
        }
        NK_UP(s->rbf_inv, (size_t)d.rbf_P * d.rbf_P, &d.rbf_inv);
        ctx->h_rbf_inv.assign(s->rbf_inv, s->rbf_inv + (size_t)d.rbf_P * d.rbf_P);
        nk_rbf_coefficients(ctx, t0);
    }
    const double *t;
    NK_UP(t0.data(), t0.size(), &t);
    d.T_sv = (double *)t;
    if (s->kind == 0 && s->S > 1) {
        double c0 = s->centers[s->axis], c1 = s->centers[3 + s->axis];
        NK_ARG(c1 > c0, "nk_set_subvolumes: slice centres must ascend along the axis");
        double Lx = c1 - c0;
        d.sv_lo = c0 - 0.5 * Lx; d.sv_invL = 1.0 / Lx;
    } else { d.sv_lo = 0.0; d.sv_invL = 0.0; }
    ctx->have_sv = true;
    nk_track_T(ctx, T_sv_init, s->S);
    int rc = nk_update_tau_window(ctx, false);
    if (rc) return rc;
    return nk_alloc_tally(ctx);
}

int nk_set_reservoirs(nk_ctx *ctx, const nk_reservoirs *r) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && r, "nk_set_reservoirs: NULL argument");
    NK_ARG(ctx->have_material && ctx->have_mesh && ctx->have_sv, "nk_set_reservoirs: set material, mesh, subvolumes first");
    NK_ARG(r->R >= 0 && r->R <= 64, "nk_set_reservoirs: R must be in [0, 64]");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    { int rc0 = nk_entry_tables_drop(ctx, false); if (rc0) return rc0; }
    d.R = r->R; d.res_gen = r->gen;
    d.res_nf = 0; d.res_lds = 0;
    NK_ARG((int64_t)d.R * d.M < (1ll << 28), "nk_set_reservoirs: R*Q*J too large for the particle id layout");
    if (r->R > 0) {
        NK_UP(r->facet, (size_t)r->R, &d.res_facet);
        NK_UP(r->T, (size_t)r->R, &d.res_T);
        NK_UP(r->enter_prob, (size_t)r->R * d.M, &d.enter_prob);
        const double *c;
        NK_UP(r->counter, (size_t)r->R * d.M, &c);
        d.res_counter = (double *)c;
        double pmax = 0.0;
        for (size_t i = 0; i < (size_t)r->R * d.M; ++i) pmax = std::max(pmax, r->enter_prob[i]);
        ctx->h_enter_prob.assign(r->enter_prob, r->enter_prob + (size_t)r->R * d.M);
        ctx->o2o_first = 0;
        NK_ARG(r->gen >= 0 && r->gen <= 2, "nk_set_reservoirs: gen must be 0 (constant), 1 (fixed_rate) or 2 (one_to_one)");
        if (r->gen == 2) {
            // one_to_one: cumulative enter_prob per reservoir (np.cumsum, then / max: Population.py:467-468) and the
            // first step's emission; the list must hold whatever leaves in one step
            NK_ARG(r->n_leaving, "nk_set_reservoirs: one_to_one needs n_leaving");
            std::vector<double> roul((size_t)r->R * d.M);
            std::vector<int32_t> nl((size_t)r->R);
            int64_t first = 0;
            for (int i = 0; i < r->R; ++i) {
                double run = 0.0, mx = 0.0;
                for (int m = 0; m < d.M; ++m) { run += r->enter_prob[(size_t)i * d.M + m]; roul[(size_t)i * d.M + m] = run; mx = std::max(mx, run); }
                NK_ARG(mx > 0.0, "nk_set_reservoirs: one_to_one reservoir with zero entry probability");
                for (int m = 0; m < d.M; ++m) roul[(size_t)i * d.M + m] /= mx;
                NK_ARG(r->n_leaving[i] >= 0 && r->n_leaving[i] < (1ll << 24), "nk_set_reservoirs: n_leaving out of range");
                nl[(size_t)i] = (int32_t)r->n_leaving[i];
                first += r->n_leaving[i];
            }
            NK_UP(roul.data(), roul.size(), &d.res_roulette);
            const int32_t *pn;
            NK_UP(nl.data(), nl.size(), &pn);
            d.nleave_prev = (int32_t *)pn;
            ctx->o2o_first = first;
        }
        NK_ARG(pmax < 4094.0, "nk_set_reservoirs: more than 4094 particles of one mode per step (id layout)");
        for (int i = 0; i < r->R; ++i) {
            NK_ARG(r->facet[i] >= 0 && r->facet[i] < d.Fc, "nk_set_reservoirs: facet index");
            ctx->host_facets[r->facet[i]].res = i;
        }
        // sampling tables of the reservoir facets, gathered per reservoir (faces in Mesh.sample_surface order)
        std::vector<int32_t> off((size_t)r->R + 1, 0);
        std::vector<double> rcdf, rverts;
        for (int i = 0; i < r->R; ++i) {
            const int f0 = ctx->h_ffo[r->facet[i]], f1 = ctx->h_ffo[r->facet[i] + 1];
            NK_ARG(f1 > f0, "nk_set_reservoirs: reservoir facet without faces");
            for (int a = f0; a < f1; ++a) {
                rcdf.push_back(ctx->h_fcdf[a]);
                const double *v = &ctx->h_verts[(size_t)ctx->h_ffi[a] * 9];
                rverts.insert(rverts.end(), v, v + 9);
            }
            off[(size_t)i + 1] = (int32_t)rcdf.size();
        }
        NK_UP(off.data(), off.size(), &d.res_face_off);
        NK_UP(rcdf.data(), rcdf.size(), &d.res_face_cdf);
        NK_UP(rverts.data(), rverts.size(), &d.res_face_verts);
        d.res_nf = (int32_t)rcdf.size();
        d.res_lds = d.res_nf <= NK_LDS_RESFACES ? 1 : 0;
        int rc = nk_patch_facets(ctx);
        if (rc) return rc;
        if ((rc = nk_entry_tables_build(ctx))) return rc;
    }
    return nk_alloc_tally(ctx);
}

int nk_set_rough(nk_ctx *ctx, const nk_rough *r) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && r, "nk_set_rough: NULL argument");
    NK_ARG(ctx->have_material && ctx->have_mesh, "nk_set_rough: set material and mesh first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.Fr = r->Fr;
    ctx->g_sweep = 0;
    if (r->Fr > 0) {
        size_t n = (size_t)r->Fr * d.M;
        NK_UP(r->specularity, n, &d.specularity);
        NK_UP(r->true_spec, n, &d.true_spec);
        NK_UP(r->spec_map, n, &d.spec_map);
        NK_UP(r->roulette, n, &d.roulette);
        {   // bucket index of the roulette search (nk_reflect): first position with roulette >= k / nlut * last
            int nlut = 1024;
            while (nlut < 65536 && (int64_t)nlut * 4 < d.M) nlut *= 2;
            d.roul_nlut = nlut;
            std::vector<int32_t> lut((size_t)r->Fr * (nlut + 1));
            for (int f = 0; f < r->Fr; ++f) {
                const double *ro = r->roulette + (size_t)f * d.M;
                const double last = ro[d.M - 1];
                int pos = 0;                                     // thresholds ascend: one merge pass per facet
                for (int k = 0; k < nlut; ++k) {
                    const double thr = ((double)k / (double)nlut) * last;
                    while (pos < d.M && ro[pos] < thr) ++pos;
                    lut[(size_t)f * (nlut + 1) + k] = pos;
                }
                lut[(size_t)f * (nlut + 1) + nlut] = d.M;        // everything is below the end of the last bucket
            }
            NK_UP(lut.data(), lut.size(), &d.roul_lut);
        }
        if (r->degen_j2) NK_UP(r->degen_j2, (size_t)d.M, &d.degen_j2); else d.degen_j2 = nullptr;
        for (int i = 0; i < r->Fr; ++i) {
            NK_ARG(r->facet[i] >= 0 && r->facet[i] < d.Fc, "nk_set_rough: facet index");
            ctx->host_facets[r->facet[i]].rough = i;
        }
        return nk_patch_facets(ctx);
    }
    return NK_OK;
}

int nk_set_params(nk_ctx *ctx, const nk_params *p) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && p, "nk_set_params: NULL argument");
    NK_ARG(p->dt > 0, "nk_set_params: dt must be positive");
    ctx->params = *p;
    NkDev &d = ctx->d;
    d.dt = p->dt; d.inv_dt = 1.0 / p->dt; d.norm_fixed = p->norm_fixed; d.particle_density = p->particle_density;
    d.T_ref_local = p->T_ref_local; d.T_ref = p->T_ref;
    ctx->have_params = true;
    return NK_OK;
}

// Host copy of the live particles, segment by segment (used by download and by re-layouts).
struct NkHostParticles {
    std::vector<double> x, y, z, occ, nts;
    std::vector<int32_t> mode, facet;
    std::vector<uint64_t> pid;
};
// Every kernel but the sweep and the emission expects a segment's particles from slot 0 of its range (NkDev::seg_lo): move them
// down after alternating sweeps, before anything else reads the store.  Needed once per download, regrow, contains_check step
// (every 100th) or flush of the deferred relaxation; the kernel moves only what is not there already.
static int nk_normalize(nk_ctx *ctx, int honor_halt = 0) {
    NkDev &d = ctx->d;
    d.down = 0;
    if (!ctx->walked || !d.seg_lo) return NK_OK;
    k_anchor<<<ctx->num_cu * 8, NK_WG, 0, ctx->stream>>>(d, honor_halt);
    NK_HIP(hipGetLastError());
    if (!honor_halt) ctx->walked = false;             // (inside a batch the launch may have been skipped by a halt: stay cautious)
    return NK_OK;
}
static int nk_gather_live(nk_ctx *ctx, NkHostParticles &h, bool want_all) {
    NkDev &d = ctx->d;
    if (d.cap == 0) return NK_OK;
    { int rcn_ = nk_normalize(ctx); if (rcn_) return rcn_; }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> cnt((size_t)d.nseg);
    NK_HIP(hipMemcpy(cnt.data(), d.seg_count, (size_t)d.nseg * 4, hipMemcpyDeviceToHost));
    int64_t live = 0;
    for (int c : cnt) live += c;
    ctx->h_seg_count = cnt;
    if (!want_all) { h.x.resize((size_t)live); return NK_OK; }
    std::vector<double> bd;
    std::vector<uint32_t> bw;
    std::vector<uint64_t> bu;
    // a field comes back whole (slot order), then the live head of every segment is kept
    auto keep = [&](const auto &all, auto *dst) {
        for (int sgm = 0; sgm < d.nseg; ++sgm) {
            memcpy(dst, all.data() + (size_t)sgm * d.segcap, (size_t)cnt[sgm] * sizeof(*dst));
            dst += cnt[sgm];
        }
    };
    h.x.resize(live); h.y.resize(live); h.z.resize(live); h.occ.resize(live); h.nts.resize(live);
    h.mode.resize(live); h.facet.resize(live); h.pid.assign(live, 0);
    std::vector<uint32_t> w0((size_t)live);
    int rc;
    if ((rc = nk_field_download(ctx, d, d.x, bd))) return rc; keep(bd, h.x.data());
    if ((rc = nk_field_download(ctx, d, d.y, bd))) return rc; keep(bd, h.y.data());
    if ((rc = nk_field_download(ctx, d, d.z, bd))) return rc; keep(bd, h.z.data());
    if ((rc = nk_field_download(ctx, d, d.occ, bd))) return rc; keep(bd, h.occ.data());
    std::vector<int32_t> bfc;
    if (d.nts) { if ((rc = nk_field_download(ctx, d, d.nts, bd))) return rc; keep(bd, h.nts.data()); }
    else {
        // box store: the next hit is not kept -- cast it where the particles stand (k_next_hit; needs the tables)
        if (ctx->have_material && ctx->have_mesh && ctx->have_sv && d.modetab) {
            double *dn = nullptr; int32_t *df = nullptr;
            NK_HIP(hipMalloc((void **)&dn, (size_t)d.cap * 8));
            if (hipMalloc((void **)&df, (size_t)d.cap * 4) != hipSuccess) { hipFree(dn); ctx->err = "nk_gather_live: out of memory"; return NK_ERR_HIP; }
            hipMemsetAsync(dn, 0, (size_t)d.cap * 8, ctx->stream); hipMemsetAsync(df, 0xff, (size_t)d.cap * 4, ctx->stream);
            k_next_hit<1><<<nk_sweep_grid(ctx), NK_WG, nk_lds(ctx, true), ctx->stream>>>(d, dn, df);
            bd.resize((size_t)d.cap); bfc.resize((size_t)d.cap);
            hipError_t e1 = hipGetLastError();
            if (e1 == hipSuccess) e1 = hipMemcpy(bd.data(), dn, (size_t)d.cap * 8, hipMemcpyDeviceToHost);
            if (e1 == hipSuccess) e1 = hipMemcpy(bfc.data(), df, (size_t)d.cap * 4, hipMemcpyDeviceToHost);
            hipFree(dn); hipFree(df);
            if (e1 != hipSuccess) { ctx->err = std::string("nk_gather_live: ") + hipGetErrorString(e1); return NK_ERR_HIP; }
            keep(bd, h.nts.data());
        } else { std::fill(h.nts.begin(), h.nts.end(), 0.0); }
    }
    if ((rc = nk_field_download(ctx, d, d.w0, bw))) return rc; keep(bw, w0.data());
    if (d.pid) { if ((rc = nk_field_download(ctx, d, d.pid, bu))) return rc; keep(bu, h.pid.data()); }
    // packed word -> (mode, facet): mode = idx * nseg + segment when the modes are partitioned
    const uint32_t lbmask = (1u << d.lb) - 1u;
    int64_t k = 0;
    for (int sgm = 0; sgm < d.nseg; ++sgm)
        for (int j = 0; j < cnt[sgm]; ++j, ++k) {
            const uint32_t idx = w0[(size_t)k] & lbmask;
            h.mode[(size_t)k] = d.part ? ctx->h_s2m[(size_t)sgm * d.nlmax + idx] : (int32_t)idx;
            if (d.nts) h.facet[(size_t)k] = (int32_t)(w0[(size_t)k] >> d.lb) - 1;
            else h.facet[(size_t)k] = (w0[(size_t)k] & NK_LOST) ? -1 : (bfc.empty() ? 0 : bfc[(size_t)sgm * d.segcap + j]);
        }
    return NK_OK;
}

// Lay N particles out over the segments.  Partitioned modes: every particle goes to the segment that owns its mode.
// Global mode indices (rough facets): equal shares, sorted by mode (stable), so that a tile shares a few mode records.
static int nk_scatter(nk_ctx *ctx, int64_t N, const double *x, const double *y, const double *z, const int32_t *mode,
                      const double *occ, const double *n_ts, const int32_t *facet, const uint64_t *pid, uint64_t pid_offset) {
    if (ctx) ctx->emitted_for = -1;
    NkDev &d = ctx->d;
    const int M = d.M;
    for (int64_t i = 0; i < N; ++i) NK_ARG(mode[i] >= 0 && mode[i] < M, "nk_upload_particles: mode index out of range");
    std::vector<int64_t> order((size_t)N);
    std::vector<int64_t> start((size_t)d.nseg + 1, 0);
    if (d.part) {
        const std::vector<int32_t> &m2s = ctx->h_m2s;
        for (int64_t i = 0; i < N; ++i) start[(size_t)(m2s[(size_t)mode[i]] % d.nseg) + 1] += 1;
        for (int sgm = 0; sgm < d.nseg; ++sgm) start[(size_t)sgm + 1] += start[(size_t)sgm];
        // inside a segment: by local mode index (stable), so that a tile shares a few records
        std::vector<int64_t> key((size_t)N);
        for (int64_t i = 0; i < N; ++i) { order[(size_t)i] = i; key[(size_t)i] = (int64_t)(m2s[(size_t)mode[i]] % d.nseg) * ((int64_t)d.nlmax + 1) + m2s[(size_t)mode[i]] / d.nseg; }
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
    } else {
        std::vector<int64_t> mstart((size_t)M + 1, 0);               // counting sort by mode
        for (int64_t i = 0; i < N; ++i) mstart[(size_t)mode[i] + 1] += 1;
        for (int m = 0; m < M; ++m) mstart[(size_t)m + 1] += mstart[(size_t)m];
        std::vector<int64_t> cur(mstart.begin(), mstart.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[(size_t)cur[(size_t)mode[i]]++] = i;
        for (int sgm = 0; sgm <= d.nseg; ++sgm) start[(size_t)sgm] = (N * sgm) / d.nseg;
    }
    std::vector<int32_t> cnt((size_t)d.nseg);
    for (int sgm = 0; sgm < d.nseg; ++sgm) {
        cnt[(size_t)sgm] = (int32_t)(start[(size_t)sgm + 1] - start[(size_t)sgm]);
        NK_ARG(cnt[(size_t)sgm] <= d.segcap, "nk_upload_particles: segment capacity too small");
    }
    // a field is laid out whole on the host (slot order: every segment's particles from its start), then uploaded
    auto put = [&](const auto *src, const auto &field) -> int {
        typedef typename std::remove_const<typename std::remove_pointer<decltype(src)>::type>::type T;
        std::vector<T> all((size_t)d.cap, T(0));
        for (int sgm = 0; sgm < d.nseg; ++sgm) {
            const int64_t lo = start[(size_t)sgm], n = cnt[(size_t)sgm];
            T *o = all.data() + (size_t)sgm * d.segcap;
            for (int64_t k = 0; k < n; ++k) o[k] = src[order[(size_t)(lo + k)]];
        }
        return nk_field_upload(ctx, d, field, all.data());
    };
    int rc;
    if ((rc = put(x, d.x)) || (rc = put(y, d.y)) || (rc = put(z, d.z)) || (rc = put(occ, d.occ))) return rc;
    if (n_ts && d.nts && (rc = put(n_ts, d.nts))) return rc;
    {
        std::vector<uint32_t> w0((size_t)N);
        for (int64_t i = 0; i < N; ++i) {
            const int32_t fc = facet ? facet[i] : -1;
            NK_ARG(fc >= -1 && fc < d.Fc, "nk_upload_particles: facet index out of range");
            const uint32_t idx = d.part ? (uint32_t)(ctx->h_m2s[(size_t)mode[i]] / d.nseg) : (uint32_t)mode[i];
            // (box store: the facet is not kept, only whether the last cast missed)
            w0[(size_t)i] = d.nts ? (((uint32_t)(fc + 1) << d.lb) | idx) : (((facet && fc < 0) ? NK_LOST : 0u) | idx);
        }
        if ((rc = put((const uint32_t *)w0.data(), d.w0))) return rc;
    }
    if (d.pid) {
        if (pid) { if ((rc = put(pid, d.pid))) return rc; }
        else {
            std::vector<uint64_t> ids((size_t)N);
            for (int64_t i = 0; i < N; ++i) ids[(size_t)i] = pid_offset + (uint64_t)i;
            if ((rc = put((const uint64_t *)ids.data(), d.pid))) return rc;
        }
    }
    NK_HIP(hipMemcpy(d.seg_count, cnt.data(), (size_t)d.nseg * 4, hipMemcpyHostToDevice));
    if (d.seg_lo) NK_HIP(hipMemset(d.seg_lo, 0, (size_t)d.nseg * 4));      // (the particles were dealt from slot 0 of their segments)
    ctx->walked = false;
    ctx->h_seg_count = cnt;
    return NK_OK;
}

// Does this configuration draw random numbers per particle (then ids are tracked)?  Are the modes partitioned over the
// segments (always, but for a developer probe; with rough facets a particle whose mode changed migrates, k_deliver)?
static inline bool nk_want_pid(const nk_ctx *ctx) { return ctx->d.Fr > 0 || ctx->params.track_ids != 0; }
// Split sweep (events through per-segment queues and k_events): large meshes, whose events are tree walks.  NK_SPLIT=0/1 forces.
static inline bool nk_want_split(const nk_ctx *ctx) {
    if (const char *e = getenv("NK_SPLIT")) return atoi(e) != 0;
    return ctx->have_mesh && !(ctx->d.F <= NK_LDS_FACES && ctx->d.Fc <= NK_LDS_FACES);
}
static inline bool nk_want_part(const nk_ctx *ctx) { return !getenv("NK_NO_PARTITION"); }   // env: developer probe

// Persistent grid of the sweep = what the device keeps resident of the instantiation this configuration uses (before the
// tables are known: four workgroups per CU, the common case).
static int nk_sweep_blocks(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    if (!(ctx->have_material && ctx->have_mesh && ctx->have_sv)) return ctx->num_cu * NK_SWEEP_OCC;
    const int gm_ = nk_geom_mode(ctx);
    const bool rough_ = d.Fr > 0, rbf_ = d.sv_interp == 3, pid_ = nk_want_pid(ctx), split_ = nk_want_split(ctx);
    const bool lrec_ = nk_want_lrec(ctx);
    const int key = gm_ | (rough_ << 2) | (rbf_ << 3) | (pid_ << 4) | (split_ << 5) | (lrec_ << 6) | (nk_sweep_fast(ctx) << 7) | ((d.box ? 1 : 0) << 9);
    const size_t lds_w = nk_lds(ctx, true, pid_ ? 3 : 2);
    if (ctx->g_sweep == 0 || ctx->g_sweep_key != key || ctx->g_sweep_lds != lds_w) {
        int per_cu = 0;
        hipError_t e_ = hipSuccess;
        NK_SWEEP_DISPATCH(gm_, rough_, rbf_, pid_, split_, lrec_, (e_ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, NK_WG, lds_w)));
        if (e_ != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        { const int bound = NK_SWEEP_BOUND(gm_, rough_, rbf_, split_); if (per_cu > bound) per_cu = bound; }   // what it was compiled for
        if (const char *e = getenv("NK_SWEEP_PER_CU")) { int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }   // developer probe
        ctx->g_sweep = ctx->num_cu * per_cu;
        ctx->g_sweep_key = key;
        ctx->g_sweep_lds = lds_w;
        if (getenv("NK_VERBOSE")) fprintf(stderr, "[nanokappa_hip] sweep: %d workgroups per CU (occupancy query rc %d), %zu B LDS each\n", per_cu, (int)e_, lds_w);
    }
    return ctx->g_sweep;
}

// The segment that owns mode m (host copy of the mode map; before a map exists for this segmentation: the plain deal).
static inline int nk_mode_seg(const nk_ctx *ctx, int m, int nseg) {
    return (ctx->map_nseg == nseg && !ctx->h_m2s.empty()) ? ctx->h_m2s[(size_t)m] % nseg : m % nseg;
}

// Deal the modes over the segments (NkDev::m2s / s2m / seg_nl, nk_device.h).  A mode's work per step = its particles (every
// active mode holds the same number) + its boundary events, sum_a |v_a| dt / extent_a of the mesh's bounding box per particle,
// an event weighing NK_EVENT_WEIGHT (0.45: from the stamps' cycles per tile) of a mean particle-step.  In the order of their
// event rate every mode goes to the segment that is furthest behind its SHARE of the work (longest-processing-time greedy).
// The shares follow the dispatch order: the SIMD issues the oldest wave first, so the k-th workgroup a CU received runs ahead
// of the (k+1)-th; share = 1 + NK_AGE_SKEW (0.14; 0.20 until the box sweep lost its general ray cast, profiles/r04_notes.txt (21)) x (1 - 2 k / (per_cu - 1)) -- speed only: if workgroups were dispatched in
// another order the sweep would merely be as uneven as with equal shares.  Modes that cannot move (never populated) fill up
// the shortest segments.  d.nlmax is the capacity of a segment's slot range (set by the caller).
static int nk_build_mode_map(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    if (ctx->m2s_dev) { hipFree(ctx->m2s_dev); ctx->m2s_dev = nullptr; }
    if (ctx->s2m_dev) { hipFree(ctx->s2m_dev); ctx->s2m_dev = nullptr; }
    if (ctx->nl_dev) { hipFree(ctx->nl_dev); ctx->nl_dev = nullptr; }
    d.m2s = nullptr; d.s2m = nullptr; d.seg_nl = nullptr;
    ctx->h_m2s.clear(); ctx->h_s2m.clear(); ctx->map_nseg = 0;
    if (d.M <= 0 || d.nseg <= 0) return NK_OK;
    const int M = d.M, nseg = d.nseg, nlmax = d.nlmax;
    const bool plain = !d.part;          // developer probe without the partition: the modes' emission is dealt m % nseg, forwards
    std::vector<int32_t> nl((size_t)nseg, 0);
    ctx->h_m2s.assign((size_t)M, 0);
    ctx->h_s2m.assign((size_t)nseg * nlmax, -1);
    auto place = [&](int m, int seg) {
        const int l = nl[(size_t)seg]++;
        ctx->h_m2s[(size_t)m] = l * nseg + seg;
        ctx->h_s2m[(size_t)seg * nlmax + l] = m;
    };
    if (plain) {
        for (int m = 0; m < M; ++m) place(m, m % nseg);
    } else {
        std::vector<double> rate((size_t)M, 0.0);
        std::vector<char> active((size_t)M, 1);
        const bool have_v = ctx->h_vg.size() == (size_t)M * 3;
        if (have_v) {
            double inv[3] = {0, 0, 0};
            if (ctx->have_mesh) for (int a = 0; a < 3; ++a) { const double e = d.bbox[3 + a] - d.bbox[a]; inv[a] = e > 0.0 ? 1.0 / e : 0.0; }
            for (int m = 0; m < M; ++m) {
                const double *v = &ctx->h_vg[3 * (size_t)m];
                active[(size_t)m] = (v[0] != 0.0 || v[1] != 0.0 || v[2] != 0.0) ? 1 : 0;
                rate[(size_t)m] = fabs(v[0]) * inv[0] + fabs(v[1]) * inv[1] + fabs(v[2]) * inv[2];
            }
        }
        double rsum = 0.0; int64_t nact = 0;
        for (int m = 0; m < M; ++m) if (active[(size_t)m]) { rsum += rate[(size_t)m]; ++nact; }
        const double rmean = nact > 0 && rsum > 0.0 ? rsum / (double)nact : 1.0;
        const double beta = getenv("NK_EVENT_WEIGHT") ? atof(getenv("NK_EVENT_WEIGHT")) : 0.45;
        // shares by dispatch age: only when every resident wave of the sweep has exactly one segment
        std::vector<double> share((size_t)nseg, 1.0);
        const int g = ctx->g_sweep, per_cu = ctx->num_cu > 0 ? g / ctx->num_cu : 0;
        const double skew = getenv("NK_AGE_SKEW") ? atof(getenv("NK_AGE_SKEW")) : 0.14;
        if (per_cu > 1 && g == per_cu * ctx->num_cu && nseg == g * (NK_WG / 64) && skew != 0.0)
            for (int sg = 0; sg < nseg; ++sg) {
                const int k = (sg / (NK_WG / 64)) / ctx->num_cu;
                share[(size_t)sg] = 1.0 + skew * (1.0 - 2.0 * (double)k / (double)(per_cu - 1));
            }
        std::vector<int32_t> order((size_t)M);
        for (int m = 0; m < M; ++m) order[(size_t)m] = m;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            if (active[(size_t)a] != active[(size_t)b]) return active[(size_t)a] > active[(size_t)b];
            return rate[(size_t)a] > rate[(size_t)b];
        });
        // least relative load first; ties to the lower segment index (deterministic)
        typedef std::pair<double, int> Key;
        std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
        std::vector<double> load((size_t)nseg, 0.0);
        for (int sg = 0; sg < nseg; ++sg) heap.push(Key(0.0, sg));
        int k = 0;
        for (; k < M && active[(size_t)order[(size_t)k]]; ++k) {
            const int m = order[(size_t)k];
            int sg = -1;
            while (!heap.empty()) { const Key top = heap.top(); heap.pop(); if (nl[(size_t)top.second] < nlmax) { sg = top.second; break; } }
            if (sg < 0) { ctx->err = "nk_build_mode_map: the segments' slot ranges are full (internal sizing error)"; return NK_ERR_ARG; }
            place(m, sg);
            load[(size_t)sg] += 1.0 + beta * rate[(size_t)m] / rmean;
            heap.push(Key(load[(size_t)sg] / share[(size_t)sg], sg));
        }
        // the modes that never hold a particle: wherever there is most room
        std::priority_queue<std::pair<int, int>, std::vector<std::pair<int, int>>, std::greater<std::pair<int, int>>> room;
        for (int sg = 0; sg < nseg; ++sg) room.push(std::make_pair((int)nl[(size_t)sg], sg));
        for (; k < M; ++k) {
            const std::pair<int, int> top = room.top(); room.pop();
            if (top.first >= nlmax) { ctx->err = "nk_build_mode_map: no room for the inactive modes (internal sizing error)"; return NK_ERR_ARG; }
            place(order[(size_t)k], top.second);
            room.push(std::make_pair(top.first + 1, top.second));
        }
    }
    NK_HIP(hipMalloc((void **)&ctx->m2s_dev, (size_t)M * 4));
    NK_HIP(hipMalloc((void **)&ctx->s2m_dev, (size_t)nseg * nlmax * 4));
    NK_HIP(hipMalloc((void **)&ctx->nl_dev, (size_t)nseg * 4));
    NK_HIP(hipMemcpy(ctx->m2s_dev, ctx->h_m2s.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->s2m_dev, ctx->h_s2m.data(), (size_t)nseg * nlmax * 4, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->nl_dev, nl.data(), (size_t)nseg * 4, hipMemcpyHostToDevice));
    d.m2s = ctx->m2s_dev; d.s2m = ctx->s2m_dev; d.seg_nl = ctx->nl_dev;
    ctx->map_nseg = nseg;
    return NK_OK;
}

// Upper bound of the particles that can enter one segment in one step (the sweep's halt criterion uses the same sum).
static int64_t nk_spawn_bound(const nk_ctx *ctx, int nseg) {
    const NkDev &d = ctx->d;
    if (d.R <= 0) return 0;
    if (d.res_gen == 2) return 4 * (ctx->o2o_first / std::max(nseg, 1) + 64);
    std::vector<int64_t> per((size_t)nseg, 0);
    for (int r = 0; r < d.R; ++r)
        for (int m = 0; m < d.M; ++m) per[(size_t)(nk_mode_seg(ctx, m, nseg))] += (int64_t)floor(ctx->h_enter_prob[(size_t)r * d.M + m]) + 1;
    int64_t mx = 0;
    for (int64_t v : per) mx = std::max(mx, v);
    return (mx + d.nranks - 1) / d.nranks + d.R;
}

// Where the store's allocation lies in memory changes how fast everything that streams it runs -- the sweep and a plain copy
// alike: three levels, 4.97 / 5.2 / 5.65 TB/s for the in-place copy of a 44-byte store; of 200 successive 833 MB allocations on
// one box 119 were slow, 70 in between and 11 fast, scattered (profiles/r03_notes.txt (9), (17), (26)).  Round 3 searched: the
// store that is about to be used was timed with k_probe_place, then up to 95 further allocations of the same size, all held at
// once, and the fastest kept.  Round 4: the search is OPT-IN (NK_PLACE_TRIES=n > 1; default 1 = the store is timed once, for the
// record, and never moved).  Reasons: with the box store the sweep is bound by its instructions, not its memory -- 0.175-0.178 ms
// on the slowest level against 0.170-0.172 on the fastest (profiles/r04_notes.txt (6)) -- and a search that holds up to a third
// of the free memory, runs on every regrow and on eight ranks at once is no way to buy 3 %.  What the levels are NOT (r04
// notes (7), counters over the same copy on the slowest and the fastest of 40 allocations): not the TLB (UTCL1 misses: 0 on both),
// not the request count or size (identical), not DRAM credit stalls (equal); and with the launches serialised by the profiler
// the two buffers run equally fast -- the difference only exists between back-to-back launches.
// With a search: ends as soon as one candidate is 12 % faster than the slowest (relative: no absolute rate of a particular
// device), never more than a third of the free memory or a second; a probe that fails leaves the store where it is.
static int nk_place_store(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    ctx->timing.place_tries = 0; ctx->timing.place_gbps = 0.0; ctx->timing.place_worst_gbps = 0.0;
    const int tries = getenv("NK_PLACE_TRIES") ? atoi(getenv("NK_PLACE_TRIES")) : 1;
    const size_t bytes = nk_store_bytes(d.cap, ctx->store_pid, ctx->store_nts);
    const int pbytes = nk_particle_bytes(ctx->store_pid, ctx->store_nts);
    // test hooks: NK_PLACE_MIN_MB (stores below it are not timed; default 64), NK_PLACE_FORCE=1 (always move into the last candidate)
    const size_t min_mb = getenv("NK_PLACE_MIN_MB") ? (size_t)atol(getenv("NK_PLACE_MIN_MB")) : 64;
    const bool force = getenv("NK_PLACE_FORCE") != nullptr;
    if (tries < 1 || !ctx->store_buf || ctx->store_pad != 0 || bytes < (min_mb << 20) || d.nseg <= 0 || d.segcap < 64) return NK_OK;
    size_t free_b = 0, total_b = 0;
    NK_HIP(hipMemGetInfo(&free_b, &total_b));
    const auto wall0 = std::chrono::steady_clock::now();
    const int g = std::max(1, std::min(nk_sweep_blocks(ctx), (d.nseg + NK_WG / 64 - 1) / (NK_WG / 64)));
    const int tiles = d.segcap / 64;
    hipEvent_t e0, e1;
    NK_HIP(hipEventCreate(&e0)); NK_HIP(hipEventCreate(&e1));
    auto probe = [&](void *buf, double &ms) -> int {
        NkDev t = d;
        nk_point_fields(t, (double *)buf, d.cap, ctx->store_pid, ctx->store_nts);
        k_probe_place<<<g, NK_WG, 0, ctx->stream>>>(t, tiles);           // untimed: page tables, caches
        NK_HIP(hipEventRecord(e0, ctx->stream));
        for (int k = 0; k < 3; ++k) k_probe_place<<<g, NK_WG, 0, ctx->stream>>>(t, tiles);
        NK_HIP(hipEventRecord(e1, ctx->stream));
        NK_HIP(hipGetLastError());
        NK_HIP(hipStreamSynchronize(ctx->stream));
        float f = 0.f;
        NK_HIP(hipEventElapsedTime(&f, e0, e1));
        ms = (double)f / 3.0;
        return NK_OK;
    };
    std::vector<void *> cand(1, ctx->store_buf);
    std::vector<double> ms(1, 0.0);
    int rc = probe(cand[0], ms[0]);
    while (!rc && (int)cand.size() < tries) {
        double lo = ms[0], hi = ms[0];
        for (double v : ms) { lo = std::min(lo, v); hi = std::max(hi, v); }
        if (hi > 1.12 * lo && !force) break;             // a candidate 12 % faster than the slowest has been seen
        if (bytes * cand.size() > free_b / 3) break;                     // the extra ones: never more than a third of what is free
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count() > 1000.0) break;   // ... nor more than a second (large stores: 0.1 s per allocation)
        void *buf = nullptr;
        if (hipMalloc(&buf, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        cand.push_back(buf); ms.push_back(0.0);
        if (hipMemsetAsync(buf, 0, bytes, ctx->stream) != hipSuccess) { rc = NK_ERR_HIP; ctx->err = "nk_place_store: hipMemsetAsync failed"; break; }
        rc = probe(buf, ms.back());
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    size_t best = 0, worst = 0;
    for (size_t k = 1; k < cand.size(); ++k) { if (ms[k] < ms[best]) best = k; if (ms[k] > ms[worst]) worst = k; }
    if (force) best = cand.size() - 1;
    if (!rc && best != 0 && (ms[best] < 0.97 * ms[0] || force)) {        // move in
        if (hipMemcpyAsync(cand[best], cand[0], bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = NK_ERR_HIP; ctx->err = "nk_place_store: copy into the chosen store failed"; }
        else {
            for (void *&p : ctx->pallocs) if (p == cand[0]) p = cand[best];
            ctx->store_buf = cand[best];
            nk_point_fields(d, (double *)cand[best], d.cap, ctx->store_pid, ctx->store_nts);
            std::swap(cand[0], cand[best]);                              // cand[0] stays, the rest goes
        }
    } else best = 0;
    for (size_t k = 1; k < cand.size(); ++k) hipFree(cand[k]);
    const double gb = 2.0 * (double)d.nseg * tiles * 64 * (pbytes - (ctx->store_pid ? 8 : 0)) / 1e9;      // read + written by one probe launch (it leaves the ids alone)
    ctx->timing.place_tries = (int64_t)cand.size();
    ctx->timing.place_gbps = ms[best] > 0.0 ? gb / (ms[best] * 1e-3) : 0.0;
    ctx->timing.place_worst_gbps = ms[worst] > 0.0 ? gb / (ms[worst] * 1e-3) : 0.0;
    if (rc) {                                            // a failed probe is no reason to fail an allocation or a regrow
        if (getenv("NK_VERBOSE")) fprintf(stderr, "[nanokappa_hip] store placement: probe failed (%s); the store stays where it is\n", ctx->err.c_str());
        (void)hipGetLastError();
        rc = NK_OK;
    }
    if (getenv("NK_VERBOSE")) {
        fprintf(stderr, "[nanokappa_hip] store placement: %zu allocation(s) of %.0f MB probed:", cand.size(), (double)bytes / 1048576.0);
        for (size_t k = 0; k < ms.size(); ++k) fprintf(stderr, " %.1f", ms[k] * 1e3);
        fprintf(stderr, " us per pass; kept number %zu (%.2f TB/s); %.1f ms\n", best, ctx->timing.place_gbps / 1e3,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    }
    return rc;
}

// (Re)allocate the particle store for `capacity` particles, `max_seg` of them at most in one segment when the modes are
// partitioned (0 = unknown: even spread assumed).
// particles of the tiled rule (particle p has mode umodes[p % nu]) per unique mode, for p in [pid_lo, pid_lo + N)
static inline int64_t nk_tiled_count(int64_t u, int64_t nu, int64_t pid_lo, int64_t N) {
    const int64_t first = pid_lo % nu;
    return N / nu + (((u - first + nu) % nu) < N % nu ? 1 : 0);
}
static int nk_alloc_particles(nk_ctx *ctx, int64_t capacity, const int32_t *mode = nullptr, int64_t N = 0,
                              const int32_t *umodes = nullptr, int64_t nu = 0, int64_t pid_lo = 0) {
    if (ctx) ctx->emitted_for = -1;
    NkDev &d = ctx->d;
    { int rc0 = nk_entry_tables_drop(ctx, true); if (rc0) return rc0; }
    for (void *p : ctx->pallocs) hipFree(p);
    ctx->pallocs.clear();
    // Segments = load-balance granularity of the persistent sweep.  For large ensembles exactly ONE segment per resident wave
    // (long segments amortise the per-segment costs: first-tile latency, the partial last event batch and spawn tile, and
    // a whole number per wave keeps the waves level).  Small ensembles: more, shorter segments, down to 128 slots, until
    // every resident wave has one -- a wave's serial chain (tiles, event passes, entering particles) is what a small
    // sweep waits for.
    d.part = nk_want_part(ctx) ? 1 : 0;
    d.box = nk_want_box(ctx) ? 1 : 0;
    d.nlrec = 0;
    int64_t nseg = 0;
    // the sweep's LDS (hence its residency, hence the segment count) depends on the record area, which depends on the segment
    // count: settle in a few rounds
    for (int round = 0; round < 4; ++round) {
        const int64_t waves = (int64_t)nk_sweep_blocks(ctx) * (NK_WG / 64);
        int64_t ns = capacity / 128;
        if (const char *e = getenv("NK_SEGMENTS")) ns = atoi(e) > 0 ? atoi(e) : ns;          // developer probe
        else if (ns >= waves) ns = waves;
        ns = ns < 64 ? 64 : (ns > NK_MAX_SEGMENTS ? NK_MAX_SEGMENTS : ns);
        const bool same = ns == nseg;
        nseg = ns;
        d.nseg = (int32_t)nseg;
        // slots per segment: the mean share of the modes plus head room for the uneven shares of nk_build_mode_map
        d.nlmax = d.M > 0 ? (int32_t)((d.M + nseg - 1) / nseg) : 1;
        if (d.part && d.M > nseg) d.nlmax += d.nlmax / 6 + 2;
        // the segments' mode records in LDS where a segment's share fits and the LDS they take costs no resident workgroup
        d.nlrec = 0;
        if (d.part && d.nlmax <= NK_LREC && !getenv("NK_NO_LREC")) {
            const int without = nk_sweep_blocks(ctx);
            d.nlrec = d.nlmax;
            if (nk_sweep_blocks(ctx) < without) d.nlrec = 0;
        }
        if (same) break;
    }
    NK_ARG(d.nlmax < (1 << 14), "too many modes per segment for k_emit's packed entry word: use more particles (segments) or fewer modes");
    { int rcm = nk_build_mode_map(ctx); if (rcm) return rcm; }
    {   // bits of the stored mode index; the rest of the 32-bit word holds facet + 1
        const int64_t maxidx = d.part ? d.nlmax - 1 : (d.M > 0 ? d.M - 1 : 0);
        int lb = 1;
        while ((1ll << lb) <= maxidx) ++lb;
        NK_ARG(lb < 30 && (int64_t)d.Fc + 1 < (1ll << (31 - lb)), "mesh has too many facets for the packed particle word (facets x modes)");
        d.lb = lb;
    }
    int64_t segcap = (capacity + nseg - 1) / nseg;
    if (d.part && (mode || umodes) && N > 0) {           // room for the fullest segment of this population + a step's arrivals
        std::vector<int64_t> per((size_t)nseg, 0);
        if (mode) { for (int64_t i = 0; i < N; ++i) if (mode[i] >= 0 && mode[i] < d.M) per[(size_t)nk_mode_seg(ctx, mode[i], (int)nseg)] += 1; }
        else for (int64_t u = 0; u < nu; ++u) per[(size_t)nk_mode_seg(ctx, umodes[u], (int)nseg)] += nk_tiled_count(u, nu, pid_lo, N);
        int64_t mx = 0;
        for (int64_t v : per) mx = std::max(mx, v);
        segcap = std::max(segcap, mx + mx / 4 + 64);
    }
    const bool tight = getenv("NK_TIGHT_STORE") != nullptr;      // test hook: barely enough room, so that a growing ensemble halts soon
    if (ctx->have_material && d.R > 0 && !ctx->h_enter_prob.empty())
        segcap = (tight ? segcap : std::max(segcap, segcap / 8 * 9)) + (tight ? 1 : 2) * nk_spawn_bound(ctx, (int)nseg) + 2 * NK_TILE;
    segcap = ((segcap + 63) / 64) * 64;
    NK_ARG(nseg * segcap < (1ll << 31), "particle store too large for 32-bit slot arithmetic");
    d.segcap = (int32_t)segcap;
    d.cap = nseg * segcap;
    ctx->timing.slots = d.cap;
    const double *pd; const uint32_t *pw; const int32_t *pi; const uint64_t *pu;
#define NK_PALLOC(T, field, ptr, count)                                                                \
    do { int rc_ = nk_upload<T>(ctx, nullptr, (size_t)(count), &ptr, true); if (rc_) return rc_; d.field = (T *)ptr; } while (0)
    { int rcf = nk_alloc_fields(ctx, d, d.cap, nk_want_pid(ctx), !d.box); if (rcf) return rcf; }
    NK_PALLOC(int32_t, seg_count, pi, d.nseg);
    NK_PALLOC(int32_t, seg_new, pi, d.nseg);
    NK_PALLOC(int32_t, seg_bound, pi, d.nseg);
    d.seg_lo = nullptr; d.down = 0; ctx->walked = false;
    NK_PALLOC(int32_t, seg_lo, pi, d.nseg);                 // (zeroed: the particles start at slot 0 of their segments)
    d.qx = d.qy = d.qz = d.qocc = d.qnts = nullptr; d.qw0 = nullptr; d.qpid = nullptr; d.seg_evq = nullptr;
    if (nk_want_split(ctx)) {
        NK_PALLOC(double, qx, pd, d.cap); NK_PALLOC(double, qy, pd, d.cap); NK_PALLOC(double, qz, pd, d.cap);
        NK_PALLOC(double, qocc, pd, d.cap); NK_PALLOC(double, qnts, pd, d.cap);
        NK_PALLOC(uint32_t, qw0, pw, d.cap);
        if (d.pid) NK_PALLOC(uint64_t, qpid, pu, d.cap);
        NK_PALLOC(int32_t, seg_evq, pi, 2 * (size_t)d.nseg + 1);
    }
#ifdef NK_STAMPS
    { const unsigned long long *ps; int rc_ = nk_upload<unsigned long long>(ctx, nullptr, (size_t)d.nseg * 16, &ps, true); if (rc_) return rc_; d.stamps = (unsigned long long *)ps; }
#endif
#undef NK_PALLOC
    ctx->layout_key = (d.part ? 1 : 0) | (d.pid ? 2 : 0) | (d.qx ? 4 : 0) | (d.box ? 8 : 0);
    // tables that follow the segmentation: permuted mode records, 'one_to_one' inboxes
    if (ctx->have_material) { int rc = nk_update_tau_window(ctx, true); if (rc) return rc; }
    if (ctx->inbox) { hipFree(ctx->inbox); ctx->inbox = nullptr; }
    if (ctx->inbox_n) { hipFree(ctx->inbox_n); ctx->inbox_n = nullptr; }
    d.sp_inbox = nullptr; d.sp_inbox_n = nullptr; d.sp_icap = 0;
    if (ctx->mig_buf) { hipFree(ctx->mig_buf); ctx->mig_buf = nullptr; }
    if (ctx->mig_n) { hipFree(ctx->mig_n); ctx->mig_n = nullptr; }
    d.mig_buf = nullptr; d.mig_n = nullptr; d.mig_cap = 0;
    { int rcp = nk_place_store(ctx); if (rcp) return rcp; }
    return nk_entry_tables_build(ctx);
}

// Rough facets with partitioned modes: per-segment inboxes of 64-byte records for the particles that change segment
// (`cap` records each; 0 = half a segment).  The inboxes are empty between steps, so resizing them loses nothing.
static int nk_ensure_migration(nk_ctx *ctx, int64_t cap) {
    NkDev &d = ctx->d;
    if (!(d.Fr > 0 && d.part)) return NK_OK;
    if (cap <= 0) cap = std::max<int64_t>(256, d.segcap / 2);
    if (d.mig_buf && d.mig_cap >= cap) return NK_OK;
    if (ctx->mig_buf) { hipFree(ctx->mig_buf); ctx->mig_buf = nullptr; }
    if (ctx->mig_n) { hipFree(ctx->mig_n); ctx->mig_n = nullptr; }
    d.mig_buf = nullptr; d.mig_n = nullptr; d.mig_cap = 0;
    NK_HIP(hipMalloc(&ctx->mig_buf, (size_t)d.nseg * cap * 64));
    NK_HIP(hipMalloc(&ctx->mig_n, (size_t)d.nseg * 4));
    NK_HIP(hipMemset(ctx->mig_n, 0, (size_t)d.nseg * 4));
    d.mig_buf = (double2 *)ctx->mig_buf; d.mig_n = (int32_t *)ctx->mig_n; d.mig_cap = (int32_t)cap;
    return NK_OK;
}

// 'one_to_one': per-segment inboxes for the records of k_emit_one_to_one
static int nk_ensure_inbox(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    if (!(d.R > 0 && d.res_gen == 2) || d.sp_inbox) return NK_OK;
    const int64_t icap = 4 * (ctx->o2o_first / d.nseg + 64);
    NK_HIP(hipMalloc(&ctx->inbox, (size_t)d.nseg * icap * 8));
    NK_HIP(hipMalloc(&ctx->inbox_n, (size_t)d.nseg * 4));
    NK_HIP(hipMemset(ctx->inbox_n, 0, (size_t)d.nseg * 4));
    d.sp_inbox = (uint64_t *)ctx->inbox; d.sp_inbox_n = (int32_t *)ctx->inbox_n; d.sp_icap = (int32_t)icap;
    return NK_OK;
}

// Grow every segment to `segcap_new` slots on the device (same segmentation: the particles keep their segments).
static int nk_regrow(nk_ctx *ctx, int64_t segcap_new) {
    if (ctx) ctx->emitted_for = -1;
    { int rcn_ = nk_normalize(ctx); if (rcn_) return rcn_; }
    NkDev &d = ctx->d;
    NK_HIP(hipStreamSynchronize(ctx->stream));
    const NkDev old = d;
    std::vector<void *> old_allocs;
    old_allocs.swap(ctx->pallocs);
    void *const old_store = ctx->store_buf; const size_t old_pad = ctx->store_pad;
    segcap_new = ((segcap_new + 63) / 64) * 64;
    NK_ARG((int64_t)old.nseg * segcap_new < (1ll << 31), "particle store too large for 32-bit slot arithmetic");
    d.segcap = (int32_t)segcap_new;
    d.cap = (int64_t)d.nseg * d.segcap;
    const double *pd; const uint32_t *pw; const int32_t *pi; const uint64_t *pu;
    int rc = NK_OK;
#define NK_PALLOC(T, field, ptr, count)                                                                \
    do { if (!rc) { rc = nk_upload<T>(ctx, nullptr, (size_t)(count), &ptr, true); if (!rc) d.field = (T *)ptr; } } while (0)
    if (!rc) rc = nk_alloc_fields(ctx, d, d.cap, (bool)old.pid, (bool)old.nts);
    NK_PALLOC(int32_t, seg_count, pi, d.nseg);
    NK_PALLOC(int32_t, seg_new, pi, d.nseg);
    NK_PALLOC(int32_t, seg_bound, pi, d.nseg);
    if (old.seg_lo) NK_PALLOC(int32_t, seg_lo, pi, d.nseg);   // (the old store was moved down above; k_regrow copies from slot 0)
    if (old.qx) {                                       // the event queues are empty between steps: new ones, nothing to copy
        NK_PALLOC(double, qx, pd, d.cap); NK_PALLOC(double, qy, pd, d.cap); NK_PALLOC(double, qz, pd, d.cap);
        NK_PALLOC(double, qocc, pd, d.cap); NK_PALLOC(double, qnts, pd, d.cap);
        NK_PALLOC(uint32_t, qw0, pw, d.cap);
        if (old.pid) NK_PALLOC(uint64_t, qpid, pu, d.cap);
        NK_PALLOC(int32_t, seg_evq, pi, 2 * (size_t)d.nseg + 1);
    }
#undef NK_PALLOC
    if (rc) {                                            // out of memory: keep the old store
        for (void *p : ctx->pallocs) hipFree(p);
        ctx->pallocs.swap(old_allocs);
        ctx->store_buf = old_store; ctx->store_pad = old_pad;
        d = old;
        return rc;
    }
    ctx->timing.regrows += 1;
    k_regrow<<<ctx->num_cu * 8, NK_WG, 0, ctx->stream>>>(old, d);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    for (void *p : old_allocs) hipFree(p);
    ctx->timing.slots = d.cap;
    return nk_place_store(ctx);       // (the old store's memory is free again: often the better place)
}

int nk_reserve(nk_ctx *ctx, int64_t capacity) {
    NK_ARG(ctx, "nk_reserve: NULL context");
    NK_ARG(capacity > 0 && capacity < (1ll << 31) - (1 << 20), "nk_reserve: capacity out of range");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    if (capacity <= d.cap) return NK_OK;
    if (d.cap == 0) return nk_alloc_particles(ctx, capacity);
    // Same number of segments before and after (every large ensemble: one segment per resident wave): the segments grow in
    // place on the device.  Otherwise the particles are re-dealt through the host.
    int64_t nseg_new = capacity / 128;
    const int64_t waves = (int64_t)nk_sweep_blocks(ctx) * (NK_WG / 64);
    if (nseg_new >= waves) nseg_new = waves;
    nseg_new = nseg_new < 64 ? 64 : (nseg_new > NK_MAX_SEGMENTS ? NK_MAX_SEGMENTS : nseg_new);
    if (nseg_new == d.nseg || getenv("NK_SEGMENTS")) return nk_regrow(ctx, (capacity + d.nseg - 1) / d.nseg);
    NkHostParticles h;
    int rc = nk_gather_live(ctx, h, true);
    if (rc) return rc;
    rc = nk_alloc_particles(ctx, capacity, h.mode.data(), (int64_t)h.mode.size());
    if (rc) return rc;
    if (!h.x.empty())
        return nk_scatter(ctx, (int64_t)h.x.size(), h.x.data(), h.y.data(), h.z.data(), h.mode.data(), h.occ.data(),
                          h.nts.data(), h.facet.data(), h.pid.data(), 0);
    return NK_OK;
}

int nk_upload_particles(nk_ctx *ctx, int64_t N, const double *x, const double *y, const double *z, const int32_t *mode,
                        const double *occ, const double *n_ts, const int32_t *facet, const uint64_t *pid,
                        uint64_t pid_offset) {
    NK_ARG(ctx && N >= 0, "nk_upload_particles: bad arguments");
    NK_ARG(N == 0 || (x && y && z && mode && occ), "nk_upload_particles: x, y, z, mode, occ are required");
    NK_ARG(ctx->have_material, "nk_upload_particles: call nk_set_material first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (n_ts && facet && ctx->mesh_box && !ctx->box_forbidden) {
        // a state handed over WITH its cached hits: a box store re-derives them from the positions, which is the same thing
        // only for particles inside the box (nk_init_boundaries has the other half of this rule)
        const double *k = d.box_k;
        for (int64_t i = 0; i < N && !ctx->box_forbidden; ++i)
            if (facet[i] >= 0 && !(x[i] >= k[0] && x[i] <= -k[1] && y[i] >= k[2] && y[i] <= -k[3] && z[i] >= k[4] && z[i] <= -k[5])) ctx->box_forbidden = true;
    }
    bool fits = d.cap > 0 && ctx->layout_key == ((nk_want_part(ctx) ? 1 : 0) | (nk_want_pid(ctx) ? 2 : 0) | (nk_want_split(ctx) ? 4 : 0) | (nk_want_box(ctx) ? 8 : 0)) && N + N / 5 + 1024 <= d.cap;
    if (fits && d.part) {                              // every segment must hold its modes' particles with head room
        std::vector<int64_t> per((size_t)d.nseg, 0);
        for (int64_t i = 0; i < N; ++i) if (mode[i] >= 0 && mode[i] < d.M) per[(size_t)nk_mode_seg(ctx, mode[i], d.nseg)] += 1;
        const int64_t room = (int64_t)d.segcap - nk_spawn_bound(ctx, d.nseg) - 2 * NK_TILE;
        for (int64_t v : per) fits = fits && v + v / 16 <= room;
    }
    if (!fits) {                                       // forget old contents and size for 1.5 N
        const int64_t want = getenv("NK_TIGHT_STORE") ? N + N / 8 + 1024 : N + N / 2 + 65536;
        int rc = nk_alloc_particles(ctx, std::max(want, d.cap), mode, N);
        if (rc) return rc;
    }
    int rc = nk_scatter(ctx, N, x, y, z, mode, occ, n_ts, facet, pid, pid_offset);
    if (rc) return rc;
    int32_t zero6[6] = {0, 0, 0, 0, 0, 0};
    NK_HIP(hipMemcpy(d.halt, zero6, 24, hipMemcpyHostToDevice));     // halt[4], overflow, ticket
    ctx->pending_relax = false;
    return NK_OK;
}

static int nk_check_ready(nk_ctx *ctx) {
    NK_ARG(ctx->have_material && ctx->have_mesh && ctx->have_sv && ctx->have_params,
           "engine not configured: need material, mesh, subvolumes and params");
    NkDev &d = ctx->d;
    // every rough / reservoir facet must be backed by its table, otherwise a kernel would index garbage
    for (int f = 0; f < d.Fc; ++f) {
        const NkFacet &hf = ctx->host_facets[f];
        NK_ARG(hf.bc != 'R' || hf.rough >= 0, "a facet has BC 'R' but nk_set_rough did not cover it");
        NK_ARG(!(hf.bc == 'T' || hf.bc == 'F') || hf.res >= 0, "a facet has BC 'T' but nk_set_reservoirs did not cover it");
    }
    NK_ARG(d.cap > 0, "no particle storage: call nk_reserve / nk_upload_particles");
    NK_ARG(nk_lds(ctx, true, 3) <= 160 * 1024 && nk_lds(ctx, true, 1) <= 160 * 1024, "tables do not fit the 160 KiB LDS");
    // the store's layout follows the configuration (ids, partitioned modes): tables set after the upload re-deal it
    const int want = (nk_want_part(ctx) ? 1 : 0) | (nk_want_pid(ctx) ? 2 : 0) | (nk_want_split(ctx) ? 4 : 0) | (nk_want_box(ctx) ? 8 : 0);
    if (ctx->layout_key != want) {
        NkHostParticles h;
        int rc = nk_gather_live(ctx, h, true);
        if (rc) return rc;
        const bool had_pid = (bool)d.pid;
        const int64_t cap_old = d.cap;
        if ((rc = nk_alloc_particles(ctx, cap_old, h.mode.data(), (int64_t)h.mode.size()))) return rc;
        if (!h.x.empty() && (rc = nk_scatter(ctx, (int64_t)h.x.size(), h.x.data(), h.y.data(), h.z.data(), h.mode.data(), h.occ.data(),
                                             h.nts.data(), h.facet.data(), had_pid ? h.pid.data() : nullptr, 0)))
            return rc;
    }
    return NK_OK;
}

// Mesh._count_crossings of the host geometry on a device (no context needed): see k_mesh_crossings.
int nk_mesh_crossings(int device, int64_t n_rays, const double *origins, const double *dirs, int64_t n_faces, const double *v0,
                      const double *e1, const double *e2, int skip_self, int32_t *counts) {
    if (n_rays <= 0 || n_faces <= 0 || !origins || !dirs || !v0 || !e1 || !e2 || !counts) return NK_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return NK_ERR_HIP;
    double *buf = nullptr;
    int32_t *dc = nullptr;
    const size_t nr = (size_t)n_rays * 3, nf = (size_t)n_faces * 3;
    if (hipMalloc((void **)&buf, (2 * nr + 3 * nf) * sizeof(double)) != hipSuccess) return NK_ERR_HIP;
    if (hipMalloc((void **)&dc, (size_t)n_rays * sizeof(int32_t)) != hipSuccess) { hipFree(buf); return NK_ERR_HIP; }
    double *po = buf, *pd = po + nr, *p0 = pd + nr, *p1 = p0 + nf, *p2 = p1 + nf;
    hipError_t e = hipMemcpy(po, origins, nr * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pd, dirs, nr * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p0, v0, nf * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p1, e1, nf * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p2, e2, nf * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_mesh_crossings<<<(int)((n_rays + NK_WG - 1) / NK_WG), NK_WG>>>(n_rays, po, pd, n_faces, p0, p1, p2, skip_self, dc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(counts, dc, (size_t)n_rays * sizeof(int32_t), hipMemcpyDeviceToHost);
    hipFree(buf); hipFree(dc);
    return e == hipSuccess ? NK_OK : NK_ERR_HIP;
}

int nk_init_boundaries(nk_ctx *ctx) {
    NK_ARG(ctx, "nk_init_boundaries: NULL context");
    int rc = nk_check_ready(ctx);
    if (rc) return rc;
    NK_HIP(hipSetDevice(ctx->device));
    if ((rc = nk_normalize(ctx))) return rc;
    for (int pass = 0; pass < 2; ++pass) {
        NK_HIP(hipMemsetAsync(ctx->anomalies, 0, 4, ctx->stream));
        NK_GEOM_LAUNCH(k_init_boundaries, nk_sweep_grid(ctx), nk_lds(ctx, true), ctx->d, ctx->anomalies);
        NK_HIP(hipGetLastError());
        int32_t bad = 0;
        NK_HIP(hipMemcpyAsync(&bad, ctx->anomalies, 4, hipMemcpyDeviceToHost, ctx->stream));
        NK_HIP(hipStreamSynchronize(ctx->stream));
        if (!(ctx->d.box && bad > 0)) break;
        // Particles that start OUTSIDE the box and would meet a wall from behind (a resumed state with particles behind their
        // reservoir face, test ensembles planted outside): the reference runs their event when they reach that wall, a store
        // without cached hits would let them fly in.  This context keeps the cached layout from here on.
        ctx->box_forbidden = true;
        if ((rc = nk_check_ready(ctx))) return rc;       // re-deals the store (layout key changed)
    }
    return NK_OK;
}

static int nk_flush_relax(nk_ctx *ctx, int honor_halt) {
    if (!ctx->pending_relax) return NK_OK;
    { int rcn_ = nk_normalize(ctx); if (rcn_) return rcn_; }
    k_relax<<<nk_sweep_grid(ctx), NK_WG, nk_lds(ctx, false), ctx->stream>>>(ctx->d, honor_halt);
    NK_HIP(hipGetLastError());
    ctx->pending_relax = false;
    return NK_OK;
}

// Enqueue up to `nsteps` timesteps without host synchronisation, drain the stream, copy the history rows back.  A sweep that
// sees a segment which COULD overflow at the following step raises the halt word; the remaining steps of the batch then do
// nothing, *done < nsteps comes back, and nk_step grows the store (state intact, nothing dropped) and carries on.
static int nk_step_batch(nk_ctx *ctx, int32_t nsteps, std::vector<double> &h, int32_t *done) {
    NkDev &d = ctx->d;
    const int S = d.S, R = d.R, NB = d.NB;
    const int HROW = NB + 2 * S + 8;
    if (nsteps > ctx->hist_cap) {
        if (ctx->hist) hipHostFree(ctx->hist);
        ctx->hist = nullptr;
        const int rows_alloc = nsteps < 1024 ? 1024 : nsteps;   // generous: a later, longer call must not pay a realloc
        // The history rows live in pinned HOST memory that the device writes directly (1.1 KB per step over PCIe by the one
        // workgroup of the update): no copy back, no fill kernel -- a driver that steps one by one pays for every operation of a
        // call (round 3: fill + two copies = a tenth of such a call).
        NK_HIP(hipHostMalloc((void **)&ctx->hist, (size_t)rows_alloc * HROW * sizeof(double), hipHostMallocMapped));
        ctx->hist_cap = rows_alloc;
    }
    ctx->timing.batches += 1;
    memset(ctx->hist, 0, (size_t)nsteps * HROW * sizeof(double));   // row_valid = 0 (host memory; the stream is idle between calls)
    const size_t lds_g = nk_lds(ctx, true), lds_w = nk_lds(ctx, true, d.pid ? 3 : 2), lds_e = nk_lds(ctx, true, 1);
    const int gm_ = nk_geom_mode(ctx);
    const bool rough_ = d.Fr > 0, rbf_ = d.sv_interp == 3, pid_ = (bool)d.pid, split_ = d.qx != nullptr;
    const bool lrec_ = nk_want_lrec(ctx);
    (void)nk_sweep_blocks(ctx);
    const int g_sweep = ctx->g_sweep < (d.nseg + 3) / 4 ? ctx->g_sweep : (d.nseg + 3) / 4;
    const int g_emit = ctx->num_cu * 8 < (d.nseg + 3) / 4 ? ctx->num_cu * 8 : (d.nseg + 3) / 4;
    const int g_ev = split_ ? ctx->num_cu * NK_EV_PER_CU : 0;    // k_events: resident waves drawing from all queues
    // k_events keeps the top of the face tree in LDS: as many whole levels as fit beside its tables at NK_EVENTS_OCC workgroups per CU
    // (160 KB per CU; NK_EVENTS_TREE_LDS = bytes to use at most, 0 = none: developer probe)
    size_t lds_ev = lds_g;
    d.tree_lds_fam0 = d.tree_nfam; d.tree_lds_off = 0;
    if (split_ && gm_ == 2 && d.NG > 0 && d.tree_nfam > 0) {
        const size_t off = (lds_g + 15) & ~(size_t)15, fixed = off + (size_t)NK_EVENTS_PEND * NK_EV_WG * 4 + 1024;
        size_t budget = (size_t)160 * 1024 / NK_EV_PER_CU > fixed ? (size_t)160 * 1024 / NK_EV_PER_CU - fixed : 0;
        if (const char *e = getenv("NK_EVENTS_TREE_LDS")) { const size_t v = (size_t)atol(e); if (v < budget) budget = v; }
        for (int l = 0; l <= d.tree_top; ++l) {
            const int fam0 = d.tree_base[l] >> 2;
            const size_t bytes = (size_t)(d.tree_nfam - fam0) * NK_TREE_FAMILY_FLOATS * 4;
            if (bytes <= budget) { d.tree_lds_fam0 = fam0; d.tree_lds_off = (int32_t)off; lds_ev = off + bytes; break; }
        }
        if (getenv("NK_VERBOSE") && ctx->timing.batches <= 1)
            fprintf(stderr, "[nanokappa_hip] k_events: families %d .. %d of the face tree in LDS (%zu B of %zu B per workgroup)\n", d.tree_lds_fam0, d.tree_nfam, lds_ev - off, lds_ev);
    }
    const int rows = g_sweep + g_ev;
    // per-kernel timing on the first 4 steps of a batch; none for the short calls of a driver that steps one by one (six event
    // records are a tenth of such a call).  Every record is a marker packet between two dependent kernels: measured ~2 us each
    // on the stream -- with records on 16 of a region's 20 steps (round 3) a step took 0.213 ms, in 100-step batches 0.206.
    const int nev = nsteps < 4 ? 0 : 4;
    if (ctx->evpool.empty()) {                           // events are created once and reused
        ctx->evpool.resize(16 * 4 + 2);
        for (auto &e : ctx->evpool) NK_HIP(hipEventCreate(&e));
    }
    hipEvent_t *ev = ctx->evpool.data();
    hipEvent_t t0 = ctx->evpool[64], t1 = ctx->evpool[65];
    const bool relax0 = ctx->pending_relax;
    NK_HIP(hipEventRecord(t0, ctx->stream));
    bool pending = ctx->pending_relax;
    std::vector<char> relax_flushed((size_t)nsteps, 0);       // steps whose deferred relaxation k_relax ran before them
    // The next step's emission rides in the same launch as this step's reduce / update (k_tail) wherever it depends on neither
    // (no rough facets, not 'one_to_one').  (On a second stream instead, the two cross-stream event waits per step cost more
    // than the 12 us they hid: 0.289 against 0.270 ms per step on config 2, profiles/r03_notes.txt (11).)
    const bool tail_emit = R > 0 && d.res_gen != 2 && !getenv("NK_NO_TAIL_EMIT");
    const size_t lds_t = lds_e > (size_t)(NK_WG * 8 + 16) ? lds_e : (size_t)(NK_WG * 8 + 16);
    bool emitted_ahead = tail_emit && ctx->emitted_for == ctx->step;   // this step's emission already ran in the previous step's k_tail (maybe of the call before)
    ctx->emitted_for = -1;
    ctx->timing.emit_fused = tail_emit ? 1 : 0;
    // the fused sweeps of small meshes (box store or cached store) alternate between walking their segments upwards and downwards
    // (NkDev::down; nk_device.h), on every rank alike (the direction follows the step's parity).  Not the split sweep of large meshes:
    // measured no gain there (config 4's store is ten times the memory-side cache).  NK_NO_ALTERNATE=1: never.
    const bool alt = d.seg_lo && gm_ == 1 && !split_ && !getenv("NK_NO_ALTERNATE");
    if (!alt) { int rcn_ = nk_normalize(ctx); if (rcn_) return rcn_; }
    d.down = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int64_t stepno = ctx->step + s;
        const uint32_t step = (uint32_t)stepno;
        if (ctx->params.contains_every > 0 && (stepno % ctx->params.contains_every) == 0 && d.nS > 0) {
            { int rcn_ = nk_normalize(ctx, 1); if (rcn_) return rcn_; }       // (k_relax, k_contains: particles from slot 0)
            if (pending) {
                k_relax<<<nk_sweep_grid(ctx), NK_WG, nk_lds(ctx, false), ctx->stream>>>(d, 1);
                pending = false;
                relax_flushed[(size_t)s] = 1;
            }
            NK_GEOM_LAUNCH(k_contains, nk_sweep_grid(ctx), lds_g, d, step);
        }
        const int fe = ctx->params.flux_every;
        const int do_flux = (fe > 0 && ((stepno + 1) % fe) == 0) ? 1 : 0;
        if (s < nev) NK_HIP(hipEventRecord(ev[4 * s], ctx->stream));
        if (R > 0 && d.res_gen == 2) k_emit_one_to_one<<<ctx->num_cu * 4, NK_WG, 0, ctx->stream>>>(d, step);
        if (R > 0 && !emitted_ahead) NK_EMIT_LAUNCH(k_emit, g_emit, lds_e, d, step);
        if (s < nev) NK_HIP(hipEventRecord(ev[4 * s + 1], ctx->stream));
        {
            const int rl = pending ? 1 : 0;
            if (alt) { d.down = (int32_t)(stepno & 1); ctx->walked = true; }
            NK_SWEEP_DISPATCH(gm_, rough_, rbf_, pid_, split_, lrec_, (KERNEL<<<g_sweep, NK_WG, lds_w, ctx->stream>>>(d, step, rl, do_flux)));
            d.down = 0;
            if (split_) {
                k_events_begin<<<1, 1024, 0, ctx->stream>>>(d);
                const int64_t ev_key = (int64_t)lds_ev * 64 + (gm_ | (rough_ << 2) | (rbf_ << 3) | (pid_ << 4));
                if (ctx->ev_lds_set != ev_key) {        // more than the default 64 KB of dynamic LDS has to be asked for, per kernel
                    hipError_t ea_ = hipSuccess;
                    NK_EVENTS_DISPATCH(gm_, rough_, rbf_, pid_, (ea_ = hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ev)));
                    NK_HIP(ea_);
                    ctx->ev_lds_set = ev_key;
                }
                NK_EVENTS_DISPATCH(gm_, rough_, rbf_, pid_, (KERNEL<<<g_ev, NK_EV_WG, lds_ev, ctx->stream>>>(d, step, do_flux, g_sweep)));
                k_events_end<<<(d.nseg + 255) / 256, 256, 0, ctx->stream>>>(d);
            }
        }
        // rough facets: the migrants of this step go to their segments before the tallies are reduced (and before the next step's
        // emission, in the tail launch, appends behind them)
        if (d.mig_buf) k_deliver<<<g_emit, NK_WG, 0, ctx->stream>>>(d, 1);
        if (s < nev) NK_HIP(hipEventRecord(ev[4 * s + 2], ctx->stream));
        double *hrow = ctx->hist + (size_t)s * HROW;
        const bool ahead = tail_emit;                          // the next step's emission beside this step's tail (the last step's too: for the next call)
        if (ctx->comm) {
            if (ahead) NK_EMIT_LAUNCH(k_tail, NB + g_emit, lds_t, d, step + 1u, rows, ctx->acc, hrow, do_flux, 0, NB);
            else k_reduce<<<NB, NK_WG, 0, ctx->stream>>>(d, rows, ctx->acc, hrow, do_flux, 0);
            ncclResult_t nrc = ctx->rccl.AllReduce(ctx->acc, ctx->acc, (size_t)NB + 2, ncclDouble, ncclSum, ctx->comm, ctx->stream);
            if (nrc != ncclSuccess) { ctx->err = "ncclAllReduce failed"; return NK_ERR_COMM; }
            k_update<<<1, NK_WG, 0, ctx->stream>>>(d, ctx->acc, hrow, do_flux);
        } else if (ahead) {
            NK_EMIT_LAUNCH(k_tail, NB + g_emit, lds_t, d, step + 1u, rows, ctx->acc, hrow, do_flux, 1, NB);
        } else {
            k_reduce<<<NB, NK_WG, 0, ctx->stream>>>(d, rows, ctx->acc, hrow, do_flux, 1);
        }
        emitted_ahead = ahead;
        if (s < nev) NK_HIP(hipEventRecord(ev[4 * s + 3], ctx->stream));
        pending = true;
        if ((s & 63) == 0) NK_HIP(hipGetLastError());   // a bad launch configuration shows at the first step of a batch (every step launches the same)
    }
    NK_HIP(hipEventRecord(t1, ctx->stream));
    NK_HIP(hipGetLastError());
    // the history rows and the halt words come back through pinned memory, enqueued behind the steps: ONE wait per call (a
    // Population.run_timestep is one such call per step -- the reference driver's granularity, nanokappa.py:91-98)
    const size_t hbytes = (size_t)nsteps * HROW * sizeof(double);
    // a call of a few steps (a driver that steps one by one) is over in a fraction of a millisecond: poll instead of sleeping on the
    // stream -- the wake-up of a blocked wait is a tenth of such a call
    if (nsteps <= 4) { hipError_t q_; while ((q_ = hipStreamQuery(ctx->stream)) == hipErrorNotReady) { } if (q_ != hipSuccess) { ctx->err = std::string("hipStreamQuery: ") + hipGetErrorString(q_); return NK_ERR_HIP; } }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    h.resize((size_t)nsteps * HROW);
    memcpy(h.data(), ctx->hist, hbytes);
    {   // the halt words as the last step that ran left them (its update wrote them behind the row); no row: ask the device
        int last_ = -1;
        for (int q_ = 0; q_ < nsteps && h[(size_t)q_ * HROW + NB + 2 * S + 1] != 0.0; ++q_) last_ = q_;
        if (last_ >= 0) for (int k_ = 0; k_ < 4; ++k_) ctx->halt_words[k_] = (int32_t)h[(size_t)last_ * HROW + NB + 2 * S + 4 + k_];
        else NK_HIP(hipMemcpy(ctx->halt_words, d.halt, 16, hipMemcpyDeviceToHost));
    }
#ifdef NK_STAMPS
    if (d.stamps) {                                     // developer build: section shares of the LAST sweep of the batch
        std::vector<unsigned long long> st((size_t)d.nseg * 16);
        NK_HIP(hipMemcpy(st.data(), d.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double sum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int sgm = 0; sgm < d.nseg; ++sgm) for (int k = 0; k < 7; ++k) sum[k] += (double)st[(size_t)sgm * 8 + k];
        double tot = 0; for (int k = 0; k < 6; ++k) tot += sum[k];
        {   // in-kernel shader clock of the sweep (s_memtime against the 100 MHz s_memrealtime, per segment; median)
            std::vector<unsigned long long> clk;
            for (int sgm = 0; sgm < d.nseg; ++sgm) clk.push_back(st[(size_t)sgm * 8 + 7]);
            std::sort(clk.begin(), clk.end());
            // wall-clock marks (10 ns ticks of the 100 MHz counter): where the launch's time goes outside the tile loops
            unsigned long long e0 = ~0ull, x1 = 0;
            std::vector<double> pro, loop, epi, fin;
            for (int sgm = 0; sgm < d.nseg; ++sgm) {
                const unsigned long long *w = &st[((size_t)d.nseg + sgm) * 8];
                if (!w[0] || !w[3]) continue;
                e0 = std::min(e0, w[0]); x1 = std::max(x1, w[3]);
            }
            for (int sgm = 0; sgm < d.nseg; ++sgm) {
                const unsigned long long *w = &st[((size_t)d.nseg + sgm) * 8];
                if (!w[0] || !w[3]) continue;
                pro.push_back((double)(w[1] - w[0]) * 0.01); loop.push_back((double)(w[2] - w[1]) * 0.01);
                epi.push_back((double)(w[3] - w[2]) * 0.01); fin.push_back((double)(w[3] - e0) * 0.01);
            }
            auto q = [](std::vector<double> v, const char *nm) {
                if (v.empty()) return;
                std::sort(v.begin(), v.end());
                double sm = 0; for (double x : v) sm += x;
                fprintf(stderr, "[stamps] %s [us]: min %.1f  p10 %.1f  median %.1f  mean %.1f  p90 %.1f  max %.1f\n", nm, v.front(), v[v.size() / 10], v[v.size() / 2], sm / v.size(), v[v.size() * 9 / 10], v.back());
            };
            {   // k_emit's marks of the same step (words 4-7 of the second row): entry, tables in LDS, first entries evaluated, done
                unsigned long long m0 = ~0ull, m3 = 0;
                std::vector<double> a, b, c, fin2;
                for (int sgm = 0; sgm < d.nseg; ++sgm) { const unsigned long long *w = &st[((size_t)d.nseg + sgm) * 8 + 4]; if (w[0] && w[3]) { m0 = std::min(m0, w[0]); m3 = std::max(m3, w[3]); } }
                for (int sgm = 0; sgm < d.nseg; ++sgm) {
                    const unsigned long long *w = &st[((size_t)d.nseg + sgm) * 8 + 4];
                    if (!w[0] || !w[3]) continue;
                    a.push_back((double)(w[1] - w[0]) * 0.01); b.push_back((double)(w[2] - w[1]) * 0.01); c.push_back((double)(w[3] - w[2]) * 0.01);
                    fin2.push_back((double)(w[3] - m0) * 0.01);
                }
                auto q2 = [](std::vector<double> v, const char *nm) {
                    if (v.empty()) return;
                    std::sort(v.begin(), v.end());
                    double sm = 0; for (double x : v) sm += x;
                    fprintf(stderr, "[stamps] k_emit %s [us]: min %.1f  median %.1f  mean %.1f  p90 %.1f  max %.1f\n", nm, v.front(), v[v.size() / 2], sm / v.size(), v[v.size() * 9 / 10], v.back());
                };
                if (!a.empty()) {
                    fprintf(stderr, "[stamps] k_emit first entry -> last wave done: %.1f us\n", (double)(m3 - m0) * 0.01);
                    q2(a, "entry -> tables in LDS"); q2(b, "-> first chunk of entries evaluated"); q2(c, "-> particles built and stored"); q2(fin2, "wave done, after the first entry");
                }
            }
            fprintf(stderr, "[stamps] first entry -> last exit: %.1f us over %zu waves\n", (double)(x1 - e0) * 0.01, pro.size());
            {   // where do the slow waves sit?  mean tile-loop time by workgroup index mod 8 (the XCD under round-robin placement),
                // by wave within the workgroup, and by thirds of the grid
                double xs[8] = {0}, xn[8] = {0}, ws[4] = {0}, wn[4] = {0}, gs[4] = {0}, gn[4] = {0}, rs[8] = {0}, rn[8] = {0};
                for (int sgm = 0; sgm < d.nseg; ++sgm) {
                    const unsigned long long *w = &st[((size_t)d.nseg + sgm) * 8];
                    if (!w[0] || !w[3]) continue;
                    const double t = (double)(w[2] - w[1]) * 0.01;
                    const int wg = sgm / 4;
                    xs[wg % 8] += t; xn[wg % 8] += 1; ws[sgm % 4] += t; wn[sgm % 4] += 1;
                    const int third = (int)((int64_t)sgm * 4 / d.nseg); gs[third] += t; gn[third] += 1;
                    const int rk = wg / ctx->num_cu; if (rk < 8) { rs[rk] += t; rn[rk] += 1; }
                }
                fprintf(stderr, "[stamps] tile loop mean [us] by workgroup %% 8:");
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %.1f", xn[k] ? xs[k] / xn[k] : 0.0);
                fprintf(stderr, " | by wave of the workgroup:");
                for (int k = 0; k < 4; ++k) fprintf(stderr, " %.1f", wn[k] ? ws[k] / wn[k] : 0.0);
                fprintf(stderr, " | by quarter of the grid:");
                for (int k = 0; k < 4; ++k) fprintf(stderr, " %.1f", gn[k] ? gs[k] / gn[k] : 0.0);
                fprintf(stderr, " | by workgroup / CUs (the k-th workgroup of a CU under in-order dispatch):");
                for (int k = 0; k < 8; ++k) if (rn[k]) fprintf(stderr, " %.1f", rs[k] / rn[k]);
                fprintf(stderr, "\n");
            }
            q(pro, "entry -> tile loop (tables into LDS, records)"); q(loop, "tile loop"); q(epi, "tile loop end -> wave through (flush, workgroup barrier, tally row)");
            q(fin, "wave through, after the first entry");
            fprintf(stderr, "[stamps] shader clock inside k_sweep: median %.0f MHz (min %.0f, max %.0f)\n", clk[clk.size() / 2] / 10.0, clk.front() / 10.0, clk.back() / 10.0);
        }
        fprintf(stderr, "[stamps] cycles per tile: arrive %.0f  relax+drift %.0f  tally+store %.0f  pack/merge %.0f  event %.0f  event tally/store/repack %.0f  | total %.0f (tiles %.0f)\n",
                sum[0] / sum[6], sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6], sum[4] / sum[6], sum[5] / sum[6], tot / sum[6], sum[6]);
        if (d.qx) {                                     // split sweep: k_events' per-segment clocks (words 3-5)
            std::vector<double> cyc, wk, qq, ps;
            for (int sgm = 0; sgm < d.nseg; ++sgm) {
                cyc.push_back((double)st[(size_t)sgm * 8 + 3]); wk.push_back((double)st[(size_t)sgm * 8 + 4]);
                qq.push_back((double)(st[(size_t)sgm * 8 + 5] >> 32)); ps.push_back((double)(st[(size_t)sgm * 8 + 5] & 0xffffffffull));
            }
            auto stat = [](std::vector<double> v, const char *nm) {
                std::sort(v.begin(), v.end());
                double sm = 0; for (double x : v) sm += x;
                fprintf(stderr, "[stamps] k_events per segment %s: min %.0f  median %.0f  mean %.0f  p99 %.0f  max %.0f\n", nm, v.front(), v[v.size() / 2], sm / v.size(), v[(size_t)(v.size() * 0.99)], v.back());
            };
            stat(cyc, "cycles"); stat(wk, "cycles in walks"); stat(qq, "queue entries"); stat(ps, "walk passes");
            std::vector<double> lc, ln;
            for (int sgm = 0; sgm < d.nseg; ++sgm) { lc.push_back((double)(st[(size_t)sgm * 8 + 6] & ((1ull << 40) - 1))); ln.push_back((double)(st[(size_t)sgm * 8 + 6] >> 40)); }
            stat(lc, "cycles in faces passes"); stat(ln, "faces passes");
        }
    }
#endif
    int32_t nd = 0;
    while (nd < nsteps && h[(size_t)nd * HROW + NB + 2 * S + 1] != 0.0) ++nd;
    *done = nd;
    {   // the emission that ran ahead in the last tail is good for the next call unless the batch halted (the store grows first)
        int32_t hw_[4];
        memcpy(hw_, ctx->halt_words, 16);
        ctx->emitted_for = (tail_emit && nd == nsteps && !hw_[0] && !hw_[2] && !hw_[3]) ? ctx->step + nsteps : -1;
    }
    // the deferred relaxation as the device left it: pending after any completed step; if none ran, whatever it was before,
    // unless a k_relax ahead of the first step flushed it (that kernel honours the halt word, which was clear then)
    if (nd > 0) ctx->pending_relax = true;
    else ctx->pending_relax = relax0 && !(nsteps > 0 && relax_flushed[0]);
    float ms = 0.f;
    double sk = 0.0, ek = 0.0, vk = 0.0;
    const int nt = nd < nev ? nd : nev;
    for (int s = 0; s < nt; ++s) {
        NK_HIP(hipEventElapsedTime(&ms, ev[4 * s], ev[4 * s + 1])); ek += ms;
        NK_HIP(hipEventElapsedTime(&ms, ev[4 * s + 1], ev[4 * s + 2])); sk += ms;
        NK_HIP(hipEventElapsedTime(&ms, ev[4 * s + 2], ev[4 * s + 3])); vk += ms;
    }
    NK_HIP(hipEventElapsedTime(&ms, t0, t1));
    if (nt > 0 && nd == nsteps) {
        ctx->timing.step_kernel_ms = sk / nt;
        ctx->timing.emit_kernel_ms = ek / nt;
        ctx->timing.events_kernel_ms = vk / nt;
        ctx->timing.total_ms = ms;
    }
    return NK_OK;
}

// Small ensembles: many steps per launch (k_resident, nk_kernels.h).  OPT-IN (NK_RESIDENT=1): measured SLOWER than the launches
// it replaces -- 1e5 particles: 0.081 (9^3 x 6 modes) / 0.092 ms (31^3 x 6) per step against 0.032 / 0.037 ms; 1e6: 0.130-0.136
// against 0.046-0.050 (profiles/r04_notes.txt (9)): a step's device-scope barrier and its FP64 atomics on 111 shared addresses
// cost more than the two launches and the reduce chain they save.  One rank, no rough facets, tables in LDS, not 'one_to_one', no
// RBF temperatures, at most NK_RESIDENT_MAX slots (default 1.2e6).  Kept, with its parity tests, as the measured record of
// that design.
static inline bool nk_want_resident(const nk_ctx *ctx) {
    const NkDev &d = ctx->d;
    if (!getenv("NK_RESIDENT") || getenv("NK_NO_RESIDENT") || ctx->comm || d.nranks != 1 || d.Fr > 0 || d.mig_buf || d.qx || nk_geom_mode(ctx) != 1) return false;
    if (d.res_gen == 2 || d.sv_interp == 3 || d.NB > 254 || d.S > 128 || d.nseg <= 0) return false;
    const int64_t lim = getenv("NK_RESIDENT_MAX") ? atoll(getenv("NK_RESIDENT_MAX")) : 1200000;
    return d.cap <= lim && nk_lds(ctx, true, 5) <= 160 * 1024;
}
#define NK_RESIDENT_CASE(B, P, lrec, STMT) { if (lrec) { auto KERNEL = k_resident<B, P, true>; STMT; } else { auto KERNEL = k_resident<B, P, false>; STMT; } }
#define NK_RESIDENT_DISPATCH(box, pid, lrec, STMT)                                                    \
    do {                                                                                               \
        if (box) { if (pid) NK_RESIDENT_CASE(true, true, lrec, STMT) else NK_RESIDENT_CASE(true, false, lrec, STMT) }   \
        else { if (pid) NK_RESIDENT_CASE(false, true, lrec, STMT) else NK_RESIDENT_CASE(false, false, lrec, STMT) }      \
    } while (0)
static int nk_step_resident(nk_ctx *ctx, int32_t nsteps, std::vector<double> &h, int32_t *done) {
    NkDev &d = ctx->d;
    { int rcn_ = nk_normalize(ctx); if (rcn_) return rcn_; }
    const int S = d.S, NB = d.NB;
    const int HROW = NB + 2 * S + 8;
    if (nsteps > ctx->hist_cap) {
        if (ctx->hist) hipHostFree(ctx->hist);
        ctx->hist = nullptr;
        const int rows_alloc = nsteps < 1024 ? 1024 : nsteps;
        NK_HIP(hipHostMalloc((void **)&ctx->hist, (size_t)rows_alloc * HROW * sizeof(double), hipHostMallocMapped));
        ctx->hist_cap = rows_alloc;
    }
    ctx->timing.batches += 1;
    ctx->emitted_for = -1;
    memset(ctx->hist, 0, (size_t)nsteps * HROW * sizeof(double));
    (void)nk_sweep_blocks(ctx);
    const bool pid_ = (bool)d.pid, lrec_ = nk_want_lrec(ctx), box_ = d.box != 0;
    const size_t lds = nk_lds(ctx, true, pid_ ? 5 : 4);
    // every workgroup must be resident for the grid barrier: no more than the device holds at once (occupancy query), and no
    // more than there are segments
    if (lds > 64 * 1024) {                                        // more dynamic LDS than the default limit of a launch
        hipError_t ea = hipSuccess;
        NK_RESIDENT_DISPATCH(box_, pid_, lrec_, (ea = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
        if (ea != hipSuccess) (void)hipGetLastError();            // (the launch itself will say if it does not fit)
    }
    int per_cu = 1;
    {
        hipError_t eo = hipSuccess;
        NK_RESIDENT_DISPATCH(box_, pid_, lrec_, (eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, NK_WG, lds)));
        if (eo != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
        if (per_cu > 2) per_cu = 2;
    }
    int G = std::min(ctx->num_cu * per_cu, (d.nseg + NK_WG / 64 - 1) / (NK_WG / 64));
    if (const char *e = getenv("NK_RESIDENT_GRID")) { const int v = atoi(e); if (v >= 1 && v < G) G = v; }   // developer probe
    G = std::max(1, G);
    if (ctx->evpool.empty()) { ctx->evpool.resize(16 * 4 + 2); for (auto &e : ctx->evpool) NK_HIP(hipEventCreate(&e)); }
    hipEvent_t t0 = ctx->evpool[64], t1 = ctx->evpool[65];
    const bool relax0 = ctx->pending_relax;
    bool pending = ctx->pending_relax, flushed_first = false;
    ctx->timing.emit_fused = 2;                                   // (2: emission inside the resident kernel)
    if (!ctx->rbar) NK_HIP(hipMalloc((void **)&ctx->rbar, (32 + 4096) * sizeof(unsigned int)));
    NK_HIP(hipEventRecord(t0, ctx->stream));
    const int ce = ctx->params.contains_every;
    int launches = 0;
    for (int k = 0; k < nsteps;) {
        const int64_t stepno = ctx->step + k;
        if (ce > 0 && (stepno % ce) == 0 && d.nS > 0) {
            if (pending) {
                k_relax<<<nk_sweep_grid(ctx), NK_WG, nk_lds(ctx, false), ctx->stream>>>(d, 1);
                pending = false;
                if (k == 0) flushed_first = true;
            }
            NK_GEOM_LAUNCH(k_contains, nk_sweep_grid(ctx), nk_lds(ctx, true), d, (uint32_t)stepno);
        }
        int n = nsteps - k;
        if (ce > 0 && d.nS > 0) n = std::min<int64_t>(n, ce - (stepno % ce));
        NK_HIP(hipMemsetAsync(ctx->rbar, 0, (32 + 1024 + 128) * sizeof(unsigned int), ctx->stream));     // the flags count from 1 in every launch
        NK_RESIDENT_DISPATCH(box_, pid_, lrec_, (KERNEL<<<G, NK_WG, lds, ctx->stream>>>(d, (uint32_t)stepno, n, pending ? 1 : 0, ctx->params.flux_every,
                                                                                       ctx->hist + (size_t)k * HROW, HROW, ctx->rbar)));
        NK_HIP(hipGetLastError());
        pending = true;
        k += n;
        ++launches;
    }
    NK_HIP(hipEventRecord(t1, ctx->stream));
    const size_t hbytes = (size_t)nsteps * HROW * sizeof(double);
    NK_HIP(hipStreamSynchronize(ctx->stream));
    h.resize((size_t)nsteps * HROW);
    memcpy(h.data(), ctx->hist, hbytes);
    {   // the halt words as the last step that ran left them (its update wrote them behind the row); no row: ask the device
        int last_ = -1;
        for (int q_ = 0; q_ < nsteps && h[(size_t)q_ * HROW + NB + 2 * S + 1] != 0.0; ++q_) last_ = q_;
        if (last_ >= 0) for (int k_ = 0; k_ < 4; ++k_) ctx->halt_words[k_] = (int32_t)h[(size_t)last_ * HROW + NB + 2 * S + 4 + k_];
        else NK_HIP(hipMemcpy(ctx->halt_words, d.halt, 16, hipMemcpyDeviceToHost));
    }
    int32_t nd = 0;
    while (nd < nsteps && h[(size_t)nd * HROW + NB + 2 * S + 1] != 0.0) ++nd;
    *done = nd;
    if (getenv("NK_VERBOSE") && ctx->timing.batches % 8 == 2) {   // developer probe: the clock marks of the launch's last step (10 ns ticks)
        unsigned long long m[12];
        NK_HIP(hipMemcpy(m, ctx->rbar + 8, sizeof(m), hipMemcpyDeviceToHost));
        if (m[0] && m[6] >= m[0])
            fprintf(stderr, "[nanokappa_hip] resident step, workgroup 0 [us]: sweep %.2f  row + flag %.2f  emission %.2f  all flags seen %.2f  rows summed %.2f  update published %.2f | workgroup G/2: sweep %.2f  row + emission %.2f  update seen %.2f\n",
                    (m[1] - m[0]) * 0.01, (m[2] - m[1]) * 0.01, (m[3] - m[2]) * 0.01, (m[4] - m[3]) * 0.01, (m[5] - m[4]) * 0.01, (m[6] - m[5]) * 0.01,
                    (m[9] - m[8]) * 0.01, (m[10] - m[9]) * 0.01, (m[11] - m[10]) * 0.01);
    }
    if (nd > 0) ctx->pending_relax = true;
    else ctx->pending_relax = relax0 && !flushed_first;
    float ms = 0.f;
    NK_HIP(hipEventElapsedTime(&ms, t0, t1));
    if (nd == nsteps && nd > 0) {
        ctx->timing.step_kernel_ms = (double)ms / nd;              // the resident kernel IS the step
        ctx->timing.emit_kernel_ms = 0.0;
        ctx->timing.events_kernel_ms = 0.0;
        ctx->timing.total_ms = ms;
    }
    (void)launches;
    return NK_OK;
}

int nk_step(nk_ctx *ctx, int32_t nsteps, nk_tally *out) {
    NK_ARG(ctx && nsteps > 0, "nk_step: bad arguments");
    int rc = nk_check_ready(ctx);
    if (rc) return rc;
    NK_HIP(hipSetDevice(ctx->device));
    if ((rc = nk_update_tau_window(ctx, false))) return rc;
    if ((rc = nk_ensure_inbox(ctx))) return rc;
    if ((rc = nk_ensure_migration(ctx, 0))) return rc;
    NkDev &d = ctx->d;
    const int S = d.S, R = d.R, NB = d.NB;
    const int HROW = NB + 2 * S + 8;
    ctx->stepped = true;
    int32_t s_out = 0;                 // rows delivered so far
    int overflow = 0;      // reason mask: 2 / 4 a segment filled up (tile commit / event survivors), 8 one_to_one inbox full,
                           // 16 one_to_one index overflow
    int grown = 0;
    std::vector<double> h;
    double last_T[2] = {0, 0};
    while (s_out < nsteps) {
        int32_t nd = 0;
        if ((rc = nk_want_resident(ctx) ? nk_step_resident(ctx, nsteps - s_out, h, &nd) : nk_step_batch(ctx, nsteps - s_out, h, &nd))) return rc;
        for (int s = 0; s < nd; ++s) {
            const double *row = &h[(size_t)s * HROW];
            if (row[NB + 2 * S + 3] != 0.0) overflow |= (int)row[NB + 2 * S + 3];
            if (!out) continue;
            const size_t q = (size_t)(s_out + s);
            if (out->E_raw) memcpy(out->E_raw + q * S, row, S * 8);
            if (out->N_sv) memcpy(out->N_sv + q * S, row + S, S * 8);
            if (out->flux_raw) {
                if (row[NB + 2 * S] != 0.0) memcpy(out->flux_raw + q * 3 * S, row + 2 * S, 3 * S * 8);
                else for (int k = 0; k < 3 * S; ++k) out->flux_raw[q * 3 * S + k] = NAN;
            }
            if (out->N_leaving && R) memcpy(out->N_leaving + q * R, row + 5 * S, R * 8);
            if (out->res_energy && R) memcpy(out->res_energy + q * R, row + 5 * S + R, R * 8);
            if (out->res_flux && R) memcpy(out->res_flux + q * 3 * R, row + 5 * S + 2 * R, 3 * R * 8);
            if (out->N_emitted) out->N_emitted[q] = row[NB - 1];
            if (out->T_sv) memcpy(out->T_sv + q * S, row + NB, S * 8);
            if (out->E_sv) memcpy(out->E_sv + q * S, row + NB + S, S * 8);
        }
        if (nd > 0) {
            const double *last = &h[(size_t)(nd - 1) * HROW];
            nk_track_T(ctx, last + NB, S);
            double live = 0.0;
            for (int k = 0; k < S; ++k) live += last[S + k];
            ctx->timing.live = (int64_t)live;
            (void)last_T;
        }
        ctx->step += nd;
        s_out += nd;
        int32_t hw[4];
        memcpy(hw, ctx->halt_words, 16);                 // as the batch left them (copied with the history rows)
        if (s_out < nsteps || hw[0] || hw[2] || hw[3]) {
            // halted: a segment could overflow at the next step (or could not take its migrants, which then wait in its
            // inbox).  Grow every segment by half (on the device, state intact), deliver, and carry on.
            ctx->timing.halts += 1;
            if (++grown > 40) { ctx->err = "nk_step: the particle store keeps filling up"; return NK_ERR_CAPACITY; }
            const bool only_inbox = hw[3] && !hw[0] && !hw[2] && s_out == nsteps;
            if (!only_inbox) {
                const int64_t need = (int64_t)d.segcap + d.segcap / 2 + 2 * nk_spawn_bound(ctx, d.nseg) + 2 * NK_TILE;
                if ((rc = nk_regrow(ctx, need))) {
                    ctx->err = "particle store nearly full after step " + std::to_string((long long)ctx->step) +
                               " and it could not be grown (" + ctx->err + "); the state is intact";
                    return NK_ERR_CAPACITY;
                }
            }
            int32_t zero4[4] = {0, 0, 0, 0};
            NK_HIP(hipMemcpy(d.halt, zero4, 16, hipMemcpyHostToDevice));
            // (hw[2] comes out of the all-reduced vector: with several ranks every one of them is here at the same step, has
            // grown its segments by the same amount, and delivers whatever waits in its own inboxes)
            if (hw[2]) {                                 // migrants that did not fit: they do now
                k_deliver<<<ctx->num_cu * 8, NK_WG, 0, ctx->stream>>>(d, 0);
                NK_HIP(hipGetLastError());
                NK_HIP(hipStreamSynchronize(ctx->stream));
                int32_t again[4];
                NK_HIP(hipMemcpy(again, d.halt, 16, hipMemcpyDeviceToHost));
                if (again[2]) { ctx->err = "nk_step: migrating particles do not fit their segment after growing it"; return NK_ERR_CAPACITY; }
                NK_HIP(hipMemcpy(d.halt, zero4, 16, hipMemcpyHostToDevice));
            }
            if (d.mig_buf && (hw[3] || !only_inbox)) {   // inboxes follow the segments (or double when they ran half full)
                const int64_t want = std::max<int64_t>(hw[3] ? 2 * (int64_t)d.mig_cap : 0, d.segcap / 2);
                if ((rc = nk_ensure_migration(ctx, want))) return rc;
            }
            if ((rc = nk_update_tau_window(ctx, false))) return rc;
        }
    }
    ctx->timing.slots = d.cap;
    if (overflow & 256) {                                // k_resident: its grid barrier was not met (a workgroup was not resident?)
        uint32_t z2[2] = {0u, 0u};
        NK_HIP(hipMemcpy(ctx->anomalies + 1, z2, 8, hipMemcpyHostToDevice));
        int32_t z = 0;
        NK_HIP(hipMemcpy(d.overflow, &z, 4, hipMemcpyHostToDevice));
        ctx->err = "nk_step: the resident kernel's grid barrier timed out; set NK_NO_RESIDENT=1";
        return NK_ERR_HIP;
    }
    if (overflow) {
        ctx->err = "particle capacity exceeded during nk_step (reason mask " + std::to_string(overflow) +
                   "): particles were dropped; call nk_reserve with a larger capacity";
        return NK_ERR_CAPACITY;
    }
    return NK_OK;
}

int nk_download_particles(nk_ctx *ctx, int64_t capacity, double *x, double *y, double *z, int32_t *mode, double *occ,
                          double *n_ts, int32_t *facet, uint64_t *pid, int64_t *N_out) {
    NK_ARG(ctx && N_out, "nk_download_particles: NULL argument");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    *N_out = 0;
    if (d.cap == 0) return NK_OK;
    if (ctx->have_material && ctx->have_sv && ctx->have_mesh) { int rc = nk_flush_relax(ctx, 0); if (rc) return rc; }
    NkHostParticles h;
    int rc = nk_gather_live(ctx, h, capacity != 0);
    if (rc) return rc;
    const int64_t live = (int64_t)h.x.size();
    *N_out = live;
    if (capacity == 0) return NK_OK;
    NK_ARG(capacity >= live, "nk_download_particles: capacity smaller than the live particle count");
    if (x) memcpy(x, h.x.data(), live * 8);
    if (y) memcpy(y, h.y.data(), live * 8);
    if (z) memcpy(z, h.z.data(), live * 8);
    if (occ) memcpy(occ, h.occ.data(), live * 8);
    if (n_ts) memcpy(n_ts, h.nts.data(), live * 8);
    if (mode) memcpy(mode, h.mode.data(), live * 4);
    if (facet) memcpy(facet, h.facet.data(), live * 4);
    if (pid) memcpy(pid, h.pid.data(), live * 8);
    return NK_OK;
}

int nk_get_subvol_temperature(nk_ctx *ctx, double *T_sv) {
    NK_ARG(ctx && T_sv && ctx->have_sv, "nk_get_subvol_temperature: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(hipMemcpy(T_sv, ctx->d.T_sv, (size_t)ctx->d.S * 8, hipMemcpyDeviceToHost));
    return NK_OK;
}
int nk_set_subvol_temperature(nk_ctx *ctx, const double *T_sv) {
    NK_ARG(ctx && T_sv && ctx->have_sv, "nk_set_subvol_temperature: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<double> t(T_sv, T_sv + ctx->d.S);
    if (ctx->d.sv_interp == 3) nk_rbf_coefficients(ctx, t);
    NK_HIP(hipMemcpy(ctx->d.T_sv, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    nk_track_T(ctx, T_sv, ctx->d.S);
    return NK_OK;
}
int nk_get_step(nk_ctx *ctx, int64_t *step) {
    NK_ARG(ctx && step, "nk_get_step: NULL argument");
    *step = ctx->step;
    return NK_OK;
}
int nk_get_timing(nk_ctx *ctx, nk_timing *t) {
    NK_ARG(ctx && t, "nk_get_timing: NULL argument");
    *t = ctx->timing;
    t->box_store = (ctx->d.cap > 0 && ctx->d.box) ? 1 : 0;
    return NK_OK;
}

// ------------------------------------------------------------------------------------------ RCCL
static int nk_load_rccl(NkRccl &r, std::string &err) {
    if (r.lib) return NK_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) { err = std::string("cannot load librccl: ") + dlerror(); return NK_ERR_COMM; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");         // (optional)
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.CommCount || !r.CommUserRank) { err = "librccl lacks expected symbols"; return NK_ERR_COMM; }
    return NK_OK;
}
int nk_comm_unique_id(void *id128) {
    static NkRccl r;
    std::string err;
    if (!id128) return NK_ERR_ARG;
    if (nk_load_rccl(r, err)) { g_create_error = err; return NK_ERR_COMM; }
    ncclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (r.GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return NK_ERR_COMM; }
    memcpy(id128, &id, 128);
    return NK_OK;
}
int nk_comm_init(nk_ctx *ctx, const void *id128, int rank, int nranks) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "nk_comm_init: bad arguments");
    NK_ARG(!ctx->stepped, "nk_comm_init: must be called before the first nk_step (emission ownership is per rank)");
    NK_HIP(hipSetDevice(ctx->device));
    ctx->d.rank = rank;
    ctx->d.nranks = nranks;
    if (nranks == 1 && !getenv("NK_FORCE_COMM")) return NK_OK;   // NK_FORCE_COMM: exercise RCCL with a 1-rank communicator
    if (getenv("NK_COMM_DRYRUN")) return NK_OK;   // test hook: this rank's share of the emission, no communicator (tallies stay local)
    if (nk_load_rccl(ctx->rccl, ctx->err)) return NK_ERR_COMM;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    memcpy(&id, id128, 128);
    // every way out of here that is not NK_OK leaves the context as a single rank without communicator (a failed
    // self-test used to leave ctx->comm set and d.nranks = nranks behind)
    auto fail = [&](const std::string &why, bool hung) -> int {
        ctx->err = why;
        if (ctx->comm) {
            if (hung && ctx->rccl.CommAbort) ctx->rccl.CommAbort(ctx->comm);       // peers missing: a destroy would wait for them
            else if (!hung && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
        }
        ctx->comm = nullptr; ctx->comm_rank = -1; ctx->comm_nranks = 0; ctx->comm_selftest = 0.0;
        ctx->d.rank = 0; ctx->d.nranks = 1;
        return NK_ERR_COMM;
    };
    if (ctx->rccl.CommInitRank(&ctx->comm, nranks, id, rank) != ncclSuccess) { ctx->comm = nullptr; return fail("ncclCommInitRank failed", false); }
    // The communicator proves itself before anything relies on it: RCCL's own view of the rank count, then an all-reduce of
    // {1, rank + 1} on the engine's stream -- every expected rank took part exactly once iff the sums are nranks and
    // nranks (nranks + 1) / 2.
    int cn = 0, cr = -1;
    if (ctx->rccl.CommCount(ctx->comm, &cn) != ncclSuccess || ctx->rccl.CommUserRank(ctx->comm, &cr) != ncclSuccess) {
        return fail("ncclCommCount / ncclCommUserRank failed", false);
    }
    ctx->comm_nranks = cn; ctx->comm_rank = cr;
    if (cn != nranks || cr != rank) {
        return fail("RCCL reports rank " + std::to_string(cr) + " of " + std::to_string(cn) + ", expected rank " + std::to_string(rank) + " of " + std::to_string(nranks), false);
    }
    double *probe = nullptr, hp[2] = {1.0, (double)(rank + 1)};
    double *hpin = nullptr;                              // pinned: the copy back must really be asynchronous for the deadline below
    if (hipMalloc((void **)&probe, 16) != hipSuccess || hipHostMalloc((void **)&hpin, 16, hipHostMallocDefault) != hipSuccess) {
        if (probe) hipFree(probe);
        return fail("self-test of the new communicator: out of memory", false);
    }
    hpin[0] = hp[0]; hpin[1] = hp[1];
    hipError_t he = hipMemcpyAsync(probe, hpin, 16, hipMemcpyHostToDevice, ctx->stream);
    ncclResult_t nrc = he == hipSuccess ? ctx->rccl.AllReduce(probe, probe, 2, ncclDouble, ncclSum, ctx->comm, ctx->stream) : ncclSystemError;
    if (he == hipSuccess && nrc == ncclSuccess) he = hipMemcpyAsync(hpin, probe, 16, hipMemcpyDeviceToHost, ctx->stream);
    bool hung = false;
    if (he == hipSuccess && nrc == ncclSuccess) {
        // a rank that never joins would leave this all-reduce waiting for ever: poll with a deadline (NK_COMM_TIMEOUT_S, 180 s)
        const double limit = getenv("NK_COMM_TIMEOUT_S") ? atof(getenv("NK_COMM_TIMEOUT_S")) : 180.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            he = hipStreamQuery(ctx->stream);
            if (he != hipErrorNotReady) break;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) { hung = true; break; }
            struct timespec ts = {0, 200000};
            nanosleep(&ts, nullptr);
        }
    }
    if (hung) return fail("self-test all-reduce of the new communicator did not finish in time: a rank is missing", true);   // (probe stays allocated: the stream may still touch it)
    hp[0] = hpin[0]; hp[1] = hpin[1];
    hipFree(probe); hipHostFree(hpin);
    if (he != hipSuccess || nrc != ncclSuccess) return fail("self-test all-reduce of the new communicator failed", false);
    ctx->comm_selftest = hp[0];
    if (hp[0] != (double)nranks || hp[1] != 0.5 * (double)nranks * (double)(nranks + 1)) {
        return fail("self-test all-reduce returned " + std::to_string(hp[0]) + " / " + std::to_string(hp[1]) + " instead of " +
                    std::to_string(nranks) + " / " + std::to_string(nranks * (nranks + 1) / 2) + ": not every rank took part", false);
    }
    return NK_OK;
}
int nk_comm_allreduce(nk_ctx *ctx, double *inout, int64_t n) {
    NK_ARG(ctx && inout && n >= 0, "nk_comm_allreduce: bad arguments");
    if (!ctx->comm || n == 0) return NK_OK;
    NK_HIP(hipSetDevice(ctx->device));
    double *buf = nullptr;
    NK_HIP(hipMalloc((void **)&buf, (size_t)n * 8));
    hipError_t he = hipMemcpyAsync(buf, inout, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
    ncclResult_t nrc = he == hipSuccess ? ctx->rccl.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, ctx->comm, ctx->stream) : ncclSystemError;
    if (he == hipSuccess && nrc == ncclSuccess) he = hipMemcpyAsync(inout, buf, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess && nrc == ncclSuccess) he = hipStreamSynchronize(ctx->stream);
    hipFree(buf);
    if (he != hipSuccess || nrc != ncclSuccess) { ctx->err = "nk_comm_allreduce failed"; return NK_ERR_COMM; }
    return NK_OK;
}
int nk_comm_info(nk_ctx *ctx, nk_comm_report *out) {
    NK_ARG(ctx && out, "nk_comm_info: NULL argument");
    memset(out, 0, sizeof(*out));
    out->rank = ctx->d.rank; out->nranks = ctx->d.nranks;
    out->comm_rank = ctx->comm ? ctx->comm_rank : -1;
    out->comm_nranks = ctx->comm ? ctx->comm_nranks : 0;
    out->device = ctx->device;
    out->selftest_sum = ctx->comm ? ctx->comm_selftest : 0.0;
    out->selftest_ok = (ctx->comm && ctx->comm_selftest == (double)ctx->d.nranks) ? 1 : 0;
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), ctx->device) != hipSuccess) out->pci_bus_id[0] = 0;
    return NK_OK;
}

// ------------------------------------------------------------------------------------- parity taps
}  // extern "C"
// Scratch device buffer of a tap: freed on every return path.
template <class T>
struct NkDevBuf {
    T *p = nullptr;
    hipError_t err = hipSuccess;
    NkDevBuf(const T *src, int64_t count) {
        err = hipMalloc((void **)&p, (size_t)(count > 0 ? count : 1) * sizeof(T));
        if (err == hipSuccess && src) err = hipMemcpy(p, src, (size_t)count * sizeof(T), hipMemcpyHostToDevice);
    }
    ~NkDevBuf() { if (p) hipFree(p); }
    hipError_t get(T *dst, int64_t count) const { return dst ? hipMemcpy(dst, p, (size_t)count * sizeof(T), hipMemcpyDeviceToHost) : hipSuccess; }
    NkDevBuf(const NkDevBuf &) = delete;
    NkDevBuf &operator=(const NkDevBuf &) = delete;
};
#define NK_BUF(T, name, src, count) NkDevBuf<T> name(src, count); NK_HIP(name.err)
extern "C" {

// Population.initialise_all_particles on the device (k_init_particles): N particles with ids pid_lo .. pid_lo + N - 1, modes
// tiled over the ids, positions uniform in the solid ('random_domain': sv_first NULL) or in the subvolume the index
// belongs to ('random_subvol': sv_first[S + 1], ascending, sv_first[0] = 0).  Replaces nk_reserve + nk_upload_particles.
int nk_init_particles(nk_ctx *ctx, int64_t N, int64_t capacity, uint64_t pid_lo, const int32_t *unique_modes, int64_t n_unique,
                      const int64_t *sv_first) {
    if (ctx) ctx->emitted_for = -1;
    NK_ARG(ctx && N >= 0 && unique_modes && n_unique > 0, "nk_init_particles: bad arguments");
    NK_ARG(ctx->have_material && ctx->have_mesh && ctx->have_sv, "nk_init_particles: set the material, the mesh and the subvolumes first");
    NkDev &d = ctx->d;
    NK_ARG(d.nS > 0, "nk_init_particles: the mesh was set without its volume tables (simplices)");
    NK_ARG(nk_want_part(ctx), "nk_init_particles: needs the mode-partitioned store");
    for (int64_t u = 0; u < n_unique; ++u) NK_ARG(unique_modes[u] >= 0 && unique_modes[u] < d.M, "nk_init_particles: mode index out of range");
    NK_HIP(hipSetDevice(ctx->device));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    int rc = nk_alloc_particles(ctx, std::max<int64_t>(capacity, N + N / 2 + 65536), nullptr, N, unique_modes, n_unique, (int64_t)(pid_lo % (uint64_t)n_unique));
    if (rc) return rc;
    NK_BUF(int32_t, um, unique_modes, n_unique);
    NK_BUF(int64_t, sf, sv_first, sv_first ? d.S + 1 : 0);
    NK_HIP(hipMemsetAsync(d.seg_count, 0, (size_t)d.nseg * sizeof(int32_t), ctx->stream));
    int32_t zero6[6] = {0, 0, 0, 0, 0, 0};
    NK_HIP(hipMemcpyAsync(d.halt, zero6, 24, hipMemcpyHostToDevice, ctx->stream));     // halt[4], overflow, ticket
    if (N > 0) k_init_particles<<<ctx->num_cu * 8, NK_WG, nk_lds(ctx, false), ctx->stream>>>(d, N, pid_lo, um.p, (int32_t)n_unique, sv_first ? sf.p : nullptr);
    NK_HIP(hipGetLastError());
    int32_t ovf = 0;
    NK_HIP(hipMemcpyAsync(&ovf, d.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_ARG(!(ovf & 64), "nk_init_particles: a particle did not find its subvolume in 4096 draws ('random_subvol' with very small "
                         "subvolumes): create the particles on the host (NK_HOST_INIT=1)");
    NK_ARG(ovf == 0, "nk_init_particles: a segment overflowed (internal sizing error)");
    ctx->pending_relax = false;
    return NK_OK;
}

// The tallies of the particles where they stand (the t = 0 row of the reference: calculate_energy and the flux sums before
// normalisation, Population.py:704-717, :734-736): E_raw[S], N[S], flux_raw[3 S] (subvolume-major).  This rank's particles only.
int nk_tally_state(nk_ctx *ctx, double *E_raw, double *N_sv, double *flux_raw) {
    NK_ARG(ctx && E_raw && N_sv && flux_raw, "nk_tally_state: bad arguments");
    int rc = nk_check_ready(ctx);
    if (rc) return rc;
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    if ((rc = nk_normalize(ctx))) return rc;
    if (ctx->pending_relax) {                          // the state the caller means includes the deferred relaxation
        k_relax<<<nk_sweep_grid(ctx), NK_WG, nk_lds(ctx, false), ctx->stream>>>(d, 0);
        ctx->pending_relax = false;
    }
    const int g = nk_sweep_grid(ctx);
    NK_BUF(double, acc, nullptr, d.NB + 2);
    k_tally_state<<<g, NK_WG, nk_lds(ctx, false), ctx->stream>>>(d);
    k_reduce<<<d.NB, NK_WG, 0, ctx->stream>>>(d, g, acc.p, nullptr, 1, 0);
    NK_HIP(hipGetLastError());
    std::vector<double> h((size_t)d.NB + 1);
    NK_HIP(hipMemcpyAsync(h.data(), acc.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    const int S = d.S;
    for (int s_ = 0; s_ < S; ++s_) { E_raw[s_] = h[(size_t)s_]; N_sv[s_] = h[(size_t)S + s_]; }
    for (int k = 0; k < 3 * S; ++k) flux_raw[k] = h[(size_t)2 * S + k];
    return NK_OK;
}

int nk_find_boundary(nk_ctx *ctx, int64_t n, const double *x, const double *v, double *xc, double *tc, int32_t *fc) {
    NK_ARG(ctx && ctx->have_mesh && ctx->have_sv && n > 0 && x && v, "nk_find_boundary: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    NK_BUF(double, dx, x, n * 3); NK_BUF(double, dv, v, n * 3);
    NK_BUF(double, dxc, nullptr, n * 3); NK_BUF(double, dtc, nullptr, n); NK_BUF(int32_t, dfc, nullptr, n);
    const bool verbose = getenv("NK_VERBOSE") != nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (verbose) { NK_HIP(hipEventCreate(&e0)); NK_HIP(hipEventCreate(&e1)); NK_HIP(hipEventRecord(e0, ctx->stream)); }
    NK_GEOM_LAUNCH(k_tap_find_boundary, (int)((n + NK_WG - 1) / NK_WG), nk_lds(ctx, true), d, n, dx.p, dv.p, dxc.p, dtc.p, dfc.p);
    NK_HIP(hipGetLastError());
    if (verbose) NK_HIP(hipEventRecord(e1, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (verbose) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[nanokappa_hip] find_boundary: %lld rays in %.3f ms\n", (long long)n, ms);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    NK_HIP(dxc.get(xc, n * 3)); NK_HIP(dtc.get(tc, n)); NK_HIP(dfc.get(fc, n));
    return NK_OK;
}
int nk_classify(nk_ctx *ctx, int64_t n, const double *x, int32_t *id) {
    NK_ARG(ctx && ctx->have_sv && n > 0 && x && id, "nk_classify: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NK_BUF(double, dx, x, n * 3); NK_BUF(int32_t, did, nullptr, n);
    k_tap_classify<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, nk_lds(ctx, false), ctx->stream>>>(ctx->d, n, dx.p, did.p);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(did.get(id, n));
    return NK_OK;
}
int nk_eval(nk_ctx *ctx, int32_t what, int64_t n, const double *a, const int32_t *mode, double *out) {
    NK_ARG(ctx && ctx->have_material && n > 0 && a && out && what >= 0 && what <= 5, "nk_eval: bad arguments");
    NK_ARG(what > 1 || mode, "nk_eval: mode required");
    NK_ARG(ctx->have_sv, "nk_eval: subvolumes required");
    NK_HIP(hipSetDevice(ctx->device));
    const int64_t na = what == 4 ? 3 * n : n;
    NK_BUF(double, da, a, na); NK_BUF(int32_t, dm, mode, n); NK_BUF(double, dout, nullptr, n);
    k_tap_eval<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, nk_lds(ctx, false), ctx->stream>>>(ctx->d, what, n, da.p, mode ? dm.p : nullptr, dout.p);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(dout.get(out, n));
    return NK_OK;
}
int nk_reflect(nk_ctx *ctx, int64_t n, const int32_t *facet, const int32_t *mode_in, const double *col_pos,
               const double *n_in, const double *omega_in, const double *r_spec, const double *r_deg,
               const double *r_diff, int32_t *mode_out, double *n_out, double *omega_out) {
    NK_ARG(ctx && ctx->have_material && ctx->have_sv && ctx->have_mesh && ctx->d.Fr > 0 && n > 0, "nk_reflect: engine not configured");
    NK_ARG(facet && mode_in && col_pos && n_in && omega_in && r_spec && r_diff, "nk_reflect: NULL input");
    for (int64_t i = 0; i < n; ++i)
        NK_ARG(facet[i] >= 0 && facet[i] < ctx->d.Fc && ctx->host_facets[facet[i]].rough >= 0, "nk_reflect: facet is not rough");
    NK_HIP(hipSetDevice(ctx->device));
    NK_BUF(int32_t, df, facet, n); NK_BUF(int32_t, dm, mode_in, n); NK_BUF(double, dc, col_pos, n * 3);
    NK_BUF(double, dn, n_in, n); NK_BUF(double, dom, omega_in, n); NK_BUF(double, drs, r_spec, n);
    NK_BUF(double, drd, r_deg, n); NK_BUF(double, drf, r_diff, n);
    NK_BUF(int32_t, dmo, nullptr, n); NK_BUF(double, dno, nullptr, n); NK_BUF(double, doo, nullptr, n);
    k_tap_reflect<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, nk_lds(ctx, false), ctx->stream>>>(
        ctx->d, n, df.p, dm.p, dc.p, dn.p, dom.p, drs.p, r_deg ? drd.p : nullptr, drf.p, dmo.p, dno.p, doo.p);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(dmo.get(mode_out, n)); NK_HIP(dno.get(n_out, n)); NK_HIP(doo.get(omega_out, n));
    return NK_OK;
}
int nk_calibrate_stream(nk_ctx *ctx, int32_t launches, int64_t *bytes_read, int64_t *bytes_written) {
    NK_ARG(ctx && launches > 0 && ctx->d.cap > 0, "nk_calibrate_stream: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NkHostParticles h;
    { int rc = nk_gather_live(ctx, h, false); if (rc) return rc; }
    const int64_t ns = (int64_t)h.x.size();
    for (int k = 0; k < launches; ++k) k_cal_stream<<<nk_sweep_grid(ctx), NK_WG, 0, ctx->stream>>>(ctx->d);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (const char *pc = getenv("NK_PROBE_COPY")) {      // developer probe: copy floor of the sweep's wave structure
        hipEvent_t e0, e1;
        NK_HIP(hipEventCreate(&e0)); NK_HIP(hipEventCreate(&e1));
        const int g = ctx->g_sweep > 0 ? ctx->g_sweep : ctx->num_cu * 3;
        for (int w = 8; w <= 16; w += 8) {
            if (atoi(pc) != w && atoi(pc) != 0) continue;
            for (int rep = 0; rep < 3; ++rep) {
                NK_HIP(hipEventRecord(e0, ctx->stream));
                for (int k = 0; k < 10; ++k) { if (w == 8) k_probe_copy<8><<<g, NK_WG, 0, ctx->stream>>>(ctx->d); else k_probe_copy<16><<<g, NK_WG, 0, ctx->stream>>>(ctx->d); }
                NK_HIP(hipEventRecord(e1, ctx->stream));
                NK_HIP(hipStreamSynchronize(ctx->stream));
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                fprintf(stderr, "[nanokappa_hip] copy probe, %d-byte accesses: %.1f us per pass over %lld particles (%.2f TB/s)\n", w, 100.0 * ms,
                        (long long)ns, (ctx->d.nts ? 88.0 : 72.0) * ns / (ms * 1e-4) / 1e12);
            }
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    if (bytes_read) *bytes_read = ns * (ctx->d.nts ? 44 : 36);
    if (bytes_written) *bytes_written = ns * (ctx->d.nts ? 32 : 24);
    return NK_OK;
}
int nk_uniform2(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *u0, double *u1) {
    if (!u0 || !u1) return NK_ERR_ARG;
    double *dout = nullptr, h[2];
    if (hipMalloc((void **)&dout, 16) != hipSuccess) return NK_ERR_HIP;
    k_tap_uniform<<<1, 1>>>(seed, pid, step, tag, dout);
    if (hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost) != hipSuccess) { hipFree(dout); return NK_ERR_HIP; }
    hipFree(dout);
    *u0 = h[0]; *u1 = h[1];
    return NK_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------ set-up table builder
static void nk_specular_free(nk_ctx *ctx) {
    for (void *p : {(void *)ctx->spec_v, (void *)ctx->spec_om, (void *)ctx->spec_dl, (void *)ctx->spec_modes,
                    (void *)ctx->spec_in, (void *)ctx->spec_out, (void *)ctx->spec_count, (void *)ctx->spec_svx,
                    (void *)ctx->spec_rank, (void *)ctx->ks_kv, (void *)ctx->ks_mat})
        if (p) hipFree(p);
    ctx->ks_kv = ctx->ks_mat = nullptr; ctx->ks_Q = 0;
    ctx->spec_v = ctx->spec_om = ctx->spec_dl = ctx->spec_svx = nullptr;
    ctx->spec_rank = nullptr;
    ctx->spec_modes = nullptr; ctx->spec_in = ctx->spec_out = nullptr; ctx->spec_count = nullptr;
    ctx->spec_M = 0; ctx->spec_cap = 0;
}
int nk_specular_begin(nk_ctx *ctx, int64_t M, const double *group_vel, const double *omega, const double *delta_omega) {
    NK_ARG(ctx && M > 0 && M < (1ll << 31) && group_vel && omega && delta_omega, "nk_specular_begin: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    nk_specular_free(ctx);
    NK_HIP(hipMalloc((void **)&ctx->spec_v, (size_t)M * 24));
    NK_HIP(hipMalloc((void **)&ctx->spec_om, (size_t)M * 8));
    NK_HIP(hipMalloc((void **)&ctx->spec_dl, (size_t)M * 8));
    NK_HIP(hipMalloc((void **)&ctx->spec_modes, (size_t)M * sizeof(NkSpecMode)));
    NK_HIP(hipMalloc((void **)&ctx->spec_count, 8));
    NK_HIP(hipMemcpy(ctx->spec_v, group_vel, (size_t)M * 24, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->spec_om, omega, (size_t)M * 8, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->spec_dl, delta_omega, (size_t)M * 8, hipMemcpyHostToDevice));
    // order of the x-velocities (the pair search only scans a window of it), rank of every mode in it, largest |v|
    std::vector<int32_t> order((size_t)M), rank((size_t)M);
    for (int64_t m = 0; m < M; ++m) order[(size_t)m] = (int32_t)m;
    std::stable_sort(order.begin(), order.end(), [&](int32_t p, int32_t q) { return group_vel[3 * (size_t)p] < group_vel[3 * (size_t)q]; });
    std::vector<double> svx((size_t)M);
    double vmax = 0.0;
    for (int64_t k = 0; k < M; ++k) {
        const size_t m = (size_t)order[(size_t)k];
        rank[m] = (int32_t)k;
        svx[(size_t)k] = group_vel[3 * m];
        const double vx = group_vel[3 * m], vy = group_vel[3 * m + 1], vz = group_vel[3 * m + 2];
        vmax = std::max(vmax, sqrt((vx * vx + vy * vy) + vz * vz));
    }
    NK_HIP(hipMalloc((void **)&ctx->spec_rank, (size_t)M * 4));
    NK_HIP(hipMalloc((void **)&ctx->spec_svx, (size_t)M * 8));
    NK_HIP(hipMemcpy(ctx->spec_rank, rank.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->spec_svx, svx.data(), (size_t)M * 8, hipMemcpyHostToDevice));
    ctx->spec_vmax = vmax;
    ctx->spec_M = M;
    return NK_OK;
}
int nk_specular_pairs(nk_ctx *ctx, const double *normal, double crit, int64_t cap, int32_t *pair_in, int32_t *pair_out,
                      int64_t *n_pairs) {
    NK_ARG(ctx && normal && n_pairs && cap >= 0 && ((pair_in != nullptr) == (pair_out != nullptr)), "nk_specular_pairs: bad arguments");
    NK_ARG(ctx->spec_M > 0, "nk_specular_pairs: call nk_specular_begin first");
    NK_HIP(hipSetDevice(ctx->device));
    const int M = (int)ctx->spec_M;
    if (cap > ctx->spec_cap) {
        if (ctx->spec_in) hipFree(ctx->spec_in);
        if (ctx->spec_out) hipFree(ctx->spec_out);
        ctx->spec_in = ctx->spec_out = nullptr;
        NK_HIP(hipMalloc((void **)&ctx->spec_in, (size_t)cap * 4));
        NK_HIP(hipMalloc((void **)&ctx->spec_out, (size_t)cap * 4));
        ctx->spec_cap = cap;
    }
    const int blocks = (M + NK_WG - 1) / NK_WG;
    NK_HIP(hipMemsetAsync(ctx->spec_count, 0, 8, ctx->stream));
    k_specular_prepare<<<blocks, NK_WG, 0, ctx->stream>>>(M, ctx->spec_v, ctx->spec_om, ctx->spec_dl, ctx->spec_rank, normal[0],
                                                          normal[1], normal[2], ctx->spec_modes);
    k_specular_pairs<<<blocks, NK_WG, 0, ctx->stream>>>(M, ctx->spec_modes, ctx->spec_svx, ctx->spec_vmax, normal[0], normal[1],
                                                        normal[2], crit, cap, ctx->spec_in, ctx->spec_out, ctx->spec_count);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long n = 0;
    NK_HIP(hipMemcpy(&n, ctx->spec_count, 8, hipMemcpyDeviceToHost));
    *n_pairs = (int64_t)n;
    const int64_t got = (int64_t)n < cap ? (int64_t)n : cap;
    ctx->spec_last = (int64_t)n <= cap ? (int64_t)n : -1;      // -1: truncated, the caller asks again with more room
    if (got > 0 && pair_in) {                          // (NULL arrays: the pairs stay on the device for nk_rough_pairs)
        NK_HIP(hipMemcpy(pair_in, ctx->spec_in, (size_t)got * 4, hipMemcpyDeviceToHost));
        NK_HIP(hipMemcpy(pair_out, ctx->spec_out, (size_t)got * 4, hipMemcpyDeviceToHost));
    }
    return NK_OK;
}
// 'k' / wavevector model (Population.py:1056-1240): the same calls as the 'velocity' search, between nk_specular_begin and
// nk_specular_end; the pairs land where nk_rough_pairs expects them.
int nk_kspec_begin(nk_ctx *ctx, int64_t Q, const double *wavevectors, const double *k_to_q, const double *q_to_k, const double *tol) {
    NK_ARG(ctx && Q > 0 && wavevectors && k_to_q && q_to_k && tol, "nk_kspec_begin: bad arguments");
    NK_ARG(ctx->spec_M > 0 && ctx->spec_M % Q == 0, "nk_kspec_begin: call nk_specular_begin first (M = Q x J modes)");
    NK_HIP(hipSetDevice(ctx->device));
    if (ctx->ks_kv) hipFree(ctx->ks_kv);
    if (ctx->ks_mat) hipFree(ctx->ks_mat);
    ctx->ks_kv = ctx->ks_mat = nullptr;
    NK_HIP(hipMalloc((void **)&ctx->ks_kv, (size_t)Q * 24));
    NK_HIP(hipMalloc((void **)&ctx->ks_mat, 18 * 8));
    NK_HIP(hipMemcpy(ctx->ks_kv, wavevectors, (size_t)Q * 24, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->ks_mat, k_to_q, 72, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->ks_mat + 9, q_to_k, 72, hipMemcpyHostToDevice));
    for (int c = 0; c < 3; ++c) ctx->ks_tol[c] = tol[c];
    ctx->ks_Q = Q;
    return NK_OK;
}
int nk_kspec_pairs(nk_ctx *ctx, const double *normal, int64_t cap, int32_t *pair_in, int32_t *pair_out, int64_t *n_pairs) {
    NK_ARG(ctx && normal && n_pairs && cap >= 0 && ((pair_in != nullptr) == (pair_out != nullptr)), "nk_kspec_pairs: bad arguments");
    NK_ARG(ctx->ks_Q > 0 && ctx->spec_M > 0, "nk_kspec_pairs: call nk_specular_begin and nk_kspec_begin first");
    NK_HIP(hipSetDevice(ctx->device));
    const int Q = (int)ctx->ks_Q, J = (int)(ctx->spec_M / ctx->ks_Q);
    if (cap > ctx->spec_cap) {
        if (ctx->spec_in) hipFree(ctx->spec_in);
        if (ctx->spec_out) hipFree(ctx->spec_out);
        ctx->spec_in = ctx->spec_out = nullptr;
        NK_HIP(hipMalloc((void **)&ctx->spec_in, (size_t)cap * 4));
        NK_HIP(hipMalloc((void **)&ctx->spec_out, (size_t)cap * 4));
        ctx->spec_cap = cap;
    }
    NK_HIP(hipMemsetAsync(ctx->spec_count, 0, 8, ctx->stream));
    k_kspec_pairs<<<(Q + NK_WG - 1) / NK_WG, NK_WG, 0, ctx->stream>>>(Q, J, ctx->spec_v, ctx->spec_om, ctx->ks_kv, ctx->ks_mat, ctx->ks_mat + 9,
                                                                      ctx->ks_tol[0], ctx->ks_tol[1], ctx->ks_tol[2], normal[0], normal[1], normal[2],
                                                                      cap, ctx->spec_in, ctx->spec_out, ctx->spec_count);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long n = 0;
    NK_HIP(hipMemcpy(&n, ctx->spec_count, 8, hipMemcpyDeviceToHost));
    *n_pairs = (int64_t)n;
    const int64_t got = (int64_t)n < cap ? (int64_t)n : cap;
    ctx->spec_last = (int64_t)n <= cap ? (int64_t)n : -1;
    if (got > 0 && pair_in) {
        NK_HIP(hipMemcpy(pair_in, ctx->spec_in, (size_t)got * 4, hipMemcpyDeviceToHost));
        NK_HIP(hipMemcpy(pair_out, ctx->spec_out, (size_t)got * 4, hipMemcpyDeviceToHost));
    }
    return NK_OK;
}
int nk_specular_end(nk_ctx *ctx) {
    NK_ARG(ctx, "nk_specular_end: NULL context");
    NK_HIP(hipSetDevice(ctx->device));
    nk_specular_free(ctx);
    return NK_OK;
}

// ---- the rough-facet tables built on the device (header: nk_rough_begin / nk_rough_pairs / nk_rough_finish)
static void nk_rough_build_free(nk_ctx *ctx, bool tables) {
    for (void *p : {(void *)ctx->rb_k2, (void *)ctx->rb_nin, (void *)ctx->rb_eta, (void *)ctx->rb_sub}) if (p) hipFree(p);
    ctx->rb_k2 = ctx->rb_nin = ctx->rb_eta = ctx->rb_sub = nullptr;
    if (tables) {
        for (void *p : {(void *)ctx->rb_spec, (void *)ctx->rb_roul, (void *)ctx->rb_ts, (void *)ctx->rb_map}) if (p) hipFree(p);
        ctx->rb_spec = ctx->rb_roul = nullptr; ctx->rb_ts = nullptr; ctx->rb_map = nullptr;
    }
}
int nk_rough_begin(nk_ctx *ctx, int32_t Fr, const int32_t *facet, const double *normal_in, const double *eta, const double *k_norm) {
    NK_ARG(ctx && Fr > 0 && facet && normal_in && eta && k_norm, "nk_rough_begin: bad arguments");
    NK_ARG(ctx->have_material && ctx->have_mesh, "nk_rough_begin: set material and mesh first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    nk_rough_build_free(ctx, true);
    for (int i = 0; i < Fr; ++i) NK_ARG(facet[i] >= 0 && facet[i] < d.Fc, "nk_rough_begin: facet index");
    const size_t n = (size_t)Fr * d.M;
    std::vector<double> k2((size_t)d.Q);
    for (int q = 0; q < d.Q; ++q) k2[(size_t)q] = k_norm[q] * k_norm[q];
    NK_HIP(hipMalloc((void **)&ctx->rb_k2, (size_t)d.Q * 8));
    NK_HIP(hipMalloc((void **)&ctx->rb_nin, (size_t)Fr * 24));
    NK_HIP(hipMalloc((void **)&ctx->rb_eta, (size_t)Fr * 8));
    NK_HIP(hipMalloc((void **)&ctx->rb_sub, n * 8));
    NK_HIP(hipMalloc((void **)&ctx->rb_spec, n * 8));
    NK_HIP(hipMalloc((void **)&ctx->rb_roul, n * 8));
    NK_HIP(hipMalloc((void **)&ctx->rb_ts, n));
    NK_HIP(hipMalloc((void **)&ctx->rb_map, n * 4));
    NK_HIP(hipMemcpy(ctx->rb_k2, k2.data(), (size_t)d.Q * 8, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->rb_nin, normal_in, (size_t)Fr * 24, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(ctx->rb_eta, eta, (size_t)Fr * 8, hipMemcpyHostToDevice));
    NK_HIP(hipMemsetAsync(ctx->rb_sub, 0, n * 8, ctx->stream));
    NK_HIP(hipMemsetAsync(ctx->rb_ts, 0, n, ctx->stream));
    k_fill_u32<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(ctx->rb_map, (int64_t)n, 0xFFFFFFFFu);
    NK_HIP(hipGetLastError());
    ctx->rb_Fr = Fr;
    ctx->rb_facet.assign(facet, facet + Fr);
    return NK_OK;
}
int nk_rough_pairs(nk_ctx *ctx, int32_t nf, const int32_t *fidx) {
    NK_ARG(ctx && nf > 0 && fidx && ctx->rb_Fr > 0, "nk_rough_pairs: call nk_rough_begin first");
    NK_ARG(ctx->spec_last >= 0, "nk_rough_pairs: the last nk_specular_pairs call was truncated");
    for (int i = 0; i < nf; ++i) NK_ARG(fidx[i] >= 0 && fidx[i] < ctx->rb_Fr, "nk_rough_pairs: rough-facet index");
    if (ctx->spec_last == 0) return NK_OK;
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    NK_BUF(int32_t, df, fidx, nf);
    k_rough_pairs<<<(int)((ctx->spec_last + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(d.M, d.J, ctx->spec_v, ctx->rb_k2, ctx->spec_last,
                                                                                         ctx->spec_in, ctx->spec_out, nf, df.p, ctx->rb_nin,
                                                                                         ctx->rb_eta, ctx->rb_ts, ctx->rb_map, ctx->rb_sub);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    return NK_OK;
}
int nk_rough_finish(nk_ctx *ctx) { return nk_rough_finish_k(ctx, 0, nullptr, nullptr); }
// the same with the 'k' model's degenerate branches (find_degeneracies, Population.py:1017-1040: rows q, j1, j2): their
// creation rates are averaged (:926-930) and the reflection flips a coin between them (:963-969; degen_j2 [M] as in nk_rough)
int nk_rough_finish_k(nk_ctx *ctx, int32_t nd, const int32_t *degen, const int32_t *degen_j2) {
    NK_ARG(ctx && ctx->rb_Fr > 0 && ctx->spec_v, "nk_rough_finish: call nk_rough_begin (inside nk_specular_begin .. end) first");
    NK_ARG(nd >= 0 && (nd == 0 || degen), "nk_rough_finish_k: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    const int Fr = ctx->rb_Fr;
    const size_t n = (size_t)Fr * d.M;
    k_rough_finish<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(Fr, d.M, d.J, ctx->spec_v, ctx->rb_k2, ctx->rb_nin, ctx->rb_eta,
                                                                              ctx->rb_ts, (int32_t *)ctx->rb_map, ctx->rb_sub, ctx->rb_spec,
                                                                              ctx->rb_roul, nd == 0 ? 1 : 0);
    const int32_t *dj2 = nullptr;
    if (nd > 0) {
        for (int i = 0; i < nd; ++i)
            NK_ARG(degen[3 * i] >= 0 && degen[3 * i] < d.M / d.J && degen[3 * i + 1] >= 0 && degen[3 * i + 1] < d.J && degen[3 * i + 2] >= 0 &&
                       degen[3 * i + 2] < d.J, "nk_rough_finish_k: degeneracy row out of range");
        NK_BUF(int32_t, dg, degen, (int64_t)nd * 3);
        k_rough_degen<<<(Fr + 63) / 64, 64, 0, ctx->stream>>>(Fr, d.M, d.J, nd, dg.p, ctx->rb_roul);
        k_rough_round<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>((int64_t)n, ctx->rb_roul);
        NK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (degen_j2) NK_UP(degen_j2, (size_t)d.M, &dj2);
    k_rough_cumsum<<<Fr, 64, 0, ctx->stream>>>(d.M, ctx->rb_roul);
    int nlut = 1024;
    while (nlut < 65536 && (int64_t)nlut * 4 < d.M) nlut *= 2;
    int32_t *lut = nullptr;
    NK_HIP(hipMalloc((void **)&lut, (size_t)Fr * (nlut + 1) * 4));
    ctx->allocs.push_back(lut);
    k_rough_lut<<<(int)(((size_t)Fr * (nlut + 1) + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(Fr, d.M, nlut, ctx->rb_roul, lut);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    // install, as nk_set_rough does (the tables now belong to the context)
    d.Fr = Fr;
    ctx->g_sweep = 0;
    d.specularity = ctx->rb_spec; d.true_spec = ctx->rb_ts; d.spec_map = (const int32_t *)ctx->rb_map; d.roulette = ctx->rb_roul;
    d.roul_lut = lut; d.roul_nlut = nlut; d.degen_j2 = dj2;
    for (void *p : {(void *)ctx->rb_spec, (void *)ctx->rb_roul, (void *)ctx->rb_ts, (void *)ctx->rb_map}) ctx->allocs.push_back(p);
    ctx->rb_spec = ctx->rb_roul = nullptr; ctx->rb_ts = nullptr; ctx->rb_map = nullptr;
    nk_rough_build_free(ctx, false);
    for (int i = 0; i < Fr; ++i) ctx->host_facets[ctx->rb_facet[(size_t)i]].rough = i;
    ctx->rb_Fr = 0;
    return nk_patch_facets(ctx);
}
int nk_rough_download(nk_ctx *ctx, double *specularity, uint8_t *true_spec, int32_t *spec_map, double *roulette) {
    NK_ARG(ctx && ctx->d.Fr > 0, "nk_rough_download: no rough tables");
    NK_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->d.Fr * ctx->d.M;
    if (specularity) NK_HIP(hipMemcpy(specularity, ctx->d.specularity, n * 8, hipMemcpyDeviceToHost));
    if (true_spec) NK_HIP(hipMemcpy(true_spec, ctx->d.true_spec, n, hipMemcpyDeviceToHost));
    if (spec_map) NK_HIP(hipMemcpy(spec_map, ctx->d.spec_map, n * 4, hipMemcpyDeviceToHost));
    if (roulette) NK_HIP(hipMemcpy(roulette, ctx->d.roulette, n * 8, hipMemcpyDeviceToHost));
    return NK_OK;
}
int nk_build_enter_prob(nk_ctx *ctx, int32_t R, const double *normal_in, const double *thickness, double dt, double *out) {
    NK_ARG(ctx && R > 0 && normal_in && thickness && out && ctx->have_material, "nk_build_enter_prob: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    const size_t n = (size_t)R * d.M;
    NK_BUF(double, dn, normal_in, (int64_t)R * 3); NK_BUF(double, dth, thickness, R); NK_BUF(double, dout, nullptr, (int64_t)n);
    k_enter_prob<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(R, d.M, ctx->d_vg, dn.p, dth.p, dt, dout.p);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(dout.get(out, (int64_t)n));
    return NK_OK;
}
