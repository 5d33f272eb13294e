// nk_kernels.h -- the HIP kernels of libnanokappa_hip.so (gfx950 / MI355X only).
//
// Stream order of one timestep (reference Population.run_timestep, Population.py:1724-1769):
//   [k_relax + k_contains every `contains_every` steps]            contains_check       :1712-1722
//   [k_emit_one_to_one: only for the 'one_to_one' generator]        fill_reservoirs      :457-489
//   k_emit        per segment (one wave): the reservoir particles of the segment's own modes, appended behind its live ones
//                                                                   fill_reservoirs      :356-455
//                                                                   add_reservoir_particles :525-552
//   k_sweep       per segment (one wave): relax(previous step) -> drift -> boundary events -> tally -> compaction
//                                                                   lifetime_scattering  :1701-1710 (deferred)
//                                                                   drift                :790-795
//                                                                   boundary_scattering  :1546-1683
//                                                                   calculate_energy     :704-717
//   k_reduce      deterministic column sums of the per-workgroup tally rows; single rank: the last workgroup also
//                 normalises, inverts E -> T and writes the history row    calculate_energy :719-728, refresh_temperatures :692
//   (nranks > 1: RCCL all-reduce of the tally vector, then k_update does that part)
//   (k_tail = k_reduce and the NEXT step's k_emit in one launch, wherever the emission depends on neither the update nor
//    k_deliver: no rough facets, not 'one_to_one')
//
// Deferred relaxation: the reference relaxes occupations at the END of step k with the temperatures of step k.  Those
// need the global tally of step k, so the relaxation is carried into the BEGINNING of the sweep of step k+1 (same
// positions, same T_sv): one pass over the particles per step instead of two.  A pending relaxation is flushed by
// k_relax before anything observes the particles (download, contains_check).
//
// The sweep (history: DESIGN.md):
//   * a WAVE owns a segment and walks it tile by tile (64 particles, coalesced loads, next tile prefetched);
//   * particles that meet a boundary inside the step (a third of them in a 20 nm box) are parked in the wave's carry, 64
//     slots in LDS; whenever 64 are there the whole wave runs one boundary event per particle with all lanes busy; the few
//     that meet another wall go back into the carry.  No second trip through HBM, no divergence against the streaming
//     lanes.  (The carry was a second set of registers filled by cross-lane permutes first: 26-40 permutes per tile and
//     15 registers live across the tile loop; in LDS it is 8 writes per tile and 8 reads per pass.)
//   * survivors are written back compacted IN PLACE (write cursor <= read cursor), absorbed particles simply vanish;
//   * the modes a segment owns (nk_device.h) enter through the reservoirs in the same loop: the wave evaluates its
//     (reservoir, mode) entries, keeps their counts in LDS and builds the entering particles in whole tiles.
#pragma once
// non-template kernels get internal linkage in a second translation unit that only wants a template (nk_sweep_plain.hip)
#ifndef NK_KERNEL_LINKAGE
#define NK_KERNEL_LINKAGE
#endif
#include "nk_device.h"

#define NK_TILE 64           // particles per tile = lanes of a wave
// NK_NT=1 builds the sweep's particle loads / stores as non-temporal accesses (developer comparison; measured SLOWER on
// MI355X: k_sweep 0.29-0.32 ms against 0.263-0.267 ms with plain accesses at 1e7 particles, profiles/r02_notes.txt).
#ifndef NK_NT
#define NK_NT 0
#endif
#if NK_NT
#define NK_LD(p) __builtin_nontemporal_load(p)
#define NK_ST(p, v) __builtin_nontemporal_store(v, p)
#else
#define NK_LD(p) (*(p))
#define NK_ST(p, v) (*(p) = (v))
#endif
// Developer build (make stamps -> libnanokappa_hip_stamps.so, env NK_STAMPS=1): s_memtime stamps around the sections of
// the sweep's tile loop, summed per wave; shares only (the stamps' waits forbid overlaps the real kernel has).
#ifdef NK_STAMPS
#define NK_STAMP(k)                                                                                    \
    do {                                                                                               \
        unsigned long long t_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        st_acc[k] += t_ - st_last;                                                                     \
        st_last = t_;                                                                                  \
    } while (0)
#else
#define NK_STAMP(k) do { } while (0)
#endif

// lanes below this one whose bit is set in `m` (a ballot): v_mbcnt_lo / v_mbcnt_hi, no mask of the lower lanes to keep
__device__ __forceinline__ int nk_rank(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// =================================================================================== LDS carve-up
struct NkLds {
    NkSvTab tb;                          // centres, temperatures (+ RBF coefficients), per-subvolume records
    NkBins bins;
    const double *planes, *faces;
    const NkFacet *facets;
    const double *resT;                  // [2R] reservoir {T, 1/T}
    const int *rf_off;                   // reservoir face tables (CSR), LDS copies when d.res_lds
    const double *rf_cdf, *rf_verts;
    // emission scratch of k_emit, one slice of NK_EMIT_CHUNK entries per wave
    unsigned int *sp_pref, *sp_cnt, *sp_rm;
    double *sp_cv, *sp_pr;
    // mode records of the sweep's segments, d.nlrec per wave (16-byte aligned)
    double *lrec;
    // the sweep's carry: up to 64 parked particles per wave
    double *carry;
    // output ring of the sweep, NK_ORING particles per wave: x y z occ nts [pid] (doubles), then w0
    double *oring;
    unsigned int *oring_w;
    double *colsum;                      // k_resident: column sums of the tally rows
};

// geom: 0 = no ray-casting tables, 1 = planes/faces/facets staged in LDS, 2 = read from global memory (large meshes)
// nrf: faces of the reservoir sampling tables staged in LDS (0 = not staged); kind: 0 plain, 1 + k_emit's scratch,
// 2 + the sweep's mode records (nlrec per wave), carry (unless the sweep is split) and output ring, 3 the same with particle ids,
// 4 / 5 = 2 / 3 + k_emit's scratch + a column-sum area (the resident kernel of small ensembles, k_resident, does both)
#ifndef NK_LREC_STRIDE
#define NK_LREC_STRIDE 5     // 16-byte units between the LDS copies of two mode records: 4 = packed (64 B), 5 spreads the banks
#endif
#ifndef NK_OUT_RING
#define NK_OUT_RING 0        // 1: finished particles go through an LDS ring and leave in whole aligned tiles (NkOut below)
#endif
#define NK_ORING (NK_OUT_RING ? 128 : 0)
// doubles of one wave's carry: x y z occ nts cts (64 each), w0 + evc (64 words each); with ids: + pid (64), gm and slot (64 words each)
#define NK_CARRY_DOUBLES(kind) (((kind) == 3 || (kind) == 5) ? 576 : 448)
#ifndef NK_BOX_GENERAL_CAST
#define NK_BOX_GENERAL_CAST 0    // 1: the box store's event pass casts its rays through the general search over the LDS planes (developer probe)
#endif
#define NK_COLSUM_DOUBLES 768   // k_resident: column sums of the tally rows, two halves (NB <= 384)
__host__ __device__ inline size_t nk_lds_bytes(int S, int R, int F, int NP, int Fc, int geom, int kind, int nrf, int rbfP, int nlrec = 0, int carry = 0) {
    const bool emit = kind == 1 || kind >= 4;
    int Fl = geom == 1 ? F : 0;
    int Pl = Fl ? NP : 0;
    int Fcl = geom == 1 ? Fc : 0;
    size_t nd = (size_t)S + (size_t)((rbfP + 1) & ~1) + 3 * S + ((3 * S) & 1) + 4 * (size_t)S + NK_NREP * S + NK_NREP * 3 * S + 4 * R +
                (size_t)Fl * NK_FACE_DOUBLES + (size_t)Pl * NK_PLANE_DOUBLES + 2 * (size_t)R + 10 * (size_t)nrf +
                (emit ? 2 * (size_t)(NK_WG / 64) * NK_EMIT_CHUNK : 0) + (kind >= 2 ? (size_t)(NK_WG / 64) * ((size_t)nlrec * 2 * NK_LREC_STRIDE + (carry ? NK_CARRY_DOUBLES(kind) : 0) + NK_ORING * ((kind == 3 || kind == 5) ? 6 : 5)) : 0) +
                (kind >= 4 ? NK_COLSUM_DOUBLES : 0) + 4;
    size_t bytes = nd * 8 + (size_t)Fcl * sizeof(NkFacet) +
                   (size_t)(NK_NREP * S + R + 1 + (R + 1) + (emit ? 3 * (NK_WG / 64) * NK_EMIT_CHUNK : 0) + (kind >= 2 ? (NK_WG / 64) * NK_ORING : 0)) * 4 + 32;
    return (bytes + 15) & ~(size_t)15;
}

// The pointer arithmetic.  GEOM as in nk_lds_bytes (a compile-time choice, so that the table pointers are provably LDS and
// are read with ds_read, not flat loads).  Plane, face, facet and subvolume records start on 16-byte boundaries.
template <int GEOM, int KIND>
__device__ __forceinline__ void nk_lds_carve(const NkDev &d, unsigned char *smem, NkLds &L) {
    constexpr bool EMIT = KIND == 1 || KIND >= 4;
    const int S = d.S, R = d.R;
    const int Fl = GEOM == 1 ? d.F : 0;
    const int Pl = Fl ? d.NP : 0;
    const int Fcl = GEOM == 1 ? d.Fc : 0;
    const int nrf = (GEOM == 1 && d.res_lds) ? d.res_nf : 0;
    double *p = (double *)smem;
    L.tb.Tsv = p; p += S + ((d.rbf_P + 1) & ~1);       // temperatures, then the RBF coefficients (even count)
    L.tb.cen = p; p += 3 * S + ((3 * S) & 1);
    p += ((size_t)S & 1);                              // 16-byte alignment of the records below
    L.tb.sv = (const NkSv *)p; p += 4 * S;
    L.bins.E = p; p += NK_NREP * S;
    L.bins.flux = p; p += NK_NREP * 3 * S;
    L.bins.resb = p; p += 4 * R;
    p += ((size_t)(p - (double *)smem) & 1);
    double *faces = p; p += (size_t)Fl * NK_FACE_DOUBLES;
    double *planes = p; p += (size_t)Pl * NK_PLANE_DOUBLES;
    double *resT = p; p += 2 * R;
    double *rf_cdf = p; p += nrf;
    double *rf_verts = p; p += 9 * (size_t)nrf;
    if (EMIT) { L.sp_cv = p; p += (NK_WG / 64) * NK_EMIT_CHUNK; L.sp_pr = p; p += (NK_WG / 64) * NK_EMIT_CHUNK; } else L.sp_cv = L.sp_pr = nullptr;
    p += ((size_t)(p - (double *)smem) & 1);           // keep the records and the facet table 16-byte aligned
    if (KIND >= 2) {
        L.lrec = p; p += (NK_WG / 64) * (size_t)d.nlrec * 2 * NK_LREC_STRIDE;
        L.carry = p; p += d.qx ? 0 : (NK_WG / 64) * NK_CARRY_DOUBLES(KIND);
        L.oring = p; p += (NK_WG / 64) * NK_ORING * ((KIND == 3 || KIND == 5) ? 6 : 5);
    } else { L.lrec = nullptr; L.carry = nullptr; L.oring = nullptr; }
    if (KIND >= 4) { L.colsum = p; p += NK_COLSUM_DOUBLES; } else L.colsum = nullptr;
    NkFacet *facets = (NkFacet *)p;
    unsigned int *u = (unsigned int *)(facets + Fcl);
    L.bins.N = u; u += NK_NREP * S;
    L.bins.nleave = u; u += R;
    L.bins.misc = u; u += 1;
    int *rf_off = (int *)u; u += R + 1;
    if (EMIT) {
        L.sp_pref = u; u += (NK_WG / 64) * NK_EMIT_CHUNK;
        L.sp_cnt = u; u += (NK_WG / 64) * NK_EMIT_CHUNK;
        L.sp_rm = u; u += (NK_WG / 64) * NK_EMIT_CHUNK;
    } else { L.sp_pref = L.sp_cnt = L.sp_rm = nullptr; }
    if (KIND >= 2) { L.oring_w = u; u += (NK_WG / 64) * NK_ORING; } else L.oring_w = nullptr;
    L.resT = resT;
    L.rf_off = rf_off; L.rf_cdf = rf_cdf; L.rf_verts = rf_verts;
    if (GEOM == 1) { L.faces = faces; L.planes = planes; L.facets = facets; }
    else { L.faces = d.faces; L.planes = d.planes; L.facets = d.facets; }
}
// Cooperative fill of the read-only tables and zeroing of the bins; ends with a barrier.
template <int GEOM, int KIND>
__device__ __forceinline__ void nk_lds_setup(const NkDev &d, unsigned char *smem, NkLds &L) {
    nk_lds_carve<GEOM, KIND>(d, smem, L);
    const int S = d.S, R = d.R;
    const int Fl = GEOM == 1 ? d.F : 0;
    const int Pl = Fl ? d.NP : 0;
    const int Fcl = GEOM == 1 ? d.Fc : 0;
    const int nrf = (GEOM == 1 && d.res_lds) ? d.res_nf : 0;
    double *faces = const_cast<double *>(L.faces), *planes = const_cast<double *>(L.planes);
    NkFacet *facets = const_cast<NkFacet *>(L.facets);
    double *resT = const_cast<double *>(L.resT), *rf_cdf = const_cast<double *>(L.rf_cdf), *rf_verts = const_cast<double *>(L.rf_verts);
    int *rf_off = const_cast<int *>(L.rf_off);
    double *Tsv = const_cast<double *>(L.tb.Tsv), *cen = const_cast<double *>(L.tb.cen);
    NkSv *sv = const_cast<NkSv *>(L.tb.sv);
    const int t = threadIdx.x, nth = blockDim.x;
    // Every table is small (a box: 20 + 60 + 192 + 36 + ... entries), so a thread's share of each is one element -- and one
    // loop per table means one memory round trip per table, nine of them in a row (stamps: 6.3 us of k_emit's 36, 8.5 us of
    // the sweep's prologue).  The FIRST element of every table is therefore requested up front, all loads in flight
    // together, and stored afterwards; whatever a table has beyond blockDim.x elements follows in ordinary loops.
    const int nT = S + d.rbf_P, a = d.sv_axis, nface = Fl * NK_FACE_DOUBLES, nplane = Pl * NK_PLANE_DOUBLES;
    const int nfw = Fcl * (int)(sizeof(NkFacet) / 4);
    const int jn = t + 1 < S ? t + 1 : t;
    const double l_T = t < nT ? d.T_sv[t] : 0.0;
    const double l_cen = t < 3 * S ? d.centers[t] : 0.0;
    const double l_c = t < S ? d.centers[3 * t + a] : 0.0, l_cn = t < S ? d.centers[3 * jn + a] : 0.0, l_Tn = t < S ? d.T_sv[jn] : 0.0;
    const double l_rT = t < R ? d.res_T[t] : 1.0;
    const double l_face = t < nface ? d.faces[t] : 0.0;
    const double l_plane = t < nplane ? d.planes[t] : 0.0;
    const int l_off = (nrf > 0 && t <= R) ? d.res_face_off[t] : 0;
    const double l_cdf = t < nrf ? d.res_face_cdf[t] : 0.0;
    const double l_vert = t < 9 * nrf ? d.res_face_verts[t] : 0.0;
    const int32_t l_fw = t < nfw ? ((const int32_t *)d.facets)[t] : 0;
    if (t < nT) Tsv[t] = l_T;
    if (t < 3 * S) cen[t] = l_cen;
    if (t < S) {
        // per-subvolume record: centre along the slice axis, T, slope of interp1d's bracket (i, i+1), 1 / T
        NkSv q;
        q.c = l_c; q.T = l_T; q.slope = jn > t ? (l_Tn - l_T) / (l_cn - l_c) : 0.0; q.invT = 1.0 / l_T;
        sv[t] = q;
    }
    if (t < R) { L.bins.nleave[t] = 0u; resT[2 * t] = l_rT; resT[2 * t + 1] = 1.0 / l_rT; }
    if (t < nface) faces[t] = l_face;
    if (t < nplane) planes[t] = l_plane;
    if (nrf > 0 && t <= R) rf_off[t] = l_off;
    if (t < nrf) rf_cdf[t] = l_cdf;
    if (t < 9 * nrf) rf_verts[t] = l_vert;
    if (t < nfw) ((int32_t *)facets)[t] = l_fw;
    for (int i = t; i < NK_NREP * S; i += nth) { L.bins.E[i] = 0.0; L.bins.N[i] = 0u; }
    for (int i = t; i < NK_NREP * 3 * S; i += nth) L.bins.flux[i] = 0.0;
    for (int i = t; i < 4 * R; i += nth) L.bins.resb[i] = 0.0;
    if (t == 0) L.bins.misc[0] = 0u;
    // the rest of tables that are longer than the workgroup
    for (int i = t + nth; i < nT; i += nth) Tsv[i] = d.T_sv[i];
    for (int i = t + nth; i < 3 * S; i += nth) cen[i] = d.centers[i];
    for (int i = t + nth; i < S; i += nth) {
        const int j = i + 1 < S ? i + 1 : i;
        const double c = d.centers[3 * i + a], cn = d.centers[3 * j + a], T = d.T_sv[i], Tn = d.T_sv[j];
        NkSv q;
        q.c = c; q.T = T; q.slope = j > i ? (Tn - T) / (cn - c) : 0.0; q.invT = 1.0 / T;
        sv[i] = q;
    }
    for (int i = t + nth; i < R; i += nth) { L.bins.nleave[i] = 0u; const double T = d.res_T[i]; resT[2 * i] = T; resT[2 * i + 1] = 1.0 / T; }
    for (int i = t + nth; i < nface; i += nth) faces[i] = d.faces[i];
    for (int i = t + nth; i < nplane; i += nth) planes[i] = d.planes[i];
    if (nrf > 0) {
        for (int i = t + nth; i <= R; i += nth) rf_off[i] = d.res_face_off[i];
        for (int i = t + nth; i < nrf; i += nth) rf_cdf[i] = d.res_face_cdf[i];
        for (int i = t + nth; i < 9 * nrf; i += nth) rf_verts[i] = d.res_face_verts[i];
    }
    for (int i = t + nth; i < nfw; i += nth) ((int32_t *)facets)[i] = ((const int32_t *)d.facets)[i];
    __syncthreads();
}

// Row layout: E[S] N[S] flux[3S] nleave[R] resE[R] resF[3R] emitted[1]
__device__ __forceinline__ void nk_lds_flush(const NkDev &d, const NkLds &L, int64_t row) {
    __syncthreads();
    const int S = d.S, R = d.R;
    double *out = d.partials + row * d.NB;
    for (int b = threadIdx.x; b < d.NB; b += blockDim.x) {
        double v = 0.0;
        if (b < S) { for (int r = 0; r < NK_NREP; ++r) v += L.bins.E[r * S + b]; }
        else if (b < 2 * S) { unsigned int c = 0; for (int r = 0; r < NK_NREP; ++r) c += L.bins.N[r * S + (b - S)]; v = (double)c; }
        else if (b < 5 * S) { int k = b - 2 * S; for (int r = 0; r < NK_NREP; ++r) v += L.bins.flux[r * 3 * S + k]; }
        else if (b < 5 * S + R) v = (double)L.bins.nleave[b - 5 * S];
        else if (b < 5 * S + 2 * R) v = L.bins.resb[4 * (b - 5 * S - R)];
        else if (b < 5 * S + 5 * R) { int k = b - 5 * S - 2 * R; v = L.bins.resb[4 * (k / 3) + 1 + (k % 3)]; }
        else v = (double)L.bins.misc[0];
        out[b] = v;
    }
}

// Particle <-> mode bookkeeping of a segment (nk_device.h: modes are partitioned over the segments).
struct NkSegModes {
    const NkMode *rec;      // record of stored index i at rec[i]
    const int32_t *gm;      // global mode of stored index i at gm[i] (null: the stored index is the mode itself)
    int nl;                 // modes this segment owns (whose reservoir particles it creates)
    int estride, eoff;      // without the partition (developer probe): the l-th of those modes is l * estride + eoff
    __device__ __forceinline__ int mode(int i) const { return gm ? gm[i] : i; }                         // of a stored index
    __device__ __forceinline__ int entry_mode(int l) const { return gm ? gm[l] : l * estride + eoff; }     // of an emission entry
};
struct NkPlainModes { __device__ __forceinline__ int mode(int i) const { return i; } };   // the stored index is the mode
// slot of a mode -> (segment, local index)
__device__ __forceinline__ void nk_mode_home(const NkDev &d, int mode, int &seg, int &idx) {
    const uint32_t slot = (uint32_t)d.m2s[mode], n = (uint32_t)d.nseg;
    const uint32_t q = slot / n;
    idx = (int)q;
    seg = (int)(slot - q * n);
}
__device__ __forceinline__ NkSegModes nk_seg_modes(const NkDev &d, int seg) {
    NkSegModes sm;
    sm.rec = d.part ? d.modetab_p + (int64_t)seg * d.nlmax : d.modetab;
    sm.gm = d.part ? d.s2m + (int64_t)seg * d.nlmax : nullptr;
    sm.nl = d.seg_nl[seg];
    sm.estride = d.nseg; sm.eoff = seg;
    return sm;
}

// Deferred lifetime_scattering (Population.py:1701-1710) for one particle.
// The mode record is passed as two 32-byte halves {omega, vx, vy, vz} {E0, tau0..tau2} (two dwordx4 pairs, no struct copy).
template <bool RBF = true>
__device__ __forceinline__ double nk_relax(const NkDev &d, const NkLds &L, const NkTauWin &tw, const double4 &ra, const double4 &rb, double x,
                                           double y, double z, double occ, const NkSegModes &sm, int idx) {
    double invT;
    const double T = nk_interp_T<RBF>(d, L.tb, x, y, z, invT);
    const double tau = nk_lifetime(d, tw, rb.y, rb.z, rb.w, T, sm, idx);
    const double n0 = (T > 0.0) ? nk_be(ra.x * d.c_hk, rb.x, invT, d.invT0) : 0.0;
    // exp(-dt / tau): where every lane of the wave has dt / tau < 1/8 (lifetimes of 8 timesteps and more -- all of the synthetic
    // Si / Ge at dt = 1 ps, most modes of a real material) the short series of nk_exp_small does without the range reduction, the
    // longer polynomial and the ldexp of nk_exp (truncation 3e-18 either way); a wave with one short-lived mode takes the general one
    const double xr = -d.dt * nk_rcp(tau);
    double er;
    if (__builtin_expect(__ballot(tau > 0.0 && !(xr > -0.125)) == 0ull, 1)) er = nk_exp_small(xr);
    else er = nk_exp(xr);
    return (tau > 0.0) ? n0 + (occ - n0) * er : n0;
}

// Mesh.sample_surface on one reservoir facet (Mesh.py:923-951): face by area (np.random.choice, :937), then a uniform
// point of the triangle (:945-947).  `off/cdf/verts` are the reservoir face tables (LDS or global).
__device__ __forceinline__ void nk_sample_res_face(const int *off, const double *cdf, const double *verts, int r, double uf,
                                                   double us, double ur, double &x0, double &y0, double &z0) {
    const int f0 = off[r], nf = off[r + 1] - f0;
    int a = nk_ss_right(cdf + f0, nf, uf);
    a = a > nf - 1 ? nf - 1 : a;
    const double *fv = verts + 9 * (f0 + a);
    const double sq = sqrt(us);
    const double a0 = 1.0 - sq, a1 = (1.0 - ur) * sq, a2 = ur * sq;
    x0 = a0 * fv[0] + a1 * fv[3] + a2 * fv[6];
    y0 = a0 * fv[1] + a1 * fv[4] + a2 * fv[7];
    z0 = a0 * fv[2] + a1 * fv[5] + a2 * fv[8];
}

// Ray cast as the kernels call it: tree walk for meshes whose tables stay in global memory, plain sweep over LDS.
#define NK_RAY(GEOM, d, L, skip, x, y, z, vx, vy, vz, tc, fc)                                                        \
    do {                                                                                                              \
        if ((GEOM) == 2 && (d).NG > 0) nk_find_boundary_tree(d, skip, x, y, z, vx, vy, vz, tc, fc);                      \
        else nk_find_boundary((L).planes, (L).faces, (d).NP, (d).tol, x, y, z, vx, vy, vz, tc, fc);                   \
    } while (0)

// ========================================================================================= kernels
// Which modes enter at each reservoir at `step`, and how many particles of each: fill_reservoirs 'constant'
// (Population.py:358-370) / 'fixed_rate' (:408-420) for one (reservoir, mode) entry.  c = particles entering,
// c_mine = those this rank owns (emission_owner: (rm + level + step) % nranks); cv = the counter / dice value that the
// level-1 particle's entry time uses.
__device__ __forceinline__ void nk_emit_entry(const NkDev &d, uint32_t step, int64_t rm, int64_t at, double prob, int &c, int &c_mine, double &cv) {
    const double fixed = floor(prob);
    int mask;
    if (d.res_gen == 0) {
        cv = d.rc_p[(int64_t)(step & 1u) * d.rc_len + at] + (prob - fixed);
        mask = cv >= 1.0;
        cv -= (double)mask;
        d.rc_p[(int64_t)((step + 1u) & 1u) * d.rc_len + at] = cv;
    } else {
        double d1;
        nk_uniform2_dev(d.seed, (uint64_t)rm | 0xFFFFFFFF00000000ull, step, NK_TAG_DICE, cv, d1);
        mask = cv <= (prob - fixed);
    }
    c = (int)fixed + mask;
    c_mine = c;
    if (d.nranks > 1) {
        c_mine = 0;
        for (int level = c; level >= 1; --level)                   // rm < 2^28, so 32-bit arithmetic is exact
            c_mine += (((uint32_t)rm + (uint32_t)level + step) % (uint32_t)d.nranks) == (uint32_t)d.rank;
    }
}

// fill_reservoirs 'one_to_one' (Population.py:457-489): one particle in for every particle that left through the
// reservoir at the previous step (all ranks; nleave_prev is written by the update after the all-reduce).  One thread
// per candidate: owner test, mode from the cumulative enter_prob (np.searchsorted :472), record (i << 40 | rm << 12)
// appended to the inbox of the segment that owns the mode.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_emit_one_to_one(NkDev d, uint32_t step) {
    if (d.halt[0]) return;
    int64_t total = 0;
    for (int r = 0; r < d.R; ++r) total += d.nleave_prev[r];
    for (int64_t c = (int64_t)blockIdx.x * NK_WG + threadIdx.x; c < total; c += (int64_t)gridDim.x * NK_WG) {
        int r = 0;
        int64_t i = c;
        while (r < d.R - 1 && i >= d.nleave_prev[r]) { i -= d.nleave_prev[r]; ++r; }
        if (((uint32_t)((uint64_t)i + step) % (uint32_t)d.nranks) != (uint32_t)d.rank) continue;
        const uint64_t pid = ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)r << 32) | (uint64_t)i;
        double um, u1;
        nk_uniform2_dev(d.seed, pid, step, NK_TAG_DICE, um, u1);
        int m = nk_ss_left(d.res_roulette + (int64_t)r * d.M, d.M, um);
        m = m > d.M - 1 ? d.M - 1 : m;
        if (i >= (1ll << 24)) { atomicOr(d.overflow, 16); continue; }          // one_to_one index beyond 2^24
        int seg, lidx;
        if (d.part) nk_mode_home(d, m, seg, lidx); else seg = m % d.nseg;
        const int at = atomicAdd(d.sp_inbox_n + seg, 1);
        if (at < d.sp_icap) d.sp_inbox[(int64_t)seg * d.sp_icap + at] = ((uint64_t)i << 40) | ((uint64_t)((int64_t)r * d.M + m) << 12);
        else atomicOr(d.overflow, 8);                                           // inbox full
    }
}

// add_reservoir_particles (Population.py:535-536): an entering particle that left its reservoir at (x0, y0, z0) dt_in before the
// end of the step and meets its first boundary after tc: where it stands at the end of the step, and its timesteps to that boundary.
__device__ __forceinline__ void nk_newborn_place(const NkDev &d, double x0, double y0, double z0, double vx, double vy, double vz,
                                                 double dt_in, double tc, double &x, double &y, double &z, double &nts) {
    x = x0 + vx * dt_in; y = y0 + vy * dt_in; z = z0 + vz * dt_in;       // :536
    nts = tc / d.dt - dt_in / d.dt;                                      // :535
}

// Reservoir emission as its own (small) kernel: fill_reservoirs + add_reservoir_particles for the modes a segment owns.
// A wave evaluates its segment's (reservoir, mode) entries 128 at a time ('one_to_one': reads the segment's inbox), builds
// the entering particles in whole tiles (Mesh.sample_surface, Mesh.py:923-951; entry times Population.py:391-394 /
// :440-443; add_reservoir_particles :525-552) and appends them BEHIND the segment's live particles, marked newborn; the
// sweep of the same step takes them in (tally, boundary events) without relaxing or drifting them.
// BOX (box store): no first ray cast -- the sweep reads a newborn particle's first event off its position like anyone's.
// ahead: the emission runs in the tail launch of the step before (k_tail), i.e. before that step's update has decided on a
// halt.  If that step's sweep has asked for one (halt[1]), this emission will be run again after the store has grown: a
// segment it cannot fit into now is not a loss and must not raise the (sticky) overflow word.
// One segment's emission, by the wave that owns it: evaluates the segment's (reservoir, mode) entries, builds the entering
// particles behind the segment's live ones, writes seg_new / seg_bound; returns the number appended.  The five scratch arrays
// hold NK_EMIT_CHUNK entries each and belong to the calling wave (k_emit: its slice of the emission scratch; a sweep that does
// its own emission: the wave's carry, which is empty then).
template <int GEOM, bool BOX>
__device__ __forceinline__ int nk_emit_one(const NkDev &d, const NkLds &L, uint32_t step, int seg, int lane, unsigned int *sp_pref,
                                           unsigned int *sp_cnt, unsigned int *sp_rm, double *sp_cv, double *sp_pr, bool ahead, bool to_queue,
                                           int &bound_out
#ifdef NK_STAMPS
                                           , unsigned long long em_t0, unsigned long long em_t1, unsigned long long &em_t2
#endif
) {
    const int plo = d.seg_lo ? d.seg_lo[seg] : 0;                 // (the live particles begin at slot seg_lo of the segment's range, NkDev)
    const int64_t base = (int64_t)seg * d.segcap;
    const int count = d.seg_count[seg];
    const NkSegModes sm = nk_seg_modes(d, seg);
    const int nent = d.res_gen != 2 ? d.R * sm.nl : 0;
    int made = 0;                                 // particles appended so far
    int sp_bound = 0;                             // per-lane partial sums of the entries' upper bounds
    for (int e0 = 0; e0 < nent || (d.res_gen == 2 && e0 == 0); e0 += NK_EMIT_CHUNK) {
        int spn = 0;
        if (d.res_gen == 2) {
            spn = d.sp_inbox_n[seg];
            spn = spn < d.sp_icap ? spn : d.sp_icap;
            sp_bound = lane == 0 ? 2 * spn + 64 : 0;
        } else {
            // ---- this chunk of the segment's (reservoir, mode) entries: entry e = r * nl + l, two per lane
            unsigned int run = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = e0 + k * 64 + lane;
                int c = 0, cmine = 0;
                double cv = 0.0, prob = 0.0;
                unsigned int rm32 = 0, rl = 0;
                if (e < nent) {
                    const int r = e / sm.nl, l = e - r * sm.nl;
                    rl = ((unsigned int)r << 12) | ((unsigned int)l << 18);      // c < 4096, R <= 64, l < 2^14
                    const int64_t rm = (int64_t)r * d.M + sm.entry_mode(l);
                    const int64_t at = ((int64_t)seg * d.R + r) * d.nlmax + l;
                    prob = d.ep_p[at];
                    nk_emit_entry(d, step, rm, at, prob, c, cmine, cv);
                    rm32 = (unsigned int)rm;
                    sp_bound += ((int)floor(prob) + 1 + d.nranks - 1) / d.nranks;
                }
                unsigned int v = (unsigned int)cmine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
                sp_cnt[k * 64 + lane] = (unsigned int)c | rl;
                sp_rm[k * 64 + lane] = rm32;
                sp_cv[k * 64 + lane] = cv;
                sp_pr[k * 64 + lane] = prob;
                sp_pref[k * 64 + lane] = v + run - (unsigned int)cmine;
                run += __shfl(v, 63, 64);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // LDS is in order per wave; keep the compiler honest
            spn = (int)run;
        }
#ifdef NK_STAMPS
        if (!em_t2) em_t2 = __builtin_amdgcn_s_memrealtime();     // the first chunk of entries is evaluated
#endif
        for (int spj = 0; spj < spn; spj += NK_TILE) {
            const int j = spj + lane;
            if (j >= spn) continue;
            int64_t rm;
            int level, r, idx;
            double prob = 0.0, cval = 0.0;
            uint64_t o2o = 0;
            if (d.res_gen == 2) {
                const uint64_t recd = d.sp_inbox[(int64_t)seg * d.sp_icap + j];
                rm = (int64_t)((recd >> 12) & 0xFFFFFFFull);
                level = 0;                              // 'one_to_one': entry time uniform in the step
                o2o = recd >> 40;
                r = (int)((uint32_t)rm / (uint32_t)d.M);                        // rm < 2^28
                const int mode = (int)((uint32_t)rm - (uint32_t)r * (uint32_t)d.M);
                idx = d.part ? (int)((uint32_t)d.m2s[mode] / (uint32_t)d.nseg) : mode;
            } else {
                // the entry this particle belongs to: the last one whose exclusive prefix is <= j
                int lo = 0, hi = NK_EMIT_CHUNK;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)sp_pref[mid] <= j) lo = mid; else hi = mid; }
                const unsigned int cw = sp_cnt[lo];
                const int q = j - (int)sp_pref[lo], c = (int)(cw & 0xFFFu);
                r = (int)((cw >> 12) & 0x3Fu);
                rm = (int64_t)sp_rm[lo];
                idx = d.part ? (int)(cw >> 18) : (int)((uint32_t)rm - (uint32_t)r * (uint32_t)d.M);
                cval = sp_cv[lo];
                prob = sp_pr[lo];
                // the q-th level this rank owns, counted down from c (nk_emit_entry's order)
                if (d.nranks == 1) level = c - q;
                else {
                    const uint32_t n = (uint32_t)d.nranks;
                    const uint32_t tq = ((uint32_t)d.rank + n - (((uint32_t)rm + step) % n)) % n;   // owned levels = tq mod n
                    const uint32_t top = (uint32_t)c - (((uint32_t)c + n - tq) % n);                // largest owned level <= c
                    level = (int)(top - (uint32_t)q * n);
                }
            }
            const NkMode *rec = sm.rec + idx;
            const double4 ra = *reinterpret_cast<const double4 *>(rec);
            const double E0 = rec->E0;
            const uint64_t pid = level > 0 ? ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)rm << 12) | (uint64_t)level
                                           : ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)r << 32) | o2o;
            double uf, us, ur, ut;
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT, uf, us);
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT + 1, ur, ut);
            const double iprob = level > 0 ? nk_rcp(prob) : 0.0;
            const double dt_in = (level == 0) ? d.dt * ut                               // one_to_one :482
                               : (level == 1) ? d.dt * (1.0 - cval * iprob)
                                              : d.dt * (1.0 - ((double)(level - 1) + ut) * iprob);
            double x0, y0, z0;
            if (GEOM == 1 && d.res_lds) nk_sample_res_face(L.rf_off, L.rf_cdf, L.rf_verts, r, uf, us, ur, x0, y0, z0);
            else nk_sample_res_face(d.res_face_off, d.res_face_cdf, d.res_face_verts, r, uf, us, ur, x0, y0, z0);
            const double omega = ra.x, vx = ra.y, vy = ra.z, vz = ra.w;
            const double occ = nk_be(omega * d.c_hk, E0, L.resT[2 * r + 1], d.invT0);   // Population.py:506
            if (GEOM == 2 && to_queue) {
                // split sweep over a face tree: the particle's first ray cast is a tree walk like any other, and k_events runs
                // those with its lanes interleaved.  The particle goes into the segment's event queue as it stands on the
                // reservoir -- position, entry time in the nts field, the reservoir's facet in the packed word -- and
                // k_events finishes what follows here (nk_newborn_place)
                const int o = made + j;
                if (o < d.segcap) {
                    const int64_t i = base + o;
                    const int rf = d.res_facet[r];
                    const NkFacet &fq = d.facets[rf];
                    // the facet field: the reservoir's facet if the walk may skip the nodes that hold only its faces, else none
                    const int skip = nk_tree_skip(d, rf, fq.cx, fq.cy, fq.cz, fq.nx, fq.ny, fq.nz, x0, y0, z0, vx, vy, vz);
                    d.qx[i] = x0; d.qy[i] = y0; d.qz[i] = z0; d.qocc[i] = occ; d.qnts[i] = dt_in;
                    d.qw0[i] = NK_NEWBORN | ((uint32_t)((skip == rf ? rf : -1) + 1) << d.lb) | (uint32_t)idx;
                    if (d.qpid) d.qpid[i] = pid;
                } else if (!(ahead && d.halt[1])) atomicOr(d.overflow, 1);
                continue;
            }
            double tc = 0.0;
            int facet = -1;
            int skip = NK_TREE_NO_SKIP;               // the particle starts on its reservoir's facet
            if (GEOM == 2 && d.NG > 0) {
                const int rf = d.res_facet[r];
                const NkFacet &fq = d.facets[rf];
                skip = nk_tree_skip(d, rf, fq.cx, fq.cy, fq.cz, fq.nx, fq.ny, fq.nz, x0, y0, z0, vx, vy, vz);
            }
            if (!BOX) NK_RAY(GEOM, d, L, skip, x0, y0, z0, vx, vy, vz, tc, facet);
            const int o = count + made + j;
            if (plo + o < d.segcap) {
                const int64_t i = base + plo + o;
                const NkSlot q = nk_slot(d, i);
                double xa, ya, za, na;
                nk_newborn_place(d, x0, y0, z0, vx, vy, vz, dt_in, tc, xa, ya, za, na);
                d.x.p[q.od] = xa; d.y.p[q.od] = ya; d.z.p[q.od] = za;
                d.occ.p[q.od] = occ;
                if (!BOX) d.nts.p[q.od] = na;
                d.w0.p[q.ow] = BOX ? (NK_NEWBORN | (uint32_t)idx) : (NK_NEWBORN | ((uint32_t)(facet + 1) << d.lb) | (uint32_t)idx);
                if (d.pid) d.pid.p[q.od] = pid;
            } else if (!(ahead && d.halt[1])) atomicOr(d.overflow, 1);         // more entering particles than free slots
        }
        made += spn;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // the next chunk overwrites the scratch
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sp_bound += __shfl_xor(sp_bound, o, 64);
    const int room = d.segcap - plo - count;
    const int put = (GEOM == 2 && to_queue) ? 0 : (made < room ? made : (room > 0 ? room : 0));   // entering particles behind the segment's live ones
    if (lane == 0) {
        if (GEOM == 2 && to_queue) { d.seg_new[seg] = 0; d.seg_evq[seg] = made < d.segcap ? made : d.segcap; }
        else d.seg_new[seg] = put;
        d.seg_bound[seg] = sp_bound;
        if (d.res_gen == 2) d.sp_inbox_n[seg] = 0;
#ifdef NK_STAMPS
        if (d.stamps) { unsigned long long *w = d.stamps + ((int64_t)d.nseg + seg) * 8 + 4; w[0] = em_t0; w[1] = em_t1; w[2] = em_t2; w[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    }
    bound_out = sp_bound;
    return put;
}

template <int GEOM, bool BOX = false>
__device__ __forceinline__ void nk_emit_segments(const NkDev &d, NkLds L, uint32_t step, int bid, int nblocks, bool ahead
#ifdef NK_STAMPS
                                                 , unsigned long long em_t0, unsigned long long em_t1
#endif
) {
#ifdef NK_STAMPS
    unsigned long long em_t2 = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: segment bookkeeping lives in scalar registers
    unsigned int *sp_pref = L.sp_pref + wave * NK_EMIT_CHUNK, *sp_cnt = L.sp_cnt + wave * NK_EMIT_CHUNK,
                 *sp_rm = L.sp_rm + wave * NK_EMIT_CHUNK;
    double *sp_cv = L.sp_cv + wave * NK_EMIT_CHUNK, *sp_pr = L.sp_pr + wave * NK_EMIT_CHUNK;
    const int nwaves = nblocks * (NK_WG / 64);
    const bool to_queue = GEOM == 2 && d.NG > 0 && d.qx != nullptr;      // first casts by k_events (below)
    for (int seg = bid * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        int bound_;
#ifdef NK_STAMPS
        (void)nk_emit_one<GEOM, BOX>(d, L, step, seg, lane, sp_pref, sp_cnt, sp_rm, sp_cv, sp_pr, ahead, to_queue, bound_, em_t0, em_t1, em_t2);
#else
        (void)nk_emit_one<GEOM, BOX>(d, L, step, seg, lane, sp_pref, sp_cnt, sp_rm, sp_cv, sp_pr, ahead, to_queue, bound_);
#endif
    }
}

template <int GEOM, bool BOX = false>
__device__ __forceinline__ void nk_emit_body(const NkDev &d, uint32_t step, unsigned char *smem, int bid, int nblocks, bool ahead = false) {
#ifdef NK_STAMPS
    const unsigned long long em_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (d.halt[0]) return;
    NkLds L;
    nk_lds_setup<GEOM, 1>(d, smem, L);
#ifdef NK_STAMPS
    nk_emit_segments<GEOM, BOX>(d, L, step, bid, nblocks, ahead, em_t0, __builtin_amdgcn_s_memrealtime());
#else
    nk_emit_segments<GEOM, BOX>(d, L, step, bid, nblocks, ahead);
#endif
}

template <int GEOM, bool BOX = false>
__global__ __launch_bounds__(NK_WG) void k_emit(NkDev d, uint32_t step) {
    extern __shared__ __align__(16) unsigned char smem[];
    nk_emit_body<GEOM, BOX>(d, step, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Output ring of a sweep wave: finished particles are staged in LDS and leave for HBM in whole, aligned tiles of 64 --
// every store instruction then writes full cache lines.  (Storing each tile's survivors where the write cursor stands
// writes ragged pieces; the partly written lines at their ends are first fetched from HBM: measured 0.14 GB of reads per
// step on top of the 0.44 GB the particles themselves need.)
template <bool PID>
struct NkOut {
    double *x, *y, *z, *occ, *nts, *pid;
    unsigned int *w0;
    int on, ob, wout;            // staged particles, ring position of the oldest, particles already in HBM
    int64_t m0; int ms, mroom;   // the o-th finished particle goes to slot m0 + ms * o (ms = -1: a DOWN sweep, NkDev::down), o < mroom
    __device__ __forceinline__ void place(int64_t first, int sign, int room) { m0 = first; ms = sign; mroom = room; }
    __device__ __forceinline__ void init(const NkLds &L, int wave) {
        double *p = L.oring + wave * NK_ORING * (PID ? 6 : 5);
        x = p; y = p + NK_ORING; z = p + 2 * NK_ORING; occ = p + 3 * NK_ORING; nts = p + 4 * NK_ORING; pid = PID ? p + 5 * NK_ORING : nullptr;
        w0 = L.oring_w + wave * NK_ORING;
        on = ob = wout = 0;
    }
    // the same in two halves (plain cursor only): the slots are taken now, the stores are issued later (the sweep's tile commit)
    __device__ __forceinline__ int reserve(int rank, int n) { const int o = wout + rank; wout += n; return o; }
    // one slot's fields from ONE address (NkBlock): the stores differ by immediate offsets
    template <bool BOX>
    static __device__ __forceinline__ void put_slot(const NkDev &d, int64_t i, double px, double py, double pz, double pocc, double pnts,
                                                    uint32_t pw0, unsigned long long ppid) {
        typedef NkBlock<!BOX, PID> B;
        double *q = B::slot(d.x.p, i);
        NK_ST(q, px); NK_ST(q + B::O_Y, py); NK_ST(q + B::O_Z, pz); NK_ST(q + B::O_OCC, pocc);
        if (!BOX) NK_ST(q + B::O_NTS, pnts);
        if (PID) NK_ST(reinterpret_cast<unsigned long long *>(q + B::O_PID), ppid);
        NK_ST(B::word(d.x.p, i), pw0);
    }
    template <bool BOX = false>
    __device__ __forceinline__ void store(const NkDev &d, int64_t base, bool put, int o, double px, double py, double pz, double pocc,
                                          double pnts, uint32_t pw0, unsigned long long ppid) {
        if (put) {
            if (o < mroom) put_slot<BOX>(d, m0 + ms * (int64_t)o, px, py, pz, pocc, pnts, pw0, ppid);
            else atomicOr(d.overflow, 2);     // segment full
        }
    }
    // lanes with `put` append their particle (rank = position among them, n = how many); a full tile leaves at once
    template <bool BOX = false>
    __device__ __forceinline__ void push(const NkDev &d, int64_t base, int lane, bool put, int rank, int n, double px, double py,
                                         double pz, double pocc, double pnts, uint32_t pw0, unsigned long long ppid) {
        if (!NK_OUT_RING) {                           // straight to the write cursor
            if (put) {
                const int o = wout + rank;
                if (o < mroom) put_slot<BOX>(d, m0 + ms * (int64_t)o, px, py, pz, pocc, pnts, pw0, ppid);
                else atomicOr(d.overflow, 2);     // segment full
            }
            wout += n;
            return;
        }
        if (put) {
            const int e = (ob + on + rank) & (NK_ORING - 1);
            x[e] = px; y[e] = py; z[e] = pz; occ[e] = pocc; nts[e] = pnts; w0[e] = pw0;
            if (PID) pid[e] = __longlong_as_double((long long)ppid);
        }
        on += n;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (on >= NK_TILE) flush(d, base, lane, NK_TILE);
    }
    __device__ __forceinline__ void flush(const NkDev &d, int64_t base, int lane, int n) {
        if (wout + n <= d.segcap) {
            if (lane < n) {
                const int e = (ob + lane) & (NK_ORING - 1);
                const int64_t i = base + wout + lane;
                d.x[i] = x[e]; d.y[i] = y[e]; d.z[i] = z[e]; d.occ[i] = occ[e]; d.nts[i] = nts[e]; d.w0[i] = w0[e];
                if (PID) d.pid[i] = (unsigned long long)__double_as_longlong(pid[e]);
            }
            wout += n;
        } else if (lane == 0) atomicOr(d.overflow, 2);      // segment full: the tile is dropped
        ob = (ob + n) & (NK_ORING - 1);
        on -= n;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
};

// LDS copies of mode records are read through pointers that carry the LDS address space in their type.  With plain
// pointers the compiler merges "record from LDS" and "record from the table in HBM" into one FLAT load behind a selected
// pointer; a flat load counts in vmcnt, which retires in order, so waiting for the record would also wait for the next
// tile's prefetch issued just before it -- the prefetch would hide nothing.
typedef double nk_v2d __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) nk_v2d *NkLdsRec;
__device__ __forceinline__ NkLdsRec nk_lds_rec(const double2 *q) { return (NkLdsRec)(const void *)q; }

// The sweep: persistent WAVES, each taking segments w, w + n_waves, ...  A wave owns its segment, so the loop needs no
// workgroup barrier: the four waves of a workgroup only share the read-only tables and the tally bins (LDS atomics).
// Per segment ONE loop over 64-particle tiles (the particles that were there, then the newborn ones k_emit appended), and
// one empty tile that drains the carry.  Every tile ends at the same commit site: final particles are tallied and stored
// compacted at the write cursor, particles that meet a boundary inside the step join the carry, and whenever the carry
// holds 64 the whole wave runs one boundary event for each.
// The segment's mode records are staged in LDS first (<= NK_LREC of them): the loop's only vector-memory operations are
// then the prefetch of the next tile and the stores of the current one.  That matters beyond bandwidth: vmcnt retires
// in order, so a gather issued after a tile's stores would wait for the stores' acknowledgement before its data counts as
// returned -- a full memory round trip per tile in front of the arithmetic.
#ifndef NK_SWEEP_OCC
#define NK_SWEEP_OCC 3          // workgroups per CU the sweep is compiled for (3 x 4 waves = 3 waves per SIMD)
#endif
// Residency per instantiation: a register spilled inside the tile loop is reloaded through the same in-order memory queue
// as the next tile's prefetch, so the reload waits for the prefetch (stamps: every section of the loop 2.5 x slower on the
// rough box at 33 spilled VGPRs).  The variants that need more than 168 VGPRs therefore run two workgroups per CU without
// spills rather than three with.
// SPLIT (large meshes): the sweep only streams -- a particle that meets a boundary goes to its segment's event queue in HBM
// (coalesced, one extra round trip for those particles) and k_events runs the events, whose tree walks are chains of
// dependent loads, at twice the residency; the fused form keeps the events in registers (boxes: a third of the particles
// have one every step, a second trip through HBM would cost more than the residency gains).
#ifndef NK_PREFETCH2
#define NK_PREFETCH2 0
#endif
#ifndef NK_PARK_TAU
#define NK_PARK_TAU 1           // the sweep keeps the lifetime window's five doubles in vector registers (NkTauWin)
#endif
#ifndef NK_PRIO_ROT
#define NK_PRIO_ROT 0           // log2 of the tiles between two steps of the sweep's issue-priority rotation (0 = off)
#endif
#ifndef NK_DEFER_STORE
#define NK_DEFER_STORE 0      // measured: no gain (profiles/r03_notes.txt); the switch stays in the source
#endif
#ifndef NK_SWEEP_OCC_BIG
#define NK_SWEEP_OCC_BIG 2      // the variants with rough facets, RBF temperatures or large meshes (more than 168 VGPRs)
#endif
#ifndef NK_SWEEP_OCC_SPLIT
#define NK_SWEEP_OCC_SPLIT 4
#endif
// workgroups per CU an instantiation is compiled for AND launched at (nk_sweep_blocks caps the grid there: with the machine
// LICM off the plain sweep needs 128 VGPRs and the hardware would take four, which measured no faster than three -- more,
// shorter segments -- and less evenly): rough facets on a small mesh fit three (166 VGPRs), RBF temperatures and large
// meshes in the fused form two.
#define NK_SWEEP_BOUND(GEOM, ROUGH, RBF, SPLIT) ((SPLIT) ? NK_SWEEP_OCC_SPLIT : (((GEOM) == 2 || (RBF)) ? NK_SWEEP_OCC_BIG : NK_SWEEP_OCC))
// FAST: the commonest configuration compiled without the general branches -- slice subvolumes, 'nearest' (1) or 'linear' (2)
// particle temperatures, local reference temperature: the run-time switches become constants of a copy of the parameter
// block (worth 2-3 % of the sweep: fewer instructions in every classification, interpolation and tally).
// BOX (axis-aligned box meshes, NkDev::box): the store holds no cached next hit -- no nts field to stream, no facet bits; a
// particle has an event when its end-of-step position lies beyond a wall it flies towards (nk_box_out), and the event pass
// evaluates that hit itself (nk_box_first_hit) before it runs the event.  72 B moved per phonon-step instead of 88.
// The body between the LDS set-up and the flush of the tally row, for workgroup `wg` of `nwg` (k_sweep: blockIdx.x of gridDim.x;
// the resident kernel of small ensembles calls it once per step from its own loop, k_resident).
// (Tried in round 4 and dropped: the wave making its segment's entering particles itself before it sweeps the segment -- nk_emit_one
// with the carry as scratch, no k_emit launch, a reduce-only tail.  The emission's scalars stay live across the tile loop: 229
// instead of 91 scalar registers parked in vector lanes, v_readlane in the tile loop 42 -> 153, and vector registers in scratch.)
template <int GEOM, bool ROUGH, bool RBF, bool PID, bool SPLIT, bool LREC, int FAST = 0, bool BOX = false>
__device__ __forceinline__ void nk_sweep_body(const NkDev &d, NkLds L, uint32_t step, int do_relax, int flags, int wg, int nwg
#ifdef NK_STAMPS
                                              , unsigned long long st_entry_r
#endif
) {
    static_assert(!BOX || (GEOM == 1 && !SPLIT && !NK_PREFETCH2 && !NK_OUT_RING), "the box store goes with the fused sweep over LDS tables");
    const bool do_flux = (flags & 1) != 0;          // flags: 1 = heat-flux step
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: segment bookkeeping lives in scalar registers
    const int rep = lane & (NK_NREP - 1);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    NkBoxWalls bw;
    if (BOX) bw.load(d);
    NkTauWin tw;
    tw.load(d, NK_PARK_TAU != 0);
    double2 *lrec = reinterpret_cast<double2 *>(L.lrec) + wave * d.nlrec * NK_LREC_STRIDE;
    // the wave's carry in LDS: x y z occ nts cts [64] each, then (ids) pid [64], then the words w0, evc and (ids) gm [64] each
    double *const cX = L.carry ? L.carry + wave * NK_CARRY_DOUBLES(PID ? 3 : 2) : nullptr;
    double *const cP = cX + 6 * 64;
    uint32_t *const cW = reinterpret_cast<uint32_t *>(cX + (PID ? 7 : 6) * 64);
    const int nwaves = nwg * (NK_WG / 64);
#if NK_PRIO_ROT
    const int prio_phase = (int)(wg / (nwg > 3 ? (nwg + 3) / 4 : 1));     // quarter of the grid = dispatch age
#endif
    for (int seg = wg * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int nnew = d.R > 0 ? d.seg_new[seg] : 0;
        const int count = d.seg_count[seg] + nnew;
        if (lane == 0 && nnew > 0) atomicAdd(&L.bins.misc[0], (unsigned int)nnew);          // "emitted" column
        const NkSegModes sm = nk_seg_modes(d, seg);
        constexpr bool use_lrec = LREC;             // host: the modes are partitioned and every segment's share fits (nk_want_lrec)
        if (use_lrec) {
            for (int r0 = lane; r0 < sm.nl; r0 += 64) {
                const double4 *g = reinterpret_cast<const double4 *>(sm.rec + r0);
                const double4 a = g[0], b = g[1];
                double2 *q = lrec + r0 * NK_LREC_STRIDE;
                q[0] = make_double2(a.x, a.y); q[1] = make_double2(a.z, a.w); q[2] = make_double2(b.x, b.y); q[3] = make_double2(b.z, b.w);
                if (NK_LREC_STRIDE > 4 && (ROUGH || PID)) *reinterpret_cast<int *>(q + 4) = sm.mode(r0);   // the mode itself, in the record's spare 16 bytes
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        // global mode of a stored index: from the LDS copy of the record where there is one (no memory access in the tile loop)
        auto mode_of = [&](int i) -> int {
            if (use_lrec && NK_LREC_STRIDE > 4) return *(const __attribute__((address_space(3))) int *)(const void *)(lrec + i * NK_LREC_STRIDE + 4);
            return sm.mode(i);
        };
        // the slots that hold the segment's particles, [lo, hi) of its range, and the order of the walk (NkDev::seg_lo / down): tiles
        // = aligned blocks of 64 slots, from the one with the first particle up or from the one with the last particle down
        constexpr bool ALT = GEOM == 1 || SPLIT;       // (the fused sweep over a face tree always walks upwards from slot 0)
        const bool down = ALT && d.down != 0;
        const int lo = (ALT && d.seg_lo) ? d.seg_lo[seg] : 0, hi = lo + count;
        const int blo = lo & ~(NK_TILE - 1), bhi = count > 0 ? ((hi - 1) & ~(NK_TILE - 1)) : blo;
        const int nA = count > 0 ? (bhi - blo) / NK_TILE + 1 : 0;
        // an UP sweep packs upwards from lo -- from 0 once lo has used up half of the head room; a DOWN sweep downwards from hi - 1
        const int wlo = (!down && lo > (d.segcap - count) / 2) ? 0 : lo;
        NkOut<PID> O;                                 // finished particles on their way back to the segment
        O.init(L, wave);
        O.place(down ? base + hi - 1 : base + wlo, down ? -1 : 1, down ? count : d.segcap - wlo);
        int qn = SPLIT ? d.seg_evq[seg] : 0;          // SPLIT: entries in the segment's event queue (k_emit may have put the entering particles there:
        if (SPLIT && lane == 0 && qn > 0) atomicAdd(&L.bins.misc[0], (unsigned int)qn);     //  they count as "emitted" here)
        int cn = 0;                                   // particles in the carry (lanes [0, cn))
#ifdef NK_STAMPS
        unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
        const unsigned long long st_c0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();   // shader clock against the 100 MHz counter
#endif
        // The tiles come through two register sets that take turns (NkTileBuf): a set is read once its tile has arrived and at once
        // refilled with the tile two turns ahead (NK_PREFETCH2; else one set, one tile ahead).  Nothing ever copies a set that a
        // load is still to fill -- such a copy waits for that load, which is what made an earlier "second prefetched tile" (one set
        // shifted into the other every iteration) a second tile in name only: the compiler had to drain the counter at the top of
        // every iteration (s_waitcnt vmcnt(0)).  Every turn issues the same loads (beyond the segment's particles the address is
        // clamped into its slots and the lanes are masked), after the deferred stores of the previous turn: "all but the newest
        // loads" (vmcnt(6), (7) with ids) then leaves exactly the other set's tile in flight.
        constexpr bool PF2 = NK_PREFETCH2 || SPLIT;
        constexpr bool DEFER = (NK_DEFER_STORE || PF2) && !NK_OUT_RING;
        struct NkTileBuf { uint32_t w0; double x, y, z, occ, nts; unsigned long long pid; };
        NkTileBuf bA = {0u, 0, 0, 0, 0, 0, 0ull}, bB = bA;
        auto fetch = [&](NkTileBuf &b, int r) {
            int rr = down ? bhi - r : blo + r;                    // r = 64 x the tile's number in the walk
            rr = rr < 0 ? 0 : (rr < d.segcap - NK_TILE ? rr : d.segcap - NK_TILE);
            const int64_t i0 = base + rr;
            if (PF2) {
                // Loads the compiler does not know to be loads: it keeps the memory counter itself, and with the event pass's loops
                // and branches between a load and its use it gives up and drains the counter (vmcnt(0)) at the top of every turn.
                // These leave the counting to `arrived` below.  What the compiler issues itself only makes its own waits stricter
                // (the counter retires in order), and it never touches a set between its loads and the wait: the set is live, and
                // only `arrived` reads it (tests/test_isa_inflight.py walks the assembly for any instruction that does; a copy of a
                // set in flight would also be garbage in every GPU parity test).
                typedef NkBlock<!BOX, PID> B;
                const double *px = B::tile(d.x.p, i0, lane), *py = px + B::O_Y, *pz = px + B::O_Z, *po = px + B::O_OCC, *pn = px + B::O_NTS;
                const uint32_t *pw = B::word(d.x.p, i0) + lane;
                asm volatile("global_load_dword %0, %1, off" : "=v"(b.w0) : "v"(pw) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.x) : "v"(px) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.y) : "v"(py) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.z) : "v"(pz) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.occ) : "v"(po) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.nts) : "v"(pn) : "memory");
                if (PID) { const double *pp = px + B::O_PID; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b.pid) : "v"(pp) : "memory"); }
                return;
            }
            if (rr + lane >= lo && rr + lane < hi && r < nA * NK_TILE) {   // (one tile ahead, the compiler's own counting: only what will be used)
                typedef NkBlock<!BOX, PID> B;
                const double *q = B::tile(d.x.p, i0, lane);     // ONE address; the fields at immediate offsets
                b.w0 = NK_LD(B::word(d.x.p, i0) + lane); b.x = NK_LD(q); b.y = NK_LD(q + B::O_Y); b.z = NK_LD(q + B::O_Z); b.occ = NK_LD(q + B::O_OCC);
                if (!BOX) b.nts = NK_LD(q + B::O_NTS);
                if (PID) b.pid = NK_LD(reinterpret_cast<const unsigned long long *>(q + B::O_PID));
            }
        };
        // the set's tile is there: everything but the loads issued last -- the other set's -- has retired
        auto arrived = [&](NkTileBuf &b) {
            if (!PF2) return;
            if (PID) asm volatile("s_waitcnt vmcnt(7)" : "+v"(b.w0), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.occ), "+v"(b.nts), "+v"(b.pid) : : "memory");
            else asm volatile("s_waitcnt vmcnt(6)" : "+v"(b.w0), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.occ), "+v"(b.nts) : : "memory");
        };
        auto drained = [&](NkTileBuf &b) {
            if (PF2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b.w0), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.occ), "+v"(b.nts), "+v"(b.pid) : : "memory");
        };
        fetch(bA, 0);
        if (PF2) fetch(bB, NK_TILE);
        // The finished particles of a tile are STORED at the top of the next iteration, between the wait for that iteration's
        // tile and the next prefetch (NK_DEFER_STORE): a wave's memory counter retires in order, so the wait for a tile also
        // waits for every store issued after its loads -- stores issued at the end of the previous iteration are the youngest
        // operations in the queue and their acknowledgement is a full memory round trip; issued an iteration earlier they
        // have long retired.  Their slots are taken at once (reserve), so the event pass appends behind them as before.
        bool sdone = false, sev = false;
        int so = 0, sq = 0;
        double sx = 0, sy = 0, sz = 0, socc = 0, snts = 0;
        uint32_t sw0 = 0u;
        unsigned long long spid = 0;
        auto turn = [&](NkTileBuf &buf, const int t) __attribute__((always_inline)) {
            const bool flush = t == nA;               // one empty tile: drains the carry
            arrived(buf);
#if NK_PRIO_ROT
            // The SIMD's issue arbiter prefers the OLDEST wave: of the workgroups resident on a CU the first-dispatched one runs
            // ahead and the last one behind (stamps: tile loops of 155 / 165 / 179 / 196 us by quarter of the grid for equal
            // work), and a sweep ends with its slowest wave.  Every wave therefore rotates its issue priority with the tiles,
            // the phase taken from its workgroup's position in the dispatch order: over a segment every wave has held every level.
            if ((t & ((1 << NK_PRIO_ROT) - 1)) == 0) {
                switch (((t >> NK_PRIO_ROT) + prio_phase) & 3) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
#endif
            bool act = false;
            double x = 0, y = 0, z = 0, occ = 0, nts = 0, omega = 0, E0 = 0, vx = 0, vy = 0, vz = 0;
            uint32_t w0 = 0u;
            unsigned long long pid = 0;
            if (!flush) {
                // ---- relax (deferred from the previous step), drift
                const int r = t * NK_TILE;
                { const int rr = down ? bhi - r : blo + r; act = rr + lane >= lo && rr + lane < hi; }
                w0 = buf.w0; x = buf.x; y = buf.y; z = buf.z; occ = buf.occ; nts = buf.nts; pid = buf.pid;
            }
            if (DEFER) {                                // the previous tile's finished particles leave now
                asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(occ), "+v"(nts), "+v"(w0));   // behind the wait for this tile
                O.template store<BOX>(d, base, sdone, so, sx, sy, sz, socc, snts, sw0, spid);
                sdone = false;
                if (SPLIT && sev) {                     // ... and its event particles, for the queue
                    if (sq < d.segcap) {
                        const int64_t i = base + sq;
                        d.qx[i] = sx; d.qy[i] = sy; d.qz[i] = sz; d.qocc[i] = socc; d.qnts[i] = snts; d.qw0[i] = sw0;
                        if (PID) d.qpid[i] = spid;
                    }
                }
                sev = false;
            }
            if (PF2) fetch(buf, (t + 2) * NK_TILE);   // every turn, also the empty one: the count of loads in flight is fixed
            if (!flush) {
                const int r = t * NK_TILE;
                if (!PF2) fetch(buf, r + NK_TILE);
                const bool newborn = (w0 & NK_NEWBORN) != 0u;
                w0 &= ~NK_NEWBORN;
                const int idx = act ? (int)(w0 & lbmask) : 0;
                double4 ra, rb;                                                      // {omega, v} {E0, tau rows}
                if (use_lrec) {
                    const NkLdsRec q = nk_lds_rec(lrec + idx * NK_LREC_STRIDE);
                    const nk_v2d q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                    ra = make_double4(q0.x, q0.y, q1.x, q1.y); rb = make_double4(q2.x, q2.y, q3.x, q3.y);
                }
                else { const double4 *g = reinterpret_cast<const double4 *>(sm.rec + idx); ra = g[0]; rb = g[1]; }
                omega = ra.x; vx = ra.y; vy = ra.z; vz = ra.w; E0 = rb.x;
#ifdef NK_STAMPS
                { const double fence_ = x + omega; asm volatile("" ::"v"(fence_)); }   // the tile's data and record have arrived
#endif
                NK_STAMP(0);
                if (act && !newborn) {
                    if (do_relax) occ = nk_relax<RBF>(d, L, tw, ra, rb, x, y, z, occ, sm, idx);
                    x += vx * d.dt; y += vy * d.dt; z += vz * d.dt;                 // drift, Population.py:793
                    if (!BOX) nts -= 1.0;                                           // :795
                }
#ifdef NK_STAMPS
                { const double fence_ = x + occ; asm volatile("" ::"v"(fence_)); }
#endif
                NK_STAMP(1);
            }
            // ---- commit: final particles -> tally + compacted store; boundary particles -> the carry
            // (box store: the reference's "n_timesteps < 0" read off the position; a particle whose last cast missed -- the
            // reference's n_timesteps = inf -- never has an event)
            const bool ev = BOX ? (act && (w0 & NK_LOST) == 0u && nk_box_out(bw, x, y, z, vx, vy, vz)) : (act && nts < 0.0);
            const bool done = act && !ev;
            const unsigned long long mD = __ballot(done), mE = __ballot(ev);
            if (done) nk_tally_one(d, L.tb, L.bins, x, y, z, occ, omega, E0, vx, vy, vz, do_flux, rep);
            if (DEFER) {
                so = O.reserve(nk_rank(mD), __popcll(mD));
                sdone = done; sx = x; sy = y; sz = z; socc = occ; snts = nts; sw0 = w0; spid = pid;
            } else O.template push<BOX>(d, base, lane, done, nk_rank(mD), __popcll(mD), x, y, z, occ, nts, w0, pid);
            NK_STAMP(2);
            if (SPLIT) {                               // the tile's event particles leave for the queue; k_events takes over
                if (DEFER) { sev = ev; sq = qn + nk_rank(mE); }     // (a particle is final or has an event: the same held values)
                else if (ev) {
                    const int o = qn + nk_rank(mE);
                    if (o < d.segcap) {
                        const int64_t i = base + o;
                        d.qx[i] = x; d.qy[i] = y; d.qz[i] = z; d.qocc[i] = occ; d.qnts[i] = nts; d.qw0[i] = w0;
                        if (PID) d.qpid[i] = pid;
                    }
                }
                qn += __popcll(mE);
                return;
            }
            // ---- drain (Population.py:1546-1683): the tile's event particles are parked in the wave's carry (LDS); whenever it
            // holds 64 (or on the last, empty tile) the whole wave runs one boundary event per particle; finished particles
            // are tallied and appended, absorbed ones vanish, the few that meet another wall go back into the carry
            const int pn = __popcll(mE), erank = nk_rank(mE);
            int taken = 0;                            // event particles of this tile already parked
            while (taken < pn || (flush && cn > 0)) {
                if (taken < pn) {
                    const int k = pn - taken < 64 - cn ? pn - taken : 64 - cn;
                    if (ev && erank >= taken && erank < taken + k) {
                        const int sl = cn + erank - taken;
                        cX[sl] = x; cX[64 + sl] = y; cX[128 + sl] = z; cX[192 + sl] = occ; cX[256 + sl] = nts; cX[320 + sl] = 0.0;
                        cW[sl] = w0; cW[64 + sl] = 0u;
                        if (PID) { cP[sl] = __longlong_as_double((long long)pid); cW[128 + sl] = (uint32_t)mode_of((int)(w0 & lbmask)); cW[192 + sl] = (w0 & lbmask) * (uint32_t)d.nseg + (uint32_t)seg; }
                    }
                    cn += k; taken += k;
                }
                if (cn < 64 && !(flush && taken == pn)) break;
                NK_STAMP(3);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const bool eact = lane < cn;
                NkParticle p;
                double cts = 0.0;
                uint32_t evc = 0u, cw0 = 0u, cgm = 0u, cslot = (uint32_t)seg;
                unsigned long long cpid = 0ull;
                int st = NK_EV_DEAD;
                p.x = p.y = p.z = p.occ = p.nts = 0.0;
                if (eact) {
                    p.x = cX[lane]; p.y = cX[64 + lane]; p.z = cX[128 + lane]; p.occ = cX[192 + lane]; p.nts = cX[256 + lane]; cts = cX[320 + lane];
                    cw0 = cW[lane]; evc = cW[64 + lane];
                    if (PID) { cpid = (unsigned long long)__double_as_longlong(cP[lane]); cgm = cW[128 + lane]; cslot = cW[192 + lane]; }
                }
                const int idx0 = eact ? (int)(cw0 & lbmask) : 0;
                if (ROUGH) {                              // the carried particle may be in a mode another segment owns
                    p.mode = eact ? (int)cgm : mode_of(0);
                    p.slot = (int)cslot;
                    const NkMode *rec = d.modetab + p.mode;
                    const double4 ra = *reinterpret_cast<const double4 *>(rec);
                    p.omega = ra.x; p.vx = ra.y; p.vy = ra.z; p.vz = ra.w;
                    p.E0 = rec->E0;
                } else {
                    if (use_lrec) {
                        const NkLdsRec lq = nk_lds_rec(lrec + idx0 * NK_LREC_STRIDE);
                        const nk_v2d l0 = lq[0], l1 = lq[1];
                        const double4 ra = make_double4(l0.x, l0.y, l1.x, l1.y);
                        p.omega = ra.x; p.vx = ra.y; p.vy = ra.z; p.vz = ra.w;
                        p.E0 = lq[2].x;
                    } else {
                        const double4 ra = *reinterpret_cast<const double4 *>(sm.rec + idx0);
                        p.omega = ra.x; p.vx = ra.y; p.vy = ra.z; p.vz = ra.w;
                        p.E0 = sm.rec[idx0].E0;
                    }
                    p.mode = PID ? (int)cgm : idx0;                          // (nothing reads it without rough facets)
                    p.slot = (int)cslot;
                }
                p.facet = (int)(cw0 >> d.lb) - 1;
                // box store: a particle's FIRST event of the step is the wall it lies beyond (the carry holds no hit for it yet)
                if (BOX && eact && evc == 0u) nk_box_first_hit(bw, d.inv_dt, p.x, p.y, p.z, p.vx, p.vy, p.vz, p.nts, p.facet);
                if (eact) st = nk_event_one<ROUGH, RBF>(d, GEOM == 2 ? d.NG : 0, L.planes, L.faces, L.facets, L.tb, L.resT, L.bins, p, cts, evc, cpid, step,
                                                        (BOX && !NK_BOX_GENERAL_CAST) ? &bw : nullptr);
                const bool alive = eact && st == NK_EV_DONE, more = eact && st == NK_EV_MORE;
#ifdef NK_STAMPS
                { const double fence_ = p.x + p.nts; asm volatile("" ::"v"(fence_)); }
#endif
                NK_STAMP(4);
                if (alive) nk_tally_one(d, L.tb, L.bins, p.x, p.y, p.z, p.occ, p.omega, p.E0, p.vx, p.vy, p.vz, do_flux, rep);
                // the particle's segment after the event: a reflection may have handed it to another one
                bool stay = true;
                uint32_t idxe = (uint32_t)idx0;
                int hseg = seg;                            // the segment that owns the particle's mode now
                if (ROUGH) {
                    if (d.part) { const uint32_t hi = (uint32_t)p.slot / (uint32_t)d.nseg; hseg = (int)((uint32_t)p.slot - hi * (uint32_t)d.nseg); stay = hseg == seg; idxe = hi; }
                    else idxe = (uint32_t)p.mode;
                }
                const uint32_t w0e = ((uint32_t)(p.facet + 1) << d.lb) | idxe;       // (the carry's own format, also in a box store)
                const uint32_t w0s = BOX ? (idxe | (p.facet < 0 ? NK_LOST : 0u)) : w0e;  // what the store keeps
                const bool home = alive && stay, away = alive && !stay;
                const unsigned long long mA = __ballot(home), mM = __ballot(more);
                O.template push<BOX>(d, base, lane, home, nk_rank(mA), __popcll(mA), p.x, p.y, p.z, p.occ, p.nts, w0s, cpid);
                if (ROUGH && away) {                       // one 64-byte record into the inbox of the segment that owns the new mode
                    const int dst = hseg;
                    const int at = atomicAdd(d.mig_n + dst, 1);
                    if (at < d.mig_cap) {
                        double2 *r = d.mig_buf + ((int64_t)dst * d.mig_cap + at) * 4;
                        r[0] = make_double2(p.x, p.y); r[1] = make_double2(p.z, p.occ);
                        r[2] = make_double2(p.nts, __longlong_as_double((long long)cpid));
                        r[3] = make_double2(__longlong_as_double((long long)w0s), 0.0);
                    } else atomicOr(d.overflow, 32);      // inbox full: the particle is lost (k_deliver asks for larger inboxes long before)
                }
                cn = __popcll(mM);
                if (more) {                               // back into the carry, packed (every entry was read above: slots <= lane)
                    const int sl = nk_rank(mM);
                    cX[sl] = p.x; cX[64 + sl] = p.y; cX[128 + sl] = p.z; cX[192 + sl] = p.occ; cX[256 + sl] = p.nts; cX[320 + sl] = cts;
                    cW[sl] = w0e; cW[64 + sl] = evc;
                    if (PID) { cP[sl] = __longlong_as_double((long long)cpid); cW[128 + sl] = (uint32_t)p.mode; cW[192 + sl] = (uint32_t)p.slot; }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                NK_STAMP(5);
            }
            NK_STAMP(3);
        };
        if (PF2) { for (int t = 0; t <= nA; t += 2) { turn(bA, t); if (t + 1 <= nA) turn(bB, t + 1); } drained(bA); drained(bB); }
        else for (int t = 0; t <= nA; ++t) turn(bA, t);
#ifdef NK_STAMPS
        if (lane == 0 && d.stamps) {
            unsigned long long *o = d.stamps + (int64_t)seg * 8;
            for (int k = 0; k < 6; ++k) o[k] = st_acc[k];
            o[6] = (unsigned long long)nA;
            // in-kernel clock: shader cycles per tick of the constant 100 MHz counter over this segment, x 1000 (MHz x 10)
            unsigned long long c1_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1_)::"memory");
            const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();
            o[7] = r1_ > st_r0 ? ((c1_ - st_c0) * 1000ull) / (r1_ - st_r0) : 0ull;
            // wall-clock marks of this wave (10 ns ticks): kernel entry, tile loop begin, tile loop end
            unsigned long long *w = d.stamps + ((int64_t)d.nseg + seg) * 8;
            w[0] = st_entry_r; w[1] = st_r0; w[2] = r1_;
        }
#endif
        if (O.on > 0) O.flush(d, base, lane, O.on);
        const int w = O.wout;
        if (SPLIT && lane == 0) d.seg_evq[seg] = qn < d.segcap ? qn : d.segcap;
        const int nlo = down ? hi - (w < count ? w : count) : wlo;                  // where the live particles begin now
        if (lane == 0) {
            d.seg_count[seg] = w < d.segcap ? w : d.segcap;
            if (ALT && d.seg_lo) d.seg_lo[seg] = nlo;
            // could the next step overflow this segment?  then nothing after this step runs until the host has grown the store
            if (d.R > 0) {
                d.seg_new[seg] = 0;
                if (!SPLIT && (int64_t)nlo + w + d.seg_bound[seg] + NK_TILE > (int64_t)d.segcap) atomicOr(d.halt + 1, 1);
            }
        }
    }
}
template <int GEOM, bool ROUGH, bool RBF, bool PID, bool SPLIT, bool LREC, int FAST = 0, bool BOX = false>
__global__ __launch_bounds__(NK_WG, NK_SWEEP_BOUND(GEOM, ROUGH, RBF, SPLIT)) void k_sweep(NkDev d, uint32_t step, int do_relax, int flags) {
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef NK_STAMPS
    const unsigned long long st_entry_r = __builtin_amdgcn_s_memrealtime();   // 100 MHz, the same counter on every CU
#endif
    if (d.halt[0]) return;                          // an earlier step of this call asked for a larger store (nk_device.h)
    if (FAST) { d.sv_kind = 0; d.sv_interp = FAST - 1; d.T_ref_local = 1; }
    NkLds L;
    nk_lds_setup<GEOM, PID ? 3 : 2>(d, smem, L);
#ifdef NK_STAMPS
    nk_sweep_body<GEOM, ROUGH, RBF, PID, SPLIT, LREC, FAST, BOX>(d, L, step, do_relax, flags, (int)blockIdx.x, (int)gridDim.x, st_entry_r);
#else
    nk_sweep_body<GEOM, ROUGH, RBF, PID, SPLIT, LREC, FAST, BOX>(d, L, step, do_relax, flags, (int)blockIdx.x, (int)gridDim.x);
#endif
    nk_lds_flush(d, L, blockIdx.x);
#ifdef NK_STAMPS
    if (d.stamps && (threadIdx.x & 63) == 0) {
        const int seg0 = blockIdx.x * (NK_WG / 64) + (threadIdx.x >> 6);
        if (seg0 < d.nseg) d.stamps[((int64_t)d.nseg + seg0) * 8 + 3] = __builtin_amdgcn_s_memrealtime();   // the wave is through
    }
#endif
}

// The events of a split sweep (Population.py:1546-1683), at four waves per SIMD: the tree walks are chains of dependent loads.
#ifndef NK_EVENTS_OCC
#define NK_EVENTS_OCC 4          // waves per SIMD; the 5000-triangle wire, ms per step, batch form of the kernel: 2 -> 7.7, 3 -> 7.2, 4 -> 5.9-6.2,
                                 // 5 -> 6.4, 6 -> 7.1, 8 -> 11.3; the state-machine form below: 3 -> 7.7, 4 -> 6.0 (before its other changes)
#endif
// (Tried on the 5000-triangle wire and dropped: one workgroup of 1024 threads per CU that stages the face tree's boxes in LDS
// -- 6.00 against 6.03 ms per step; the walk's box reads are not what the events wait for.)
// Two imbalances shape it.  Lanes finish their walks after very different numbers of visits (5000-triangle wire: 13-27 per
// ray, 31-59 for the slowest of 64), so a wave does not work in batches of 64: every lane is a small state machine (NEED an
// entry -> PRE: the event up to the ray cast -> WALK: one visit per pass -> POST: the cast's result, then either another
// event or tally + append), and the lanes that are through take new entries while the others keep walking; the pre / post
// / refill blocks only run when the walking lanes have dropped to NK_EVENTS_LOW (or no entries are left), so they run
// with many lanes and the walk loop stays full.  And segments own different modes, hence see different event rates
// (5e7 particles in 4096 segments: 353 to 1181 queue entries, mean 668), so waves do not own segments here: the queues are
// one concatenated list (prefix sums by k_events_begin) that all waves draw from through a ticket counter; a finished
// particle takes its slot in its segment with an atomic on the segment's count.  k_events_end closes the step per segment.
// One workgroup of NK_EV_WG threads per NK_EV_WG / (64 x 4 x NK_EVENTS_OCC) of a CU: the larger the workgroup, the more LDS it has for the face
// tree's boxes (1024 threads at four waves per SIMD: the CU's whole 160 KB).
#ifndef NK_EV_WG
#define NK_EV_WG 1024
#endif
#define NK_EV_PER_CU ((NK_EVENTS_OCC * 256) / NK_EV_WG)
#ifndef NK_EVENTS_LOW
#define NK_EVENTS_LOW 16         // (round 4, wire at 5e7: 8 -> 3.636 ms per step, 16 -> 3.577, 24 -> 3.573; NK_EVENTS_LEAVES 16 / 32 / 48: 3.69 / 3.64 / 3.66)
#endif
#ifndef NK_EVENTS_LEAVES
#define NK_EVENTS_LEAVES 32      // lanes holding a noted leaf that make a faces pass worth its instructions
#endif
#ifndef NK_EVENTS_PEND
#define NK_EVENTS_PEND 4         // leaves a lane may note before it has to wait for a faces pass
#endif
#ifndef NK_EVENTS_STUCK
#define NK_EVENTS_STUCK 16       // ... or that many walking lanes unable to go on through the boxes
#endif
#define NK_PH_NEED 0
#define NK_PH_PRE 1
#define NK_PH_WALK 2
#define NK_PH_POST 3
NK_KERNEL_LINKAGE __global__ __launch_bounds__(1024) void k_events_begin(NkDev d) {
    __shared__ int part[1024];
    if (d.halt[0]) return;
    const int t = threadIdx.x, per = (d.nseg + 1023) / 1024;
    const int lo = t * per, hi = lo + per < d.nseg ? lo + per : d.nseg;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += d.seg_evq[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {              // inclusive scan of the 1024 partial sums
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t > 0 ? part[t - 1] : 0;
    int32_t *pf = d.seg_evq + d.nseg;
    for (int i = lo; i < hi; ++i) { pf[i] = run; run += d.seg_evq[i]; }
    if (t == 1023) pf[d.nseg] = part[1023];
    if (t == 0) *d.ev_ticket = 0;
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(256) void k_events_end(NkDev d) {
    if (d.halt[0]) return;
    const int seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= d.nseg) return;
    int w = d.seg_count[seg];
    const int slo = d.seg_lo ? d.seg_lo[seg] : 0;
    if (slo + w > d.segcap) { w = d.segcap - slo; d.seg_count[seg] = w; }   // (the surplus was dropped and flagged by k_events)
    d.seg_evq[seg] = 0;
    // could the next step overflow this segment?  then nothing after this step runs until the host has grown the store
    if (d.R > 0 && (int64_t)slo + w + d.seg_bound[seg] + NK_TILE > (int64_t)d.segcap) atomicOr(d.halt + 1, 1);
}
template <int GEOM, bool ROUGH, bool RBF, bool PID>
__global__ __launch_bounds__(NK_EV_WG, NK_EV_PER_CU) void k_events(NkDev d, uint32_t step, int flags, int row0) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (d.halt[0]) return;
    NkLds L;
    nk_lds_setup<GEOM, 0>(d, smem, L);
    const bool do_flux = (flags & 1) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int rep = lane & (NK_NREP - 1);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    const bool tree = GEOM == 2 && d.NG > 0;
    const int32_t *pf = d.seg_evq + d.nseg;           // exclusive prefix sums of the queue lengths
    const int total = pf[d.nseg];
    bool more = total > 0;                            // entries may be left to hand out
    int phase = NK_PH_NEED;
    NkParticle p;
    NkWalk W;
    uint32_t evc = 0u;
    unsigned long long pid = 0;
    double cts = 0.0;
    int idx0 = 0, seg = 0;
    bool first = false;                               // the lane holds an entering particle before its first ray cast (k_emit)
    int skipl = NK_TREE_NO_SKIP;                      // facet whose nodes this lane's walk skips (nk_tree_skip)
    __shared__ int pend[NK_EVENTS_PEND * NK_EV_WG];      // the leaves a lane has noted and not tested yet
    int npend = 0;
    bool wdone = false;                               // the lane's walk is through the boxes (noted leaves may be left)
    p.x = p.y = p.z = p.occ = p.nts = p.omega = p.E0 = p.vx = p.vy = p.vz = 0.0; p.mode = 0; p.facet = -1;
    nk_walk_begin(d, W, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0);
    // the top levels of the face tree's boxes into LDS (as many whole levels as the host found room for): two thirds of a walk's
    // visits, and every one of them a dependent load that the L1 serves in turn with the leaves' misses
    const float4 *tl = reinterpret_cast<const float4 *>(smem + d.tree_lds_off);
    const int tl_fam0 = tree ? d.tree_lds_fam0 : 0x7fffffff;
    if (tree && d.tree_lds_fam0 < d.tree_nfam) {
        float4 *dst = reinterpret_cast<float4 *>(smem + d.tree_lds_off);
        const float4 *src = reinterpret_cast<const float4 *>(d.tree_boxes + (size_t)d.tree_lds_fam0 * NK_TREE_FAMILY_FLOATS);
        const int n4 = (d.tree_nfam - d.tree_lds_fam0) * (NK_TREE_FAMILY_FLOATS / 4);
        for (int i = tid; i < n4; i += NK_EV_WG) dst[i] = src[i];
        __syncthreads();
    }
#ifdef NK_STAMPS
    unsigned long long st_t0, st_walk = 0, st_pass = 0, st_got = 0, st_leaf = 0, st_nleaf = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");
#endif
    for (;;) {
        // ---- lanes without a particle take the next entries of the concatenated queues
        if (more) {
            const unsigned long long mN = __ballot(phase == NK_PH_NEED);
            if (mN != 0ull) {
                const int n = __popcll(mN);
                int g0 = 0;
                if (lane == 0) g0 = atomicAdd(d.ev_ticket, n);
                g0 = __builtin_amdgcn_readfirstlane(g0);
                if (g0 + n >= total) more = false;
                const int g = g0 + nk_rank(mN);
                if (phase == NK_PH_NEED && g < total) {
                    int lo = 0, hi = d.nseg;              // the segment whose queue holds entry g: the last one with pf[s] <= g
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pf[mid] <= g) lo = mid; else hi = mid; }
                    seg = lo;
                    const int64_t i = (int64_t)seg * d.segcap + (g - pf[seg]);
                    p.x = d.qx[i]; p.y = d.qy[i]; p.z = d.qz[i]; p.occ = d.qocc[i]; p.nts = d.qnts[i];
                    const uint32_t w0 = d.qw0[i];
                    if (PID) pid = d.qpid[i];
                    idx0 = (int)(w0 & lbmask);
                    // record from the segments' copy of the table; the mode itself (rough reflections key their tables by it) is
                    // fetched beside it, not in front of it
                    p.mode = d.part ? d.s2m[(int64_t)seg * d.nlmax + idx0] : idx0;
                    p.slot = idx0 * d.nseg + seg;
                    const NkMode *rec = d.part ? d.modetab_p + (int64_t)seg * d.nlmax + idx0 : d.modetab + idx0;
                    const double4 ra = *reinterpret_cast<const double4 *>(rec);
                    p.omega = ra.x; p.vx = ra.y; p.vy = ra.z; p.vz = ra.w;
                    p.E0 = rec->E0;
                    p.facet = (int)((w0 & ~NK_NEWBORN) >> d.lb) - 1;
                    cts = 0.0; evc = 0u;
                    phase = NK_PH_PRE;
                    first = (w0 & NK_NEWBORN) != 0u;
                    skipl = NK_TREE_NO_SKIP;
                    if (first) {
                        // an entering particle as k_emit left it on its reservoir (position, entry time in nts, in the facet field
                        // the facet its walk may skip): its first ray cast is this walk; nk_newborn_place follows in the POST block
                        skipl = p.facet >= 0 ? p.facet : NK_TREE_NO_SKIP;
                        nk_walk_begin(d, W, p.x, p.y, p.z, p.vx, p.vy, p.vz);
                        npend = 0; wdone = false;
                        phase = NK_PH_WALK;
                    }
#ifdef NK_STAMPS
                    st_got += 1;
#endif
                }
            }
        }
        // ---- the event up to its ray cast
        if (phase == NK_PH_PRE) {
            const int st = nk_event_pre<ROUGH, RBF>(d, L.facets, L.tb, L.resT, L.bins, p, cts, evc, pid, step);
            if (st == NK_EV_DEAD) phase = NK_PH_NEED;
            else if (tree) { nk_walk_begin(d, W, p.x, p.y, p.z, p.vx, p.vy, p.vz); npend = 0; wdone = false; skipl = NK_TREE_NO_SKIP; phase = NK_PH_WALK; }
            else {
                nk_find_boundary(L.planes, L.faces, d.NP, d.tol, p.x, p.y, p.z, p.vx, p.vy, p.vz, W.h.t, W.h.facet);
                phase = NK_PH_POST;
            }
        }
        // ---- the walks, until few enough lanes are left in one (all of them, once no entries are left)
        // A lane that reaches a leaf does not wait for a faces pass: it notes the leaf (up to NK_EVENTS_PEND of them, in LDS) and goes on
        // through the boxes; the faces pass runs when enough lanes hold a leaf or cannot go on (list full, or boxes done), every lane
        // taking the leaf it noted last.  A noted leaf is tested later than it would be in the walk proper, so the hits that would
        // have shortened the ray come later and a few more boxes are entered -- the result is the same (earliest hit, lowest face
        // index among equals, whatever the order).  Before: at most one leaf per lane, and up to NK_EVENTS_LEAVES - 1 lanes idle at
        // theirs during every boxes pass (22 of 64 lanes active in the boxes passes of the 5000-triangle wire).
        {
            const int low = more ? NK_EVENTS_LOW : 0;
#ifdef NK_STAMPS
            unsigned long long st_a, st_b;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_a)::"memory");
#endif
            for (;;) {
                const bool walking = phase == NK_PH_WALK;
                const bool boxes = walking && !wdone && npend < NK_EVENTS_PEND;
                const unsigned long long mW = __ballot(walking), mB = __ballot(boxes), mL = __ballot(walking && npend > 0);
                const int nw = __popcll(mW), nb = __popcll(mB), nl = __popcll(mL);
                if (nw <= low) break;
                if (nb == 0 || nl >= NK_EVENTS_LEAVES || nw - nb >= NK_EVENTS_STUCK) {        // the faces of one noted leaf per lane
#ifdef NK_STAMPS
                    unsigned long long st_c, st_d;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c)::"memory");
#endif
                    if (walking && npend > 0) {
                        --npend;
                        W.leaf = pend[npend * NK_EV_WG + tid];
                        nk_walk_leaf(d, W, p.x, p.y, p.z, p.vx, p.vy, p.vz);
                        if (wdone && npend == 0) phase = NK_PH_POST;
                    }
#ifdef NK_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_d)::"memory");
                    st_leaf += st_d - st_c; st_nleaf += 1;
#endif
                } else {
                    // (the entering particles sit together at the head of every segment's queue, so few passes hold a lane whose walk
                    // skips a facet: those passes alone pay for the facet tags)
                    bool over = false;
                    if (__ballot(boxes && skipl != NK_TREE_NO_SKIP) != 0ull) { if (boxes) over = nk_walk_boxes(d, skipl, W, tl, tl_fam0); }
                    else if (boxes) over = nk_walk_boxes(d, NK_TREE_NO_SKIP, W, tl, tl_fam0);
                    if (boxes) {
                        if (W.leaf >= 0) { pend[npend * NK_EV_WG + tid] = W.leaf; ++npend; W.leaf = -1; }
                        if (over) { wdone = true; if (npend == 0) phase = NK_PH_POST; }
                    }
                }
#ifdef NK_STAMPS
                st_pass += 1;
#endif
            }
#ifdef NK_STAMPS
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_b)::"memory");
            st_walk += st_b - st_a;
#endif
        }
        // ---- the cast's result: another event, or the particle is final
        bool fin = false;
        if (phase == NK_PH_POST) {
            if (first) {                                  // the entering particle's first cast is in: end-of-step position, timesteps to go;
                first = false;                            // inside this step -> its first event, else it is final (as the sweep decides, k_sweep)
                double xa, ya, za, na;
                nk_newborn_place(d, p.x, p.y, p.z, p.vx, p.vy, p.vz, p.nts, W.h.t, xa, ya, za, na);
                p.x = xa; p.y = ya; p.z = za; p.nts = na;
                p.facet = W.h.facet;
                if (na < 0.0) phase = NK_PH_PRE; else { fin = true; phase = NK_PH_NEED; }
            } else {
                const int st = nk_event_post(d, p, cts, evc, W.h.t, W.h.facet);
                if (st == NK_EV_MORE) phase = NK_PH_PRE; else { fin = true; phase = NK_PH_NEED; }
            }
        }
        if (fin) {
            nk_tally_one(d, L.tb, L.bins, p.x, p.y, p.z, p.occ, p.omega, p.E0, p.vx, p.vy, p.vz, do_flux, rep);
            bool stay = true;
            uint32_t idxe = (uint32_t)idx0;
            int hseg = seg;                               // the segment that owns the particle's mode now
            if (ROUGH) {
                if (d.part) { const uint32_t hi = (uint32_t)p.slot / (uint32_t)d.nseg; hseg = (int)((uint32_t)p.slot - hi * (uint32_t)d.nseg); stay = hseg == seg; idxe = hi; }
                else idxe = (uint32_t)p.mode;
            }
            const uint32_t w0e = ((uint32_t)(p.facet + 1) << d.lb) | idxe;
            if (stay) {                                   // its slot in its segment (above the particles the sweep left: NkDev::seg_lo)
                const int slo = d.seg_lo ? d.seg_lo[seg] : 0;
                const int o = atomicAdd(d.seg_count + seg, 1);
                if (slo + o < d.segcap) {
                    const int64_t i = (int64_t)seg * d.segcap + slo + o;
                    const NkSlot q = nk_slot(d, i);
                    d.x.p[q.od] = p.x; d.y.p[q.od] = p.y; d.z.p[q.od] = p.z; d.occ.p[q.od] = p.occ; d.nts.p[q.od] = p.nts; d.w0.p[q.ow] = w0e;
                    if (PID) d.pid.p[q.od] = pid;
                } else atomicOr(d.overflow, 4);
            } else {                                      // (ROUGH) one 64-byte record into the inbox of the segment that owns the new mode
                const int dst = hseg;
                const int at = atomicAdd(d.mig_n + dst, 1);
                if (at < d.mig_cap) {
                    double2 *r = d.mig_buf + ((int64_t)dst * d.mig_cap + at) * 4;
                    r[0] = make_double2(p.x, p.y); r[1] = make_double2(p.z, p.occ);
                    r[2] = make_double2(p.nts, __longlong_as_double((long long)pid));
                    r[3] = make_double2(__longlong_as_double((long long)w0e), 0.0);
                } else atomicOr(d.overflow, 32);
            }
        }
        if (!more && __ballot(phase != NK_PH_NEED) == 0ull) break;
    }
#ifdef NK_STAMPS
    if (lane == 0 && d.stamps) {                      // developer build: this wave's clocks (words 3-5 of a row of its own)
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        const int wv = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
        if (wv < d.nseg) {
            unsigned long long *o = d.stamps + (int64_t)wv * 8;
            o[3] = t1 - st_t0; o[4] = st_walk; o[5] = (st_got << 32) | st_pass; o[6] = (st_nleaf << 40) | st_leaf;
        }
    }
#endif
    nk_lds_flush(d, L, row0 + blockIdx.x);
}

// Box store after alternating sweeps: every segment's particles (and the entering ones appended above them, if the emission has
// run ahead) down to slot 0 of its range (NkDev::seg_lo), which is where every kernel but the sweep and the emission expects them.
// One wave per segment, 64 slots at a time upwards: the destination lies below the source, a chunk is read before it is written.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_anchor(NkDev d, int honor_halt) {
    if (!d.seg_lo || (honor_halt && d.halt[0])) return;       // (inside a batch: a halted batch's remaining launches do nothing)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * (NK_WG / 64);
    for (int seg = blockIdx.x * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        const int lo = d.seg_lo[seg];
        if (lo <= 0) continue;
        const int64_t base = (int64_t)seg * d.segcap;
        int n = d.seg_count[seg] + (d.R > 0 ? d.seg_new[seg] : 0);
        n = lo + n <= d.segcap ? n : d.segcap - lo;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const bool on = j0 + lane < n;
            const NkSlot qs = nk_slot(d, base + lo + j0 + (on ? lane : 0)), qd = nk_slot(d, base + j0 + (on ? lane : 0));
            double vx = 0, vy = 0, vz = 0, vo = 0, vn = 0;
            uint32_t vw = 0u;
            uint64_t vp = 0;
            if (on) {
                vx = d.x.p[qs.od]; vy = d.y.p[qs.od]; vz = d.z.p[qs.od]; vo = d.occ.p[qs.od]; vw = d.w0.p[qs.ow];
                if (d.nts.p) vn = d.nts.p[qs.od];
                if (d.pid.p) vp = d.pid.p[qs.od];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (on) {
                d.x.p[qd.od] = vx; d.y.p[qd.od] = vy; d.z.p[qd.od] = vz; d.occ.p[qd.od] = vo; d.w0.p[qd.ow] = vw;
                if (d.nts.p) d.nts.p[qd.od] = vn;
                if (d.pid.p) d.pid.p[qd.od] = vp;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) d.seg_lo[seg] = 0;
    }
}

// nk_reserve with an unchanged number of segments: every segment's particles move to the start of its longer successor.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_regrow(NkDev o, NkDev n) {
    for (int seg = blockIdx.x; seg < o.nseg; seg += gridDim.x) {
        const int cnt = o.seg_count[seg];
        const int64_t a = (int64_t)seg * o.segcap, b = (int64_t)seg * n.segcap;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            n.x[b + i] = o.x[a + i]; n.y[b + i] = o.y[a + i]; n.z[b + i] = o.z[a + i];
            n.occ[b + i] = o.occ[a + i]; if (o.nts && n.nts) n.nts[b + i] = o.nts[a + i];
            n.w0[b + i] = o.w0[a + i];
            if (o.pid && n.pid) n.pid[b + i] = o.pid[a + i];
        }
        if (threadIdx.x == 0) n.seg_count[seg] = cnt;
    }
}

// Rough facets: the particles whose reflection moved them to a mode of another segment wait in that segment's inbox; after
// the step's update they are appended to the segment (one wave per segment, coalesced stores).  A segment that cannot
// take its migrants keeps them in the inbox and raises the halt word: the host grows the store and delivers again.
// in_step = 1: launched right behind the step's sweep, BEFORE its reduce / update (round 4; so that the next step's emission can ride
// in the tail launch: it appends behind the delivered migrants): a segment that cannot take its migrants leaves them in the inbox,
// k_reduce sees them there and the halt travels with the tally vector.  in_step = 0: launched by the host on its own, after the
// store has grown: raises the halt words itself if something still does not fit.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_deliver(NkDev d, int in_step) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (NK_WG / 64);
    for (int seg = blockIdx.x * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        int n = d.mig_n[seg];
        if (n <= 0) continue;
        if (n > d.mig_cap) n = d.mig_cap;
        const int count = d.seg_count[seg];
        const int lo = d.seg_lo ? d.seg_lo[seg] : 0;          // (box store: the live particles begin at slot seg_lo, NkDev)
        // no room: the migrants stay in the inbox.  In a step k_reduce sees that before the tallies are summed over the ranks, so
        // halt[0] and halt[2] go up on EVERY rank (nk_update_body) and all of them stop after this step
        if (lo + count + n > d.segcap) { if (lane == 0 && !in_step) { atomicOr(d.halt + 2, 1); atomicOr(d.halt, 1); } continue; }
        const int64_t base = (int64_t)seg * d.segcap + lo + count;
        for (int j = lane; j < n; j += 64) {
            const double2 *r = d.mig_buf + ((int64_t)seg * d.mig_cap + j) * 4;
            const double2 a = r[0], b = r[1], c = r[2], e = r[3];
            d.x[base + j] = a.x; d.y[base + j] = a.y; d.z[base + j] = b.x; d.occ[base + j] = b.y; if (d.nts) d.nts[base + j] = c.x;
            d.pid[base + j] = (uint64_t)__double_as_longlong(c.y);
            d.w0[base + j] = (uint32_t)__double_as_longlong(e.x);
        }
        if (lane == 0) {
            d.seg_count[seg] = count + n;
            d.mig_n[seg] = 0;
            // room for the next TWO steps (emission and about as many migrants again): the request travels with this step's
            // tally vector, so that every rank halts at the same step
            const int bound = d.R > 0 ? d.seg_bound[seg] : 0;
            if ((int64_t)lo + count + n + 2 * (bound + 2 * n) + NK_TILE > (int64_t)d.segcap) atomicOr(d.halt + 1, 1);
            if (2 * n > d.mig_cap) atomicOr(d.halt + 3, 1);
        }
    }
}

// One subvolume of the update: normalise the raw energy sum, add the reference E(T_old), invert to T (calculate_energy
// Population.py:719-728, refresh_temperatures :692).  Ta, Tb, Tz, Ea, Ez: first two / last entries of the E(T) table.
__device__ __forceinline__ void nk_update_sv(const NkDev &d, double Eraw, double Ns, double Told, int t, double Ta, double Tb, double Tz,
                                             double Ea, double Ez, double &Tnew_out, double &E_out) {
    const int n = d.nE;
        double norm;
        if (d.norm_fixed) norm = d.active_modes / (d.particle_density * d.sv_volume[t]);
        else { norm = d.active_modes / Ns; if (isnan(norm)) norm = 0.0; }
        double E = Eraw * norm / d.QV;
        int it = (int)((Told - Ta) / (Tb - Ta));               // uniform-grid guess of searchsorted(Tarr, Told)
        it = it < 0 ? 0 : (it > n - 1 ? n - 1 : it);
        int base = it - 1;
        base = base < 0 ? 0 : (base > n - 4 ? n - 4 : base);
        double wT[4], wE[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int q = n >= 4 ? base + j : 0; wT[j] = d.Tarr[q]; wE[j] = d.Earr[q]; }
        // E(T_old), crystal_energy_function
        double ref;
        if (Told < Ta) ref = Ea;
        else if (Told > Tz) ref = Ez;
        else if (n >= 4 && wT[0] < Told && !(wT[3] < Told)) {
            const int c = (wT[1] < Told) + (wT[2] < Told);     // searchsorted-left = base + 1 + c
            const double xlo = c == 0 ? wT[0] : (c == 1 ? wT[1] : wT[2]), xhi = c == 0 ? wT[1] : (c == 1 ? wT[2] : wT[3]);
            const double ylo = c == 0 ? wE[0] : (c == 1 ? wE[1] : wE[2]), yhi = c == 0 ? wE[1] : (c == 1 ? wE[2] : wE[3]);
            ref = (yhi - ylo) / (xhi - xlo) * (Told - xlo) + ylo;
        } else { int io; ref = nk_interp_lin_hint(d.Tarr, d.Earr, n, Told, it, io); }
        E += ref;
        // T(E), temperature_function
        double Tnew;
        if (E < Ea) Tnew = d.Tfill_lo;
        else if (E > Ez) Tnew = d.Tfill_hi;
        else if (n >= 4 && wE[0] < E && !(wE[3] < E)) {
            const int c = (wE[1] < E) + (wE[2] < E);
            const double xlo = c == 0 ? wE[0] : (c == 1 ? wE[1] : wE[2]), xhi = c == 0 ? wE[1] : (c == 1 ? wE[2] : wE[3]);
            const double ylo = c == 0 ? wT[0] : (c == 1 ? wT[1] : wT[2]), yhi = c == 0 ? wT[1] : (c == 1 ? wT[2] : wT[3]);
            Tnew = (yhi - ylo) / (xhi - xlo) * (E - xlo) + ylo;
        } else { int io; Tnew = nk_interp_lin_hint(d.Earr, d.Tarr, n, E, it, io); }
        Tnew_out = Tnew;
        E_out = E;
}
// Normalise, invert E(T), publish the new subvolume temperatures, history row: calculate_energy (Population.py:719-728)
// + refresh_temperatures (:692), run by ONE workgroup.
// History row: acc[NB] | T_sv[S] | E_sv[S] | flux_valid, row_valid, halt, overflow | the four halt words after this step
// acc[NB] / acc[NB + 1] (summed over the ranks like the tallies) > 0: some segment could overflow at the next step / cannot
// take its migrants -> raise the halt word,
// on every rank at the same step.
__device__ __forceinline__ void nk_update_body(const NkDev &d, const double *acc, double *hist_row, int do_flux) {
    // One workgroup, so everything here is a chain of memory latencies: the E(T) / T(E) tables are read as one 4-point
    // window around the old temperature (T moves by a small fraction of the 0.1 K table step per timestep; the general
    // search is the fallback).
    const int tid = threadIdx.x, nth = blockDim.x;
    const int S = d.S, NB = d.NB, n = d.nE;
    const double Ta = d.Tarr[0], Tb = d.Tarr[1], Tz = d.Tarr[n - 1], Ea = d.Earr[0], Ez = d.Earr[n - 1];
    for (int t = tid; t < S; t += nth) {
        double Tnew, E;
        nk_update_sv(d, acc[t], acc[S + t], d.T_ref_local ? d.T_sv[t] : d.T_ref, t, Ta, Tb, Tz, Ea, Ez, Tnew, E);
        hist_row[NB + t] = Tnew;
        hist_row[NB + S + t] = E;
        d.T_sv[t] = Tnew;
    }
    for (int b = tid; b < NB; b += nth) hist_row[b] = acc[b];
    if (d.sv_interp == 3) {                          // RBF coefficients of the new temperatures: [w; p] = inv[:, :S] @ T_sv
        __syncthreads();
        const int P = d.rbf_P;
        for (int j = tid; j < P; j += nth) {
            double a = 0.0;
            for (int i = 0; i < S; ++i) a += d.rbf_inv[(int64_t)j * P + i] * d.T_sv[i];
            d.T_sv[S + j] = a;
        }
    }
    if (d.res_gen == 2) for (int r = tid; r < d.R; r += nth) d.nleave_prev[r] = (int32_t)acc[5 * S + r];
    if (tid == 0) {
        hist_row[NB + 2 * S + 0] = (double)do_flux;
        hist_row[NB + 2 * S + 1] = 1.0;              // this step ran
        const bool hdel = acc[NB + 1] > 0.0;         // on some rank a segment cannot take the migrants waiting in its inbox
        const bool hreq = acc[NB] > 0.0 || hdel;
        hist_row[NB + 2 * S + 2] = hreq ? 1.0 : 0.0;
        hist_row[NB + 2 * S + 3] = (double)*d.overflow;
        if (hreq) d.halt[0] = 1;
        if (hdel) d.halt[2] = 1;
        // the four halt words as this step leaves them, behind the row (the host reads them there: no copy of their own)
        hist_row[NB + 2 * S + 4] = hreq ? 1.0 : 0.0;
        hist_row[NB + 2 * S + 5] = (double)d.halt[1];
        hist_row[NB + 2 * S + 6] = (hdel || d.halt[2]) ? 1.0 : 0.0;
        hist_row[NB + 2 * S + 7] = (double)d.halt[3];
    }
}

// Column sums of the tally rows, fixed order -> bitwise reproducible for a given grid.  One workgroup per column.
// fuse != 0 (single rank): the workgroup that finishes last also runs the update, saving a launch.
// Column NB is the halt request of this step's sweep (halt[1]); halt[0] is only raised by the update at the END of a step,
// so every kernel of a step sees the same value.
__device__ __forceinline__ void nk_reduce_body(const NkDev &d, int rows, double *acc, double *hist_row, int do_flux, int fuse, int b,
                                               int nblocks, double *sh, int &last) {
    if (d.halt[0]) return;
    const int NB = d.NB;
    double v = 0.0;
    for (int r = threadIdx.x; r < rows; r += NK_WG) v += d.partials[(int64_t)r * NB + b];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = NK_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[b] = sh[0];
    if (b == 0) {
        // the two halt requests that travel with the tally vector (summed over the ranks, so that every rank stops at the same
        // step): the sweep's "a segment could overflow at the next step", and -- rough facets -- "a segment cannot take the
        // migrants that wait in its inbox" (what k_deliver will find after this step's update, checked here, before the sum)
        int stuck = 0;
        if (d.mig_n) for (int sg = threadIdx.x; sg < d.nseg; sg += NK_WG) { const int n = d.mig_n[sg]; stuck |= (n > 0 && d.seg_count[sg] + (n < d.mig_cap ? n : d.mig_cap) > d.segcap) ? 1 : 0; }
        stuck = __syncthreads_or(stuck);
        if (threadIdx.x == 0) { acc[NB] = (double)d.halt[1]; acc[NB + 1] = (double)stuck; }
    }
    if (!fuse) return;
    if (threadIdx.x == 0) {
        __threadfence();
        last = (atomicAdd(d.ticket, 1) == nblocks - 1);
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *d.ticket = 0;
    __threadfence();
    nk_update_body(d, acc, hist_row, do_flux);
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_reduce(NkDev d, int rows, double *acc, double *hist_row, int do_flux, int fuse) {
    __shared__ double sh[NK_WG];
    __shared__ int last;
    nk_reduce_body(d, rows, acc, hist_row, do_flux, fuse, (int)blockIdx.x, (int)gridDim.x, sh, last);
}
// The step's tail and the NEXT step's reservoir emission in ONE launch: the first n_reduce workgroups are k_reduce (+ the fused
// update), the others k_emit for step_next.  The emission needs nothing of this step's tally -- only the segments' ends, which
// the sweep has left -- and the reduce / update are latency chains of a hundred workgroups that leave the chip idle: side by
// side they cost the longer of the two (and one launch less).  The update may raise the halt word while the emission of the
// step that will then NOT run is under way: harmless, its counters are double-buffered (NkDev::rc_len) and it is simply run
// again once the store has grown.  Not 'one_to_one' (emits what the update says left).  With rough facets k_deliver runs between the
// sweep and this launch (round 4: it used to run after the update and moved the segments' ends under the emission).
template <int GEOM, bool BOX = false>
__global__ __launch_bounds__(NK_WG) void k_tail(NkDev d, uint32_t step_next, int rows, double *acc, double *hist_row, int do_flux, int fuse,
                                                int n_reduce) {
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int)blockIdx.x < n_reduce) {
        double *sh = reinterpret_cast<double *>(smem);
        int &last = *reinterpret_cast<int *>(smem + NK_WG * sizeof(double));
        nk_reduce_body(d, rows, acc, hist_row, do_flux, fuse, (int)blockIdx.x, n_reduce, sh, last);
    } else {
        nk_emit_body<GEOM, BOX>(d, step_next, smem, (int)blockIdx.x - n_reduce, (int)gridDim.x - n_reduce, true);
    }
}

// =============================================================================== small ensembles: the resident kernel
// Config 1's own size (1e5 particles: 3.6 MB of state) spends a step in launches and latency chains -- k_sweep 22 us + k_tail
// 10 us + the gaps between them for a few microseconds of arithmetic.  Here ONE launch runs many steps: the workgroups stay
// resident, the read-only tables are staged in LDS once, a segment is swept by the same wave at every step (so its particles stay
// in that XCD's L2), and a step is
//     sweep -> the workgroup's tally row to global memory, its flag raised -> the NEXT step's emission (depends on nothing of this
//     step's update, like in k_tail) -> workgroup 0: waits for all flags, sums the rows column by column in a fixed order, inverts
//     E -> T (nk_update_sv), writes the history row, publishes T and the step's number; the others: wait for that number, read T.
// What crosses workgroups is the tally rows and the S temperatures (the S-vector coupling of refresh_temperatures,
// Population.py:685-702) -- all of it through agent-scope atomic loads and stores (write-through / read-through at the level where
// the eight XCDs agree), ordered by "my stores have completed" (a workgroup-scope fence = s_waitcnt) before the flag goes up.
// No __threadfence(): an agent-scope release writes back, an acquire invalidates, the XCD's WHOLE L2 -- the particles with it.
// (Round 4's first version added every workgroup's 111 sums to one accumulator with FP64 atomics and met at a counter: 28 000
// atomics on 111 addresses + 256 on one, every one of them serialised where the XCDs agree, plus those fences: a floor of 75 us
// per step whatever the ensemble.  The version before that: every workgroup summing all G rows itself.)
// Same device functions as the launch-per-step path (nk_emit_segments, nk_sweep_body, nk_update_sv); the rows are summed in a
// fixed order (deterministic), though not in k_reduce's: subvolume energies agree with the launch path to the last bits, not bit
// for bit.
// Conditions (host, nk_step_resident): one rank, no rough facets (their migrants cross segments), small mesh, not 'one_to_one',
// no RBF temperatures; contains_check steps start a new launch.  Every workgroup must be resident: the grid is at most what the
// occupancy query allows.  A wait that is not met within ~2 s gives up (overflow bit 256) instead of hanging the device.
// bar[1]: this launch's update asked for a halt; bar[32 + g]: steps of this launch whose row workgroup g has written; bar[1056 + t]:
// steps of this launch whose temperature of subvolume t is published.  Zeroed by the host before every launch.
__device__ __forceinline__ bool nk_wait_word(const unsigned int *w, unsigned int atleast) {
    int spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < atleast) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 1000000) return false;
    }
    return true;
}
template <bool BOX, bool PID, bool LREC>
__global__ __launch_bounds__(NK_WG, 2) void k_resident(NkDev d, uint32_t step0, int nsteps, int relax0, int flux_every, double *hist, int hrow,
                                                       unsigned int *bar) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (d.halt[0]) return;
    NkLds L;
    nk_lds_setup<1, PID ? 5 : 4>(d, smem, L);
    const int tid = threadIdx.x, wg = (int)blockIdx.x, G = (int)gridDim.x;
    const int S = d.S, R = d.R, NB = d.NB, a = d.sv_axis;
    double *Tsv = const_cast<double *>(L.tb.Tsv);
    NkSv *sv = const_cast<NkSv *>(L.tb.sv);
    const int nE = d.nE;
    const double Ta = d.Tarr[0], Tb = d.Tarr[1], Tz = d.Tarr[nE - 1], Ea = d.Earr[0], Ez = d.Earr[nE - 1];
    unsigned int *flags = bar + 32, *flagsT = bar + 32 + 1024;      // (G <= 1024 workgroups, S <= 128 subvolumes)
    unsigned long long *rows = reinterpret_cast<unsigned long long *>(d.partials);
    unsigned long long *Tpub = reinterpret_cast<unsigned long long *>(d.T_sv);
    const int NBP = (NB + 1) & ~1;                                // a row's length in memory (even: pairs of columns, 16-byte loads)
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(bar + 8);   // developer probe: 100 MHz clock marks of the launch's last step
#ifdef NK_STAMPS
    if (R > 0) nk_emit_segments<1, BOX>(d, L, step0, wg, G, false, 0ull, 0ull);
#else
    if (R > 0) nk_emit_segments<1, BOX>(d, L, step0, wg, G, false);
#endif
    for (int s = 0; s < nsteps; ++s) {
        const uint32_t step = step0 + (uint32_t)s;
        const int do_flux = (flux_every > 0 && ((step + 1u) % (uint32_t)flux_every) == 0u) ? 1 : 0;
        const bool mark = s == nsteps - 1 && tid == 0 && (wg == 0 || wg == G / 2);
        unsigned long long *mk = dbg + (wg == 0 ? 0 : 8);
        if (mark) mk[0] = __builtin_amdgcn_s_memrealtime();
#ifdef NK_STAMPS
        nk_sweep_body<1, false, false, PID, false, LREC, 0, BOX>(d, L, step, s == 0 ? relax0 : 1, do_flux, wg, G, 0ull);
#else
        nk_sweep_body<1, false, false, PID, false, LREC, 0, BOX>(d, L, step, s == 0 ? relax0 : 1, do_flux, wg, G);
#endif
        // ---- this workgroup's tally sums (the row of nk_lds_flush), then its flag
        __syncthreads();
        if (mark) mk[1] = __builtin_amdgcn_s_memrealtime();
        for (int b = tid; b < NB; b += NK_WG) {
            double v = 0.0;
            if (b < S) { for (int r = 0; r < NK_NREP; ++r) v += L.bins.E[r * S + b]; }
            else if (b < 2 * S) { unsigned int c = 0; for (int r = 0; r < NK_NREP; ++r) c += L.bins.N[r * S + (b - S)]; v = (double)c; }
            else if (b < 5 * S) { const int k = b - 2 * S; for (int r = 0; r < NK_NREP; ++r) v += L.bins.flux[r * 3 * S + k]; }
            else if (b < 5 * S + R) v = (double)L.bins.nleave[b - 5 * S];
            else if (b < 5 * S + 2 * R) v = L.bins.resb[4 * (b - 5 * S - R)];
            else if (b < 5 * S + 5 * R) { const int k = b - 5 * S - 2 * R; v = L.bins.resb[4 * (k / 3) + 1 + (k % 3)]; }
            else v = (double)L.bins.misc[0];
            __hip_atomic_store(rows + (size_t)wg * NBP + b, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0 && NBP > NB) __hip_atomic_store(rows + (size_t)wg * NBP + NB, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the row's stores have completed (no cache is written back)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + wg, (unsigned int)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bins back to zero (what nk_lds_setup does at a launch's start): the next step's emission tallies into them
        for (int i = tid; i < NK_NREP * S; i += NK_WG) { L.bins.E[i] = 0.0; L.bins.N[i] = 0u; }
        for (int i = tid; i < NK_NREP * 3 * S; i += NK_WG) L.bins.flux[i] = 0.0;
        for (int i = tid; i < 4 * R; i += NK_WG) L.bins.resb[i] = 0.0;
        for (int i = tid; i < R; i += NK_WG) L.bins.nleave[i] = 0u;
        if (tid == 0) L.bins.misc[0] = 0u;
        __syncthreads();
        if (mark && wg == 0) mk[2] = __builtin_amdgcn_s_memrealtime();
        // ---- the next step's emission, while the rows travel (ahead of this step's update: it may run again after a halt)
#ifdef NK_STAMPS
        if (R > 0 && s + 1 < nsteps) nk_emit_segments<1, BOX>(d, L, step + 1u, wg, G, true, 0ull, 0ull);
#else
        if (R > 0 && s + 1 < nsteps) nk_emit_segments<1, BOX>(d, L, step + 1u, wg, G, true);
#endif
        int ok = 1, hreq = 0;
        if (mark) mk[wg == 0 ? 3 : 2] = __builtin_amdgcn_s_memrealtime();
        // ---- the update, dealt over the workgroups by COLUMN of the tally rows: task k < S = subvolume k (its energy and particle
        // columns -> E -> T, nk_update_sv needs nothing else), task k >= S = one of the other columns (history row only); task k goes to
        // workgroup k % G.  An owner waits for every workgroup's row, sums its column(s) -- one row per thread: ONE round of loads --
        // in a fixed order, and publishes T_k behind its own flag; then everybody waits for the S flags and reads the S
        // temperatures.  (One workgroup summing all 111 columns took four dependent rounds: 6.6 us of a 28 us step.)
        double *hrowp = hist + (size_t)s * hrow;
        const int ntask = S + (NB - 2 * S);
        if (wg < ntask) {
            for (int g = tid; g < G; g += NK_WG) if (!nk_wait_word(flags + g, (unsigned int)(s + 1))) ok = 0;
            ok = __syncthreads_and(ok);
            if (!ok) {          // a workgroup never arrived: give up (everybody times out on the flags below)
                if (tid == 0) atomicOr(d.overflow, 256);
                return;
            }
            if (mark) mk[4] = __builtin_amdgcn_s_memrealtime();
            if (wg == 0) hreq = __hip_atomic_load(d.halt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a sweep of this step asked for head room
            for (int k = wg; k < ntask; k += G) {
                const int c0 = k < S ? k : 2 * S + (k - S), c1 = k < S ? S + k : -1;
                double v0 = 0.0, v1 = 0.0;
                for (int g = tid; g < G; g += NK_WG) {
                    v0 += __longlong_as_double((long long)__hip_atomic_load(rows + (size_t)g * NBP + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (c1 >= 0) v1 += __longlong_as_double((long long)__hip_atomic_load(rows + (size_t)g * NBP + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                // the workgroup's sum in a fixed order: a tree over the thread index
                double *red = L.colsum;
                red[tid] = v0; red[NK_WG + tid] = v1;
                __syncthreads();
                for (int o = NK_WG / 2; o > 0; o >>= 1) {
                    if (tid < o) { red[tid] += red[tid + o]; red[NK_WG + tid] += red[NK_WG + tid + o]; }
                    __syncthreads();
                }
                if (tid == 0) {
                    const double t0 = red[0], t1 = red[NK_WG];
                    if (k < S) {
                        double Tnew, E;
                        nk_update_sv(d, t0, t1, d.T_ref_local ? Tsv[k] : d.T_ref, k, Ta, Tb, Tz, Ea, Ez, Tnew, E);
                        __hip_atomic_store(Tpub + k, (unsigned long long)__double_as_longlong(Tnew), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (k == 0 && hreq) { d.halt[0] = 1; __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // T (and the halt word) are out before the flag
                        __hip_atomic_store(flagsT + k, (unsigned int)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // the history row (pinned host memory: slow writes) behind the flag, off everybody's path
                        hrowp[k] = t0; hrowp[S + k] = t1; hrowp[NB + k] = Tnew; hrowp[NB + S + k] = E;
                        if (k == 0) {                     // (workgroup 0: the step's control words)
                            hrowp[NB + 2 * S + 0] = (double)do_flux;
                            hrowp[NB + 2 * S + 1] = 1.0;
                            hrowp[NB + 2 * S + 2] = hreq ? 1.0 : 0.0;
                            hrowp[NB + 2 * S + 3] = (double)__hip_atomic_load(d.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hrowp[NB + 2 * S + 4] = hreq ? 1.0 : 0.0;
                            hrowp[NB + 2 * S + 5] = (double)hreq;
                            hrowp[NB + 2 * S + 6] = (double)d.halt[2];
                            hrowp[NB + 2 * S + 7] = (double)d.halt[3];
                        }
                    } else hrowp[c0] = t0;
                }
                __syncthreads();
            }
            if (mark) mk[5] = __builtin_amdgcn_s_memrealtime();
        }
        for (int t = tid; t < S; t += NK_WG) {
            if (!nk_wait_word(flagsT + t, (unsigned int)(s + 1))) ok = 0;
            else Tsv[t] = __longlong_as_double((long long)__hip_atomic_load(Tpub + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        ok = __syncthreads_and(ok);
        if (!ok) { if (tid == 0) atomicOr(d.overflow, 256); return; }
        if (wg != 0) hreq = (int)__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hreq = __syncthreads_or(hreq);
        if (mark) mk[wg == 0 ? 6 : 3] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_dcache_inv();                      // (nothing uniform that this kernel writes should sit in the scalar cache; belt and braces)
        __syncthreads();
        // per-subvolume records of the new temperatures
        for (int t = tid; t < S; t += NK_WG) {
            const int jn = t + 1 < S ? t + 1 : t;
            NkSv q;
            q.c = L.tb.cen[3 * t + a]; q.T = Tsv[t];
            q.slope = jn > t ? (Tsv[jn] - Tsv[t]) / (L.tb.cen[3 * jn + a] - L.tb.cen[3 * t + a]) : 0.0;
            q.invT = 1.0 / Tsv[t];
            sv[t] = q;
        }
        __syncthreads();
        if (hreq) return;                                     // the host grows the store and carries on
    }
}

// The update as its own launch (after the RCCL all-reduce when nranks > 1).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_update(NkDev d, const double *acc, double *hist_row, int do_flux) {
    if (d.halt[0]) return;
    nk_update_body(d, acc, hist_row, do_flux);
}

// Stand-alone lifetime_scattering (flushes the deferred relaxation).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_relax(NkDev d, int honor_halt) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (honor_halt && d.halt[0]) return;
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    NkTauWin tw;
    tw.load(d, false);
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        const NkSegModes sm = nk_seg_modes(d, seg);
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const int idx = (int)(d.w0[i] & lbmask);
            const double4 *mrec = reinterpret_cast<const double4 *>(sm.rec + idx);
            const double4 ra = mrec[0], rb = mrec[1];
            d.occ[i] = nk_relax(d, L, tw, ra, rb, d.x[i], d.y[i], d.z[i], d.occ[i], sm, idx);
        }
    }
}

// timesteps_to_boundary for every particle (Population.py:310-314)
// Box store: nothing to cache but the "lost" mark of a particle whose cast misses.  What the store cannot express -- a
// particle that starts outside the box and would meet a wall from behind -- is counted in *anomalies; the host then goes back
// to the cached layout for this context (nk_init_boundaries).
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_init_boundaries(NkDev d, int32_t *anomalies) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, 0>(d, smem, L);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        const NkSegModes sm = nk_seg_modes(d, seg);
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const uint32_t idx = d.w0[i] & lbmask;
            const NkMode *rec = sm.rec + idx;
            double tc; int fc;
            const double x = d.x[i], y = d.y[i], z = d.z[i];
            NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x, y, z, rec->vx, rec->vy, rec->vz, tc, fc);
            if (d.box) {
                const bool inside = x >= d.box_k[0] && x <= -d.box_k[1] && y >= d.box_k[2] && y <= -d.box_k[3] && z >= d.box_k[4] && z <= -d.box_k[5];
                if (!inside && fc >= 0) atomicAdd(anomalies, 1);
                d.w0[i] = (fc < 0 ? NK_LOST : 0u) | idx;
            } else {
                d.nts[i] = tc / d.dt;
                d.w0[i] = ((uint32_t)(fc + 1) << d.lb) | idx;
            }
        }
    }
}
// The next boundary hit of every live particle where it stands, in slot order: what the reference keeps in n_timesteps /
// collision_facets (Population.py:797-830), for the callers that ask for it when the store does not hold it (box store:
// nk_download_particles, re-dealing).  A particle marked lost reports (inf, -1).  A particle behind a wall it flies towards
// from outside (negative entry time, SURVEY quirk list) is cast from where it enters, as the reference cast it from its
// reservoir face, and the flight up to there is added.
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_next_hit(NkDev d, double *nts_out, int32_t *facet_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, 0>(d, smem, L);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        const NkSegModes sm = nk_seg_modes(d, seg);
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const uint32_t w0 = d.w0[i];
            if (w0 & NK_LOST) { nts_out[i] = __builtin_inf(); facet_out[i] = -1; continue; }
            const NkMode *rec = sm.rec + (w0 & lbmask);
            double x = d.x[i], y = d.y[i], z = d.z[i];
            const double vx = rec->vx, vy = rec->vy, vz = rec->vz;
            double tin = 0.0;
            if (vx > 0.0 && x < d.box_k[0]) tin = fmax(tin, (d.box_k[0] - x) / vx);
            if (vx < 0.0 && x > -d.box_k[1]) tin = fmax(tin, (-d.box_k[1] - x) / vx);
            if (vy > 0.0 && y < d.box_k[2]) tin = fmax(tin, (d.box_k[2] - y) / vy);
            if (vy < 0.0 && y > -d.box_k[3]) tin = fmax(tin, (-d.box_k[3] - y) / vy);
            if (vz > 0.0 && z < d.box_k[4]) tin = fmax(tin, (d.box_k[4] - z) / vz);
            if (vz < 0.0 && z > -d.box_k[5]) tin = fmax(tin, (-d.box_k[5] - z) / vz);
            if (tin > 0.0) { x += vx * tin; y += vy * tin; z += vz * tin; }
            double tc; int fc;
            NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x, y, z, vx, vy, vz, tc, fc);
            nts_out[i] = (tc + tin) / d.dt;
            facet_out[i] = fc;
        }
    }
}

// Key of a particle's random draws when ids are not tracked: its mode and the bits of its position (DESIGN.md "RNG").
__device__ __forceinline__ uint64_t nk_state_key(int mode, double x, double y, double z) {
    uint64_t k = (uint64_t)(uint32_t)mode;
    k = k * 0x9E3779B97F4A7C15ull + (uint64_t)__double_as_longlong(x);
    k = k * 0x9E3779B97F4A7C15ull + (uint64_t)__double_as_longlong(y);
    k = k * 0x9E3779B97F4A7C15ull + (uint64_t)__double_as_longlong(z);
    return k;
}

// Set-up helper of the host geometry (nanokappa_amd/mesh.py, Mesh._count_crossings: inside tests by ray parity for the
// outward orientation of the faces, the tetrahedra of the volume sampler and the Monte-Carlo subvolume volumes; the role of
// trimesh's ray queries in the reference's Mesh.py): triangles crossed by the open rays o + t d, t > 0, all pairs,
// Moller-Trumbore written out by components in the host's order of operations (no fused multiply-add), crossings at the
// same distance (rounded to 1e-8: a ray through a shared edge or vertex) counted once.  One thread per ray, the triangles
// pass through LDS in tiles.  A ray with more than NK_XMAX distinct crossings gets -1 (the host counts that one itself).
#define NK_XMAX 24
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_mesh_crossings(int64_t n, const double *orig, const double *dir, int64_t F, const double *v0,
                                                          const double *e1, const double *e2, int skip_self, int32_t *counts) {
#pragma clang fp contract(off)
    __shared__ double tile[NK_WG * 9];
    const int64_t r = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    const bool act = r < n;
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    if (act) { ox = orig[3 * r]; oy = orig[3 * r + 1]; oz = orig[3 * r + 2]; dx = dir[3 * r]; dy = dir[3 * r + 1]; dz = dir[3 * r + 2]; }
    double seen[NK_XMAX];
    int ns = 0;
    bool over = false;
    const double eps = 1e-9;
    for (int64_t f0 = 0; f0 < F; f0 += NK_WG) {
        __syncthreads();
        const int64_t tf = f0 + threadIdx.x;
        if (tf < F)
            for (int c = 0; c < 3; ++c) { tile[9 * threadIdx.x + c] = v0[3 * tf + c]; tile[9 * threadIdx.x + 3 + c] = e1[3 * tf + c]; tile[9 * threadIdx.x + 6 + c] = e2[3 * tf + c]; }
        __syncthreads();
        const int nt = F - f0 < NK_WG ? (int)(F - f0) : NK_WG;
        if (!act) continue;
        for (int i = 0; i < nt; ++i) {
            const double *q = tile + 9 * i;
            const double e1x = q[3], e1y = q[4], e1z = q[5], e2x = q[6], e2y = q[7], e2z = q[8];
            const double px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;      // p = d x e2
            const double det = e1x * px + e1y * py + e1z * pz;
            const double tx = ox - q[0], ty = oy - q[1], tz = oz - q[2];
            const double inv = 1.0 / det;
            const double u = (tx * px + ty * py + tz * pz) * inv;
            const double qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;      // q = tv x e1
            const double w = (dx * qx + dy * qy + dz * qz) * inv;
            const double t = (e2x * qx + e2y * qy + e2z * qz) * inv;
            const bool ok = (fabs(det) > 1e-14) && (u >= -eps) && (w >= -eps) && (u + w <= 1 + eps) && (t > 1e-9);
            if (!ok || (skip_self && f0 + i == r)) continue;
            const double tr = rint(t * 1e8) / 1e8;                                       // np.round(t, 8)
            bool dup = false;
            for (int k = 0; k < ns; ++k) dup |= seen[k] == tr;
            if (dup) continue;
            if (ns < NK_XMAX) seen[ns++] = tr; else over = true;
        }
    }
    if (act) counts[r] = over ? -1 : ns;
}

// Population.initialise_all_particles on the device (Population.py:186-321) for the common case: modes tiled over the
// particle id (:127-144, at least one particle per mode and subvolume), positions 'random_domain' (one draw of
// Mesh.sample_volume, Mesh.py:890-904) or 'random_subvol' (the particle's id fixes the subvolume -- sv_first[s] is the first
// id of subvolume s's share of the WHOLE ensemble, :222-246, so a rank's shard is a part of the single-rank ensemble --
// and draws are repeated until one falls into it), occupation = Bose-Einstein
// at the temperature of the subvolume the particle is in (:280).  A particle takes the next free slot of the segment that
// owns its mode (cursor = seg_count, zeroed before the launch).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_init_particles(NkDev d, int64_t n, uint64_t pid_lo, const int32_t *umodes, int32_t nu,
                                                          const int64_t *sv_first) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t pid = pid_lo + (uint64_t)i;
        const int mode = umodes[pid % (uint64_t)nu];
        int seg, idx;
        nk_mode_home(d, mode, seg, idx);
        int want = -1;                                // random_subvol: the subvolume this index belongs to
        if (sv_first) { int lo = 0, hi = d.S; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sv_first[mid] <= (int64_t)pid) lo = mid; else hi = mid; } want = lo; }
        double x = 0.0, y = 0.0, z = 0.0;
        int s = 0;
        for (uint32_t t = 0; t < 4096u; ++t) {         // rejection sampling: a subvolume is expected to hold well over 1 / 4096 of the volume
            double u[6];
            nk_uniform2_dev(d.seed, pid, 0u, NK_TAG_INIT + 3u * t + 0u, u[0], u[1]);
            nk_uniform2_dev(d.seed, pid, 0u, NK_TAG_INIT + 3u * t + 1u, u[2], u[3]);
            nk_uniform2_dev(d.seed, pid, 0u, NK_TAG_INIT + 3u * t + 2u, u[4], u[5]);
            int sx = nk_ss_right(d.simplex_cdf, d.nS, u[0]);
            sx = sx > d.nS - 1 ? d.nS - 1 : sx;
            double a[4], asum = 0.0;
            for (int q = 0; q < 4; ++q) { a[q] = -log(u[1 + q]); asum += a[q]; }
            const double *sp = d.simplex_pts + 12 * (int64_t)sx;
            x = y = z = 0.0;
            for (int q = 0; q < 4; ++q) { const double wq = a[q] / asum; x += wq * sp[3 * q]; y += wq * sp[3 * q + 1]; z += wq * sp[3 * q + 2]; }
            s = nk_classify(d, L.tb, x, y, z);
            if (want < 0 || s == want) break;
            if (t == 4095u) atomicOr(d.overflow, 64);     // never found its subvolume: the host refuses the ensemble (nk_init_particles)
        }
        const NkMode *rec = d.modetab + mode;
        const double occ = nk_occupation(d, L.tb.Tsv[s], rec->omega, rec->E0);
        const int slot = atomicAdd(d.seg_count + seg, 1);
        if (slot < d.segcap) {
            const int64_t o = (int64_t)seg * d.segcap + slot;
            d.x[o] = x; d.y[o] = y; d.z[o] = z; d.occ[o] = occ; if (d.nts) d.nts[o] = 0.0; d.w0[o] = (uint32_t)idx;
            if (d.pid) d.pid[o] = pid;
        } else atomicOr(d.overflow, 2);
    }
}
// calculate_energy + the heat-flux sums of the particles where they stand (Population.py:704-717, :734-736): the t = 0 row.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_tally_state(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    const int rep = threadIdx.x & (NK_NREP - 1);
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        const NkSegModes sm = nk_seg_modes(d, seg);
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const NkMode *rec = sm.rec + (d.w0[i] & lbmask);
            nk_tally_one(d, L.tb, L.bins, d.x[i], d.y[i], d.z[i], d.occ[i], rec->omega, rec->E0, rec->vx, rec->vy, rec->vz, true, rep);
        }
    }
    nk_lds_flush(d, L, blockIdx.x);
}

// contains_check (Population.py:1712-1722) + Mesh.sample_volume (Mesh.py:890-904)
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_contains(NkDev d, uint32_t step) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (d.halt[0]) return;
    NkLds L;
    nk_lds_setup<GEOM, 0>(d, smem, L);
    const uint32_t lbmask = (1u << d.lb) - 1u;
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        const NkSegModes sm = nk_seg_modes(d, seg);
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            double x = d.x[i], y = d.y[i], z = d.z[i];
            bool out = x < d.bbox[0] - 1e-10 || y < d.bbox[1] - 1e-10 || z < d.bbox[2] - 1e-10 || x > d.bbox[3] + 1e-10 ||
                       y > d.bbox[4] + 1e-10 || z > d.bbox[5] + 1e-10;
            if (!out) continue;
            const uint32_t idx = d.w0[i] & lbmask;
            const uint64_t pid = d.pid ? d.pid[i] : nk_state_key(sm.mode((int)idx), x, y, z);
            double u[6];
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_RESAMP + 0, u[0], u[1]);
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_RESAMP + 1, u[2], u[3]);
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_RESAMP + 2, u[4], u[5]);
            int s = nk_ss_right(d.simplex_cdf, d.nS, u[0]);
            s = s > d.nS - 1 ? d.nS - 1 : s;
            double a[4], asum = 0.0;
            for (int q = 0; q < 4; ++q) { a[q] = -log(u[1 + q]); asum += a[q]; }
            const double *sp = d.simplex_pts + 12 * (int64_t)s;
            x = y = z = 0.0;
            for (int q = 0; q < 4; ++q) { double wq = a[q] / asum; x += wq * sp[3 * q]; y += wq * sp[3 * q + 1]; z += wq * sp[3 * q + 2]; }
            const NkMode *rec = sm.rec + idx;
            double tc; int fc;
            NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x, y, z, rec->vx, rec->vy, rec->vz, tc, fc);
            d.x[i] = x; d.y[i] = y; d.z[i] = z;
            if (d.box) d.w0[i] = (fc < 0 ? NK_LOST : 0u) | idx;
            else { d.nts[i] = tc / d.dt; d.w0[i] = ((uint32_t)(fc + 1) << d.lb) | idx; }
        }
    }
}

// (reservoir, mode) tables between the caller's order [r * M + m] and the segments' order (nk_device.h ep_p / rc_p)
NK_KERNEL_LINKAGE __global__ void k_perm_rm(int to_seg_order, int R, int M, int nseg, int nlmax, const int32_t *m2s, double *canon, double *perm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * M) return;
    const int r = (int)(i / M), m = (int)(i - (int64_t)r * M);
    const int slot = m2s[m];
    const int64_t at = ((int64_t)(slot % nseg) * R + r) * nlmax + slot / nseg;
    if (to_seg_order) perm[at] = canon[i]; else canon[i] = perm[at];
}

// {omega, v, E0, tau[row0..row0+2]} records, by mode index and (part) in the segments' order
NK_KERNEL_LINKAGE __global__ void k_build_modetab(const double *omega, const double *vg, const double *tau, int M, int NT, int row0, double c_hk,
                                double invT0, int nseg, int nlmax, const int32_t *m2s, NkMode *out, NkMode *out_p) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    NkMode r;
    r.omega = omega[m]; r.vx = vg[3 * m]; r.vy = vg[3 * m + 1]; r.vz = vg[3 * m + 2];
    r.E0 = exp(r.omega * c_hk * invT0);
    for (int k = 0; k < NK_TAU_ROWS; ++k) {
        int row = row0 + k;
        r.tau[k] = (row >= 0 && row < NT) ? tau[(int64_t)row * M + m] : 0.0;
    }
    out[m] = r;
    if (out_p) { const int slot = m2s[m]; out_p[(int64_t)(slot % nseg) * nlmax + slot / nseg] = r; }
}

// ---- parity taps: the reference's primitives evaluated on the device
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_tap_find_boundary(NkDev d, int64_t n, const double *x, const double *v,
                                                             double *xc, double *tc, int32_t *fc) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, 0>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    double t; int f;
    NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x[3 * i], x[3 * i + 1], x[3 * i + 2], v[3 * i], v[3 * i + 1], v[3 * i + 2], t, f);
    tc[i] = t; fc[i] = f;
    for (int k = 0; k < 3; ++k) xc[3 * i + k] = x[3 * i + k] + t * v[3 * i + k];
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_tap_classify(NkDev d, int64_t n, const double *x, int32_t *id) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i < n) id[i] = nk_classify(d, L.tb, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_tap_eval(NkDev d, int what, int64_t n, const double *a, const int32_t *mode,
                                                    double *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    switch (what) {
        case 0: out[i] = nk_occupation(d, a[i], d.modetab[mode[i]].omega, d.modetab[mode[i]].E0); break;
        case 1: { const NkMode *rec = d.modetab + mode[i]; NkTauWin tw; tw.load(d, false); out[i] = nk_lifetime(d, tw, rec->tau[0], rec->tau[1], rec->tau[2], a[i], NkPlainModes(), mode[i]); break; }
        case 2: out[i] = nk_T_of_E(d, a[i]); break;
        case 3: out[i] = nk_E_of_T(d, a[i]); break;
        case 5: out[i] = nk_exp(a[i]); break;
        default: { double invT; out[i] = nk_interp_T(d, L.tb, a[3 * i], a[3 * i + 1], a[3 * i + 2], invT); break; }
    }
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_tap_reflect(NkDev d, int64_t n, const int32_t *facet, const int32_t *mode_in,
                                                       const double *col, const double *n_in, const double *om_in,
                                                       const double *r_spec, const double *r_deg, const double *r_diff,
                                                       int32_t *mode_out, double *n_out, double *om_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, 0>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    int mo; double no, oo, eo;
    nk_reflect(d, L.tb, d.facets[facet[i]].rough, mode_in[i], col[3 * i], col[3 * i + 1], col[3 * i + 2],
               n_in[i], om_in[i], d.modetab[mode_in[i]].E0, r_spec[i], r_deg ? r_deg[i] : 0.0, r_diff[i], mo, no, oo, eo);
    mode_out[i] = mo; n_out[i] = no; om_out[i] = oo;
}
// Counter calibration: coalesced 8-byte-per-lane sweeps with a KNOWN byte count (44 B read + 32 B written per live
// particle), so FETCH_SIZE / WRITE_SIZE readings of k_sweep can be scaled (MI355X_MICROARCH.md: FETCH_SIZE is
// uncalibrated for accesses other than 16 B/lane).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_cal_stream(NkDev d) {
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            uint32_t w0 = d.w0[i];
            double x = d.x[i], y = d.y[i], z = d.z[i], occ = d.occ[i], nts = d.nts ? d.nts[i] : 0.0;
            // every one of the six loads must really be issued: a load whose only use sits behind a never-taken branch is sunk
            // into that branch by the compiler (round 2's version lost its occ load that way: 36 B read, not the 44 B the
            // calibration assumed -- its "read factor" 2.44 was 2.0 x 44 / 36)
            asm volatile("" : "+v"(w0), "+v"(occ));
            d.x[i] = x; d.y[i] = y; d.z[i] = z; if (d.nts) d.nts[i] = nts;     // (box store: 36 B read + 24 B written per particle)
        }
    }
}
// Developer probe (NK_PROBE_COPY=8|16 + nk_calibrate_stream): the sweep's wave-per-segment tile loop with one tile prefetched,
// copying the particle state in place -- the bandwidth floor of that structure (8-byte accesses, one particle per lane).
template <int W>
__global__ __launch_bounds__(NK_WG, 3) void k_probe_copy(NkDev d) {
    const bool hn = (bool)d.nts;                         // (a box store has no nts field)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (NK_WG / 64);
    for (int seg = blockIdx.x * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg] & ~63;        // whole tiles only (a probe)
        double xN = 0, yN = 0, zN = 0, oN = 0, nN = 0; uint32_t wN = 0;
        if (count > 0) { const int64_t i0 = base; xN = *d.x.tile(i0, lane); yN = *d.y.tile(i0, lane); zN = *d.z.tile(i0, lane); oN = *d.occ.tile(i0, lane); if (hn) nN = *d.nts.tile(i0, lane); wN = *d.w0.tile(i0, lane); }
        for (int r = 0; r < count; r += 64) {
            const double x = xN, y = yN, z = zN, o = oN, n = nN; const uint32_t w = wN;
            if (r + 64 < count) { const int64_t i0 = base + r + 64; xN = *d.x.tile(i0, lane); yN = *d.y.tile(i0, lane); zN = *d.z.tile(i0, lane); oN = *d.occ.tile(i0, lane); if (hn) nN = *d.nts.tile(i0, lane); wN = *d.w0.tile(i0, lane); }
            const int64_t i0 = base + r;
            *d.x.tile(i0, lane) = x + 1e-300; *d.y.tile(i0, lane) = y; *d.z.tile(i0, lane) = z; *d.occ.tile(i0, lane) = o; if (hn) *d.nts.tile(i0, lane) = n; *d.w0.tile(i0, lane) = w;
        }
    }
}
// Placement probe of the particle store (nk_place_store, nk_engine.hip): the same wave-per-segment tile loop over the first
// `tiles` tiles of EVERY segment, every value written back exactly as it was read -- harmless on a live store, and its time
// tells the two speeds apart that one and the same store shows depending on where its allocation lies in memory
// (profiles/r03_notes.txt (9), (17)).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG, 3) void k_probe_place(NkDev d, int tiles) {
    const bool hn = (bool)d.nts;                         // (a box store has no nts field)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (NK_WG / 64);
    for (int seg = blockIdx.x * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int64_t step = 64;
        const int count = tiles;
        double xN = 0, yN = 0, zN = 0, oN = 0, nN = 0; uint32_t wN = 0;
        if (count > 0) { const int64_t i0 = base; xN = *d.x.tile(i0, lane); yN = *d.y.tile(i0, lane); zN = *d.z.tile(i0, lane); oN = *d.occ.tile(i0, lane); if (hn) nN = *d.nts.tile(i0, lane); wN = *d.w0.tile(i0, lane); }
        for (int r = 0; r < count; ++r) {
            double x = xN, y = yN, z = zN, o = oN, n = nN; uint32_t w = wN;
            if (r + 1 < count) { const int64_t i0 = base + (r + 1) * step; xN = *d.x.tile(i0, lane); yN = *d.y.tile(i0, lane); zN = *d.z.tile(i0, lane); oN = *d.occ.tile(i0, lane); if (hn) nN = *d.nts.tile(i0, lane); wN = *d.w0.tile(i0, lane); }
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(o), "+v"(n), "+v"(w));     // (a store of the value just loaded would be dropped)
            const int64_t i0 = base + r * step;
            *d.x.tile(i0, lane) = x; *d.y.tile(i0, lane) = y; *d.z.tile(i0, lane) = z; *d.occ.tile(i0, lane) = o; if (hn) *d.nts.tile(i0, lane) = n; *d.w0.tile(i0, lane) = w;
        }
    }
}
NK_KERNEL_LINKAGE __global__ void k_tap_uniform(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *out) {
    double a, b;
    nk_uniform2_dev(seed, pid, step, tag, a, b);
    out[0] = a; out[1] = b;
}


// ============================================================================ set-up tables
// find_specular_correspondences, 'velocity' model (Population.py:1241-1454), for ONE surface normal: all pairs
// (in-mode, out-mode) whose mirrored group velocity and frequency agree within the grid tolerance.  The reference
// walks the in-modes in Python (minutes at 31^3 q-points, per normal); here every in-mode is a thread that sweeps the
// out-modes through LDS tiles.  The arithmetic is the reference's, operation by operation and without fused
// multiply-adds, because the pair set depends on roundings (the angle test rejects a pair when the dot product of two
// unit vectors rounds above 1: arccos -> NaN -> pi, :1357-1369).
struct NkSpecMode { double vx, vy, vz, nrm, om, dl, vdn, pad; };     // 64 bytes per mode
// The records go out in the order of the modes' x-velocity (`rank` = position of mode m in that order), the original mode
// index in `pad`: a thread of k_specular_pairs then only looks at the window of that order that the first criterion
// |v_ref.x - v_out.x| / max(|v_ref|, |v_out|) < crit can reach (|v| <= vmax), instead of at all M modes.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_specular_prepare(int M, const double *v, const double *omega, const double *delta,
                                                            const int32_t *rank, double nx, double ny, double nz, NkSpecMode *out) {
#pragma clang fp contract(off)
    const int m = blockIdx.x * NK_WG + threadIdx.x;
    if (m >= M) return;
    NkSpecMode r;
    r.vx = v[3 * m]; r.vy = v[3 * m + 1]; r.vz = v[3 * m + 2];
    r.vdn = (r.vx * nx + r.vy * ny) + r.vz * nz;                       // np.sum(v * n, axis=2)
    r.nrm = sqrt((r.vx * r.vx + r.vy * r.vy) + r.vz * r.vz);           // np.linalg.norm(v_out, axis=1)
    r.om = omega[m]; r.dl = delta[m]; r.pad = (double)m;
    out[rank[m]] = r;
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_specular_pairs(int M, const NkSpecMode *modes, const double *sorted_vx, double vmax,
                                                          double nx, double ny, double nz, double crit, int64_t cap,
                                                          int32_t *pin, int32_t *pout, unsigned long long *count) {
#pragma clang fp contract(off)
    const int a = blockIdx.x * NK_WG + threadIdx.x;
    if (a >= M) return;
    const NkSpecMode me = modes[a];
    if (!(me.vdn < 0.0)) return;                                         // in-modes only
    // mirrored velocity and its norm (v_ref = v_in - 2 n (v_in . n)), unit in-velocity mirrored the same way
    const double t2x = 2.0 * nx, t2y = 2.0 * ny, t2z = 2.0 * nz;
    const double rx = me.vx - t2x * me.vdn, ry = me.vy - t2y * me.vdn, rz = me.vz - t2z * me.vdn;
    const double nrm_in = sqrt((rx * rx + ry * ry) + rz * rz);
    const double vn = sqrt((me.vx * me.vx + me.vy * me.vy) + me.vz * me.vz);
    const double ux = me.vx / vn, uy = me.vy / vn, uz = me.vz / vn;
    const double udn = (ux * nx + uy * ny) + uz * nz;
    const double tx = ux - t2x * udn, ty = uy - t2y * udn, tz = uz - t2z * udn;
    // window of the x-velocity order: everything the first criterion can accept, and a little more
    const double h = crit * fmax(vmax, nrm_in) * (1.0 + 1e-6);
    int lo = 0, hi = M;
    { int l = 0, r = M; const double key = rx - h; while (l < r) { const int mid = (l + r) >> 1; if (sorted_vx[mid] < key) l = mid + 1; else r = mid; } lo = l; }
    { int l = lo, r = M; const double key = rx + h; while (l < r) { const int mid = (l + r) >> 1; if (sorted_vx[mid] <= key) l = mid + 1; else r = mid; } hi = l; }
    const int a_orig = (int)me.pad;
    for (int k = lo; k < hi; ++k) {
        const NkSpecMode o = modes[k];
        if (!(o.vdn > 0.0)) continue;
        const double ref = fmax(nrm_in, o.nrm);
        if (!(fabs(rx - o.vx) / ref < crit)) continue;
        if (!(fabs(ry - o.vy) / ref < crit) || !(fabs(rz - o.vz) / ref < crit)) continue;
        if (!(fabs(me.om - o.om) < me.dl + o.dl)) continue;
        const double ox = o.vx / o.nrm, oy = o.vy / o.nrm, oz = o.vz / o.nrm;
        const double dot = (tx * ox + ty * oy) + tz * oz;
        const double ang = acos(dot);                               // NaN (dot rounded above 1) rejects the pair
        if (!(ang < crit)) continue;
        const unsigned long long at = atomicAdd(count, 1ull);
        if ((int64_t)at < cap) { pin[at] = a_orig; pout[at] = (int)o.pad; }
    }
}

// ---- the rough-facet tables on the device: calculate_fbz_specularity (Population.py:852-877), the specular
// map of find_specular_correspondences (:1457), diffuse_scat_probability (:879-939), built in the layout nk_set_rough uploads.
// The pairs of one normal (k_specular_pairs, still on the device) against the facets that share it: in-modes become truly
// specular, the map keeps the smallest out-mode, and what every pair takes out of its out-mode's diffuse creation rate
// (D_total * specularity of the in-mode, :912-925) is summed per out-mode.
__device__ __forceinline__ double nk_spec0(const double *v, const double *k2, int J, int m, double nx, double ny, double nz, double eta,
                                           double &vdn) {
    const double vx = v[3 * m], vy = v[3 * m + 1], vz = v[3 * m + 2];
    vdn = nx * vx + ny * vy + nz * vz;
    const double vn = sqrt(vx * vx + vy * vy + vz * vz);
    const double c = vdn / vn, e = 2.0 * eta * c;
    const double sp = exp(-(e * e) * k2[m / J]);               // exp(-(2 eta cos)^2 k^2), :873-875
    return isnan(sp) ? 0.0 : sp;
}
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_rough_pairs(int M, int J, const double *v, const double *k2, int64_t npairs, const int32_t *pin,
                                                      const int32_t *pout, int nf, const int32_t *fidx, const double *nin,
                                                      const double *eta, uint8_t *true_spec, unsigned int *spec_map, double *sub) {
    const int64_t p = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (p >= npairs) return;
    const int a = pin[p], b = pout[p];
    for (int k = 0; k < nf; ++k) {
        const int f = fidx[k];
        double vdn;
        const double sp = nk_spec0(v, k2, J, a, nin[3 * f], nin[3 * f + 1], nin[3 * f + 2], eta[f], vdn);
        true_spec[(int64_t)f * M + a] = 1;
        atomicMin(spec_map + (int64_t)f * M + a, (unsigned int)b);
        atomicAdd(sub + (int64_t)f * M + b, (vdn < 0.0 ? -vdn : 0.0) * sp);
    }
}
// per (facet, mode): specularity = true_specular * spec0 (:1459); creation rate = max(v.n, 0) - what the pairs took, rounded to
// ten decimals (np.around, :933)
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_rough_finish(int Fr, int M, int J, const double *v, const double *k2, const double *nin,
                                                       const double *eta, const uint8_t *true_spec, int32_t *spec_map, const double *sub,
                                                       double *specularity, double *rate, int round_now) {
    const int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= (int64_t)Fr * M) return;
    const int f = (int)(i / M), m = (int)(i - (int64_t)f * M);
    double vdn;
    const double sp = nk_spec0(v, k2, J, m, nin[3 * f], nin[3 * f + 1], nin[3 * f + 2], eta[f], vdn);
    specularity[i] = true_spec[i] ? sp : 0.0;
    if (!true_spec[i]) spec_map[i] = -1;
    const double c = (vdn > 0.0 ? vdn : 0.0) - sub[i];
    rate[i] = round_now ? rint(c * 1e10) / 1e10 : c;
}
// find_specular_correspondences, 'k' / wavevector model (Population.py:1056-1240), for one normal: one thread per q-point.
// An in-mode (q, j) is specular when the mirrored wavevector k - 2 n (k.n) is the shortest of its 27 reciprocal-lattice
// neighbours (a normal process: Phonon.find_min_k's first pass ends at the zero displacement, Phonon.py:209-247, with its
// first-minimum rule in np.meshgrid's 'xy' order), lies within half a grid step (per axis) of the nearest grid point,
// that point has an outgoing branch, and of the outgoing branches whose frequency interval overlaps the in-mode's the one
// with the smallest relative frequency difference is taken (the first of equals).  The nearest grid point is searched
// over ALL q-points, staged through LDS in tiles (what scipy's NearestNDInterpolator answers).
// a2q, q2k: row-major 3 x 3, q = k . a2q and k = q . q2k (Phonon.k_to_q / q_to_k).
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_kspec_pairs(int Q, int J, const double *v, const double *om, const double *kv, const double *a2q,
                                                       const double *q2k, double tx, double ty, double tz, double nx, double ny, double nz,
                                                       int64_t cap, int32_t *pin, int32_t *pout, unsigned long long *count) {
    // The normal-process test is a comparison of norms that TIE on the zone boundary; which of the equals is a hair smaller
    // is decided by how the reference's NumPy rounds: np.dot goes through dgemm, whose kernels accumulate
    // fma(a2, b2, fma(a1, b1, a0 b0)); element-wise products and np.linalg.norm do not fuse.  The same operations here.
#pragma clang fp contract(off)
    __shared__ double tile[NK_WG * 3];
    const int q = blockIdx.x * NK_WG + threadIdx.x;
    bool act = q < Q;
    double kx = 0, ky = 0, kz = 0;
    if (act) {
        bool any_in = false;
        for (int j = 0; j < J; ++j) { const double *vv = v + ((int64_t)q * J + j) * 3; any_in |= (vv[0] * nx + vv[1] * ny + vv[2] * nz) < 0.0; }
        act = any_in;
        const double s = kv[3 * q] * nx + kv[3 * q + 1] * ny + kv[3 * q + 2] * nz;
        kx = kv[3 * q] - (2.0 * nx) * s; ky = kv[3 * q + 1] - (2.0 * ny) * s; kz = kv[3 * q + 2] - (2.0 * nz) * s;
    }
    if (act) {                                        // normal process: the zero displacement is the first minimum
        const double q0 = fma(kz, a2q[6], fma(ky, a2q[3], kx * a2q[0])), q1 = fma(kz, a2q[7], fma(ky, a2q[4], kx * a2q[1])),
                     q2 = fma(kz, a2q[8], fma(ky, a2q[5], kx * a2q[2]));
        double best = __builtin_inf();
        int ibest = -1;
        for (int t = 0; t < 27; ++t) {                // np.meshgrid(a, a, a) 'xy': neighbour t = (a[j], a[i], a[k]), t = 9 i + 3 j + k
            const double a0 = q0 + (double)((t / 3) % 3 - 1), a1 = q1 + (double)(t / 9 - 1), a2 = q2 + (double)(t % 3 - 1);
            const double x = fma(a2, q2k[6], fma(a1, q2k[3], a0 * q2k[0])), y = fma(a2, q2k[7], fma(a1, q2k[4], a0 * q2k[1])),
                         z = fma(a2, q2k[8], fma(a1, q2k[5], a0 * q2k[2]));
            const double nr = sqrt((x * x + y * y) + z * z);
            if (nr < best) { best = nr; ibest = t; }
        }
        act = ibest == 13;
    }
    // nearest grid point of the mirrored wavevector
    double dbest = __builtin_inf();
    int qo = 0;
    for (int t0 = 0; t0 < Q; t0 += NK_WG) {
        __syncthreads();
        const int tq = t0 + threadIdx.x;
        if (tq < Q) { tile[3 * threadIdx.x] = kv[3 * tq]; tile[3 * threadIdx.x + 1] = kv[3 * tq + 1]; tile[3 * threadIdx.x + 2] = kv[3 * tq + 2]; }
        __syncthreads();
        const int nt = Q - t0 < NK_WG ? Q - t0 : NK_WG;
        if (act)
            for (int i = 0; i < nt; ++i) {
                const double dx = kx - tile[3 * i], dy = ky - tile[3 * i + 1], dz = kz - tile[3 * i + 2];
                const double dd = dx * dx + dy * dy + dz * dz;
                if (dd < dbest) { dbest = dd; qo = t0 + i; }
            }
    }
    if (!act) return;
    if (!(fabs(kx - kv[3 * qo]) < tx && fabs(ky - kv[3 * qo + 1]) < ty && fabs(kz - kv[3 * qo + 2]) < tz)) return;
    for (int ji = 0; ji < J; ++ji) {
        const double *vi = v + ((int64_t)q * J + ji) * 3;
        if (!((vi[0] * nx + vi[1] * ny + vi[2] * nz) < 0.0)) continue;
        const double io = om[(int64_t)q * J + ji];
        const double idl = fabs(vi[0]) * tx + fabs(vi[1]) * ty + fabs(vi[2]) * tz;
        const double iup = io + idl, idn = io - idl;
        double dmin = __builtin_inf();
        int br = -1;
        for (int jo = 0; jo < J; ++jo) {
            const double *vo = v + ((int64_t)qo * J + jo) * 3;
            if (!((vo[0] * nx + vo[1] * ny + vo[2] * nz) > 0.0)) continue;
            const double oo = om[(int64_t)qo * J + jo];
            const double odl = fabs(vo[0]) * tx + fabs(vo[1]) * ty + fabs(vo[2]) * tz;
            const double oup = oo + odl, odn = oo - odl;
            if (!(((iup < oup ? iup : oup) - (idn > odn ? idn : odn)) > 0.0)) continue;
            const double df = fabs((io - oo) / io);
            if (br < 0 || df < dmin) { dmin = df; br = jo; }
        }
        if (br < 0) continue;
        const unsigned long long at = atomicAdd(count, 1ull);
        if ((int64_t)at < cap) { pin[at] = q * J + ji; pout[at] = qo * J + br; }
    }
}
// 'k' model: creation rates of degenerate branches are averaged (Population.py:926-930), pair after pair in the list's order,
// before they are rounded.  One thread per rough facet.
NK_KERNEL_LINKAGE __global__ void k_rough_degen(int Fr, int M, int J, int nd, const int32_t *degen, double *rate) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Fr) return;
    double *r = rate + (int64_t)f * M;
    for (int i = 0; i < nd; ++i) {
        const int q = degen[3 * i], j1 = degen[3 * i + 1], j2 = degen[3 * i + 2];
        const double m = (r[(int64_t)q * J + j1] + r[(int64_t)q * J + j2]) / 2.0;
        r[(int64_t)q * J + j1] = m; r[(int64_t)q * J + j2] = m;
    }
}
NK_KERNEL_LINKAGE __global__ void k_rough_round(int64_t n, double *rate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rate[i] = rint(rate[i] * 1e10) / 1e10;
}
// creation_roulette = cumsum(rate) / max(cumsum) per facet (:938-939), the running sum in np.cumsum's own order: one wave per
// facet reads 64 rates at a time (coalesced) and every lane adds them up one by one, keeping the sum at its own position.
NK_KERNEL_LINKAGE __global__ __launch_bounds__(64) void k_rough_cumsum(int M, double *rate_roul) {
    double *r = rate_roul + (int64_t)blockIdx.x * M;
    const int lane = threadIdx.x;
    double run = 0.0, mx = -__builtin_inf();
    for (int m0 = 0; m0 < M; m0 += 64) {
        const double val = m0 + lane < M ? r[m0 + lane] : 0.0;
        double mine = 0.0;
        for (int k = 0; k < 64; ++k) {
            run += __shfl(val, k, 64);
            if (k == lane) mine = run;
            if (m0 + k < M) mx = fmax(mx, run);
        }
        if (m0 + lane < M) r[m0 + lane] = mine;
    }
    for (int m = lane; m < M; m += 64) r[m] = r[m] / mx;
}
// bucket index of the roulette search (nk_reflect): lut[f][k] = first position with roulette >= k / nlut * last
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_rough_lut(int Fr, int M, int nlut, const double *roul, int32_t *lut) {
    const int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= (int64_t)Fr * (nlut + 1)) return;
    const int f = (int)(i / (nlut + 1)), k = (int)(i - (int64_t)f * (nlut + 1));
    const double *ro = roul + (int64_t)f * M;
    if (k == nlut) { lut[i] = M; return; }
    const double thr = ((double)k / (double)nlut) * ro[M - 1];
    lut[i] = nk_ss_left(ro, M, thr);
}
NK_KERNEL_LINKAGE __global__ void k_fill_u32(unsigned int *p, int64_t n, unsigned int v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// enter_probability (Population.py:146-161): p[r, m] = max(0, v . n_in) * dt / thickness_r
NK_KERNEL_LINKAGE __global__ __launch_bounds__(NK_WG) void k_enter_prob(int R, int M, const double *v, const double *nin, const double *thick, double dt,
                                                     double *out) {
    const int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= (int64_t)R * M) return;
    const int r = (int)(i / M), m = (int)(i - (int64_t)r * M);
    const double p = (nin[3 * r] * v[3 * m] + nin[3 * r + 1] * v[3 * m + 1] + nin[3 * r + 2] * v[3 * m + 2]) * dt / thick[r];
    out[i] = p < 0.0 ? 0.0 : p;
}
