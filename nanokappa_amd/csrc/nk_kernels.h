// nk_kernels.h -- the HIP kernels of libnanokappa_hip.so (gfx950 / MI355X only).
//
// Stream order of one timestep (reference Population.run_timestep, Population.py:1724-1769):
//   [k_relax + k_contains every `contains_every` steps]            contains_check       :1712-1722
//   [k_emit_count / k_emit_one_to_one: only to prime a run, for very large (reservoir, mode) tables, or for the
//    'one_to_one' generator]                                        fill_reservoirs      :356-489
//   k_sweep       per segment (one wave): relax(previous step) -> drift -> boundary events -> tally -> compaction,
//                 then the segment's share of the entering particles; in its tail the NEXT step's emission
//                                                                   lifetime_scattering  :1701-1710 (deferred)
//                                                                   drift                :790-795
//                                                                   boundary_scattering  :1546-1683
//                                                                   add_reservoir_particles :525-552
//                                                                   calculate_energy     :704-717
//                                                                   fill_reservoirs      :356-455 (for step + 1)
//   k_reduce      deterministic column sums of the per-workgroup tally rows; single rank: the last workgroup also
//                 normalises, inverts E -> T and writes the history row    calculate_energy :719-728, refresh_temperatures :692
//   (nranks > 1: RCCL all-reduce of the tally vector, then k_update does that part)
//
// Deferred relaxation: the reference relaxes occupations at the END of step k with the temperatures of step k.  Those
// need the global tally of step k, so the relaxation is carried into the BEGINNING of the sweep of step k+1 (same
// positions, same T_sv): one pass over the particles per step instead of two.  A pending relaxation is flushed by
// k_relax before anything observes the particles (download, contains_check).
//
// Why one fused sweep (measured, profiles/r01_c2_1e7_pmc_v2_uncalibrated.json): with a separate event kernel and
// free-slot reuse the step kernel moved 2.8 GB per launch for 0.68 GB of algorithmic traffic -- random 64-B mode gathers
// (1.2 GB) and line-granular write-backs of the scattered event kernel -- and ran AT the HBM roof (6.3 TB/s).  Here
//   * a WAVE owns a segment and walks it tile by tile (64 particles, coalesced loads, next tile prefetched);
//   * particles that meet a boundary inside the step (a third of them in a 20 nm box) are parked in the wave's slice of
//     an LDS buffer and processed 64 at a time by all lanes, one event per pass (no divergence against the streaming
//     lanes, no second trip through HBM); the few that meet another wall are parked again;
//   * survivors are written back compacted IN PLACE (write cursor <= read cursor), absorbed particles simply vanish;
//   * entering particles are appended to the segment in whole tiles, in runs of (reservoir, mode) order.
#pragma once
#include "nk_device.h"

// =================================================================================== LDS carve-up
struct NkEvBuf {          // parked boundary-event particles of the workgroup
    double *x, *y, *z, *occ, *nts, *cts;     // cts: fraction of the step already consumed by earlier events
    unsigned long long *pid;
    int *mode, *facet, *evc;                 // evc: events so far in this step (numbers the RNG draws)
};
struct NkLds {
    double *Tsv, *cen;
    NkBins bins;
    const double *planes, *faces;
    const NkFacet *facets;
    const double *resT;                  // [R] reservoir temperatures
    const int *rf_off;                   // reservoir face tables (CSR), LDS copies when d.res_lds
    const double *rf_cdf, *rf_verts;
    NkEvBuf ev;
};

// geom: 0 = no ray-casting tables, 1 = planes/faces/facets staged in LDS, 2 = read from global memory (large meshes)
// nrf: faces of the reservoir sampling tables staged in LDS (0 = not staged)
__host__ __device__ inline size_t nk_lds_bytes(int S, int R, int F, int NP, int Fc, int geom, bool evbuf, int nrf, int rbfP) {
    int Fl = geom == 1 ? F : 0;
    int Pl = Fl ? NP : 0;
    int Fcl = geom == 1 ? Fc : 0;
    size_t nd = (size_t)S + (size_t)((rbfP + 1) & ~1) + 3 * S + NK_NREP * S + NK_NREP * 3 * S + 4 * R + (size_t)Fl * NK_FACE_DOUBLES +
                (size_t)Pl * NK_PLANE_DOUBLES + (evbuf ? 7 * NK_EVCAP : 0) + (size_t)R + 10 * (size_t)nrf + 2;
    size_t bytes = nd * 8 + (size_t)Fcl * sizeof(NkFacet) +
                   (size_t)(NK_NREP * S + R + 1 + (R + 1) + (evbuf ? 3 * NK_EVCAP : 0)) * 4 + 16;
    return (bytes + 15) & ~(size_t)15;
}

// Cooperative fill of the read-only tables and zeroing of the bins; ends with a barrier.  GEOM as in nk_lds_bytes (a
// compile-time choice, so that the table pointers are provably LDS and are read with ds_read, not flat loads);
// EVBUF = carve the event buffer.  Plane, face and facet tables start on 16-byte boundaries (they are read 16 bytes
// at a time).
// The pointer arithmetic alone (also what an out-of-line helper needs to find the tables again).
template <int GEOM, bool EVBUF>
__device__ __forceinline__ void nk_lds_carve(const NkDev &d, unsigned char *smem, NkLds &L) {
    const int S = d.S, R = d.R;
    const int Fl = GEOM == 1 ? d.F : 0;
    const int Pl = Fl ? d.NP : 0;
    const int Fcl = GEOM == 1 ? d.Fc : 0;
    const int nrf = (GEOM == 1 && d.res_lds) ? d.res_nf : 0;
    double *p = (double *)smem;
    L.Tsv = p; p += S + ((d.rbf_P + 1) & ~1);          // temperatures, then the RBF coefficients (even count)
    L.cen = p; p += 3 * S;
    L.bins.E = p; p += NK_NREP * S;
    L.bins.flux = p; p += NK_NREP * 3 * S;
    L.bins.resb = p; p += 4 * R;                       // 36 S + 4 R doubles so far: even
    double *faces = p; p += (size_t)Fl * NK_FACE_DOUBLES;
    double *planes = p; p += (size_t)Pl * NK_PLANE_DOUBLES;
    if (EVBUF) {
        L.ev.x = p; p += NK_EVCAP; L.ev.y = p; p += NK_EVCAP; L.ev.z = p; p += NK_EVCAP;
        L.ev.occ = p; p += NK_EVCAP; L.ev.nts = p; p += NK_EVCAP;
        L.ev.pid = (unsigned long long *)p; p += NK_EVCAP;
        L.ev.cts = p; p += NK_EVCAP;
    }
    double *resT = p; p += R;
    double *rf_cdf = p; p += nrf;
    double *rf_verts = p; p += 9 * (size_t)nrf;
    p += ((size_t)R + 10 * (size_t)nrf) & 1;           // keep the facet table 16-byte aligned
    NkFacet *facets = (NkFacet *)p;
    unsigned int *u = (unsigned int *)(facets + Fcl);
    L.bins.N = u; u += NK_NREP * S;
    L.bins.nleave = u; u += R;
    L.bins.misc = u; u += 1;
    int *rf_off = (int *)u; u += R + 1;
    if (EVBUF) { L.ev.mode = (int *)u; u += NK_EVCAP; L.ev.facet = (int *)u; u += NK_EVCAP; L.ev.evc = (int *)u; u += NK_EVCAP; }
    L.resT = resT;
    L.rf_off = rf_off; L.rf_cdf = rf_cdf; L.rf_verts = rf_verts;
    if (GEOM == 1) { L.faces = faces; L.planes = planes; L.facets = facets; }
    else { L.faces = d.faces; L.planes = d.planes; L.facets = d.facets; }
}
template <int GEOM, bool EVBUF>
__device__ __forceinline__ void nk_lds_setup(const NkDev &d, unsigned char *smem, NkLds &L) {
    nk_lds_carve<GEOM, EVBUF>(d, smem, L);
    const int S = d.S, R = d.R;
    const int Fl = GEOM == 1 ? d.F : 0;
    const int Pl = Fl ? d.NP : 0;
    const int Fcl = GEOM == 1 ? d.Fc : 0;
    const int nrf = (GEOM == 1 && d.res_lds) ? d.res_nf : 0;
    double *faces = const_cast<double *>(L.faces), *planes = const_cast<double *>(L.planes);
    NkFacet *facets = const_cast<NkFacet *>(L.facets);
    double *resT = const_cast<double *>(L.resT), *rf_cdf = const_cast<double *>(L.rf_cdf), *rf_verts = const_cast<double *>(L.rf_verts);
    int *rf_off = const_cast<int *>(L.rf_off);
    const int t = threadIdx.x;
    for (int i = t; i < S + d.rbf_P; i += NK_WG) L.Tsv[i] = d.T_sv[i];
    for (int i = t; i < 3 * S; i += NK_WG) L.cen[i] = d.centers[i];
    for (int i = t; i < NK_NREP * S; i += NK_WG) { L.bins.E[i] = 0.0; L.bins.N[i] = 0u; }
    for (int i = t; i < NK_NREP * 3 * S; i += NK_WG) L.bins.flux[i] = 0.0;
    for (int i = t; i < 4 * R; i += NK_WG) L.bins.resb[i] = 0.0;
    for (int i = t; i < R; i += NK_WG) { L.bins.nleave[i] = 0u; resT[i] = d.res_T[i]; }
    if (t == 0) L.bins.misc[0] = 0u;
    for (int i = t; i < Fl * NK_FACE_DOUBLES; i += NK_WG) faces[i] = d.faces[i];
    for (int i = t; i < Pl * NK_PLANE_DOUBLES; i += NK_WG) planes[i] = d.planes[i];
    if (nrf > 0) {
        for (int i = t; i <= R; i += NK_WG) rf_off[i] = d.res_face_off[i];
        for (int i = t; i < nrf; i += NK_WG) rf_cdf[i] = d.res_face_cdf[i];
        for (int i = t; i < 9 * nrf; i += NK_WG) rf_verts[i] = d.res_face_verts[i];
    }
    {
        const int nw = Fcl * (int)(sizeof(NkFacet) / 4);
        const int32_t *src = (const int32_t *)d.facets;
        int32_t *dst = (int32_t *)facets;
        for (int i = t; i < nw; i += NK_WG) dst[i] = src[i];
    }
    __syncthreads();
}

// Row layout: E[S] N[S] flux[3S] nleave[R] resE[R] resF[3R] emitted[1]
__device__ __forceinline__ void nk_lds_flush(const NkDev &d, const NkLds &L, int64_t row) {
    __syncthreads();
    const int S = d.S, R = d.R;
    double *out = d.partials + row * d.NB;
    for (int b = threadIdx.x; b < d.NB; b += NK_WG) {
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

// Deferred lifetime_scattering (Population.py:1701-1710) for one particle.
// The mode record is passed as two 32-byte halves {omega, vx, vy, vz} {tau0..tau3} (two dwordx4 pairs, no struct copy).
template <bool RBF = true>
__device__ __forceinline__ double nk_relax(const NkDev &d, const NkLds &L, const double4 &ra, const double4 &rb, double x,
                                           double y, double z, double occ, int mode) {
    double T = nk_interp_T<RBF>(d, L.cen, L.Tsv, x, y, z, -1);
    double tau = nk_lifetime(d, rb.x, rb.y, rb.z, rb.w, T, mode);
    double n0 = nk_occupation(d, T, ra.x);
    return (tau > 0.0) ? n0 + (occ - n0) * exp(-d.dt / tau) : n0;
}

__device__ __forceinline__ void nk_store(const NkDev &d, int64_t i, double x, double y, double z, double occ, double nts,
                                         int mode, int facet, unsigned long long pid) {
    d.x[i] = x; d.y[i] = y; d.z[i] = z; d.occ[i] = occ; d.nts[i] = nts;
    d.mode[i] = mode; d.facet[i] = facet; d.pid[i] = pid;
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

// Ray cast as the kernels call it: grouped sweep for meshes whose tables stay in global memory, plain sweep over LDS.
#define NK_RAY(GEOM, d, L, skip, x, y, z, vx, vy, vz, tc, fc)                                                        \
    do {                                                                                                              \
        if ((GEOM) == 2 && (d).NG > 0) nk_find_boundary_tree(d, skip, x, y, z, vx, vy, vz, tc, fc);                      \
        else nk_find_boundary((L).planes, (L).faces, (d).NP, (d).tol, x, y, z, vx, vy, vz, tc, fc);                   \
    } while (0)

// ========================================================================================= kernels
// Which modes enter at each reservoir at `step`, and how many particles of each: fill_reservoirs 'constant'
// (Population.py:358-370) / 'fixed_rate' (:408-420) for one (reservoir, mode) entry.  c = particles entering,
// c_mine = those this rank owns (emission_owner: (rm + level + step) % nranks).
__device__ __forceinline__ void nk_emit_entry(const NkDev &d, uint32_t step, int buf, int64_t rm, int &c, int &c_mine) {
    const double prob = d.enter_prob[rm];
    const double fixed = floor(prob);
    int mask;
    double cv;
    if (d.res_gen == 0) {
        cv = d.res_counter[rm] + (prob - fixed);
        mask = cv >= 1.0;
        cv -= (double)mask;
        d.res_counter[rm] = cv;
    } else {
        double d1;
        nk_uniform2_dev(d.seed, (uint64_t)rm | 0xFFFFFFFF00000000ull, step, NK_TAG_DICE, cv, d1);
        mask = cv <= (prob - fixed);
    }
    c = (int)fixed + mask;
    if (c > 0) d.res_cval[buf][rm] = cv;
    c_mine = 0;
    if (d.nranks == 1) c_mine = c;
    else for (int level = c; level >= 1; --level)                  // rm < 2^28, so 32-bit arithmetic is exact
        c_mine += (((uint32_t)rm + (uint32_t)level + step) % (uint32_t)d.nranks) == (uint32_t)d.rank;
}

// The entries [e0, e1) (at most KMAX * NK_WG of them) handled by one workgroup: every entering particle gets one
// 64-bit record (rm << 12 | level) in spawn_list[buf].  Space is claimed with ONE global atomic per workgroup (a hot
// counter serves ~90 atomics/us).  Must be called by all threads of the workgroup.
template <int KMAX>
__device__ __forceinline__ void nk_emit_block(const NkDev &d, uint32_t step, int buf, int64_t e0, int64_t e1, int *wsum,
                                              int *bbase) {
    static_assert(KMAX <= 4, "counts are packed 16 bits per entry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t pc = 0, pm = 0;                       // (c, c_mine) of this thread's entries, 16 bits each (c <= 4095);
    int mine = 0;                                  // rolled loops: this sits in the sweep's tail, keep the code small
#pragma unroll 1
    for (int k = 0; k < KMAX; ++k) {
        const int64_t rm = e0 + (int64_t)k * NK_WG + tid;
        int c = 0, cm = 0;
        if (rm < e1) nk_emit_entry(d, step, buf, rm, c, cm);
        pc |= (uint64_t)c << (16 * k);
        pm |= (uint64_t)cm << (16 * k);
        mine += cm;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const int total = __shfl(incl, 63, 64);
    if (lane == 63) wsum[wave] = total;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < NK_WG / 64; ++w) t += wsum[w];
        *bbase = t > 0 ? atomicAdd(d.alloc_count + buf, t) : 0;
    }
    __syncthreads();
    int base = *bbase;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int64_t g = (int64_t)base + incl - mine;
    uint64_t *list = d.spawn_list[buf];
#pragma unroll 1
    for (int k = 0; k < KMAX; ++k) {
        const int64_t rm = e0 + (int64_t)k * NK_WG + tid;
        const int c = (int)((pc >> (16 * k)) & 0xFFFFu), cm = (int)((pm >> (16 * k)) & 0xFFFFu);
        for (int level = c; level >= 1 && cm > 0; --level) {
            if (d.nranks > 1 && (((uint32_t)rm + (uint32_t)level + step) % (uint32_t)d.nranks) != (uint32_t)d.rank) continue;
            if (g < d.spawn_cap) list[g] = ((uint64_t)rm << 12) | (uint64_t)level;
            else atomicOr(d.overflow, 8);       // spawn list full
            ++g;
        }
    }
}
#define NK_EMIT_KMAX 4          // entries per thread the sweep's tail may take (else k_emit_count runs every step)

// fill_reservoirs 'one_to_one' (Population.py:457-489): one particle in for every particle that left through the
// reservoir at the previous step (all ranks; nleave_prev is written by the update after the all-reduce).  One thread
// per candidate: owner test, mode from the cumulative enter_prob (np.searchsorted :472), record
// (i << 40 | rm << 12 | 0) -- level 0 marks "entry time uniform in the step" for the sweep's phase B.
__global__ __launch_bounds__(NK_WG) void k_emit_one_to_one(NkDev d, uint32_t step) {
    __shared__ int wsum[NK_WG / 64];
    __shared__ int bbase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int buf = (int)(step & 1u);
    int64_t total = 0;
    for (int r = 0; r < d.R; ++r) total += d.nleave_prev[r];
    uint64_t *list = d.spawn_list[buf];
    for (int64_t c0 = (int64_t)blockIdx.x * NK_WG; c0 < total; c0 += (int64_t)gridDim.x * NK_WG) {
        const int64_t c = c0 + tid;
        uint64_t rec = 0;
        int mine = 0;
        if (c < total) {
            int r = 0;
            int64_t i = c;
            while (r < d.R - 1 && i >= d.nleave_prev[r]) { i -= d.nleave_prev[r]; ++r; }
            if (((uint32_t)((uint64_t)i + step) % (uint32_t)d.nranks) == (uint32_t)d.rank) {
                const uint64_t pid = ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)r << 32) | (uint64_t)i;
                double um, u1;
                nk_uniform2_dev(d.seed, pid, step, NK_TAG_DICE, um, u1);
                int m = nk_ss_left(d.res_roulette + (int64_t)r * d.M, d.M, um);
                m = m > d.M - 1 ? d.M - 1 : m;
                if (i < (1ll << 24)) { rec = ((uint64_t)i << 40) | ((uint64_t)((int64_t)r * d.M + m) << 12); mine = 1; }
                else atomicOr(d.overflow, 16);      // one_to_one index beyond 2^24
            }
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < NK_WG / 64; ++w) t += wsum[w];
            bbase = t > 0 ? atomicAdd(d.alloc_count + buf, t) : 0;
        }
        __syncthreads();
        int base = bbase;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int64_t g = (int64_t)base + incl - mine;
        if (mine) { if (g < d.spawn_cap) list[g] = rec; else atomicOr(d.overflow, 8); }
        __syncthreads();
    }
}

// Stand-alone emission pass: primes the first step after (re)configuration, and serves every step when the
// (reservoir, mode) table is too large for the sweep's tail.
__global__ __launch_bounds__(NK_WG) void k_emit_count(NkDev d, uint32_t step) {
    __shared__ int wsum[NK_WG / 64];
    __shared__ int bbase;
    const int64_t RM = (int64_t)d.R * d.M;
    const int64_t e0 = (int64_t)blockIdx.x * NK_WG;
    nk_emit_block<1>(d, step, (int)(step & 1u), e0, e0 + NK_WG < RM ? e0 + NK_WG : RM, wsum, &bbase);
}

// The sweep: persistent WAVES, each taking segments w, w + n_waves, ...  A wave owns its segment and its slice of the
// LDS event buffer, so the loop needs no workgroup barrier: the four waves of a workgroup only share the read-only
// tables and the tally bins (LDS atomics).
// Per segment one loop over 64-particle tiles: first the particles already there (phase A), then this segment's share
// of the entering particles (phase B), then one empty tile that flushes the event buffer.  Every tile ends at the same
// commit site: final particles are tallied and stored compacted at the write cursor, particles that meet a boundary
// inside the step are parked in LDS, and whenever 64 are parked the whole wave processes them.
#define NK_TILE 64
#define NK_MIN_FREE 128         // segments with fewer free slots take no entering particles this step ...
#define NK_QUANT_SEGCAP 512     // ... when segments are at least this large (tiny test populations deal exactly)
#ifndef NK_SWEEP_OCC
#define NK_SWEEP_OCC 3          // workgroups per CU the sweep is compiled for (3 x 4 waves = 3 waves per SIMD)
#endif
// Developer ablation build (make ablate -> libnanokappa_hip_ablate.so, env NK_DEBUG = mask): skip one part of the sweep
// to see what bounds it.  1 no particle stores, 2 no tally, 4 no relaxation, 8 no boundary events, 16 no entering
// particles, 32 no ray casting after an event, 64 no reservoir tally, 128 no tally of event particles.
// The production library compiles NK_ABL(b) to false.
#ifdef NK_ABLATE
#define NK_ABL(b) ((d.dbg & (b)) != 0)
#else
#define NK_ABL(b) false
#endif
#define NK_WAVE_EVCAP (NK_EVCAP / (NK_WG / 64))     // 128 parked particles per wave (< 64 pending + 64 new)
template <int GEOM, bool ROUGH, bool RBF>
__global__ __launch_bounds__(NK_WG, NK_SWEEP_OCC) void k_sweep(NkDev d, uint32_t step, int do_relax, int flags) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, true>(d, smem, L);
    const bool do_flux = (flags & 1) != 0;          // flags: 1 = heat-flux step, 2 = emit the next step's particles
    const int buf = (int)(step & 1u);
    __shared__ int emit_wsum[NK_WG / 64];
    __shared__ int emit_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rep = lane & (NK_NREP - 1);
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int eb = wave * NK_WAVE_EVCAP;                    // this wave's slice of the event buffer
    int64_t total = d.R > 0 ? (int64_t)d.alloc_count[buf] : 0;
    if (total > d.spawn_cap) total = d.spawn_cap;
    const int64_t total_free = d.seg_free_prefix[d.nseg];
    if (total > total_free) { total = total_free; if (tid == 0) atomicOr(d.overflow, 1); }
    if (NK_ABL(16)) total = 0;
    if (NK_ABL(4)) do_relax = 0;
    if (tid == 0 && blockIdx.x == 0) L.bins.misc[0] = (unsigned int)total;          // "emitted" column
    const int nwaves = gridDim.x * (NK_WG / 64);
    for (int seg = blockIdx.x * (NK_WG / 64) + wave; seg < d.nseg; seg += nwaves) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        // entering particles are dealt in proportion to the segment's free space (snapshot of the previous step):
        // self-balancing, and the share always fits
        int64_t g0 = total_free > 0 ? total * d.seg_free_prefix[seg] / total_free : 0;
        int64_t g1 = total_free > 0 ? total * d.seg_free_prefix[seg + 1] / total_free : 0;
        // whole tiles: share boundaries rounded UP to multiples of 64 and clipped at the total (a 9-lane tile costs as
        // much as a full one).  A share grows by at most 63, and a segment whose exact share is empty stays empty -- in
        // particular the excluded ones: the prefix only counts segments with >= NK_MIN_FREE free slots, so the grown
        // share still fits whenever the entering particles fill at most half of that space.
        if (d.segcap >= NK_QUANT_SEGCAP && total * 2 <= total_free && !NK_ABL(256)) {
            g0 = (g0 + 63) & ~(int64_t)63; g0 = g0 < total ? g0 : total;
            g1 = (g1 + 63) & ~(int64_t)63; g1 = g1 < total ? g1 : total;
        }
        const int nA = (count + NK_TILE - 1) / NK_TILE, nB = (int)((g1 - g0 + NK_TILE - 1) / NK_TILE);
        int w = 0, ev_n = 0;
        // next tile of phase A is requested before the current tile's arithmetic (the loop is latency-bound otherwise)
        int modeN = 0, facetN = 0;
        double xN = 0, yN = 0, zN = 0, occN = 0, ntsN = 0;
        unsigned long long pidN = 0;
        if (lane < count) {
            const int64_t i = base + lane;
            modeN = d.mode[i]; xN = d.x[i]; yN = d.y[i]; zN = d.z[i]; occN = d.occ[i]; ntsN = d.nts[i];
            facetN = d.facet[i]; pidN = d.pid[i];
        }
        for (int t = 0; t <= nA + nB; ++t) {
            bool act = false;
            double x = 0, y = 0, z = 0, occ = 0, nts = 0, omega = 0, vx = 0, vy = 0, vz = 0;
            int mode = 0, facet = -1;
            unsigned long long pid = 0;
            if (t < nA) {
                // ---- phase A: relax (deferred from the previous step), drift
                const int r = t * NK_TILE;
                act = r + lane < count;
                mode = modeN; facet = facetN; x = xN; y = yN; z = zN; occ = occN; nts = ntsN; pid = pidN;
                const double4 *mrec = reinterpret_cast<const double4 *>(d.modetab + (act ? mode : 0));
                const double4 ra = mrec[0], rb = mrec[1];                            // {omega, v} {tau rows}
                if (r + NK_TILE + lane < count) {
                    const int64_t i = base + r + NK_TILE + lane;
                    modeN = d.mode[i]; xN = d.x[i]; yN = d.y[i]; zN = d.z[i]; occN = d.occ[i]; ntsN = d.nts[i];
                    facetN = d.facet[i]; pidN = d.pid[i];
                }
                omega = ra.x; vx = ra.y; vy = ra.z; vz = ra.w;
                if (act) {
                    if (do_relax) occ = nk_relax<RBF>(d, L, ra, rb, x, y, z, occ, mode);
                    x += vx * d.dt; y += vy * d.dt; z += vz * d.dt;                 // drift, Population.py:793
                    nts -= 1.0;                                                     // :795
                }
            } else if (t < nA + nB) {
                // ---- phase B: an entering particle (Mesh.sample_surface, Mesh.py:923-951; entry times
                // Population.py:391-394 / :440-443; add_reservoir_particles :525-552)
                const int64_t g = g0 + (int64_t)(t - nA) * NK_TILE + lane;
                act = g < g1;
                if (act) {
                    const uint64_t recd = d.spawn_list[buf][g];
                    const int64_t rm = (int64_t)((recd >> 12) & 0xFFFFFFFull);
                    const int level = (int)(recd & 0xFFFu);      // 0: 'one_to_one' record (index in bits 40..63)
                    const int r = (int)(rm / d.M);
                    mode = (int)(rm - (int64_t)r * d.M);
                    // everything that hangs off the record is requested at once; the Philox rounds cover the latency
                    const double prob = d.enter_prob[rm];
                    const double cval = d.res_cval[buf][rm];
                    const double4 ra = *reinterpret_cast<const double4 *>(d.modetab + mode);
                    pid = level > 0 ? ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)rm << 12) | (uint64_t)level
                                    : ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)r << 32) | (recd >> 40);
                    double uf, us, ur, ut;
                    nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT, uf, us);
                    nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT + 1, ur, ut);
                    const double dt_in = (level == 0) ? d.dt * ut                               // one_to_one :482
                                       : (level == 1) ? d.dt * (1.0 - (cval / prob))
                                                      : d.dt * (1.0 - ((double)(level - 1) + ut) / prob);
                    double x0, y0, z0;
                    if (GEOM == 1 && d.res_lds) nk_sample_res_face(L.rf_off, L.rf_cdf, L.rf_verts, r, uf, us, ur, x0, y0, z0);
                    else nk_sample_res_face(d.res_face_off, d.res_face_cdf, d.res_face_verts, r, uf, us, ur, x0, y0, z0);
                    omega = ra.x; vx = ra.y; vy = ra.z; vz = ra.w;
                    occ = nk_occupation(d, L.resT[r], omega);                        // Population.py:506
                    double tc;
                    int skip = NK_TREE_NO_SKIP;               // the particle starts on its reservoir's facet
                    if (GEOM == 2 && d.NG > 0) {
                        const int rf = d.res_facet[r];
                        const NkFacet &fq = d.facets[rf];
                        skip = nk_tree_skip(d, rf, fq.cx, fq.cy, fq.cz, fq.nx, fq.ny, fq.nz, x0, y0, z0, vx, vy, vz);
                    }
                    NK_RAY(GEOM, d, L, skip, x0, y0, z0, vx, vy, vz, tc, facet);
                    nts = tc / d.dt - dt_in / d.dt;                                  // :535
                    x = x0 + vx * dt_in; y = y0 + vy * dt_in; z = z0 + vz * dt_in;   // :536
                }
            }
            // ---- commit: boundary particles -> LDS buffer; final particles -> tally + compacted store
            const bool ev = act && nts < 0.0 && !NK_ABL(8);
            const bool done = act && !ev;
            const unsigned long long mD = __ballot(done), mE = __ballot(ev);
            if (ev) {
                const int e = eb + ev_n + __popcll(mE & lower);
                L.ev.x[e] = x; L.ev.y[e] = y; L.ev.z[e] = z; L.ev.occ[e] = occ; L.ev.nts[e] = nts;
                L.ev.mode[e] = mode; L.ev.facet[e] = facet; L.ev.pid[e] = pid;
                L.ev.cts[e] = 0.0; L.ev.evc[e] = 0;
            }
            ev_n += __popcll(mE);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // LDS is in-order per wave; keep the compiler honest
            // whenever a full batch is parked (everything on the last, empty tile) the whole wave runs the boundary
            // event loop below; the batch's mode records are requested now, under the tally and the stores
            const int keep = (t == nA + nB) ? 0 : NK_TILE - 1;
            double4 pre = make_double4(0, 0, 0, 0);
            if (ev_n > keep) {
                const int n = ev_n >= NK_TILE ? NK_TILE : ev_n;
                if (lane < n) pre = *reinterpret_cast<const double4 *>(d.modetab + L.ev.mode[eb + ev_n - n + lane]);
            }
            if (done && !NK_ABL(2)) nk_tally_one(d, L.cen, L.Tsv, L.bins, x, y, z, occ, omega, vx, vy, vz, do_flux, rep);
            if (done) {
                const int o = w + __popcll(mD & lower);
                if (o < d.segcap) { if (!NK_ABL(1)) nk_store(d, base + o, x, y, z, occ, nts, mode, facet, pid); }
                else atomicOr(d.overflow, 2);       // segment full at the commit of a tile
            }
            w += __popcll(mD);
            // ---- drain (Population.py:1546-1683): one boundary event per particle and pass; finished particles are
            // tallied and appended, absorbed ones vanish, the few that meet another wall go back to the buffer
            bool first = true;
            while (ev_n > keep) {
                const int n = ev_n >= NK_TILE ? NK_TILE : ev_n;
                const int e = eb + ev_n - n + lane;
                const bool eact = lane < n;
                NkParticle p;
                unsigned long long ppid = 0;
                double cts = 0.0;
                uint32_t evc = 0;
                int st = NK_EV_DEAD;
                p.alive = false;
                if (eact) {
                    p.x = L.ev.x[e]; p.y = L.ev.y[e]; p.z = L.ev.z[e]; p.occ = L.ev.occ[e]; p.nts = L.ev.nts[e];
                    p.mode = L.ev.mode[e]; p.facet = L.ev.facet[e]; ppid = L.ev.pid[e];
                    cts = L.ev.cts[e]; evc = (uint32_t)L.ev.evc[e];
                    const double4 ra = first ? pre : *reinterpret_cast<const double4 *>(d.modetab + p.mode);
                    p.omega = ra.x; p.vx = ra.y; p.vy = ra.z; p.vz = ra.w;
                    p.alive = true;
                    st = nk_event_one<ROUGH, RBF>(d, GEOM == 2 ? d.NG : 0, L.planes, L.faces, L.facets, L.cen, L.Tsv, L.resT, L.bins, p, cts, evc, ppid, step);
                }
                first = false;
                const bool alive = eact && st == NK_EV_DONE, more = eact && st == NK_EV_MORE;
                if (alive && !NK_ABL(2) && !NK_ABL(128)) nk_tally_one(d, L.cen, L.Tsv, L.bins, p.x, p.y, p.z, p.occ, p.omega, p.vx, p.vy, p.vz, do_flux, rep);
                const unsigned long long mA = __ballot(alive), mM = __ballot(more);
                if (alive) {
                    const int o = w + __popcll(mA & lower);
                    if (o < d.segcap) { if (!NK_ABL(1)) nk_store(d, base + o, p.x, p.y, p.z, p.occ, p.nts, p.mode, p.facet, ppid); }
                    else atomicOr(d.overflow, 4);   // segment full while appending event survivors
                }
                w += __popcll(mA);
                ev_n -= n;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // all lanes have read their entries
                if (more) {
                    const int q = eb + ev_n + __popcll(mM & lower);
                    L.ev.x[q] = p.x; L.ev.y[q] = p.y; L.ev.z[q] = p.z; L.ev.occ[q] = p.occ; L.ev.nts[q] = p.nts;
                    L.ev.mode[q] = p.mode; L.ev.facet[q] = p.facet; L.ev.pid[q] = ppid;
                    L.ev.cts[q] = cts; L.ev.evc[q] = (int)evc;
                }
                ev_n += __popcll(mM);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        if (lane == 0 && !NK_ABL(1)) d.seg_count[seg] = w < d.segcap ? w : d.segcap;
    }
    // tail: this workgroup's slice of the (reservoir, mode) table for the NEXT step (fill_reservoirs does not look at
    // the particles), into the other spawn buffer -- no separate launch, and it fills the sweep's ragged end
    if (flags & 2) {
        const int64_t RM = (int64_t)d.R * d.M;
        const int64_t per = (RM + gridDim.x - 1) / gridDim.x;
        const int64_t e0 = (int64_t)blockIdx.x * per;
        nk_emit_block<NK_EMIT_KMAX>(d, step + 1u, buf ^ 1, e0 < RM ? e0 : RM, e0 + per < RM ? e0 + per : RM, emit_wsum, &emit_base);
    }
    nk_lds_flush(d, L, blockIdx.x);
}

// nk_reserve with an unchanged number of segments: every segment's particles move to the start of its longer successor.
__global__ __launch_bounds__(NK_WG) void k_regrow(NkDev o, NkDev n) {
    for (int seg = blockIdx.x; seg < o.nseg; seg += gridDim.x) {
        const int cnt = o.seg_count[seg];
        const int64_t a = (int64_t)seg * o.segcap, b = (int64_t)seg * n.segcap;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            n.x[b + i] = o.x[a + i]; n.y[b + i] = o.y[a + i]; n.z[b + i] = o.z[a + i];
            n.occ[b + i] = o.occ[a + i]; n.nts[b + i] = o.nts[a + i];
            n.mode[b + i] = o.mode[a + i]; n.facet[b + i] = o.facet[a + i]; n.pid[b + i] = o.pid[a + i];
        }
    }
}

// Normalise, invert E(T), publish the new subvolume temperatures, history row: calculate_energy (Population.py:719-728)
// + refresh_temperatures (:692), run by ONE workgroup.
// History row: acc[NB] | T_sv[S] | E_sv[S] | flux_valid, 0, 0, overflow
__device__ __forceinline__ void nk_update_body(const NkDev &d, const double *acc, double *hist_row, int do_flux, int buf,
                                               long long *part) {
    // One workgroup, so everything here is a chain of memory latencies: all first-round loads are issued together, the
    // E(T) / T(E) tables are read as one 4-point window around the old temperature (T moves by a small fraction of
    // the 0.1 K table step per timestep; the general search is the fallback), and the segment counts stay in registers.
    const int tid = threadIdx.x, nth = blockDim.x;
    const int S = d.S, NB = d.NB, n = d.nE;
    constexpr int KMAX = NK_MAX_SEGMENTS / NK_WG;              // segments per thread of the update workgroup
    const int lane = tid & 63, wave = tid >> 6, nw = (nth + 63) >> 6;
    const int per = (d.nseg + nth - 1) / nth;                  // <= KMAX contiguous segments per thread
    int fr[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {                           // all loads in flight at once
        const int sgm = tid * per + k;
        const int fs = (k < per && sgm < d.nseg) ? d.segcap - d.seg_count[sgm] : 0;
        fr[k] = (d.segcap < NK_QUANT_SEGCAP || fs >= NK_MIN_FREE) ? fs : 0;
    }
    const double Ta = d.Tarr[0], Tb = d.Tarr[1], Tz = d.Tarr[n - 1], Ea = d.Earr[0], Ez = d.Earr[n - 1];
    for (int t = tid; t < S; t += nth) {
        const double Eraw = acc[t], Ns = acc[S + t];
        const double Told = d.T_ref_local ? d.T_sv[t] : d.T_ref;
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
        hist_row[NB + t] = Tnew;
        hist_row[NB + S + t] = E;
        if (!NK_ABL(2)) d.T_sv[t] = Tnew;
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
    // free-space prefix over the segments for the next step's spawn distribution: thread t owns the contiguous
    // segments [t*per, (t+1)*per); one workgroup scan of the per-thread sums
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) mine += fr[k];
    long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { long long v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    long long wbase = 0, run = 0;
    for (int w = 0; w < nw; ++w) { const long long v = part[w]; if (w < wave) wbase += v; run += v; }
    long long pre = wbase + incl - mine;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int sgm = tid * per + k;
        if (k < per && sgm < d.nseg) d.seg_free_prefix[sgm] = pre;
        pre += fr[k];
    }
    if (d.res_gen == 2) for (int r = tid; r < d.R; r += nth) d.nleave_prev[r] = (int32_t)acc[5 * S + r];
    if (tid == 0) {
        d.seg_free_prefix[d.nseg] = run;
        d.alloc_count[buf] = 0;                      // consumed by this step's sweep; refilled two steps on
        hist_row[NB + 2 * S + 0] = (double)do_flux;
        hist_row[NB + 2 * S + 1] = 0.0;
        hist_row[NB + 2 * S + 2] = 0.0;
        hist_row[NB + 2 * S + 3] = (double)*d.overflow;
    }
}

// Column sums of the tally rows, fixed order -> bitwise reproducible for a given grid.  One workgroup per column.
// fuse != 0 (single rank): the workgroup that finishes last also runs the update, saving a launch.
__global__ __launch_bounds__(NK_WG) void k_reduce(NkDev d, int rows, double *acc, double *hist_row, int do_flux, int buf,
                                                  int fuse) {
    __shared__ double sh[NK_WG];
    __shared__ long long part[NK_WG];
    __shared__ int last;
    const int b = blockIdx.x, NB = d.NB;
    double v = 0.0;
    for (int r = threadIdx.x; r < rows; r += NK_WG) v += d.partials[(int64_t)r * NB + b];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = NK_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[b] = sh[0];
    if (!fuse) return;
    if (threadIdx.x == 0) {
        __threadfence();
        last = (atomicAdd(d.ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *d.ticket = 0;
    __threadfence();
    nk_update_body(d, acc, hist_row, do_flux, buf, part);
}

// The update as its own launch (after the RCCL all-reduce when nranks > 1).
__global__ __launch_bounds__(NK_WG) void k_update(NkDev d, const double *acc, double *hist_row, int do_flux, int buf) {
    __shared__ long long part[NK_WG];
    nk_update_body(d, acc, hist_row, do_flux, buf, part);
}

// Stand-alone lifetime_scattering (flushes the deferred relaxation).
__global__ __launch_bounds__(NK_WG) void k_relax(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<0, false>(d, smem, L);
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const int mode = d.mode[i];
            const double4 *mrec = reinterpret_cast<const double4 *>(d.modetab + mode);
            const double4 ra = mrec[0], rb = mrec[1];
            d.occ[i] = nk_relax(d, L, ra, rb, d.x[i], d.y[i], d.z[i], d.occ[i], mode);
        }
    }
}

// timesteps_to_boundary for every particle (Population.py:310-314)
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_init_boundaries(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, false>(d, smem, L);
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const NkMode *rec = d.modetab + d.mode[i];
            double tc; int fc;
            NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, d.x[i], d.y[i], d.z[i], rec->vx, rec->vy, rec->vz, tc, fc);
            d.nts[i] = tc / d.dt;
            d.facet[i] = fc;
        }
    }
}

// contains_check (Population.py:1712-1722) + Mesh.sample_volume (Mesh.py:890-904)
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_contains(NkDev d, uint32_t step) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, false>(d, smem, L);
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            double x = d.x[i], y = d.y[i], z = d.z[i];
            bool out = x < d.bbox[0] - 1e-10 || y < d.bbox[1] - 1e-10 || z < d.bbox[2] - 1e-10 || x > d.bbox[3] + 1e-10 ||
                       y > d.bbox[4] + 1e-10 || z > d.bbox[5] + 1e-10;
            if (!out) continue;
            const uint64_t pid = d.pid[i];
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
            const NkMode *rec = d.modetab + d.mode[i];
            double tc; int fc;
            NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x, y, z, rec->vx, rec->vy, rec->vz, tc, fc);
            d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = tc / d.dt; d.facet[i] = fc;
        }
    }
}

// {omega, v, tau[row0..row0+3]} records for one-gather-per-particle access
__global__ void k_build_modetab(const double *omega, const double *vg, const double *tau, int M, int NT, int row0,
                                NkMode *out) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    NkMode r;
    r.omega = omega[m]; r.vx = vg[3 * m]; r.vy = vg[3 * m + 1]; r.vz = vg[3 * m + 2];
    for (int k = 0; k < NK_TAU_ROWS; ++k) {
        int row = row0 + k;
        r.tau[k] = (row >= 0 && row < NT) ? tau[(int64_t)row * M + m] : 0.0;
    }
    out[m] = r;
}

// ---- parity taps: the reference's primitives evaluated on the device
template <int GEOM>
__global__ __launch_bounds__(NK_WG) void k_tap_find_boundary(NkDev d, int64_t n, const double *x, const double *v,
                                                             double *xc, double *tc, int32_t *fc) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<GEOM, false>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    double t; int f;
    NK_RAY(GEOM, d, L, NK_TREE_NO_SKIP, x[3 * i], x[3 * i + 1], x[3 * i + 2], v[3 * i], v[3 * i + 1], v[3 * i + 2], t, f);
    tc[i] = t; fc[i] = f;
    for (int k = 0; k < 3; ++k) xc[3 * i + k] = x[3 * i + k] + t * v[3 * i + k];
}
__global__ __launch_bounds__(NK_WG) void k_tap_classify(NkDev d, int64_t n, const double *x, int32_t *id) {
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i < n) id[i] = nk_classify(d, d.centers, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
}
__global__ __launch_bounds__(NK_WG) void k_tap_eval(NkDev d, int what, int64_t n, const double *a, const int32_t *mode,
                                                    double *out) {
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    switch (what) {
        case 0: out[i] = nk_occupation(d, a[i], d.modetab[mode[i]].omega); break;
        case 1: { const NkMode *rec = d.modetab + mode[i]; out[i] = nk_lifetime(d, rec->tau[0], rec->tau[1], rec->tau[2], rec->tau[3], a[i], mode[i]); break; }
        case 2: out[i] = nk_T_of_E(d, a[i]); break;
        case 3: out[i] = nk_E_of_T(d, a[i]); break;
        default: out[i] = nk_interp_T(d, d.centers, d.T_sv, a[3 * i], a[3 * i + 1], a[3 * i + 2], -1); break;
    }
}
__global__ __launch_bounds__(NK_WG) void k_tap_reflect(NkDev d, int64_t n, const int32_t *facet, const int32_t *mode_in,
                                                       const double *col, const double *n_in, const double *om_in,
                                                       const double *r_spec, const double *r_deg, const double *r_diff,
                                                       int32_t *mode_out, double *n_out, double *om_out) {
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    int mo; double no, oo;
    nk_reflect(d, d.centers, d.T_sv, d.facets[facet[i]].rough, mode_in[i], col[3 * i], col[3 * i + 1], col[3 * i + 2],
               n_in[i], om_in[i], r_spec[i], r_deg ? r_deg[i] : 0.0, r_diff[i], mo, no, oo);
    mode_out[i] = mo; n_out[i] = no; om_out[i] = oo;
}
// Counter calibration: coalesced 8-byte-per-lane sweeps with a KNOWN byte count (44 B read + 32 B written per live
// particle), so FETCH_SIZE / WRITE_SIZE readings of k_sweep can be scaled (MI355X_MICROARCH.md: FETCH_SIZE is
// uncalibrated for accesses other than 16 B/lane).
__global__ __launch_bounds__(NK_WG) void k_cal_stream(NkDev d) {
    for (int seg = blockIdx.x; seg < d.nseg; seg += gridDim.x) {
        const int64_t base = (int64_t)seg * d.segcap;
        const int count = d.seg_count[seg];
        for (int k = threadIdx.x; k < count; k += NK_WG) {
            const int64_t i = base + k;
            const int mode = d.mode[i];
            double x = d.x[i], y = d.y[i], z = d.z[i], occ = d.occ[i], nts = d.nts[i];
            if (mode == -123456789) { x += occ; }                 // keeps the occ load alive; never true
            d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = nts;
        }
    }
}
__global__ void k_tap_uniform(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *out) {
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
__global__ __launch_bounds__(NK_WG) void k_specular_prepare(int M, const double *v, const double *omega, const double *delta,
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
__global__ __launch_bounds__(NK_WG) void k_specular_pairs(int M, const NkSpecMode *modes, const double *sorted_vx, double vmax,
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
