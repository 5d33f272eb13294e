// nk_kernels.h -- the HIP kernels of libnanokappa_hip.so (gfx950 / MI355X only).
//
// Stream order of one timestep (reference Population.run_timestep, Population.py:1724-1769):
//   [k_relax + k_contains every `contains_every` steps]           contains_check       :1712-1722
//   k_step        relax(previous step) -> drift -> tally, or hand the particle to the event queue
//                                                                  lifetime_scattering  :1701-1710 (deferred, see below)
//                                                                  drift                :790-795
//                                                                  calculate_energy     :704-717
//   k_emit_count  which modes enter at each reservoir, how many    fill_reservoirs      :356-455
//   k_spawn       one lane per entering particle                   Mesh.sample_surface  Mesh.py:923-951,
//                                                                  add_reservoir_particles :525-552
//   k_events      one lane per particle that meets a boundary      boundary_scattering  :1546-1683
//   k_reduce      deterministic column sums of the per-workgroup tally rows
//   (RCCL all-reduce of the tally vector when nranks > 1)
//   k_update      normalisation, E -> T, bookkeeping, history row  calculate_energy :719-728, refresh_temperatures :692
//
// Deferred relaxation: the reference relaxes occupations at the END of step k with the temperatures of step k.  Those
// need the global tally of step k, so the relaxation is carried into the BEGINNING of the step kernel of step k+1
// (same positions, same T_sv): one streaming pass per step instead of two.  A pending relaxation is flushed by
// k_relax before anything observes the particles (download, contains_check).
//
// Event queue: in a 20 nm box a third of the particles meets a boundary every step.  Running the event loop inside
// the streaming kernel would make every wave pay for it; instead k_step appends those slots to a queue and k_events
// processes them densely (64 busy lanes per wave), with the ray-casting tables in LDS.
#pragma once
#include "nk_device.h"

// =================================================================================== LDS carve-up
struct NkLds {
    double *Tsv, *cen;
    NkBins bins;
    const double *planes, *faces;
    const NkFacet *facets;
};

__host__ __device__ inline size_t nk_lds_bytes(int S, int R, int F, int NP, int Fc, bool geom) {
    int Fl = (geom && F <= NK_LDS_FACES) ? F : 0;
    int Pl = Fl ? NP : 0;
    int Fcl = (geom && Fc <= NK_LDS_FACES) ? Fc : 0;
    size_t nd = (size_t)S + 3 * S + NK_NREP * S + NK_NREP * 3 * S + 4 * R + (size_t)Fl * NK_FACE_DOUBLES +
                (size_t)Pl * NK_PLANE_DOUBLES;
    size_t bytes = nd * 8 + (size_t)Fcl * sizeof(NkFacet) + (size_t)(NK_NREP * S + R + 1) * 4;
    return (bytes + 15) & ~(size_t)15;
}

// Cooperative fill of the read-only tables and zeroing of the bins; ends with a barrier.  GEOM = also stage the
// ray-casting tables (kernels that never cast rays skip them).
template <bool GEOM>
__device__ __forceinline__ void nk_lds_setup(const NkDev &d, unsigned char *smem, NkLds &L) {
    const int S = d.S, R = d.R;
    const int Fl = (GEOM && d.F <= NK_LDS_FACES) ? d.F : 0;
    const int Pl = Fl ? d.NP : 0;
    const int Fcl = (GEOM && d.Fc <= NK_LDS_FACES) ? d.Fc : 0;
    double *p = (double *)smem;
    L.Tsv = p; p += S;
    L.cen = p; p += 3 * S;
    L.bins.E = p; p += NK_NREP * S;
    L.bins.flux = p; p += NK_NREP * 3 * S;
    L.bins.resb = p; p += 4 * R;
    double *faces = p; p += (size_t)Fl * NK_FACE_DOUBLES;
    double *planes = p; p += (size_t)Pl * NK_PLANE_DOUBLES;
    NkFacet *facets = (NkFacet *)p;
    unsigned int *u = (unsigned int *)(facets + Fcl);
    L.bins.N = u; u += NK_NREP * S;
    L.bins.nleave = u; u += R;
    L.bins.misc = u;
    const int t = threadIdx.x;
    for (int i = t; i < S; i += NK_WG) L.Tsv[i] = d.T_sv[i];
    for (int i = t; i < 3 * S; i += NK_WG) L.cen[i] = d.centers[i];
    for (int i = t; i < NK_NREP * S; i += NK_WG) { L.bins.E[i] = 0.0; L.bins.N[i] = 0u; }
    for (int i = t; i < NK_NREP * 3 * S; i += NK_WG) L.bins.flux[i] = 0.0;
    for (int i = t; i < 4 * R; i += NK_WG) L.bins.resb[i] = 0.0;
    for (int i = t; i < R; i += NK_WG) L.bins.nleave[i] = 0u;
    if (t == 0) L.bins.misc[0] = 0u;
    for (int i = t; i < Fl * NK_FACE_DOUBLES; i += NK_WG) faces[i] = d.faces[i];
    for (int i = t; i < Pl * NK_PLANE_DOUBLES; i += NK_WG) planes[i] = d.planes[i];
    {
        const int nw = Fcl * (int)(sizeof(NkFacet) / 4);
        const int32_t *src = (const int32_t *)d.facets;
        int32_t *dst = (int32_t *)facets;
        for (int i = t; i < nw; i += NK_WG) dst[i] = src[i];
    }
    L.faces = Fl ? faces : d.faces;
    L.planes = Fl ? planes : d.planes;
    L.facets = Fcl ? facets : d.facets;
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

// ========================================================================================= kernels
// Deferred lifetime_scattering (Population.py:1701-1710) for one particle.
__device__ __forceinline__ double nk_relax(const NkDev &d, const NkLds &L, const NkMode &rec, double x, double y, double z,
                                           double occ, int mode) {
    double T = nk_interp_T(d, L.cen, L.Tsv, x, y, z, -1);
    double tau = nk_lifetime(d, rec, T, mode);
    double n0 = nk_occupation(d, T, rec.omega);
    return (tau > 0.0) ? n0 + (occ - n0) * exp(-d.dt / tau) : n0;
}

// The streaming kernel: every live slot once per step.
__global__ __launch_bounds__(NK_WG) void k_step(NkDev d, int do_relax, int do_flux) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<false>(d, smem, L);
    const int64_t n = *d.n_slots;
    const int rep = threadIdx.x & (NK_NREP - 1);
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    // Software pipeline: the next slot's state is requested before this slot's arithmetic starts, so the HBM round
    // trip of iteration k+1 overlaps the exp/divide chains of iteration k (the kernel is latency-bound: 86 % of wave
    // cycles were s_waitcnt stalls without it).
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    int modeN = -1;
    double xN = 0, yN = 0, zN = 0, occN = 0, ntsN = 0;
    if (i < n) { modeN = d.mode[i]; xN = d.x[i]; yN = d.y[i]; zN = d.z[i]; occN = d.occ[i]; ntsN = d.nts[i]; }
    for (; i < n; i += stride) {
        const int mode = modeN;
        double x = xN, y = yN, z = zN, occ = occN, nts = ntsN;
        NkMode rec = d.modetab[(mode >= 0 && !(d.dbg & 4)) ? mode : 0];   // gather first, prefetch behind it (vmcnt is in-order)
        if (d.dbg & 4) { rec.omega = 10.0 + (mode & 63); rec.vx = (mode & 7) - 3.5; rec.vy = ((mode >> 3) & 7) - 3.5; rec.vz = ((mode >> 6) & 7) - 3.5; }
        const int64_t in = i + stride;
        if (in < n) { modeN = d.mode[in]; xN = d.x[in]; yN = d.y[in]; zN = d.z[in]; occN = d.occ[in]; ntsN = d.nts[in]; }
        if (mode < 0) continue;                                   // dead slot (absorbed, not yet reused)
        if (do_relax && !(d.dbg & 16)) occ = nk_relax(d, L, rec, x, y, z, occ, mode);
        x += rec.vx * d.dt; y += rec.vy * d.dt; z += rec.vz * d.dt;                 // drift, Population.py:793
        nts -= 1.0;                                                                 // :795
        if (!(d.dbg & 8)) { d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = nts; }
        if (do_relax) d.occ[i] = occ;
        if (nts < 0.0) {
            if (!(d.dbg & 2)) nk_evq_push(d, i);                  // boundary reached inside this step -> k_events
        } else if (!(d.dbg & 1)) {
            nk_tally_one(d, L.cen, L.Tsv, L.bins, x, y, z, occ, rec.omega, rec.vx, rec.vy, rec.vz, do_flux != 0, rep);
        }
    }
    nk_lds_flush(d, L, blockIdx.x);
}

// Which modes enter at each reservoir this step, and how many particles of each:
// fill_reservoirs 'constant' (Population.py:358-370) / 'fixed_rate' (:408-420).  One lane per (reservoir, mode);
// every entering particle gets one 64-bit record (rm << 12 | level) in spawn_list.
__global__ __launch_bounds__(NK_WG) void k_emit_count(NkDev d, uint32_t step) {
    const int64_t RM = (int64_t)d.R * d.M;
    const int64_t rm = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = 0, c_mine = 0;
    if (rm < RM) {
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
        if (c > 0) d.res_cval[rm] = cv;
        if (d.nranks == 1) c_mine = c;
        else for (int level = c; level >= 1; --level) c_mine += (((rm + level + (int64_t)step) % d.nranks) == d.rank);
    }
    // wave-aggregated allocation in spawn_list: inclusive scan over the 64 lanes, one atomic per wave
    int incl = c_mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const int total = __shfl(incl, 63, 64);
    // one global atomic per workgroup (a single hot counter serves only ~90 atomics/us)
    __shared__ int wsum[NK_WG / 64];
    __shared__ int bbase;
    const int wave = threadIdx.x >> 6;
    if (lane == 63) wsum[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < NK_WG / 64; ++w) t += wsum[w];
        bbase = t > 0 ? atomicAdd(d.alloc_count, t) : 0;
    }
    __syncthreads();
    int base = bbase;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int64_t g = (int64_t)base + incl - c_mine;
    for (int level = c; level >= 1 && c_mine > 0; --level) {
        if (d.nranks > 1 && (((rm + level + (int64_t)step) % d.nranks) != d.rank)) continue;
        if (g < d.spawn_cap) d.spawn_list[g] = ((uint64_t)rm << 12) | (uint64_t)level;
        else *d.overflow = 1;
        ++g;
    }
}

// One lane per entering particle: position on the facet (Mesh.sample_surface, Mesh.py:923-951), entry time
// (Population.py:391-394 / :440-443), first boundary, advance by the time spent inside (:535-536).
__global__ __launch_bounds__(NK_WG) void k_spawn(NkDev d, uint32_t step, int do_flux, int row0) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<true>(d, smem, L);
    int64_t total = *d.alloc_count;
    if (total > d.spawn_cap) total = d.spawn_cap;
    const unsigned long long head = *d.fl_head;
    const int64_t avail = *d.fl_avail;
    const int64_t ns0 = *d.n_slots;
    const int rep = threadIdx.x & (NK_NREP - 1);
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    if (threadIdx.x == 0 && blockIdx.x == 0) L.bins.misc[0] = (unsigned int)total;   // "emitted" column
    for (int64_t g = (int64_t)blockIdx.x * NK_WG + threadIdx.x; g < total; g += stride) {
        const uint64_t recd = d.spawn_list[g];
        const int64_t rm = (int64_t)(recd >> 12);
        const int level = (int)(recd & 0xFFFu);
        const int r = (int)(rm / d.M), m = (int)(rm - (int64_t)r * d.M);
        const int64_t slot = g < avail ? (int64_t)d.free_ring[(head + (unsigned long long)g) % (unsigned long long)d.cap]
                                       : ns0 + (g - avail);
        if (slot >= d.cap) { *d.overflow = 1; continue; }
        const uint64_t pid = ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)rm << 12) | (uint64_t)level;
        double uf, us, ur, ut;
        nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT, uf, us);
        nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT + 1, ur, ut);
        const double prob = d.enter_prob[rm];
        const double dt_in = (level == 1) ? d.dt * (1.0 - (d.res_cval[rm] / prob))
                                          : d.dt * (1.0 - ((double)(level - 1) + ut) / prob);
        const int facet = d.res_facet[r];
        const int f0 = d.facet_face_off[facet], nf = d.facet_face_off[facet + 1] - f0;
        int a = nk_ss_right(d.facet_face_cdf + f0, nf, uf);                          // np.random.choice, Mesh.py:937
        a = a > nf - 1 ? nf - 1 : a;
        const double *vx = d.face_verts + 9 * (int64_t)d.facet_face_idx[f0 + a];
        const double sq = sqrt(us);
        const double a0 = 1.0 - sq, a1 = (1.0 - ur) * sq, a2 = ur * sq;              // Mesh.py:945-947
        const double x0 = a0 * vx[0] + a1 * vx[3] + a2 * vx[6];
        const double y0 = a0 * vx[1] + a1 * vx[4] + a2 * vx[7];
        const double z0 = a0 * vx[2] + a1 * vx[5] + a2 * vx[8];
        const NkMode rec = d.modetab[m];
        const double occ = nk_occupation(d, d.res_T[r], rec.omega);                  // Population.py:506
        double tc; int fcn;
        nk_find_boundary(L.planes, L.faces, d.NP, d.tol, x0, y0, z0, rec.vx, rec.vy, rec.vz, tc, fcn);
        const double nts = tc / d.dt - dt_in / d.dt;                                 // :535
        const double x = x0 + rec.vx * dt_in, y = y0 + rec.vy * dt_in, z = z0 + rec.vz * dt_in;   // :536
        d.x[slot] = x; d.y[slot] = y; d.z[slot] = z; d.occ[slot] = occ; d.nts[slot] = nts;
        d.mode[slot] = m; d.facet[slot] = fcn; d.pid[slot] = pid;
        if (nts < 0.0) {
            nk_evq_push(d, slot);
        } else {
            nk_tally_one(d, L.cen, L.Tsv, L.bins, x, y, z, occ, rec.omega, rec.vx, rec.vy, rec.vz, do_flux != 0, rep);
        }
    }
    nk_lds_flush(d, L, row0 + blockIdx.x);
}

// One lane per queued particle: the boundary event loop, then the tally (or the free ring if it was absorbed).
__global__ __launch_bounds__(NK_WG) void k_events(NkDev d, uint32_t step, int do_flux, int row0) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<true>(d, smem, L);
    // workgroup b drains shard b % SHARDS together with the other workgroups of the same residue
    const int shard = blockIdx.x & (NK_EVQ_SHARDS - 1);
    int64_t n = d.evq_count[shard * NK_EVQ_PAD];
    if (n > d.evq_seg) n = d.evq_seg;
    const int32_t *queue = d.evq + (int64_t)shard * d.evq_seg;
    const int rep = threadIdx.x & (NK_NREP - 1);
    const int64_t stride = (int64_t)(gridDim.x / NK_EVQ_SHARDS) * NK_WG;
    for (int64_t q = (int64_t)(blockIdx.x / NK_EVQ_SHARDS) * NK_WG + threadIdx.x; q < n; q += stride) {
        const int64_t i = queue[q];
        NkParticle p;
        p.x = d.x[i]; p.y = d.y[i]; p.z = d.z[i]; p.occ = d.occ[i]; p.nts = d.nts[i];
        p.mode = d.mode[i]; p.facet = d.facet[i]; p.alive = true;
        const NkMode *rec = d.modetab + p.mode;
        p.omega = rec->omega; p.vx = rec->vx; p.vy = rec->vy; p.vz = rec->vz;
        nk_events(d, L.planes, L.faces, L.facets, L.cen, L.Tsv, L.bins, p, d.pid[i], step);
        if (p.alive) {
            nk_tally_one(d, L.cen, L.Tsv, L.bins, p.x, p.y, p.z, p.occ, p.omega, p.vx, p.vy, p.vz, do_flux != 0, rep);
            d.x[i] = p.x; d.y[i] = p.y; d.z[i] = p.z; d.occ[i] = p.occ; d.nts[i] = p.nts;
            d.mode[i] = p.mode; d.facet[i] = p.facet;
        } else {
            d.mode[i] = -1;
            unsigned long long t = atomicAdd(d.fl_tail, 1ull);
            d.free_ring[t % (unsigned long long)d.cap] = (int32_t)i;
        }
    }
    nk_lds_flush(d, L, row0 + blockIdx.x);
}

// Column sums of the tally rows, fixed order -> bitwise reproducible for a given grid.
__global__ __launch_bounds__(NK_WG) void k_reduce(const double *partials, int rows, int NB, double *acc) {
    __shared__ double sh[NK_WG];
    const int b = blockIdx.x;
    double v = 0.0;
    for (int r = threadIdx.x; r < rows; r += NK_WG) v += partials[(int64_t)r * NB + b];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = NK_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[b] = sh[0];
}

// Normalise, invert E(T), publish the new subvolume temperatures, bookkeeping, history row.
// calculate_energy (Population.py:719-728) + refresh_temperatures (:692).
// History row: acc[NB] | T_sv[S] | E_sv[S] | flux_valid, n_slots, free slots, overflow
__global__ void k_update(NkDev d, const double *acc, double *hist_row, int do_flux) {
    const int t = threadIdx.x;
    const int S = d.S, NB = d.NB;
    double Tnew = 0.0;
    if (t < S) {
        double Eraw = acc[t], Ns = acc[S + t];
        double norm;
        if (d.norm_fixed) norm = d.active_modes / (d.particle_density * d.sv_volume[t]);
        else { norm = d.active_modes / Ns; if (isnan(norm)) norm = 0.0; }
        double E = Eraw * norm / d.QV;
        double ref = nk_E_of_T(d, d.T_ref_local ? d.T_sv[t] : d.T_ref);
        E += ref;
        Tnew = nk_T_of_E(d, E);
        hist_row[NB + t] = Tnew;
        hist_row[NB + S + t] = E;
    }
    for (int b = t; b < NB; b += blockDim.x) hist_row[b] = acc[b];
    __syncthreads();
    if (t < S) d.T_sv[t] = Tnew;
    if (t < NK_EVQ_SHARDS) d.evq_count[t * NK_EVQ_PAD] = 0;
    if (t == 0) {
        int64_t em = *d.alloc_count;
        if (em > d.spawn_cap) em = d.spawn_cap;
        const int64_t avail = *d.fl_avail;
        const int64_t popped = em < avail ? em : avail;
        int64_t ns = *d.n_slots + (em - popped);
        if (ns > d.cap) ns = d.cap;
        const unsigned long long head = *d.fl_head + (unsigned long long)popped;
        *d.fl_head = head;
        *d.n_slots = ns;
        *d.fl_avail = (int64_t)(*d.fl_tail - head);
        *d.alloc_count = 0;
        hist_row[NB + 2 * S + 0] = (double)do_flux;
        hist_row[NB + 2 * S + 1] = (double)ns;
        hist_row[NB + 2 * S + 2] = (double)(*d.fl_tail - head);
        hist_row[NB + 2 * S + 3] = (double)*d.overflow;
    }
}

// Stand-alone lifetime_scattering (flushes the deferred relaxation).
__global__ __launch_bounds__(NK_WG) void k_relax(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<false>(d, smem, L);
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        if (mode < 0) continue;
        const NkMode rec = d.modetab[mode];
        d.occ[i] = nk_relax(d, L, rec, d.x[i], d.y[i], d.z[i], d.occ[i], mode);
    }
}

// timesteps_to_boundary for every particle (Population.py:310-314)
__global__ __launch_bounds__(NK_WG) void k_init_boundaries(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<true>(d, smem, L);
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        if (mode < 0) continue;
        const NkMode *rec = d.modetab + mode;
        double tc; int fc;
        nk_find_boundary(L.planes, L.faces, d.NP, d.tol, d.x[i], d.y[i], d.z[i], rec->vx, rec->vy, rec->vz, tc, fc);
        d.nts[i] = tc / d.dt;
        d.facet[i] = fc;
    }
}

// contains_check (Population.py:1712-1722) + Mesh.sample_volume (Mesh.py:890-904)
__global__ __launch_bounds__(NK_WG) void k_contains(NkDev d, uint32_t step) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<true>(d, smem, L);
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        if (mode < 0) continue;
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
        for (int k = 0; k < 4; ++k) { a[k] = -log(u[1 + k]); asum += a[k]; }
        const double *sp = d.simplex_pts + 12 * (int64_t)s;
        x = y = z = 0.0;
        for (int k = 0; k < 4; ++k) { double w = a[k] / asum; x += w * sp[3 * k]; y += w * sp[3 * k + 1]; z += w * sp[3 * k + 2]; }
        const NkMode *rec = d.modetab + mode;
        double tc; int fc;
        nk_find_boundary(L.planes, L.faces, d.NP, d.tol, x, y, z, rec->vx, rec->vy, rec->vz, tc, fc);
        d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = tc / d.dt; d.facet[i] = fc;
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
__global__ __launch_bounds__(NK_WG) void k_tap_find_boundary(NkDev d, int64_t n, const double *x, const double *v,
                                                             double *xc, double *tc, int32_t *fc) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup<true>(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    double t; int f;
    nk_find_boundary(L.planes, L.faces, d.NP, d.tol, x[3 * i], x[3 * i + 1], x[3 * i + 2], v[3 * i], v[3 * i + 1],
                     v[3 * i + 2], t, f);
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
        case 1: { const NkMode rec = d.modetab[mode[i]]; out[i] = nk_lifetime(d, rec, a[i], mode[i]); break; }
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
// Counter calibration: the same coalesced 8-byte-per-lane sweep as k_step with a KNOWN byte count
// (44 B read + 32 B written per slot), so FETCH_SIZE / WRITE_SIZE readings of k_step can be scaled
// (MI355X_MICROARCH.md: FETCH_SIZE is uncalibrated for accesses other than 16 B/lane).
__global__ __launch_bounds__(NK_WG) void k_cal_stream(NkDev d) {
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        double x = d.x[i], y = d.y[i], z = d.z[i], occ = d.occ[i], nts = d.nts[i];
        if (mode == -123456789) { x += occ; }                     // keeps the occ load alive; never true
        d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = nts;
    }
}
__global__ void k_tap_uniform(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *out) {
    double a, b;
    nk_uniform2_dev(seed, pid, step, tag, a, b);
    out[0] = a; out[1] = b;
}
