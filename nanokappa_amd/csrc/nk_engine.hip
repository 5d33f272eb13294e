// nk_engine.hip -- kernels and C ABI of libnanokappa_hip.so (gfx950 / MI355X only).
//
// Stream order of one timestep (reference Population.run_timestep, Population.py:1724-1769):
//   [k_relax + k_contains every `contains_every` steps]          contains_check        :1712-1722
//   k_step    relax(previous step) -> drift -> boundary events -> energy/flux tally
//                                                                 lifetime_scattering  :1701-1710 (deferred, see below)
//                                                                 drift                :790-795
//                                                                 boundary_scattering  :1546-1683
//                                                                 calculate_energy     :704-717
//   k_emit    reservoir emission + the same event loop + tally    fill_reservoirs      :356-523, add_reservoir_particles :525-552
//   k_reduce  deterministic column sums of the per-workgroup tally rows
//   (RCCL all-reduce of the tally vector when nranks > 1)
//   k_update  normalisation, E -> T, bookkeeping, history row     calculate_energy     :719-728, refresh_temperatures :692
//
// Deferred relaxation: the reference relaxes occupations at the END of step k with the temperatures of step k.
// Those temperatures need the global tally of step k, so the relaxation is carried into the BEGINNING of the step
// kernel of step k+1 (same particle positions, same T_sv): one streaming pass per step instead of two.  A pending
// relaxation is flushed by k_relax before anything observes the particles (download, contains_check).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/nanokappa_hip.h"
#include "nk_device.h"

// =================================================================================== LDS carve-up
struct NkLds {
    double *Tsv, *cen;
    NkBins bins;
    const double *faces;
    const int32_t *face_facet;
    const NkFacet *facets;
};

__host__ __device__ inline size_t nk_lds_bytes(int S, int R, int F, int Fc) {
    int Fl = F <= NK_LDS_FACES ? F : 0;
    int Fcl = Fc <= NK_LDS_FACES ? Fc : 0;
    size_t nd = (size_t)S + 3 * S + NK_NREP * S + NK_NREP * 3 * S + 4 * R + (size_t)Fl * NK_FACE_DOUBLES;
    size_t bytes = nd * 8 + (size_t)Fcl * sizeof(NkFacet) + (size_t)Fl * 4 + (size_t)(NK_NREP * S + R + 1) * 4;
    return (bytes + 15) & ~(size_t)15;
}

// Cooperative fill of the read-only tables and zeroing of the bins; ends with a barrier.
__device__ __forceinline__ void nk_lds_setup(const NkDev &d, unsigned char *smem, NkLds &L) {
    const int S = d.S, R = d.R;
    const int Fl = d.F <= NK_LDS_FACES ? d.F : 0;
    const int Fcl = d.Fc <= NK_LDS_FACES ? d.Fc : 0;
    double *p = (double *)smem;
    L.Tsv = p; p += S;
    L.cen = p; p += 3 * S;
    L.bins.E = p; p += NK_NREP * S;
    L.bins.flux = p; p += NK_NREP * 3 * S;
    L.bins.resb = p; p += 4 * R;
    double *faces = p; p += (size_t)Fl * NK_FACE_DOUBLES;
    NkFacet *facets = (NkFacet *)p;
    int32_t *ff = (int32_t *)(facets + Fcl);
    unsigned int *u = (unsigned int *)(ff + Fl);
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
    for (int i = t; i < Fl; i += NK_WG) ff[i] = d.face_facet[i];
    {
        const int nw = Fcl * (int)(sizeof(NkFacet) / 4);
        const int32_t *src = (const int32_t *)d.facets;
        int32_t *dst = (int32_t *)facets;
        for (int i = t; i < nw; i += NK_WG) dst[i] = src[i];
    }
    L.faces = Fl ? faces : d.faces;
    L.face_facet = Fl ? ff : d.face_facet;
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
__device__ __forceinline__ double nk_relax(const NkDev &d, const NkLds &L, double x, double y, double z, double occ,
                                           double omega, int mode) {
    double T = nk_interp_T(d, L.cen, L.Tsv, x, y, z, -1);
    double tau = nk_lifetime(d, T, mode);
    double n0 = nk_occupation(d, T, omega);
    return (tau > 0.0) ? n0 + (occ - n0) * exp(-d.dt / tau) : n0;
}

__global__ __launch_bounds__(NK_WG) void k_step(NkDev d, uint32_t step, int do_relax, int do_flux) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
    const int64_t n = *d.n_slots;
    const int rep = threadIdx.x & (NK_NREP - 1);
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode0 = d.mode[i];
        if (mode0 < 0) continue;                                  // dead slot (absorbed, not yet reused)
        NkParticle p;
        p.x = d.x[i]; p.y = d.y[i]; p.z = d.z[i]; p.occ = d.occ[i]; p.nts = d.nts[i];
        p.mode = mode0; p.facet = d.facet[i]; p.alive = true;
        const double4 rec = d.modetab[mode0];
        p.omega = rec.x; p.vx = rec.y; p.vy = rec.z; p.vz = rec.w;
        if (do_relax) p.occ = nk_relax(d, L, p.x, p.y, p.z, p.occ, p.omega, p.mode);
        p.x += p.vx * d.dt; p.y += p.vy * d.dt; p.z += p.vz * d.dt;                 // drift, Population.py:793
        p.nts -= 1.0;                                                               // :795
        const bool had_event = p.nts < 0.0;
        if (had_event) nk_events(d, L.faces, L.face_facet, L.facets, L.cen, L.Tsv, L.bins, p, d.pid[i], step);
        if (p.alive) {
            nk_tally_one(d, L.cen, L.Tsv, L.bins, p, do_flux != 0, rep);
            d.x[i] = p.x; d.y[i] = p.y; d.z[i] = p.z; d.nts[i] = p.nts;
            if (do_relax || had_event) d.occ[i] = p.occ;
            if (had_event) { d.mode[i] = p.mode; d.facet[i] = p.facet; }
        } else {
            d.mode[i] = -1;
            int k = atomicAdd(d.free_top, 1);
            d.free_list[k] = (int32_t)i;
        }
    }
    nk_lds_flush(d, L, blockIdx.x);
}

// Reservoir emission: one lane per (reservoir, mode) table entry.
// fill_reservoirs 'constant' (Population.py:358-406) / 'fixed_rate' (:408-455), Mesh.sample_surface (Mesh.py:923-951),
// add_reservoir_particles (Population.py:525-552), then the shared event loop and tally.
__global__ __launch_bounds__(NK_WG) void k_emit(NkDev d, uint32_t step, int do_flux, int row0) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
    const int64_t RM = (int64_t)d.R * d.M;
    const int64_t rm = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    const int rep = threadIdx.x & (NK_NREP - 1);
    const int lane = threadIdx.x & 63;
    const int ft0 = *d.free_top;
    const int64_t ns0 = *d.n_slots;
    int c = 0, c_mine = 0;
    double prob = 0.0, cnt = 0.0;
    if (rm < RM) {
        prob = d.enter_prob[rm];
        double fixed = floor(prob);
        int mask;
        if (d.res_gen == 0) {
            double cv = d.res_counter[rm] + (prob - fixed);
            mask = cv >= 1.0;
            cv -= (double)mask;
            d.res_counter[rm] = cv;
            cnt = cv;
        } else {
            double d0, d1;
            nk_uniform2_dev(d.seed, (uint64_t)rm | 0xFFFFFFFF00000000ull, step, NK_TAG_DICE, d0, d1);
            mask = d0 <= (prob - fixed);
            cnt = d0;
        }
        c = (int)fixed + mask;
        if (d.nranks == 1) c_mine = c;
        else for (int level = c; level >= 1; --level) c_mine += (((rm + level + (int64_t)step) % d.nranks) == d.rank);
    }
    // wave-aggregated slot allocation: inclusive scan over the 64 lanes, one atomic per wave
    int incl = c_mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(d.alloc_count, total);
    base = __shfl(base, 63, 64);
    int g = base + incl - c_mine;
    if (c_mine > 0) {
        atomicAdd(&L.bins.misc[0], (unsigned int)c_mine);
        const int r = (int)(rm / d.M), m = (int)(rm - (int64_t)r * d.M);
        const int facet = d.res_facet[r];
        const int f0 = d.facet_face_off[facet], nf = d.facet_face_off[facet + 1] - f0;
        const double4 rec = d.modetab[m];
        const double Tres = d.res_T[r];
        const double occ0 = nk_occupation(d, Tres, rec.x);                       // Population.py:506
        for (int level = c; level >= 1; --level) {
            if (d.nranks > 1 && (((rm + level + (int64_t)step) % d.nranks) != d.rank)) continue;
            const int gi = g++;
            int64_t slot = gi < ft0 ? (int64_t)d.free_list[ft0 - 1 - gi] : ns0 + (gi - ft0);
            if (slot >= d.cap) { *d.overflow = 1; continue; }
            const uint64_t pid = ((uint64_t)((step + 1u) & 0xFFFFFFu) << 40) | ((uint64_t)rm << 12) | (uint64_t)level;
            double uf, us, ur, ut;
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT, uf, us);
            nk_uniform2_dev(d.seed, pid, step, NK_TAG_EMIT + 1, ur, ut);
            const double dt_in = (level == 1) ? d.dt * (1.0 - (cnt / prob))              // :391 / :440
                                              : d.dt * (1.0 - ((double)(level - 1) + ut) / prob);  // :394
            int a = nk_ss_right(d.facet_face_cdf + f0, nf, uf);                          // np.random.choice, Mesh.py:937
            a = a > nf - 1 ? nf - 1 : a;
            const double *vx = d.face_verts + 9 * (int64_t)d.facet_face_idx[f0 + a];
            const double sq = sqrt(us);
            const double a0 = 1.0 - sq, a1 = (1.0 - ur) * sq, a2 = ur * sq;              // Mesh.py:945-947
            NkParticle p;
            const double x0 = a0 * vx[0] + a1 * vx[3] + a2 * vx[6];
            const double y0 = a0 * vx[1] + a1 * vx[4] + a2 * vx[7];
            const double z0 = a0 * vx[2] + a1 * vx[5] + a2 * vx[8];
            p.omega = rec.x; p.vx = rec.y; p.vy = rec.z; p.vz = rec.w;
            p.mode = m; p.occ = occ0; p.alive = true;
            double tc; int fcn;
            nk_find_boundary(L.faces, L.face_facet, d.F, d.tol, x0, y0, z0, p.vx, p.vy, p.vz, tc, fcn);
            p.nts = tc / d.dt - dt_in / d.dt;                                            // :535
            p.x = x0 + p.vx * dt_in; p.y = y0 + p.vy * dt_in; p.z = z0 + p.vz * dt_in;   // :536
            p.facet = fcn;
            if (p.nts < 0.0) nk_events(d, L.faces, L.face_facet, L.facets, L.cen, L.Tsv, L.bins, p, pid, step);
            if (p.alive) {
                nk_tally_one(d, L.cen, L.Tsv, L.bins, p, do_flux != 0, rep);
                d.x[slot] = p.x; d.y[slot] = p.y; d.z[slot] = p.z; d.occ[slot] = p.occ; d.nts[slot] = p.nts;
                d.mode[slot] = p.mode; d.facet[slot] = p.facet; d.pid[slot] = pid;
            } else {
                // absorbed inside its entry step: the slot is already allocated, so park it at the tail of free_list;
                // k_update moves parked slots onto the free stack (the stack itself is being popped by this kernel)
                d.mode[slot] = -1;
                d.pid[slot] = pid;
                int q = atomicAdd(d.alloc_count + 1, 1);
                d.free_list[d.cap - 1 - q] = (int32_t)slot;
            }
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
// History row: acc[NB] | T_sv[S] | E_sv[S] | flux_valid, n_slots, free_top, overflow
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
    if (t == 0) {
        int ft = *d.free_top, em = *d.alloc_count, dead = d.alloc_count[1];
        int64_t ns = *d.n_slots;
        if (em >= ft) { ns += em - ft; ft = 0; } else ft -= em;
        if (ns > d.cap) ns = d.cap;
        // slots that died inside the emission kernel were parked at the tail of free_list; move them onto the stack
        for (int q = 0; q < dead; ++q) d.free_list[ft++] = d.free_list[d.cap - 1 - q];
        *d.free_top = ft; *d.n_slots = ns; d.alloc_count[0] = 0; d.alloc_count[1] = 0;
        hist_row[NB + 2 * S + 0] = (double)do_flux;
        hist_row[NB + 2 * S + 1] = (double)ns;
        hist_row[NB + 2 * S + 2] = (double)ft;
        hist_row[NB + 2 * S + 3] = (double)*d.overflow;
    }
}

// Stand-alone lifetime_scattering (flushes the deferred relaxation).
__global__ __launch_bounds__(NK_WG) void k_relax(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        if (mode < 0) continue;
        d.occ[i] = nk_relax(d, L, d.x[i], d.y[i], d.z[i], d.occ[i], d.modetab[mode].x, mode);
    }
}

// timesteps_to_boundary for every particle (Population.py:310-314)
__global__ __launch_bounds__(NK_WG) void k_init_boundaries(NkDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
    const int64_t n = *d.n_slots;
    const int64_t stride = (int64_t)gridDim.x * NK_WG;
    for (int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x; i < n; i += stride) {
        const int mode = d.mode[i];
        if (mode < 0) continue;
        const double4 rec = d.modetab[mode];
        double tc; int fc;
        nk_find_boundary(L.faces, L.face_facet, d.F, d.tol, d.x[i], d.y[i], d.z[i], rec.y, rec.z, rec.w, tc, fc);
        d.nts[i] = tc / d.dt;
        d.facet[i] = fc;
    }
}

// contains_check (Population.py:1712-1722) + Mesh.sample_volume (Mesh.py:890-904)
__global__ __launch_bounds__(NK_WG) void k_contains(NkDev d, uint32_t step) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
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
        const double4 rec = d.modetab[mode];
        double tc; int fc;
        nk_find_boundary(L.faces, L.face_facet, d.F, d.tol, x, y, z, rec.y, rec.z, rec.w, tc, fc);
        d.x[i] = x; d.y[i] = y; d.z[i] = z; d.nts[i] = tc / d.dt; d.facet[i] = fc;
    }
}

// {omega, vx, vy, vz} records for one-gather-per-particle access
__global__ void k_build_modetab(const double *omega, const double *vg, int M, double4 *out) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) out[m] = make_double4(omega[m], vg[3 * m], vg[3 * m + 1], vg[3 * m + 2]);
}

// ---- parity taps: the reference's primitives evaluated on the device
__global__ __launch_bounds__(NK_WG) void k_tap_find_boundary(NkDev d, int64_t n, const double *x, const double *v,
                                                             double *xc, double *tc, int32_t *fc) {
    extern __shared__ __align__(16) unsigned char smem[];
    NkLds L;
    nk_lds_setup(d, smem, L);
    int64_t i = (int64_t)blockIdx.x * NK_WG + threadIdx.x;
    if (i >= n) return;
    double t; int f;
    nk_find_boundary(L.faces, L.face_facet, d.F, d.tol, x[3 * i], x[3 * i + 1], x[3 * i + 2], v[3 * i], v[3 * i + 1],
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
        case 0: out[i] = nk_occupation(d, a[i], d.modetab[mode[i]].x); break;
        case 1: out[i] = nk_lifetime(d, a[i], mode[i]); break;
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
__global__ void k_tap_uniform(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, double *out) {
    double a, b;
    nk_uniform2_dev(seed, pid, step, tag, a, b);
    out[0] = a; out[1] = b;
}

// ============================================================================================ host
struct NkRccl {      // symbols resolved lazily with dlopen: a single-GPU run never loads librccl
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
};

static thread_local std::string g_create_error;

struct nk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    NkDev d;                       // device view (pointers into `allocs`)
    std::vector<void *> allocs;    // everything hipMalloc'ed except the particle arrays
    std::vector<void *> pallocs;   // particle arrays (re-allocated by nk_reserve)
    bool have_material = false, have_mesh = false, have_sv = false, have_params = false;
    int64_t step = 0;
    bool pending_relax = false;
    int grid_step = 0, rows = 0;
    double *acc = nullptr;         // [NB]
    double *hist = nullptr;        // [hist_cap][HROW]
    int hist_cap = 0;
    nk_params params;
    nk_timing timing;
    NkRccl rccl;
    ncclComm_t comm = nullptr;
    int num_cu = 256;
    std::vector<NkFacet> host_facets;   // host mirror of d.facets (patched by nk_set_reservoirs / nk_set_rough)
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

extern "C" {

const char *nk_last_error(const nk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

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
    ctx->params.dt = 1.0; ctx->params.T_ref_local = 1; ctx->params.flux_every = 10; ctx->params.contains_every = 100;
    ctx->d.dt = 1.0; ctx->d.T_ref_local = 1;
    // bookkeeping words
    {
        int rc;
        const int64_t *p64; const int32_t *p32;
        if ((rc = nk_upload<int64_t>(ctx, nullptr, 1, &p64))) { g_create_error = ctx->err; delete ctx; return rc; }
        ctx->d.n_slots = (int64_t *)p64;
        if ((rc = nk_upload<int32_t>(ctx, nullptr, 1, &p32))) { g_create_error = ctx->err; delete ctx; return rc; }
        ctx->d.free_top = (int32_t *)p32;
        if ((rc = nk_upload<int32_t>(ctx, nullptr, 2, &p32))) { g_create_error = ctx->err; delete ctx; return rc; }
        ctx->d.alloc_count = (int32_t *)p32;
        if ((rc = nk_upload<int32_t>(ctx, nullptr, 1, &p32))) { g_create_error = ctx->err; delete ctx; return rc; }
        ctx->d.overflow = (int32_t *)p32;
    }
    *out = ctx;
    return NK_OK;
}

void nk_destroy(nk_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
    for (void *p : ctx->allocs) hipFree(p);
    for (void *p : ctx->pallocs) hipFree(p);
    if (ctx->acc) hipFree(ctx->acc);
    if (ctx->hist) hipFree(ctx->hist);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

int nk_set_material(nk_ctx *ctx, const nk_material *m) {
    NK_ARG(ctx && m, "nk_set_material: NULL argument");
    NK_ARG(m->Q > 0 && m->J > 0 && m->NT >= 2 && m->nE >= 2, "nk_set_material: bad sizes");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.Q = m->Q; d.J = m->J; d.NT = m->NT; d.M = m->Q * m->J;
    const double *om, *vg;
    NK_UP(m->omega, (size_t)d.M, &om);
    NK_UP(m->group_vel, (size_t)d.M * 3, &vg);
    const double4 *mt;
    NK_UP((const double4 *)nullptr, (size_t)d.M, &mt);
    k_build_modetab<<<(d.M + 255) / 256, 256, 0, ctx->stream>>>(om, vg, d.M, (double4 *)mt);
    NK_HIP(hipGetLastError());
    d.modetab = mt;
    NK_UP(m->lifetime, (size_t)d.NT * d.M, &d.tau);
    NK_UP(m->T_grid, (size_t)d.NT, &d.Tgrid);
    d.nE = m->nE;
    NK_UP(m->T_array, (size_t)d.nE, &d.Tarr);
    NK_UP(m->energy_array, (size_t)d.nE, &d.Earr);
    d.Tfill_lo = m->T_fill_lo; d.Tfill_hi = m->T_fill_hi;
    d.hbar = m->hbar; d.kb = m->kb; d.QV = m->QV; d.active_modes = (double)m->active_modes;
    NK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_material = true;
    return NK_OK;
}

int nk_set_mesh(nk_ctx *ctx, const nk_mesh *m) {
    NK_ARG(ctx && m, "nk_set_mesh: NULL argument");
    NK_ARG(m->F > 0 && m->Fc > 0, "nk_set_mesh: empty mesh");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.F = m->F; d.Fc = m->Fc; d.tol = m->tol;
    for (int i = 0; i < 6; ++i) d.bbox[i] = m->bbox[i];
    // face records: n k lo hi o iu iw pad
    std::vector<double> rec((size_t)m->F * NK_FACE_DOUBLES, 0.0);
    for (int f = 0; f < m->F; ++f) {
        double *p = &rec[(size_t)f * NK_FACE_DOUBLES];
        for (int k = 0; k < 3; ++k) {
            p[k] = m->normals[3 * f + k]; p[4 + k] = m->bounds_lo[3 * f + k]; p[7 + k] = m->bounds_hi[3 * f + k];
            p[10 + k] = m->origins[3 * f + k];
        }
        p[3] = m->k[f];
        const double *A = m->basis + 9 * f;      // A[d][b]: columns are (e1, e2, n)
        double a00 = A[0], a01 = A[1], a02 = A[2], a10 = A[3], a11 = A[4], a12 = A[5], a20 = A[6], a21 = A[7], a22 = A[8];
        double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
        NK_ARG(det != 0.0 && isfinite(det), "nk_set_mesh: degenerate face");
        p[13] = (a11 * a22 - a12 * a21) / det; p[14] = (a02 * a21 - a01 * a22) / det; p[15] = (a01 * a12 - a02 * a11) / det;
        p[16] = (a12 * a20 - a10 * a22) / det; p[17] = (a00 * a22 - a02 * a20) / det; p[18] = (a02 * a10 - a00 * a12) / det;
    }
    NK_UP(rec.data(), rec.size(), &d.faces);
    NK_UP(m->face_facet, (size_t)m->F, &d.face_facet);
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
    std::vector<NkFacet> fct((size_t)m->Fc);
    for (int fc = 0; fc < m->Fc; ++fc) {
        NkFacet &q = fct[fc];
        memset(&q, 0, sizeof(q));
        q.cx = m->facet_centroid[3 * fc]; q.cy = m->facet_centroid[3 * fc + 1]; q.cz = m->facet_centroid[3 * fc + 2];
        q.nx = m->facet_normal[3 * fc]; q.ny = m->facet_normal[3 * fc + 1]; q.nz = m->facet_normal[3 * fc + 2];
        q.bc = m->facet_bc[fc]; q.partner = m->facet_partner[fc]; q.res = -1; q.rough = -1;
        NK_ARG(q.bc == 'T' || q.bc == 'F' || q.bc == 'P' || q.bc == 'R', "nk_set_mesh: unknown boundary condition");
        NK_ARG(q.bc != 'P' || (q.partner >= 0 && q.partner < m->Fc), "nk_set_mesh: periodic facet without partner");
    }
    ctx->host_facets = fct;
    NK_UP(fct.data(), fct.size(), &d.facets);
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
    ctx->have_mesh = true;
    return NK_OK;
}

static int nk_patch_facets(nk_ctx *ctx) {
    NK_HIP(hipMemcpy((void *)ctx->d.facets, ctx->host_facets.data(), (size_t)ctx->d.Fc * sizeof(NkFacet), hipMemcpyHostToDevice));
    return NK_OK;
}

static int nk_alloc_tally(nk_ctx *ctx) {
    NkDev &d = ctx->d;
    d.NB = 5 * d.S + 5 * d.R + 1;
    if (ctx->acc) { hipFree(ctx->acc); ctx->acc = nullptr; }
    NK_HIP(hipMalloc((void **)&ctx->acc, (size_t)d.NB * sizeof(double)));
    ctx->grid_step = ctx->num_cu * 8;
    int emit_blocks = (int)(((int64_t)d.R * d.M + NK_WG - 1) / NK_WG);
    ctx->rows = ctx->grid_step + emit_blocks;
    const double *p;
    NK_UP((const double *)nullptr, (size_t)ctx->rows * d.NB, &p);
    d.partials = (double *)p;
    return NK_OK;
}

int nk_set_subvolumes(nk_ctx *ctx, const nk_subvols *s, const double *T_sv_init) {
    NK_ARG(ctx && s && T_sv_init, "nk_set_subvolumes: NULL argument");
    NK_ARG(s->S > 0 && s->S <= 512, "nk_set_subvolumes: S must be in [1, 512]");
    NK_ARG(ctx->have_material, "nk_set_subvolumes: call nk_set_material first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.S = s->S; d.sv_kind = s->kind; d.sv_axis = s->axis; d.sv_interp = s->interp;
    NK_ARG(s->axis >= 0 && s->axis < 3, "nk_set_subvolumes: axis");
    NK_ARG(!(s->kind != 0 && s->interp != 2), "nk_set_subvolumes: slice interpolation needs slice subvolumes");
    NK_UP(s->centers, (size_t)s->S * 3, &d.centers);
    NK_UP(s->volumes, (size_t)s->S, &d.sv_volume);
    const double *t;
    NK_UP(T_sv_init, (size_t)s->S, &t);
    d.T_sv = (double *)t;
    if (s->kind == 0 && s->S > 1) {
        double c0 = s->centers[s->axis], c1 = s->centers[3 + s->axis];
        NK_ARG(c1 > c0, "nk_set_subvolumes: slice centres must ascend along the axis");
        double Lx = c1 - c0;
        d.sv_lo = c0 - 0.5 * Lx; d.sv_invL = 1.0 / Lx;
    } else { d.sv_lo = 0.0; d.sv_invL = 0.0; }
    ctx->have_sv = true;
    return nk_alloc_tally(ctx);
}

int nk_set_reservoirs(nk_ctx *ctx, const nk_reservoirs *r) {
    NK_ARG(ctx && r, "nk_set_reservoirs: NULL argument");
    NK_ARG(ctx->have_material && ctx->have_mesh && ctx->have_sv, "nk_set_reservoirs: set material, mesh, subvolumes first");
    NK_ARG(r->R >= 0 && r->R <= 64, "nk_set_reservoirs: R must be in [0, 64]");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.R = r->R; d.res_gen = r->gen;
    NK_ARG((int64_t)d.R * d.M < (1ll << 28), "nk_set_reservoirs: R*Q*J too large for the particle id layout");
    if (r->R > 0) {
        NK_UP(r->facet, (size_t)r->R, &d.res_facet);
        NK_UP(r->T, (size_t)r->R, &d.res_T);
        NK_UP(r->enter_prob, (size_t)r->R * d.M, &d.enter_prob);
        const double *c;
        NK_UP(r->counter, (size_t)r->R * d.M, &c);
        d.res_counter = (double *)c;
        NkFacet *hf = ctx->host_facets.data();
        for (int i = 0; i < r->R; ++i) {
            NK_ARG(r->facet[i] >= 0 && r->facet[i] < d.Fc, "nk_set_reservoirs: facet index");
            hf[r->facet[i]].res = i;
        }
        int rc = nk_patch_facets(ctx);
        if (rc) return rc;
    }
    return nk_alloc_tally(ctx);
}

int nk_set_rough(nk_ctx *ctx, const nk_rough *r) {
    NK_ARG(ctx && r, "nk_set_rough: NULL argument");
    NK_ARG(ctx->have_material && ctx->have_mesh, "nk_set_rough: set material and mesh first");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    d.Fr = r->Fr;
    if (r->Fr > 0) {
        size_t n = (size_t)r->Fr * d.M;
        NK_UP(r->specularity, n, &d.specularity);
        NK_UP(r->true_spec, n, &d.true_spec);
        NK_UP(r->spec_map, n, &d.spec_map);
        NK_UP(r->roulette, n, &d.roulette);
        if (r->degen_j2) NK_UP(r->degen_j2, (size_t)d.M, &d.degen_j2); else d.degen_j2 = nullptr;
        NkFacet *hf = ctx->host_facets.data();
        for (int i = 0; i < r->Fr; ++i) {
            NK_ARG(r->facet[i] >= 0 && r->facet[i] < d.Fc, "nk_set_rough: facet index");
            hf[r->facet[i]].rough = i;
        }
        return nk_patch_facets(ctx);
    }
    return NK_OK;
}

int nk_set_params(nk_ctx *ctx, const nk_params *p) {
    NK_ARG(ctx && p, "nk_set_params: NULL argument");
    NK_ARG(p->dt > 0, "nk_set_params: dt must be positive");
    ctx->params = *p;
    NkDev &d = ctx->d;
    d.dt = p->dt; d.norm_fixed = p->norm_fixed; d.particle_density = p->particle_density;
    d.T_ref_local = p->T_ref_local; d.T_ref = p->T_ref;
    ctx->have_params = true;
    return NK_OK;
}

static int nk_check_ready(nk_ctx *ctx) {
    NK_ARG(ctx->have_material && ctx->have_mesh && ctx->have_sv && ctx->have_params,
           "engine not configured: need material, mesh, subvolumes and params");
    NkDev &d = ctx->d;
    // every rough / reservoir facet must be backed by its table, otherwise a kernel would index garbage
    const NkFacet *hf = ctx->host_facets.data();
    for (int f = 0; f < d.Fc; ++f) {
        NK_ARG(hf[f].bc != 'R' || hf[f].rough >= 0, "a facet has BC 'R' but nk_set_rough did not cover it");
        NK_ARG(!(hf[f].bc == 'T' || hf[f].bc == 'F') || hf[f].res >= 0, "a facet has BC 'T' but nk_set_reservoirs did not cover it");
    }
    NK_ARG(d.cap > 0, "no particle storage: call nk_reserve / nk_upload_particles");
    NK_ARG(nk_lds_bytes(d.S, d.R, d.F, d.Fc) <= 160 * 1024, "tables do not fit the 160 KiB LDS");
    return NK_OK;
}

int nk_reserve(nk_ctx *ctx, int64_t capacity) {
    NK_ARG(ctx, "nk_reserve: NULL context");
    NK_ARG(capacity > 0 && capacity < (1ll << 31) - 1024, "nk_reserve: capacity out of range");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    if (capacity <= d.cap) return NK_OK;
    NK_HIP(hipStreamSynchronize(ctx->stream));
    int64_t ns = 0;
    int32_t ft = 0;
    if (d.cap > 0) {
        NK_HIP(hipMemcpy(&ns, d.n_slots, 8, hipMemcpyDeviceToHost));
        NK_HIP(hipMemcpy(&ft, d.free_top, 4, hipMemcpyDeviceToHost));
    }
    std::vector<void *> old = ctx->pallocs;
    ctx->pallocs.clear();
    NkDev nd = d;
    const double *pd; const int32_t *pi; const uint64_t *pu;
#define NK_PALLOC(T, field, ptr)                                                                       \
    do { int rc_ = nk_upload<T>(ctx, nullptr, (size_t)capacity, &ptr, true); if (rc_) return rc_;      \
         if (d.cap > 0 && ns > 0) NK_HIP(hipMemcpy((void *)ptr, d.field, (size_t)ns * sizeof(T), hipMemcpyDeviceToDevice)); \
         nd.field = (T *)ptr; } while (0)
    NK_PALLOC(double, x, pd); NK_PALLOC(double, y, pd); NK_PALLOC(double, z, pd);
    NK_PALLOC(double, occ, pd); NK_PALLOC(double, nts, pd);
    NK_PALLOC(int32_t, mode, pi); NK_PALLOC(int32_t, facet, pi);
    NK_PALLOC(uint64_t, pid, pu);
    {
        int rc_ = nk_upload<int32_t>(ctx, nullptr, (size_t)capacity, &pi, true);
        if (rc_) return rc_;
        if (d.cap > 0 && ft > 0) NK_HIP(hipMemcpy((void *)pi, d.free_list, (size_t)ft * 4, hipMemcpyDeviceToDevice));
        nd.free_list = (int32_t *)pi;
    }
#undef NK_PALLOC
    nd.cap = capacity;
    for (void *p : old) hipFree(p);
    d = nd;
    return NK_OK;
}

int nk_upload_particles(nk_ctx *ctx, int64_t N, const double *x, const double *y, const double *z, const int32_t *mode,
                        const double *occ, const double *n_ts, const int32_t *facet, const uint64_t *pid,
                        uint64_t pid_offset) {
    NK_ARG(ctx && N >= 0, "nk_upload_particles: bad arguments");
    NK_ARG(N == 0 || (x && y && z && mode && occ), "nk_upload_particles: x, y, z, mode, occ are required");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    if (N > d.cap) {
        // forget old contents, then grow
        int64_t zero = 0;
        if (d.cap > 0) NK_HIP(hipMemcpy(d.n_slots, &zero, 8, hipMemcpyHostToDevice));
        int rc = nk_reserve(ctx, N + N / 2 + 4096);
        if (rc) return rc;
    }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    size_t n = (size_t)N;
    if (n) {
        NK_HIP(hipMemcpy(d.x, x, n * 8, hipMemcpyHostToDevice));
        NK_HIP(hipMemcpy(d.y, y, n * 8, hipMemcpyHostToDevice));
        NK_HIP(hipMemcpy(d.z, z, n * 8, hipMemcpyHostToDevice));
        NK_HIP(hipMemcpy(d.occ, occ, n * 8, hipMemcpyHostToDevice));
        NK_HIP(hipMemcpy(d.mode, mode, n * 4, hipMemcpyHostToDevice));
        if (n_ts) NK_HIP(hipMemcpy(d.nts, n_ts, n * 8, hipMemcpyHostToDevice));
        if (facet) NK_HIP(hipMemcpy(d.facet, facet, n * 4, hipMemcpyHostToDevice));
        if (pid) NK_HIP(hipMemcpy(d.pid, pid, n * 8, hipMemcpyHostToDevice));
        else {
            std::vector<uint64_t> ids(n);
            for (size_t i = 0; i < n; ++i) ids[i] = pid_offset + i;
            NK_HIP(hipMemcpy(d.pid, ids.data(), n * 8, hipMemcpyHostToDevice));
        }
    }
    int32_t zero32[2] = {0, 0};
    NK_HIP(hipMemcpy(d.n_slots, &N, 8, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(d.free_top, zero32, 4, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(d.alloc_count, zero32, 8, hipMemcpyHostToDevice));
    NK_HIP(hipMemcpy(d.overflow, zero32, 4, hipMemcpyHostToDevice));
    ctx->pending_relax = false;
    return NK_OK;
}

static inline int nk_sweep_grid(const nk_ctx *ctx) { return ctx->num_cu * 8; }

int nk_init_boundaries(nk_ctx *ctx) {
    NK_ARG(ctx, "nk_init_boundaries: NULL context");
    int rc = nk_check_ready(ctx);
    if (rc) return rc;
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    size_t lds = nk_lds_bytes(d.S, d.R, d.F, d.Fc);
    k_init_boundaries<<<nk_sweep_grid(ctx), NK_WG, lds, ctx->stream>>>(d);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    return NK_OK;
}

static int nk_flush_relax(nk_ctx *ctx) {
    if (!ctx->pending_relax) return NK_OK;
    NkDev &d = ctx->d;
    size_t lds = nk_lds_bytes(d.S, d.R, d.F, d.Fc);
    k_relax<<<nk_sweep_grid(ctx), NK_WG, lds, ctx->stream>>>(d);
    NK_HIP(hipGetLastError());
    ctx->pending_relax = false;
    return NK_OK;
}

int nk_step(nk_ctx *ctx, int32_t nsteps, nk_tally *out) {
    NK_ARG(ctx && nsteps > 0, "nk_step: bad arguments");
    int rc = nk_check_ready(ctx);
    if (rc) return rc;
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    const int S = d.S, R = d.R, NB = d.NB;
    const int HROW = NB + 2 * S + 4;
    if (nsteps > ctx->hist_cap) {
        if (ctx->hist) hipFree(ctx->hist);
        ctx->hist = nullptr;
        NK_HIP(hipMalloc((void **)&ctx->hist, (size_t)nsteps * HROW * sizeof(double)));
        ctx->hist_cap = nsteps;
    }
    const size_t lds = nk_lds_bytes(S, R, d.F, d.Fc);
    const int emit_blocks = (int)(((int64_t)R * d.M + NK_WG - 1) / NK_WG);
    const int nev = nsteps < 64 ? nsteps : 64;          // per-kernel timing on (up to) the first 64 steps
    std::vector<hipEvent_t> ev((size_t)nev * 3);
    for (auto &e : ev) NK_HIP(hipEventCreate(&e));
    hipEvent_t t0, t1;
    NK_HIP(hipEventCreate(&t0));
    NK_HIP(hipEventCreate(&t1));
    NK_HIP(hipEventRecord(t0, ctx->stream));
    for (int s = 0; s < nsteps; ++s) {
        const uint32_t step = (uint32_t)ctx->step;
        if (ctx->params.contains_every > 0 && (ctx->step % ctx->params.contains_every) == 0 && d.nS > 0) {
            if ((rc = nk_flush_relax(ctx))) return rc;
            k_contains<<<nk_sweep_grid(ctx), NK_WG, lds, ctx->stream>>>(d, step);
        }
        const int fe = ctx->params.flux_every;
        const int do_flux = (fe > 0 && ((ctx->step + 1) % fe) == 0) ? 1 : 0;
        if (s < nev) NK_HIP(hipEventRecord(ev[3 * s], ctx->stream));
        k_step<<<ctx->grid_step, NK_WG, lds, ctx->stream>>>(d, step, ctx->pending_relax ? 1 : 0, do_flux);
        if (s < nev) NK_HIP(hipEventRecord(ev[3 * s + 1], ctx->stream));
        if (R > 0) k_emit<<<emit_blocks, NK_WG, lds, ctx->stream>>>(d, step, do_flux, ctx->grid_step);
        if (s < nev) NK_HIP(hipEventRecord(ev[3 * s + 2], ctx->stream));
        k_reduce<<<NB, NK_WG, 0, ctx->stream>>>(d.partials, ctx->grid_step + (R > 0 ? emit_blocks : 0), NB, ctx->acc);
        if (ctx->comm) {
            ncclResult_t nrc = ctx->rccl.AllReduce(ctx->acc, ctx->acc, (size_t)NB, ncclDouble, ncclSum, ctx->comm, ctx->stream);
            if (nrc != ncclSuccess) { ctx->err = "ncclAllReduce failed"; return NK_ERR_COMM; }
        }
        k_update<<<1, 512, 0, ctx->stream>>>(d, ctx->acc, ctx->hist + (size_t)s * HROW, do_flux);
        ctx->pending_relax = true;
        ctx->step += 1;
    }
    NK_HIP(hipEventRecord(t1, ctx->stream));
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    double sk = 0.0, ek = 0.0;
    for (int s = 0; s < nev; ++s) {
        NK_HIP(hipEventElapsedTime(&ms, ev[3 * s], ev[3 * s + 1])); sk += ms;
        NK_HIP(hipEventElapsedTime(&ms, ev[3 * s + 1], ev[3 * s + 2])); ek += ms;
    }
    NK_HIP(hipEventElapsedTime(&ms, t0, t1));
    for (auto &e : ev) hipEventDestroy(e);
    hipEventDestroy(t0); hipEventDestroy(t1);
    ctx->timing.step_kernel_ms = sk / nev;
    ctx->timing.emit_kernel_ms = ek / nev;
    ctx->timing.total_ms = ms;
    std::vector<double> h((size_t)nsteps * HROW);
    NK_HIP(hipMemcpy(h.data(), ctx->hist, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    bool overflow = false;
    for (int s = 0; s < nsteps; ++s) {
        const double *row = &h[(size_t)s * HROW];
        if (row[NB + 2 * S + 3] != 0.0) overflow = true;
        if (!out) continue;
        if (out->E_raw) memcpy(out->E_raw + (size_t)s * S, row, S * 8);
        if (out->N_sv) memcpy(out->N_sv + (size_t)s * S, row + S, S * 8);
        if (out->flux_raw) {
            if (row[NB + 2 * S] != 0.0) memcpy(out->flux_raw + (size_t)s * 3 * S, row + 2 * S, 3 * S * 8);
            else for (int k = 0; k < 3 * S; ++k) out->flux_raw[(size_t)s * 3 * S + k] = NAN;
        }
        if (out->N_leaving && R) memcpy(out->N_leaving + (size_t)s * R, row + 5 * S, R * 8);
        if (out->res_energy && R) memcpy(out->res_energy + (size_t)s * R, row + 5 * S + R, R * 8);
        if (out->res_flux && R) memcpy(out->res_flux + (size_t)s * 3 * R, row + 5 * S + 2 * R, 3 * R * 8);
        if (out->N_emitted) out->N_emitted[s] = row[NB - 1];
        if (out->T_sv) memcpy(out->T_sv + (size_t)s * S, row + NB, S * 8);
        if (out->E_sv) memcpy(out->E_sv + (size_t)s * S, row + NB + S, S * 8);
    }
    const double *last = &h[(size_t)(nsteps - 1) * HROW];
    ctx->timing.slots = (int64_t)last[NB + 2 * S + 1];
    double live = 0.0;
    for (int k = 0; k < S; ++k) live += last[S + k];
    ctx->timing.live = (int64_t)live;
    if (overflow) {
        ctx->err = "particle capacity exceeded during nk_step: particles were dropped; call nk_reserve with a larger capacity";
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
    if (ctx->have_material && ctx->have_sv && ctx->have_mesh) { int rc = nk_flush_relax(ctx); if (rc) return rc; }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    int64_t ns = 0;
    NK_HIP(hipMemcpy(&ns, d.n_slots, 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> hm((size_t)ns);
    if (ns) NK_HIP(hipMemcpy(hm.data(), d.mode, (size_t)ns * 4, hipMemcpyDeviceToHost));
    int64_t live = 0;
    for (int64_t i = 0; i < ns; ++i) live += hm[i] >= 0;
    *N_out = live;
    if (capacity == 0) return NK_OK;
    NK_ARG(capacity >= live, "nk_download_particles: capacity smaller than the live particle count");
    std::vector<double> buf((size_t)ns);
    auto pack_d = [&](const double *src, double *dst) -> int {
        if (!dst || !ns) return NK_OK;
        NK_HIP(hipMemcpy(buf.data(), src, (size_t)ns * 8, hipMemcpyDeviceToHost));
        int64_t w = 0;
        for (int64_t i = 0; i < ns; ++i) if (hm[i] >= 0) dst[w++] = buf[i];
        return NK_OK;
    };
    int rc;
    if ((rc = pack_d(d.x, x)) || (rc = pack_d(d.y, y)) || (rc = pack_d(d.z, z)) || (rc = pack_d(d.occ, occ)) ||
        (rc = pack_d(d.nts, n_ts)))
        return rc;
    if (facet && ns) {
        std::vector<int32_t> b((size_t)ns);
        NK_HIP(hipMemcpy(b.data(), d.facet, (size_t)ns * 4, hipMemcpyDeviceToHost));
        int64_t w = 0;
        for (int64_t i = 0; i < ns; ++i) if (hm[i] >= 0) facet[w++] = b[i];
    }
    if (pid && ns) {
        std::vector<uint64_t> b((size_t)ns);
        NK_HIP(hipMemcpy(b.data(), d.pid, (size_t)ns * 8, hipMemcpyDeviceToHost));
        int64_t w = 0;
        for (int64_t i = 0; i < ns; ++i) if (hm[i] >= 0) pid[w++] = b[i];
    }
    if (mode) { int64_t w = 0; for (int64_t i = 0; i < ns; ++i) if (hm[i] >= 0) mode[w++] = hm[i]; }
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
    NK_HIP(hipMemcpy(ctx->d.T_sv, T_sv, (size_t)ctx->d.S * 8, hipMemcpyHostToDevice));
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
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { err = "librccl lacks expected symbols"; return NK_ERR_COMM; }
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
    NK_ARG(ctx && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "nk_comm_init: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    ctx->d.rank = rank;
    ctx->d.nranks = nranks;
    if (nranks == 1) return NK_OK;
    if (nk_load_rccl(ctx->rccl, ctx->err)) return NK_ERR_COMM;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    memcpy(&id, id128, 128);
    if (ctx->rccl.CommInitRank(&ctx->comm, nranks, id, rank) != ncclSuccess) { ctx->err = "ncclCommInitRank failed"; ctx->comm = nullptr; return NK_ERR_COMM; }
    return NK_OK;
}

// ------------------------------------------------------------------------------------- parity taps
#define NK_DEV_IN(T, name, src, count)                                                                 \
    T *name = nullptr;                                                                                 \
    NK_HIP(hipMalloc((void **)&name, (size_t)(count) * sizeof(T)));                                    \
    if (src) NK_HIP(hipMemcpy(name, src, (size_t)(count) * sizeof(T), hipMemcpyHostToDevice));
#define NK_DEV_OUT(T, name, dst, count)                                                                \
    if (dst) NK_HIP(hipMemcpy(dst, name, (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost));         \
    hipFree(name);

int nk_find_boundary(nk_ctx *ctx, int64_t n, const double *x, const double *v, double *xc, double *tc, int32_t *fc) {
    NK_ARG(ctx && ctx->have_mesh && ctx->have_sv && n > 0 && x && v, "nk_find_boundary: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NkDev &d = ctx->d;
    NK_DEV_IN(double, dx, x, n * 3); NK_DEV_IN(double, dv, v, n * 3);
    NK_DEV_IN(double, dxc, (double *)nullptr, n * 3); NK_DEV_IN(double, dtc, (double *)nullptr, n);
    NK_DEV_IN(int32_t, dfc, (int32_t *)nullptr, n);
    k_tap_find_boundary<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, nk_lds_bytes(d.S, d.R, d.F, d.Fc), ctx->stream>>>(d, n, dx, dv, dxc, dtc, dfc);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_DEV_OUT(double, dxc, xc, n * 3); NK_DEV_OUT(double, dtc, tc, n); NK_DEV_OUT(int32_t, dfc, fc, n);
    hipFree(dx); hipFree(dv);
    return NK_OK;
}
int nk_classify(nk_ctx *ctx, int64_t n, const double *x, int32_t *id) {
    NK_ARG(ctx && ctx->have_sv && n > 0 && x && id, "nk_classify: bad arguments");
    NK_HIP(hipSetDevice(ctx->device));
    NK_DEV_IN(double, dx, x, n * 3); NK_DEV_IN(int32_t, did, (int32_t *)nullptr, n);
    k_tap_classify<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(ctx->d, n, dx, did);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_DEV_OUT(int32_t, did, id, n);
    hipFree(dx);
    return NK_OK;
}
int nk_eval(nk_ctx *ctx, int32_t what, int64_t n, const double *a, const int32_t *mode, double *out) {
    NK_ARG(ctx && ctx->have_material && n > 0 && a && out && what >= 0 && what <= 4, "nk_eval: bad arguments");
    NK_ARG(what > 1 || mode, "nk_eval: mode required");
    NK_ARG(what != 4 || ctx->have_sv, "nk_eval: subvolumes required");
    NK_HIP(hipSetDevice(ctx->device));
    const int64_t na = what == 4 ? 3 * n : n;
    NK_DEV_IN(double, da, a, na); NK_DEV_IN(int32_t, dm, mode, n); NK_DEV_IN(double, dout, (double *)nullptr, n);
    k_tap_eval<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(ctx->d, what, n, da, dm, dout);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_DEV_OUT(double, dout, out, n);
    hipFree(da); hipFree(dm);
    return NK_OK;
}
int nk_reflect(nk_ctx *ctx, int64_t n, const int32_t *facet, const int32_t *mode_in, const double *col_pos,
               const double *n_in, const double *omega_in, const double *r_spec, const double *r_deg,
               const double *r_diff, int32_t *mode_out, double *n_out, double *omega_out) {
    NK_ARG(ctx && ctx->have_material && ctx->have_sv && ctx->have_mesh && ctx->d.Fr > 0 && n > 0, "nk_reflect: engine not configured");
    NK_ARG(facet && mode_in && col_pos && n_in && omega_in && r_spec && r_diff, "nk_reflect: NULL input");
    const NkFacet *hf = ctx->host_facets.data();
    for (int64_t i = 0; i < n; ++i)
        NK_ARG(facet[i] >= 0 && facet[i] < ctx->d.Fc && hf[facet[i]].rough >= 0, "nk_reflect: facet is not rough");
    NK_HIP(hipSetDevice(ctx->device));
    NK_DEV_IN(int32_t, df, facet, n); NK_DEV_IN(int32_t, dm, mode_in, n); NK_DEV_IN(double, dc, col_pos, n * 3);
    NK_DEV_IN(double, dn, n_in, n); NK_DEV_IN(double, dom, omega_in, n); NK_DEV_IN(double, drs, r_spec, n);
    double *drd = nullptr;
    if (r_deg) { NK_HIP(hipMalloc((void **)&drd, (size_t)n * 8)); NK_HIP(hipMemcpy(drd, r_deg, (size_t)n * 8, hipMemcpyHostToDevice)); }
    NK_DEV_IN(double, drf, r_diff, n);
    NK_DEV_IN(int32_t, dmo, (int32_t *)nullptr, n); NK_DEV_IN(double, dno, (double *)nullptr, n); NK_DEV_IN(double, doo, (double *)nullptr, n);
    k_tap_reflect<<<(int)((n + NK_WG - 1) / NK_WG), NK_WG, 0, ctx->stream>>>(ctx->d, n, df, dm, dc, dn, dom, drs, drd, drf, dmo, dno, doo);
    NK_HIP(hipGetLastError());
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_DEV_OUT(int32_t, dmo, mode_out, n); NK_DEV_OUT(double, dno, n_out, n); NK_DEV_OUT(double, doo, omega_out, n);
    hipFree(df); hipFree(dm); hipFree(dc); hipFree(dn); hipFree(dom); hipFree(drs); hipFree(drf);
    if (drd) hipFree(drd);
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
