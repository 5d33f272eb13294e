// Developer probe: does the particle store's layout limit the sweep's streaming rate?  3072 waves, one segment each, read
// (and optionally write back) 56 B per particle from (A) eight separate arrays, as the engine does, or (B) one array of
// 64-particle blocks (x[64] y[64] z[64] occ[64] nts[64] pid[64] mode[64] facet[64] = 3584 B).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_probe scripts/probes/layout_probe.hip && /tmp/layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Soa { double *x, *y, *z, *occ, *nts; uint64_t *pid; int *mode, *facet; };

template <bool WRITE>
__global__ __launch_bounds__(256, 3) void k_soa(Soa a, Soa o, int nseg, int segcap, double *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    double acc = 0.0;
    for (int seg = blockIdx.x * 4 + wave; seg < nseg; seg += nw) {
        const int64_t base = (int64_t)seg * segcap;
        int64_t i = base + lane;
        double xN = a.x[i], yN = a.y[i], zN = a.z[i], oN = a.occ[i], nN = a.nts[i];
        uint64_t pN = a.pid[i]; int mN = a.mode[i], fN = a.facet[i];
        for (int t = 0; t < segcap; t += 64) {
            const double x = xN, y = yN, z = zN, oc = oN, n = nN; const uint64_t p = pN; const int m = mN, f = fN;
            if (t + 64 < segcap) {
                i = base + t + 64 + lane;
                xN = a.x[i]; yN = a.y[i]; zN = a.z[i]; oN = a.occ[i]; nN = a.nts[i]; pN = a.pid[i]; mN = a.mode[i]; fN = a.facet[i];
            }
            if (WRITE) {
                const int64_t j = base + t + lane;
                o.x[j] = x + 1.0; o.y[j] = y; o.z[j] = z; o.occ[j] = oc; o.nts[j] = n - 1.0; o.pid[j] = p; o.mode[j] = m; o.facet[j] = f;
            } else acc += x + y + z + oc + n + (double)(p & 3) + m + f;
        }
    }
    if (!WRITE && acc == 1.2345e-300) sink[0] = acc;
}

template <bool WRITE>
__global__ __launch_bounds__(256, 3) void k_blk(const char *a, char *o, int nseg, int segcap, double *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    double acc = 0.0;
    for (int seg = blockIdx.x * 4 + wave; seg < nseg; seg += nw) {
        const int64_t b0 = (int64_t)seg * (segcap / 64);
        const char *B = a + b0 * 3584;
        const double *D = (const double *)B;
        double xN = D[lane], yN = D[64 + lane], zN = D[128 + lane], oN = D[192 + lane], nN = D[256 + lane];
        uint64_t pN = ((const uint64_t *)B)[320 + lane]; int mN = ((const int *)B)[768 + lane], fN = ((const int *)B)[832 + lane];
        for (int t = 0; t < segcap / 64; ++t) {
            const double x = xN, y = yN, z = zN, oc = oN, n = nN; const uint64_t p = pN; const int m = mN, f = fN;
            if (t + 1 < segcap / 64) {
                B = a + (b0 + t + 1) * 3584; D = (const double *)B;
                xN = D[lane]; yN = D[64 + lane]; zN = D[128 + lane]; oN = D[192 + lane]; nN = D[256 + lane];
                pN = ((const uint64_t *)B)[320 + lane]; mN = ((const int *)B)[768 + lane]; fN = ((const int *)B)[832 + lane];
            }
            if (WRITE) {
                char *O = o + (b0 + t) * 3584; double *E = (double *)O;
                E[lane] = x + 1.0; E[64 + lane] = y; E[128 + lane] = z; E[192 + lane] = oc; E[256 + lane] = n - 1.0;
                ((uint64_t *)O)[320 + lane] = p; ((int *)O)[768 + lane] = m; ((int *)O)[832 + lane] = f;
            } else acc += x + y + z + oc + n + (double)(p & 3) + m + f;
        }
    }
    if (!WRITE && acc == 1.2345e-300) sink[0] = acc;
}

int main() {
    const int nseg = 3072, segcap = 3264;                   // 1.0027e7 particles
    const int64_t N = (int64_t)nseg * segcap;
    Soa a, o;
    char *ba, *bo; double *sink;
    CK(hipMalloc(&a.x, N * 8)); CK(hipMalloc(&a.y, N * 8)); CK(hipMalloc(&a.z, N * 8)); CK(hipMalloc(&a.occ, N * 8)); CK(hipMalloc(&a.nts, N * 8));
    CK(hipMalloc(&a.pid, N * 8)); CK(hipMalloc(&a.mode, N * 4)); CK(hipMalloc(&a.facet, N * 4));
    CK(hipMalloc(&o.x, N * 8)); CK(hipMalloc(&o.y, N * 8)); CK(hipMalloc(&o.z, N * 8)); CK(hipMalloc(&o.occ, N * 8)); CK(hipMalloc(&o.nts, N * 8));
    CK(hipMalloc(&o.pid, N * 8)); CK(hipMalloc(&o.mode, N * 4)); CK(hipMalloc(&o.facet, N * 4));
    CK(hipMalloc(&ba, N * 56)); CK(hipMalloc(&bo, N * 56)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a.x, 0, N * 8)); CK(hipMemset(a.y, 0, N * 8)); CK(hipMemset(a.z, 0, N * 8)); CK(hipMemset(a.occ, 0, N * 8)); CK(hipMemset(a.nts, 0, N * 8));
    CK(hipMemset(a.pid, 0, N * 8)); CK(hipMemset(a.mode, 0, N * 4)); CK(hipMemset(a.facet, 0, N * 4)); CK(hipMemset(ba, 0, N * 56));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 768, reps = 20;
    for (int variant = 0; variant < 4; ++variant) {
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(e0));
            if (variant == 0) k_soa<false><<<grid, 256>>>(a, o, nseg, segcap, sink);
            if (variant == 1) k_blk<false><<<grid, 256>>>(ba, bo, nseg, segcap, sink);
            if (variant == 2) k_soa<true><<<grid, 256>>>(a, o, nseg, segcap, sink);
            if (variant == 3) k_blk<true><<<grid, 256>>>(ba, bo, nseg, segcap, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        const double bytes = (double)N * 56 * (variant >= 2 ? 2 : 1);
        const char *names[4] = {"8 arrays, read", "blocks, read", "8 arrays, read + write", "blocks, read + write"};
        printf("%-26s avg %.1f us  best %.1f us  -> %.2f TB/s (avg)\n", names[variant], 1e3 * sum / reps, 1e3 * best, bytes / (sum / reps * 1e-3) / 1e12);
    }
    return 0;
}
