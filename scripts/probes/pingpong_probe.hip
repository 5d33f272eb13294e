// Developer probe (round 4): does the memory-side cache carry a launch's last-written data into the next launch?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/pingpong_probe scripts/probes/pingpong_probe.hip ; gpurun -- ./scripts/probes/pingpong_probe
// The sweep's access pattern (3072 waves, each copying its own contiguous segment in place, block by block) over a buffer of the
// box store's size, launch after launch, (a) always front to back, (b) alternating: every other launch walks its segments back to
// front, so that a launch begins with the blocks the previous launch wrote last.  If what a launch leaves in the L2s / the 256 MB
// memory-side cache survives into the next launch, (b) reads much of its data from there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define BD 288
__global__ __launch_bounds__(256, 3) void k_copy(double *b, int nseg, int blocks, int backwards) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * BD;
    double v[4], n[4]; unsigned wv = 0, wn = 0;
    auto at = [&](int r) { return p + (size_t)(backwards ? blocks - 1 - r : r) * BD; };
    { const double *q = at(0); for (int k = 0; k < 4; ++k) n[k] = q[k * 64 + lane]; wn = ((const unsigned *)(q + 256))[lane]; }
    for (int r = 0; r < blocks; ++r) {
        for (int k = 0; k < 4; ++k) v[k] = n[k];
        wv = wn;
        if (r + 1 < blocks) { const double *q = at(r + 1); for (int k = 0; k < 4; ++k) n[k] = q[k * 64 + lane]; wn = ((const unsigned *)(q + 256))[lane]; }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(wv));
        double *q = at(r);
        for (int k = 0; k < 4; ++k) q[k * 64 + lane] = v[k];
        ((unsigned *)(q + 256))[lane] = wv;
    }
}
int main() {
    CK(hipSetDevice(0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nseg = 3072;
    for (size_t mb : {90, 180, 270, 360, 540, 720, 1440}) {
        const size_t bytes = mb << 20;
        const int blocks = (int)(bytes / (BD * 8) / nseg);
        void *p = nullptr;
        CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 0, bytes));
        for (int mode = 0; mode < 2; ++mode) {
            for (int k = 0; k < 4; ++k) k_copy<<<nseg / 4, 256>>>((double *)p, nseg, blocks, mode ? (k & 1) : 0);
            CK(hipEventRecord(e0));
            const int L = 20;
            for (int k = 0; k < L; ++k) k_copy<<<nseg / 4, 256>>>((double *)p, nseg, blocks, mode ? (k & 1) : 0);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5zu MB  %-28s %8.1f us per launch   %.2f TB/s (read + written)\n", mb, mode ? "alternating direction" : "always front to back", 1e3 * ms / L,
                   2.0 * nseg * (double)blocks * BD * 8 / (ms / L * 1e-3) / 1e12);
        }
        CK(hipFree(p));
    }
    return 0;
}
