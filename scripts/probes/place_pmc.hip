// Developer probe (round 4): WHY does the same streaming kernel run at different speeds depending on the allocation it works on?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/place_pmc scripts/probes/place_pmc.hip
//   rocprofv3 --kernel-trace --pmc <counters> -d out -o p --output-format csv -- ./scripts/probes/place_pmc [MB] [count]
// Allocates `count` buffers of `MB` megabytes (all held at once), times the sweep's access pattern on each (3072 waves, each
// copying its own contiguous 1/3072 of the buffer in place, block by block, one block prefetched), then runs the SAME kernel
// under two names -- k_copy<0> on the slowest buffer, k_copy<1> on the fastest, k_copy<2> on the one in the middle -- five
// times each, so that a counter pass can tell them apart.  Prints every buffer's rate and the physical-contiguity proxy the
// driver exposes (allocation time).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define BD 288      // doubles per block of 64 particles (box store: x y z occ + 64 packed words)
template <int TAG>
__global__ __launch_bounds__(256, 3) void k_copy(double *b, int nseg, int blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * BD;
    double v[4], n[4]; unsigned wv = 0, wn = 0;
    for (int k = 0; k < 4; ++k) n[k] = p[k * 64 + lane];
    wn = ((unsigned *)(p + 256))[lane];
    for (int r = 0; r < blocks; ++r) {
        for (int k = 0; k < 4; ++k) v[k] = n[k];
        wv = wn;
        if (r + 1 < blocks) { const double *q = p + (size_t)(r + 1) * BD; for (int k = 0; k < 4; ++k) n[k] = q[k * 64 + lane]; wn = ((const unsigned *)(q + 256))[lane]; }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(wv));
        double *q = p + (size_t)r * BD;
        for (int k = 0; k < 4; ++k) q[k * 64 + lane] = v[k];
        ((unsigned *)(q + 256))[lane] = wv;
    }
}
int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 520;
    const int count = argc > 2 ? atoi(argv[2]) : 48;
    const size_t bytes = mb << 20;
    const int nseg = 3072, blocks = (int)(bytes / (BD * 8) / nseg);
    CK(hipSetDevice(0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<void *> buf;
    std::vector<double> rate, ams;
    for (int c = 0; c < count; ++c) {
        void *p = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        CK(hipMemset(p, 0, bytes));
        CK(hipDeviceSynchronize());
        ams.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        buf.push_back(p);
        k_copy<3><<<nseg / 4, 256>>>((double *)p, nseg, blocks);
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) k_copy<3><<<nseg / 4, 256>>>((double *)p, nseg, blocks);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        rate.push_back(2.0 * nseg * blocks * BD * 8.0 / (ms / 3.0 * 1e-3) / 1e12);
    }
    std::vector<int> order(buf.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return rate[a] < rate[b]; });
    for (size_t i = 0; i < buf.size(); ++i)
        printf("buffer %2zu  va %p  (offset in 2 MB: %7zu B, in 1 GB: %4zu MB)  alloc+clear %7.1f ms   %.3f TB/s\n", i, buf[i], (size_t)buf[i] & ((1u << 21) - 1),
               ((size_t)buf[i] & ((1ull << 30) - 1)) >> 20, ams[i], rate[i]);
    const int slow = order.front(), fast = order.back(), mid = order[order.size() / 2];
    printf("slowest: buffer %d %.3f TB/s   middle: buffer %d %.3f TB/s   fastest: buffer %d %.3f TB/s\n", slow, rate[slow], mid, rate[mid], fast, rate[fast]);
    for (int k = 0; k < 5; ++k) {
        k_copy<0><<<nseg / 4, 256>>>((double *)buf[slow], nseg, blocks);
        k_copy<1><<<nseg / 4, 256>>>((double *)buf[fast], nseg, blocks);
        k_copy<2><<<nseg / 4, 256>>>((double *)buf[mid], nseg, blocks);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
