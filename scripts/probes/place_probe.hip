// Developer probe: why does the same streaming kernel run at two speeds depending on the allocation it works on?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/place_probe scripts/probes/place_probe.hip
//   gpurun -- './scripts/probes/place_probe [MB] [count]'
// Allocates `count` buffers of `MB` megabytes with hipMalloc (all held at once) and as many through the virtual-memory API at
// a 1 GB-aligned address (hipMemAddressReserve / hipMemCreate / hipMemMap), and times on each (a) the sweep's access pattern:
// 3072 waves, each copying its own contiguous 1/3072 of the buffer in place in 2816-byte pieces with one piece prefetched;
// (b) a plain grid-stride copy in place.  Prints address, allocation time and both rates per buffer.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave per segment: `blocks` pieces of 352 doubles (2816 B); lane l copies doubles l, l + 64, ... l + 320 of a piece (+ 32 words)
__global__ __launch_bounds__(256, 3) void k_streams(double *b, int nseg, int blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * 352;
    double v[5], n[5]; unsigned wv = 0, wn = 0;
    for (int k = 0; k < 5; ++k) n[k] = p[k * 64 + lane];
    wn = ((unsigned *)(p + 320))[lane];
    for (int r = 0; r < blocks; ++r) {
        for (int k = 0; k < 5; ++k) v[k] = n[k];
        wv = wn;
        if (r + 1 < blocks) { const double *q = p + (size_t)(r + 1) * 352; for (int k = 0; k < 5; ++k) n[k] = q[k * 64 + lane]; wn = ((const unsigned *)(q + 320))[lane]; }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(wv));
        double *q = p + (size_t)r * 352;
        for (int k = 0; k < 5; ++k) q[k * 64 + lane] = v[k];
        ((unsigned *)(q + 320))[lane] = wv;
    }
}
__global__ __launch_bounds__(256) void k_linear(double2 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 v = b[i];
        asm volatile("" : "+v"(v.x), "+v"(v.y));
        b[i] = v;
    }
}

static double time_kernel(bool streams, void *buf, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
    const int nseg = 3072, blocks = (int)(bytes / 2816 / nseg);
    auto go = [&]() {
        if (streams) k_streams<<<nseg / 4, 256>>>((double *)buf, nseg, blocks);
        else k_linear<<<256 * 8, 256>>>((double2 *)buf, (size_t)nseg * blocks * 2816 / 16);
    };
    go();
    CK(hipEventRecord(e0));
    for (int k = 0; k < 3; ++k) go();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * nseg * blocks * 2816.0 / (ms / 3.0 * 1e-3) / 1e12;      // TB/s, read + written
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 833;
    const int count = argc > 2 ? atoi(argv[2]) : 8;
    const size_t bytes = mb << 20;
    CK(hipSetDevice(0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GB of %.1f GB; buffers of %zu MB\n", fr / 1e9, tot / 1e9, mb);
    std::vector<void *> held;
    for (int c = 0; c < count; ++c) {
        void *p = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 0, bytes));
        const double ams = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        held.push_back(p);
        const double a = time_kernel(true, p, bytes, e0, e1), b = time_kernel(false, p, bytes, e0, e1);
        printf("hipMalloc %2d  va %p (offset in 1 GB: %4zu MB)  alloc+clear %7.1f ms   streams %.2f TB/s   linear %.2f TB/s\n", c, p,
               ((size_t)p & ((1ull << 30) - 1)) >> 20, ams, a, b);
        fflush(stdout);
    }
    // virtual-memory API: 1 GB-aligned address, one physical allocation
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("no virtual-memory API\n"); return 0; }
    printf("virtual-memory API: recommended granularity %zu KB\n", gran >> 10);
    const size_t vbytes = (bytes + gran - 1) / gran * gran;
    for (int c = 0; c < count; ++c) {
        void *va = nullptr;
        hipMemGenericAllocationHandle_t h;
        auto t0 = std::chrono::steady_clock::now();
        if (hipMemAddressReserve(&va, vbytes, (size_t)1 << 30, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); break; }
        if (hipMemCreate(&h, vbytes, &prop, 0) != hipSuccess) { printf("create failed\n"); break; }
        if (hipMemMap(va, vbytes, 0, h, 0) != hipSuccess) { printf("map failed\n"); break; }
        hipMemAccessDesc ad = {};
        ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, vbytes, &ad, 1) != hipSuccess) { printf("set access failed\n"); break; }
        CK(hipMemset(va, 0, bytes));
        const double ams = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const double a = time_kernel(true, va, bytes, e0, e1), b = time_kernel(false, va, bytes, e0, e1);
        printf("vmm       %2d  va %p (offset in 1 GB: %4zu MB)  alloc+clear %7.1f ms   streams %.2f TB/s   linear %.2f TB/s\n", c, va,
               ((size_t)va & ((1ull << 30) - 1)) >> 20, ams, a, b);
        fflush(stdout);
    }
    return 0;
}
