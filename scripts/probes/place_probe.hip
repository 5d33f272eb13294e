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
// the same with TWO pieces on their way per wave: two register sets that alternate (no copies between them: a copy of a register
// that a load is still to fill waits for that load), the stores of a piece issued at the top of the next turn, BEFORE that
// turn's loads -- the wave's memory counter retires in order, so "all but the newest 6" then means: everything but the loads
// just issued
struct Piece { double v[5]; unsigned w; };
__device__ __forceinline__ void piece_load(Piece &P, const double *q, int lane) {
    for (int k = 0; k < 5; ++k) P.v[k] = q[k * 64 + lane];
    P.w = ((const unsigned *)(q + 320))[lane];
}
__device__ __forceinline__ void piece_store(const Piece &P, double *q, int lane) {
    for (int k = 0; k < 5; ++k) q[k * 64 + lane] = P.v[k];
    ((unsigned *)(q + 320))[lane] = P.w;
}
__global__ __launch_bounds__(256, 3) void k_streams2(double *b, int nseg, int blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * 352;
    Piece A, B, S;                       // S: the piece whose stores are due
    bool have = false;
    int sr = 0;
    piece_load(A, p, lane);
    piece_load(B, p + 352, lane);
    auto turn = [&](Piece &T, int r) {
        Piece C = T;                                             // arrived: out of the buffer
        asm volatile("" : "+v"(C.v[0]), "+v"(C.v[1]), "+v"(C.v[2]), "+v"(C.v[3]), "+v"(C.v[4]), "+v"(C.w));
        if (have) piece_store(S, p + (size_t)sr * 352, lane);    // the previous piece leaves
        const int rn = r + 2 < blocks ? r + 2 : blocks - 1;      // always a load (clamped): the count of loads per turn is fixed
        piece_load(T, p + (size_t)rn * 352, lane);
        S = C; sr = r; have = true;
    };
    for (int r = 0; r < blocks; r += 2) {
        turn(A, r);
        if (r + 1 < blocks) turn(B, r + 1);
    }
    if (have) piece_store(S, p + (size_t)sr * 352, lane);
}
// ... and with the loads as inline assembly and the wait written by hand (the compiler drains the counter at the loop header:
// s_waitcnt vmcnt(0) -- with its own bookkeeping the second piece is never really in flight): "vmcnt(6)" = all but the six
// loads issued last, which are the other buffer's
__device__ __forceinline__ void piece_load_asm(Piece &P, const double *q, int lane) {
    const double *a = q + lane;
    const unsigned *w = (const unsigned *)(q + 320) + lane;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(P.v[0]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off offset:512" : "=v"(P.v[1]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off offset:1024" : "=v"(P.v[2]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off offset:1536" : "=v"(P.v[3]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off offset:2048" : "=v"(P.v[4]) : "v"(a) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(P.w) : "v"(w) : "memory");
}
__global__ __launch_bounds__(256, 3) void k_streams3(double *b, int nseg, int blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * 352;
    Piece A, B, S;
    bool have = false;
    int sr = 0;
    piece_load_asm(A, p, lane);
    piece_load_asm(B, p + 352, lane);
    auto turn = [&](Piece &T, int r) {
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(T.v[0]), "+v"(T.v[1]), "+v"(T.v[2]), "+v"(T.v[3]), "+v"(T.v[4]), "+v"(T.w) : : "memory");
        Piece C = T;
        if (have) piece_store(S, p + (size_t)sr * 352, lane);
        const int rn = r + 2 < blocks ? r + 2 : blocks - 1;
        piece_load_asm(T, p + (size_t)rn * 352, lane);
        S = C; sr = r; have = true;
    };
    for (int r = 0; r < blocks; r += 2) {
        turn(A, r);
        if (r + 1 < blocks) turn(B, r + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(A.v[0]), "+v"(B.v[0]) : : "memory");
    if (have) piece_store(S, p + (size_t)sr * 352, lane);
}
// ... and with THREE register sets in rotation, the compiler keeping the counter: turn r works in place on set r % 3 (loaded two
// turns ago), then the set of turn r - 1 is stored and at once refilled with piece r + 2 (clamped, so every turn issues the same
// loads).  No set is ever copied while a load into it is pending.
__global__ __launch_bounds__(256, 3) void k_streams4(double *b, int nseg, int blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;
    double *p = b + (size_t)seg * blocks * 352;
    Piece P0, P1, P2;
    piece_load(P0, p, lane);
    piece_load(P1, p + 352, lane);
    piece_load(P2, p + 704, lane);                            // (blocks >= 3)
#define TURN(CUR, PREV, R)                                                                                             \
    {                                                                                                                  \
        asm volatile("" : "+v"(CUR.v[0]), "+v"(CUR.v[1]), "+v"(CUR.v[2]), "+v"(CUR.v[3]), "+v"(CUR.v[4]), "+v"(CUR.w)); \
        piece_store(PREV, p + (size_t)((R) - 1) * 352, lane);                                                          \
        const int rn_ = (R) + 2 < blocks ? (R) + 2 : blocks - 1;                                                       \
        piece_load(PREV, p + (size_t)rn_ * 352, lane);                                                                 \
    }
    // turn 0 by hand (nothing to store yet; set 2 is already loading)
    asm volatile("" : "+v"(P0.v[0]), "+v"(P0.v[1]), "+v"(P0.v[2]), "+v"(P0.v[3]), "+v"(P0.v[4]), "+v"(P0.w));
    int r = 1;
    for (; r + 2 < blocks; r += 3) {
        TURN(P1, P0, r);
        TURN(P2, P1, r + 1);
        TURN(P0, P2, r + 2);
    }
    // the last turns, then the last processed set
    if (r < blocks) { TURN(P1, P0, r); ++r; if (r < blocks) { TURN(P2, P1, r); piece_store(P2, p + (size_t)r * 352, lane); } else piece_store(P1, p + (size_t)(r - 1) * 352, lane); }
    else piece_store(P0, p + (size_t)(r - 1) * 352, lane);
#undef TURN
}
__global__ __launch_bounds__(256) void k_linear(double2 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 v = b[i];
        asm volatile("" : "+v"(v.x), "+v"(v.y));
        b[i] = v;
    }
}

static int g_nseg = 3072;       // streams (waves) of the stream kernels: third argument
static double time_kernel(int streams, void *buf, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
    const int nseg = g_nseg, blocks = (int)(bytes / 2816 / nseg);
    auto go = [&]() {
        if (streams == 4) k_streams4<<<nseg / 4, 256>>>((double *)buf, nseg, blocks);
        else if (streams == 3) k_streams3<<<nseg / 4, 256>>>((double *)buf, nseg, blocks);
        else if (streams == 2) k_streams2<<<nseg / 4, 256>>>((double *)buf, nseg, blocks);
        else if (streams) k_streams<<<nseg / 4, 256>>>((double *)buf, nseg, blocks);
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
    if (argc > 3) g_nseg = atoi(argv[3]);
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
        const double a = time_kernel(1, p, bytes, e0, e1), a2 = time_kernel(2, p, bytes, e0, e1), a3 = time_kernel(4, p, bytes, e0, e1), b = time_kernel(0, p, bytes, e0, e1);
        printf("hipMalloc %2d  va %p (offset in 1 GB: %4zu MB)  alloc+clear %7.1f ms   streams %.2f TB/s   two pieces ahead %.2f TB/s   three sets %.2f TB/s   linear %.2f TB/s\n", c, p,
               ((size_t)p & ((1ull << 30) - 1)) >> 20, ams, a, a2, a3, b);
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
        const double a = time_kernel(1, va, bytes, e0, e1), b = time_kernel(0, va, bytes, e0, e1);
        printf("vmm       %2d  va %p (offset in 1 GB: %4zu MB)  alloc+clear %7.1f ms   streams %.2f TB/s   linear %.2f TB/s\n", c, va,
               ((size_t)va & ((1ull << 30) - 1)) >> 20, ams, a, b);
        fflush(stdout);
    }
    return 0;
}
