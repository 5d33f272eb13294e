#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__global__ void k(const double *x, double *out, int n) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *slot = (double *)(smem + wave * 1024);
    const uint32_t dst = (uint32_t)(uintptr_t)slot;
    const uint32_t dstu = __builtin_amdgcn_readfirstlane(dst);
    const int64_t base = (int64_t)(blockIdx.x * 4 + wave) * 128;
    if (2 * lane < n) glds16(x + base, lane * 16u, dstu);
    wait_vm<0>();
    out[base + lane] = slot[lane] * 2.0;
    out[base + 64 + lane] = slot[64 + lane] * 2.0;
}
int main() {
    const int n = 128 * 4 * 8;
    double *x, *o;
    hipMalloc(&x, n * 8); hipMalloc(&o, n * 8);
    double *h = new double[n];
    for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(x, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 8, 256, 4096, 0, x, o, 128);
    hipMemcpy(h, o, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (h[i] != 2.0 * i) ++bad;
    printf("bad %d of %d\n", bad, n);
    return bad != 0;
}
