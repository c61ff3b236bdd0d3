// dev micro-test: are scalar atomics (s_atomic_add ... glc) on gfx950 coherent across the eight XCDs on ordinary
// (coarse-grained) device memory?  Every wave draws one ticket; the returned values must be a permutation of 0 .. n-1.
// No loop waits on memory anywhere: the kernel cannot hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void draw_scalar(unsigned* counter, unsigned* out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
// compare-and-swap semantics: DATA[0] = new value, DATA[1] = compare; the old value comes back in DATA[0]
__global__ void cas_check(unsigned* p, unsigned* out) {
    unsigned long long d;
    d = (0ull << 32) | 7ull;  // expect 0, store 7
    asm volatile("s_atomic_cmpswap %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(d) : "s"(p) : "memory");
    out[0] = (unsigned)d;
    d = (0ull << 32) | 9ull;  // expect 0 (it is 7 now): must fail
    asm volatile("s_atomic_cmpswap %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(d) : "s"(p) : "memory");
    out[1] = (unsigned)d;
    d = (7ull << 32) | 9ull;  // expect 7, store 9
    asm volatile("s_atomic_cmpswap %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(d) : "s"(p) : "memory");
    out[2] = (unsigned)d;
    out[3] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void draw_vector(unsigned* counter, unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(counter, 1u);
}
int main() {
    const int n = 1 << 16;
    unsigned *c, *o;
    hipMalloc(&c, 256); hipMalloc(&o, n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(c, 0, 256); hipMemset(o, 0xff, n * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(draw_vector, dim3(n), dim3(64), 0, 0, c, o);
        else hipLaunchKernelGGL(draw_scalar, dim3(n), dim3(64), 0, 0, c, o);
        hipEventRecord(e1);
        hipError_t e = hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h(n); unsigned hc = 0;
        hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != (unsigned)i;
        printf("%s: rc=%d counter=%u (want %d) values out of place=%d  %.3f ms (%.1f draws/us)\n", mode ? "scalar" : "vector", (int)e, hc, n, bad, ms, n / ms / 1e3);
    }
    hipMemset(c, 0, 256);
    hipLaunchKernelGGL(cas_check, dim3(1), dim3(64), 0, 0, c, o);
    unsigned r[4]; hipDeviceSynchronize(); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("cas: returned %u %u %u, memory %u (want 0 7 7, 9)\n", r[0], r[1], r[2], r[3]);
    return 0;
}
