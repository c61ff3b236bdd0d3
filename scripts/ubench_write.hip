// dev microbenchmark: write-only non-temporal stream (16 B per lane, 1-KiB wave stores, tiles from a ticket) — the rate
// at which a launch's drain (tape stores with no reads left to overlap) could run at best.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_write.hip -o /tmp/ubw && /tmp/ubw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int W>
__global__ __launch_bounds__(W * 64) void wr(uint4* __restrict__ out, uint32_t* ticket, uint32_t num_tiles) {
    __shared__ uint32_t s_tile;
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (;;) {
        if (t == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        __syncthreads();
        if (tile >= num_tiles) break;
        uint4* base = out + ((uint64_t)tile * W + w) * (8 * 64);   // 8 KiB per wave per tile
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32x4 x = {tile, w, (uint32_t)k, lane};
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(base + k * 64 + lane));
        }
    }
}

template <int W>
int run(uint4* out, uint64_t n, uint32_t* ticket, int bpc) {
    const uint32_t tiles = (uint32_t)(n / (W * 8192));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipMemsetAsync(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((wr<W>), dim3(256 * bpc), dim3(W * 64), 0, 0, out, ticket, tiles);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 1 && ms < best) best = ms;
    }
    printf("write only: %4.0f MiB  waves/WG %d  WGs/CU %d (%2d waves/CU)  %.4f ms  %.2f TB/s\n", n / 1048576.0, W, bpc, bpc * W, best,
           n / best / 1e9);
    return 0;
}

int main() {
    uint4* out; uint32_t* ticket;
    CHECK(hipMalloc(&out, 2ull << 30)); CHECK(hipMalloc(&ticket, 64));
    CHECK(hipMemset(out, 0, 2ull << 30));
    for (uint64_t n : {64ull << 20, 256ull << 20, 2ull << 30})
        for (int bpc : {1, 2}) { run<8>(out, n, ticket, bpc); run<4>(out, n, ticket, bpc * 2); }
    return 0;
}
