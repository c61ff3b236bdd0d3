// dev microbenchmark: VALU issue rates on gfx950 for the instruction kinds the stage-1 kernel uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;   // independent chains

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[UNROLL];
    uint64_t b[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * 977u + i * 131u; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const uint32_t n = a[(i + 1) % UNROLL];
            const uint64_t nb = b[(i + 1) % UNROLL];
            if (KIND == 0) a[i] = (a[i] ^ 0x2c2c2c2cu) + n;                               // xor lit + add  (2)
            if (KIND == 1) a[i] = (a[i] + 0x7f7f7f7fu) ^ n;                               // add lit + xor  (2)
            if (KIND == 2) a[i] = __builtin_amdgcn_udot4(a[i], 0x08040201u, n, false);    // dot4 (1)
            if (KIND == 3) a[i] = (a[i] & n) | (a[(i + 2) % UNROLL] & 0x80808080u);       // bitop3? (1-2)
            if (KIND == 4) b[i] = (b[i] + 0x7f7f7f7f7f7f7f7full) ^ nb;                    // u64 add + 2 xor
            if (KIND == 5) b[i] = (b[i] << 3) ^ nb;                                       // u64 shl + 2 xor
            if (KIND == 6) a[i] = __builtin_amdgcn_lerp(a[i], 0xf6f6f6f6u, n);            // v_lerp_u8 (1)
            if (KIND == 7) a[i] = __builtin_amdgcn_perm(a[i], n, 0x07020500u);            // v_perm (1)
            if (KIND == 8) a[i] = __builtin_popcount(a[i]) + n;                           // bcnt (1, has add)
            if (KIND == 9) a[i] = __builtin_amdgcn_mbcnt_lo(a[i], n);                     // mbcnt (1)
            if (KIND == 10) a[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false) + n;
            if (KIND == 11) a[i] = __builtin_amdgcn_sad_u8(a[i], 0x2c2c2c2cu, n);         // v_sad_u8 (1)
            if (KIND == 12) a[i] = a[i] * 0x00204081u + n;                                // mul_lo + add / mad
            if (KIND == 13) a[i] = (a[i] ^ n) ;                                           // plain xor (1)
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32);
    if (r == 0x12345678u) out[0] = r;
}

template <int KIND>
int run(const char* name, int blocks_per_cu, uint32_t* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double ops = (double)grid * 4 /*waves*/ * ITERS * UNROLL;   // source-level ops per wave
    // cycles per op per SIMD assuming 2.4 GHz and even spread over 1024 SIMDs
    const double per_simd = ops / 1024.0;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f Gop/s (wave-ops)  ~%.2f cyc/op/SIMD @2.4GHz\n", name, blocks_per_cu,
           ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 / per_simd);
    return 0;
}

int main() {
    uint32_t* d; CHECK(hipMalloc(&d, 64));
    for (int bpc : {1, 2, 8}) {
        run<0>("v_xor_b32 lit", bpc, d);
        run<1>("v_add_u32 lit", bpc, d);
        run<2>("v_dot4_u32_u8", bpc, d);
        run<3>("and/or (bitop3?)", bpc, d);
        run<4>("u64 add", bpc, d);
        run<5>("u64 shl3 xor", bpc, d);
        run<6>("v_lerp_u8", bpc, d);
        run<7>("v_perm_b32", bpc, d);
        run<8>("popcount+add", bpc, d);
        run<9>("mbcnt_lo", bpc, d);
        run<10>("dpp row_shr1 + add", bpc, d);
        run<11>("v_sad_u8", bpc, d);
        run<12>("v_mul_lo_u32", bpc, d);
        run<13>("u64 shl1 xor", bpc, d);
    }
    return 0;
}
