// dev microbenchmark: does the ADDRESS PATTERN of the stage-1 stream matter?  Same traffic as the 64x31 corpus (8 B
// written per 32 B read, nt both ways, tiles from a ticket), three ways of laying a workgroup's 8 rounds over its tile:
//   pattern 0  wave-contiguous spans (the kernel): wave w reads  tile + w * 32 KiB + r * 4 KiB
//   pattern 1  workgroup-contiguous rounds:        wave w reads  tile + r * (W * 4 KiB) + w * 4 KiB
//   pattern 2  like 1, but the lanes of the WHOLE workgroup interleave at 16 B (one 1-KiB line set per wave instruction
//              is replaced by W waves touching the same 4 x 1 KiB rows): tile + r * (W*4K) + j * (W*1K) + w * 1K
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_pattern.hip -o /tmp/ubp && /tmp/ubp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int W, int PATTERN>
__global__ __launch_bounds__(W * 64) void stream(const uint8_t* __restrict__ in, uint4* __restrict__ out, uint32_t* ticket,
                                                 uint32_t num_tiles) {
    __shared__ uint32_t s_tile;
    constexpr uint32_t kTile = W * 32768;
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (;;) {
        if (t == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        __syncthreads();
        if (tile >= num_tiles) break;
        const uint64_t tile0 = (uint64_t)tile * kTile;
        const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in) + tile0, 0, kTile, 0x00020000);
        uint4* obase = out + ((uint64_t)tile * W + w) * (8 * 64);   // 8 KiB of output per 32 KiB read, per wave
        uint32_t ostore = 0;
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += 2) {
            uint4 v[2][4];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = r0 + d;
                    uint32_t off;
                    if (PATTERN == 0) off = w * 32768 + r * 4096 + j * 1024 + lane * 16;
                    else if (PATTERN == 1) off = r * (W * 4096) + w * 4096 + j * 1024 + lane * 16;
                    else off = r * (W * 4096) + j * (W * 1024) + w * 1024 + lane * 16;
                    const auto x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 2);
                    v[d][j] = make_uint4(x[0], x[1], x[2], x[3]);
                }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                uint4 o;
                o.x = v[d][0].x ^ v[d][1].x ^ v[d][2].x ^ v[d][3].x; o.y = v[d][0].y ^ v[d][1].y ^ v[d][2].y ^ v[d][3].y;
                o.z = v[d][0].z ^ v[d][1].z ^ v[d][2].z ^ v[d][3].z; o.w = v[d][0].w ^ v[d][1].w ^ v[d][2].w ^ v[d][3].w;
                const uint32_t upto = (uint32_t)(r0 + d + 1);   // one 1-KiB store per round
                for (; ostore < upto; ++ostore) {
                    const u32x4 x = {o.x + ostore, o.y, o.z, o.w};
                    __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(obase + ostore * 64 + lane));
                }
            }
        }
    }
}

template <int W, int PATTERN>
int run(const uint8_t* in, uint4* out, uint64_t n, uint32_t* ticket, int bpc) {
    const uint32_t tiles = (uint32_t)(n / (W * 32768));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipMemsetAsync(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<W, PATTERN>), dim3(256 * bpc), dim3(W * 64), 0, 0, in, out, ticket, tiles);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 1 && ms < best) best = ms;
    }
    printf("waves/WG %d pattern %d  WGs/CU %d (%2d waves/CU)  %.3f ms  read %.2f TB/s (%.1f %% of 8)  total %.2f TB/s\n", W, PATTERN,
           bpc, bpc * W, best, n / best / 1e9, n / best / 1e9 * 12.5, 1.25 * n / best / 1e9);
    return 0;
}

int main() {
    const uint64_t n = 4ull << 30;
    uint8_t* in; uint4* out; uint32_t* ticket;
    CHECK(hipMalloc(&in, n)); CHECK(hipMalloc(&out, n / 4 + (1 << 20))); CHECK(hipMalloc(&ticket, 64));
    CHECK(hipMemset(in, 0x61, n)); CHECK(hipMemset(out, 0, n / 4));
    for (int bpc : {1, 2}) {
        run<8, 0>(in, out, n, ticket, bpc); run<8, 1>(in, out, n, ticket, bpc); run<8, 2>(in, out, n, ticket, bpc);
    }
    for (int bpc : {2, 4}) {
        run<4, 0>(in, out, n, ticket, bpc); run<4, 1>(in, out, n, ticket, bpc); run<4, 2>(in, out, n, ticket, bpc);
    }
    return 0;
}
