// dev microbenchmark (round 3): WHY does a plain copy (6.47 TB/s read + write, bench.py's plain_copy_yardstick) beat the
// ticket-driven probe (5.3-5.9 TB/s) on write-heavy mixes — and which part of the stage-1 traffic SHAPE would have to
// change for the dense corpus (1.6 B of tape per byte read) to get there?  One "chunk" = 4 KiB read by one wave
// (four fully coalesced 1-KiB loads) -> WR/16 x 4 KiB written (1-KiB wave stores, line aligned), output of chunk c at
// out + c * (WR/16) * 4 KiB, i.e. the output is laid out in input order like a copy's.  What varies is WHICH wave handles
// which chunk WHEN:
//   map 0  the stage-1 kernel's shape: a workgroup draws a tile of (waves x 8) chunks from a ticket, wave w takes the 8
//          consecutive chunks  tile*8W + 8w + r   (wave-contiguous 32-KiB spans), DEPTH chunks' loads in flight
//   map 1  same tiles, round-interleaved: chunk tile*8W + W*r + w  (the workgroup reads W*4 KiB contiguous per round)
//   map 2  no tiles, no tickets: persistent waves grid-stride over chunks (chunk = gw + k * G): at any time the chip works
//          on ONE compact window of G chunks
//   map 3  plain grid, one chunk per wave, no persistence (the copy kernel's shape, 4 KiB per wave instead of 1 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WR16>
__device__ __forceinline__ void do_chunk(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t c, uint32_t lane) {
    const u32x4* p = in + c * 256 + lane;
    u32x4 v0 = __builtin_nontemporal_load(p), v1 = __builtin_nontemporal_load(p + 64), v2 = __builtin_nontemporal_load(p + 128),
          v3 = __builtin_nontemporal_load(p + 192);
    u32x4 o = v0 ^ v1 ^ v2 ^ v3;
    // stores due for chunk c: [c * WR16 / 4, (c + 1) * WR16 / 4) in 1-KiB units  (WR16/4 KiB per 4-KiB chunk)
    const uint64_t from = c * WR16 / 4, upto = (c + 1) * WR16 / 4;
    for (uint64_t q = from; q < upto; ++q) {
        u32x4 x = o;
        x.x += (uint32_t)q;
        __builtin_nontemporal_store(x, out + q * 64 + lane);
    }
}

// two chunks' loads in flight before the first store
template <int WR16>
__device__ __forceinline__ void do_chunk2(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t c0, uint64_t c1, uint32_t lane) {
    const u32x4* p = in + c0 * 256 + lane;
    const u32x4* q = in + c1 * 256 + lane;
    u32x4 a0 = __builtin_nontemporal_load(p), a1 = __builtin_nontemporal_load(p + 64), a2 = __builtin_nontemporal_load(p + 128),
          a3 = __builtin_nontemporal_load(p + 192);
    u32x4 b0 = __builtin_nontemporal_load(q), b1 = __builtin_nontemporal_load(q + 64), b2 = __builtin_nontemporal_load(q + 128),
          b3 = __builtin_nontemporal_load(q + 192);
    u32x4 oa = a0 ^ a1 ^ a2 ^ a3, ob = b0 ^ b1 ^ b2 ^ b3;
    for (uint64_t s = c0 * WR16 / 4; s < (c0 + 1) * WR16 / 4; ++s) { u32x4 x = oa; x.x += (uint32_t)s; __builtin_nontemporal_store(x, out + s * 64 + lane); }
    for (uint64_t s = c1 * WR16 / 4; s < (c1 + 1) * WR16 / 4; ++s) { u32x4 x = ob; x.x += (uint32_t)s; __builtin_nontemporal_store(x, out + s * 64 + lane); }
}

template <int WR16, int MAP, int DEPTH, int WAVES, int CPW = 8>
__global__ __launch_bounds__(WAVES * 64) void stream(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t* ticket,
                                                    uint64_t n_chunks) {
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (MAP == 3) {
        const uint64_t c = (uint64_t)blockIdx.x * WAVES + w;
        if (c < n_chunks) do_chunk<WR16>(in, out, c, lane);
        return;
    }
    if (MAP == 2) {
        const uint64_t G = (uint64_t)gridDim.x * WAVES;
        uint64_t c = (uint64_t)blockIdx.x * WAVES + w;
        if (DEPTH == 2) {
            for (; c + G < n_chunks; c += 2 * G) do_chunk2<WR16>(in, out, c, c + G, lane);
            if (c < n_chunks) do_chunk<WR16>(in, out, c, lane);
        } else {
            for (; c < n_chunks; c += G) do_chunk<WR16>(in, out, c, lane);
        }
        return;
    }
    __shared__ uint32_t s_tile;
    const uint64_t n_tiles = n_chunks / (CPW * WAVES);
    for (;;) {
        if (MAP == 4) {  // plain grid, one tile per workgroup, no persistence, no ticket
            s_tile = blockIdx.x;
        } else {
            if (t == 0) s_tile = atomicAdd(ticket, 1u);
            __syncthreads();
        }
        const uint32_t tile = MAP == 4 ? blockIdx.x : s_tile;
        if (MAP != 4) __syncthreads();
        if (tile >= n_tiles) break;
        const uint64_t c0 = (uint64_t)tile * CPW * WAVES;
        if (MAP >= 5) {
            // reads exactly as map 0 (wave-contiguous spans); only the WRITE side differs:
            //   5: the wave's whole output (its span's share) leaves at the end of the span, in one burst
            //   6: barrier, then the workgroup writes its tile's output region cooperatively in address order
            //      (1-KiB pieces round-robin over the waves)
            //   7: no barrier: after every chunk pair the wave writes its share, interleaved over the waves at 1 KiB
            //      (the workgroup's output front advances together)
            u32x4 acc = {0, 0, 0, 0};
            constexpr int kPerWave = CPW * WR16 / 4;          // 1-KiB stores per wave per tile
            u32x4* const tile_out = out + (uint64_t)tile * WAVES * kPerWave * 64;
            int done = 0;
#pragma unroll
            for (int r = 0; r < CPW; r += 2) {
                const u32x4* p = in + (c0 + CPW * w + r) * 256 + lane;
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc ^= v[j];
                if (MAP == 7) {
                    const int upto = (r + 2) * WR16 / 4;
                    for (; done < upto; ++done) {
                        u32x4 x = acc; x.x += done;
                        __builtin_nontemporal_store(x, tile_out + ((uint64_t)done * WAVES + w) * 64 + lane);
                    }
                }
            }
            if (MAP == 5) {
                for (int q = 0; q < kPerWave; ++q) {
                    u32x4 x = acc; x.x += q;
                    __builtin_nontemporal_store(x, tile_out + ((uint64_t)w * kPerWave + q) * 64 + lane);
                }
            }
            if (MAP == 6) {
                __syncthreads();
                for (int q = 0; q < kPerWave; ++q) {
                    u32x4 x = acc; x.x += q;
                    __builtin_nontemporal_store(x, tile_out + ((uint64_t)q * WAVES + w) * 64 + lane);
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < CPW; r += DEPTH) {
            const uint64_t ca = MAP != 1 ? c0 + CPW * w + r : c0 + (uint64_t)WAVES * r + w;
            const uint64_t cb = MAP != 1 ? ca + 1 : ca + WAVES;
            if (DEPTH == 2) do_chunk2<WR16>(in, out, ca, cb, lane);
            else do_chunk<WR16>(in, out, ca, lane);
        }
        if (MAP == 4) break;
    }
}

template <int WR16, int MAP, int DEPTH, int WAVES, int CPW = 8>
int run(const u32x4* in, u32x4* out, uint64_t n, uint32_t* ticket, int bpc) {
    const uint64_t chunks = n / 4096;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    const uint32_t grid = MAP == 3 ? (uint32_t)((chunks + WAVES - 1) / WAVES) : MAP == 4 ? (uint32_t)(chunks / (CPW * WAVES)) : 256u * bpc;
    for (int rep = 0; rep < 7; ++rep) {
        CHECK(hipMemsetAsync(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<WR16, MAP, DEPTH, WAVES, CPW>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, ticket, chunks);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 1 && ms < best) best = ms;
    }
    const double rd = (double)n, wr = (double)n * WR16 / 16;
    printf("w/r %.3f  map %d cpw %d depth %d waves/WG %d  WG/CU %2d (%2d waves/CU)  %.3f ms  read %.2f TB/s = %.1f %%  total %.2f TB/s\n",
           WR16 / 16.0, MAP, CPW, DEPTH, WAVES, MAP >= 3 ? 0 : bpc, MAP >= 3 ? 0 : bpc * WAVES, best, rd / best / 1e9, rd / best / 1e9 / 80.0,
           (rd + wr) / best / 1e9);
    return 0;
}

template <int WR16>
int sweep(const u32x4* in, u32x4* out, uint64_t n, uint32_t* ticket) {
    for (int bpc : {1, 2}) {
        run<WR16, 0, 2, 8, 8>(in, out, n, ticket, bpc);   // the kernel's shape
        run<WR16, 5, 2, 8, 8>(in, out, n, ticket, bpc);   // + one burst per span
        run<WR16, 6, 2, 8, 8>(in, out, n, ticket, bpc);   // + cooperative ordered flush after a barrier
        run<WR16, 7, 2, 8, 8>(in, out, n, ticket, bpc);   // + interleaved, advancing together
        run<WR16, 0, 2, 8, 4>(in, out, n, ticket, bpc);   // half-size tiles
        run<WR16, 6, 2, 8, 4>(in, out, n, ticket, bpc);
        run<WR16, 7, 2, 8, 4>(in, out, n, ticket, bpc);
        run<WR16, 0, 2, 8, 2>(in, out, n, ticket, bpc);   // quarter-size tiles
        run<WR16, 7, 2, 8, 2>(in, out, n, ticket, bpc);
    }
    run<WR16, 0, 2, 4, 2>(in, out, n, ticket, 2);
    run<WR16, 7, 2, 4, 2>(in, out, n, ticket, 2);
    run<WR16, 7, 2, 4, 8>(in, out, n, ticket, 2);
    run<WR16, 7, 2, 4, 8>(in, out, n, ticket, 4);
    run<WR16, 3, 1, 4>(in, out, n, ticket, 0);
    return 0;
}

int main() {
    const uint64_t n = 1ull << 30;
    u32x4* in; u32x4* out; uint32_t* ticket;
    CHECK(hipMalloc(&in, n)); CHECK(hipMalloc(&out, 2 * n)); CHECK(hipMalloc(&ticket, 64));
    CHECK(hipMemset(in, 0x61, n)); CHECK(hipMemset(out, 0, 2 * n));
    printf("# 1 GiB read; 4-KiB chunks; nt loads and stores; best of 5 after 2 warm-ups\n");
    sweep<26>(in, out, n, ticket);   // 1.625: the dense corpus
    sweep<16>(in, out, n, ticket);   // copy
    sweep<8>(in, out, n, ticket);
    sweep<4>(in, out, n, ticket);    // the 64x31 corpus
    return 0;
}
