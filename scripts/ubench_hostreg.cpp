// dev microbenchmark (round 3): is pinning the CALLER's pages (hipHostRegister) cheaper than copying
// them into a pinned staging slot?  The ingest path (csvsimd_stage1_index) stages every 32-MiB chunk with a sliced
// non-temporal memcpy; the alternative is to register the chunk (or the whole mapping) and let the DMA engine read the
// caller's memory directly.  Prints: register / unregister time per size (malloc'ed, touched pages), H2D rate from
// registered vs hipHostMalloc'ed memory, and the staging memcpy it would replace.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t total = 2ull << 30;
    char* user = (char*)aligned_alloc(4096, total);
    memset(user, 0x61, total);  // touched: resident pages
    void* dev; CHECK(hipMalloc(&dev, 256u << 20));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // (1) register / unregister cost by size
    for (size_t sz : {32ull << 20, 256ull << 20, 2048ull << 20}) {
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CHECK(hipHostRegister(user, sz, hipHostRegisterDefault));
            double t1 = now();
            CHECK(hipHostUnregister(user));
            double t2 = now();
            printf("hipHostRegister %5zu MiB: register %.3f ms (%.1f GiB/s), unregister %.3f ms\n", sz >> 20, (t1 - t0) * 1e3,
                   sz / (t1 - t0) / 1073741824.0, (t2 - t1) * 1e3);
        }
    }
    // (2) sliding registration: chunk i+1 registered while chunk i is copied (what a pipelined ingest would do)
    {
        const size_t chunk = 32ull << 20, n = 512ull << 20;
        double t0 = now();
        CHECK(hipHostRegister(user, chunk, hipHostRegisterDefault));
        for (size_t off = 0; off < n; off += chunk) {
            CHECK(hipMemcpyAsync(dev, user + off, chunk, hipMemcpyHostToDevice, s));
            if (off + chunk < n) CHECK(hipHostRegister(user + off + chunk, chunk, hipHostRegisterDefault));
            CHECK(hipStreamSynchronize(s));
            CHECK(hipHostUnregister(user + off));
        }
        double t1 = now();
        printf("sliding register + H2D + unregister, 32-MiB chunks over 512 MiB: %.2f ms = %.1f GiB/s\n", (t1 - t0) * 1e3,
               n / (t1 - t0) / 1073741824.0);
    }
    // (3) H2D rate: hipHostMalloc'ed vs registered source, 256 MiB
    {
        const size_t sz = 256ull << 20;
        void* pin; CHECK(hipHostMalloc(&pin, sz, hipHostMallocDefault));
        memset(pin, 1, sz);
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, s)); CHECK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s)); CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("H2D 256 MiB from hipHostMalloc: %.3f ms = %.1f GiB/s\n", ms, sz / (ms * 1e-3) / 1073741824.0);
        }
        CHECK(hipHostRegister(user, sz, hipHostRegisterDefault));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, s)); CHECK(hipMemcpyAsync(dev, user, sz, hipMemcpyHostToDevice, s)); CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("H2D 256 MiB from registered malloc memory: %.3f ms = %.1f GiB/s\n", ms, sz / (ms * 1e-3) / 1073741824.0);
        }
        CHECK(hipHostUnregister(user));
        // pageable source straight into hipMemcpy (the runtime stages it itself)
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CHECK(hipMemcpy(dev, user, sz, hipMemcpyHostToDevice));
            double t1 = now();
            printf("hipMemcpy 256 MiB from pageable memory (runtime's own staging): %.3f ms = %.1f GiB/s\n", (t1 - t0) * 1e3,
                   sz / (t1 - t0) / 1073741824.0);
        }
        // (4) the staging memcpy this would replace: T threads, 256 MiB pageable -> pinned
        for (int T : {1, 4, 8, 12}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                std::vector<std::thread> th;
                for (int i = 0; i < T; ++i)
                    th.emplace_back([&, i] { memcpy((char*)pin + sz / T * i, user + sz / T * i, sz / T); });
                for (auto& x : th) x.join();
                double t1 = now();
                if (t1 - t0 < best) best = t1 - t0;
            }
            printf("memcpy pageable -> pinned, 256 MiB, %2d threads: %.2f ms = %.1f GiB/s\n", T, best * 1e3, sz / best / 1073741824.0);
        }
    }
    return 0;
}
