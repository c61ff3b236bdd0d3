// pool_stress.cpp — the ingest pipelines' host machinery (csv-simd_amd/host/ingest_pool.hpp: CopyPool, TaskThread) under
// ThreadSanitizer, without a GPU.  Test infrastructure.  It plays one pipeline's three roles the way capi.cpp does — a
// stager thread and an expander thread (TaskThreads) and the caller, all three slicing copies / widenings / parallel_fors
// over ONE pool, calls back to back so that workers and task threads are met hot (lingering) and cold (asleep) — and checks
// every byte.  It also plays the hand-over the round-5 bug was in: chunk 0 staged by the caller, chunks 1 .. by the stager,
// finishing in either order (capi.cpp: staged0 / staged).
// build + run: make -C tests/native pool_stress && tests/native/pool_stress   (tests/test_native_cpp.py does both)
#include <cstdio>
#include <numeric>
#include <random>

#include "ingest_pool.hpp"

using csvsimd_host::CopyPool;
using csvsimd_host::TaskThread;

static int g_failed = 0;
#define CHECK(x) do { if (!(x)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); ++g_failed; } } while (0)

int main() {
    CopyPool pool(5);
    TaskThread stager, expander;
    std::mt19937_64 rng(7);
    const size_t kMax = 6u << 20;
    std::vector<char> src(kMax), dst(kMax);
    std::vector<uint32_t> narrow(kMax / 8);
    std::vector<uint64_t> wide(kMax / 8);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (char)(i * 2654435761u >> 11);
    for (size_t i = 0; i < narrow.size(); ++i) narrow[i] = (uint32_t)(i * 40503u);
    for (int call = 0; call < 300; ++call) {
        const size_t n = 1 + (size_t)(rng() % kMax);
        const size_t chunks = 1 + (size_t)(rng() % 6);
        const size_t min_slice = (rng() & 1) ? (128u << 10) : CopyPool::kMinSlice;
        std::vector<size_t> cuts(chunks + 1);
        for (size_t c = 0; c <= chunks; ++c) cuts[c] = n * c / chunks;
        std::fill(dst.begin(), dst.begin() + n, 0);
        std::fill(wide.begin(), wide.end(), 0);
        // the hand-over: staged0 (the caller's chunk 0) and staged (the stager's count of chunks 1 ..) — two words
        std::mutex m;
        std::condition_variable cv;
        std::atomic<bool> staged0{false};
        std::atomic<size_t> staged{0}, expanded{0};
        const bool busy = (call % 3) != 0;
        if (busy) pool.busy();
        stager.post([&] {
            for (size_t c = 1; c < chunks; ++c) {
                pool.copy(dst.data() + cuts[c], src.data() + cuts[c], cuts[c + 1] - cuts[c], min_slice);
                { std::lock_guard<std::mutex> g(m); staged.store(c + 1, std::memory_order_release); }
                cv.notify_all();
            }
        });
        const size_t nw = std::min(n / 8, narrow.size());
        expander.post([&] {
            // (widening runs beside the copies, like the expander beside the stager)
            pool.expand(wide.data(), narrow.data(), nw / 2, 1000, min_slice);
            { std::lock_guard<std::mutex> g(m); expanded.store(1, std::memory_order_release); }
            cv.notify_all();
        });
        pool.copy(dst.data(), src.data(), cuts[1] - cuts[0], min_slice);
        { std::lock_guard<std::mutex> g(m); staged0.store(true, std::memory_order_release); }
        cv.notify_all();
        for (size_t c = 0; c < chunks; ++c) {  // the submitter: chunk c only once it is staged
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [&] { return c == 0 ? staged0.load() : staged.load() > c; });
            g.unlock();
            CHECK(std::memcmp(dst.data() + cuts[c], src.data() + cuts[c], cuts[c + 1] - cuts[c]) == 0);
        }
        // the caller's own share of the widening, and a parallel_for like the batch path's pack / unpack
        pool.expand(wide.data() + nw / 2, narrow.data() + nw / 2, nw - nw / 2, 1000, min_slice);
        std::vector<uint32_t> marks(257 + call % 900, 0);
        const std::function<void(size_t, size_t)> f = [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) marks[i] += (uint32_t)i + 1; };
        pool.parallel_for(marks.size(), 64, f);
        stager.wait();
        expander.wait();
        if (busy) pool.quiet();
        CHECK(expanded.load() == 1);
        bool ok = true;
        for (size_t i = 0; i < nw; ++i) ok = ok && wide[i] == 1000ull + narrow[i];
        CHECK(ok);
        for (size_t i = 0; i < marks.size(); ++i) ok = ok && marks[i] == (uint32_t)i + 1;
        CHECK(ok);
        if (call % 50 == 49) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // let everybody fall asleep: cold start next
    }
    std::printf(g_failed ? "pool_stress: %d check(s) FAILED\n" : "pool_stress ok: 300 calls, all bytes in place\n", g_failed);
    return g_failed ? 1 : 0;
}
