// ingest_pool.hpp — the host-side machinery of the ingest pipelines (csv-simd_amd/csrc/capi.cpp): streaming copies, a pool of
// copying threads that a call can keep polling, and a thread a context keeps for one role of a pipeline.  Plain C++17 +
// SSE2, no HIP: tests/native/pool_stress.cpp builds it with -fsanitize=thread and hammers it from several threads.
#pragma once
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace csvsimd_host {

// Streaming copy for the ingest pipeline's slices: 4 MiB per thread is below the size at which glibc's memcpy switches
// to non-temporal stores, so memcpy reads the DESTINATION lines too (read-for-ownership) — a third of the memory
// traffic of a copy whose destination nobody on the CPU is going to read (the DMA engine reads the staging slot, the
// caller reads the tape later).  Non-temporal stores leave that out and keep the caches for the copying threads'
// neighbours.  SSE2 only (x86-64 baseline): 16-byte streams fill whole write-combining lines just as well.
inline void copy_streaming(char* dst, const char* src, size_t n) {
#ifdef CSVSIMD_DEV_PROBES
    static const bool plain = getenv("CSVSIMD_PROBE_PLAIN_MEMCPY") != nullptr;
    if (plain) {
        memcpy(dst, src, n);
        return;
    }
#endif
    const size_t head = std::min<size_t>(n, (size_t)(-(uintptr_t)dst & 63u));  // up to the next 64-byte line of dst
    if (head) memcpy(dst, src, head);
    dst += head, src += head, n -= head;
    const size_t body = n & ~(size_t)63;
    for (size_t i = 0; i < body; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i));
        const __m128i b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32));
        const __m128i d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a);
        _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c);
        _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    _mm_sfence();
    if (n - body) memcpy(dst + body, src + body, n - body);
}

// 32-bit chunk-relative tape offsets (as the narrow kernel left them in the pinned slot) -> the caller's tape: u64, absolute.
// Streaming stores like copy_streaming: the destination is 8-byte aligned, head and tail entries go one by one.
inline void expand_streaming(uint64_t* dst, const uint32_t* src, size_t n, uint64_t base) {
    size_t i = 0;
    for (; i < n && ((uintptr_t)(dst + i) & 15u); ++i) dst[i] = base + src[i];
    const __m128i vb = _mm_set1_epi64x((long long)base), zero = _mm_setzero_si128();
    for (; i + 4 <= n; i += 4) {
        const __m128i v = _mm_loadu_si128((const __m128i*)(src + i));
        _mm_stream_si128((__m128i*)(dst + i), _mm_add_epi64(_mm_unpacklo_epi32(v, zero), vb));
        _mm_stream_si128((__m128i*)(dst + i + 2), _mm_add_epi64(_mm_unpackhi_epi32(v, zero), vb));
    }
    _mm_sfence();
    for (; i < n; ++i) dst[i] = base + src[i];
}

// Host-side copies of the ingest pipeline (user buffer -> pinned staging, pinned 32-bit tape -> user tape).
// One thread moves ~10-30 GB/s, less than the PCIe link it feeds, so the copies are sliced over a few
// persistent workers (measured on the MI355X host: 22 -> 39 GiB/s host buffer to tape, NOTEBOOK.md).
// Several threads may call copy() / expand() at once (round 4: the pipeline's staging thread and its expanding thread
// do): every call counts its own slices down, the workers serve whatever slice is next.
class CopyPool {
public:
    explicit CopyPool(int workers) {
        jobs_.reserve(128);  // copy() must not allocate once jobs are published
        try {
            for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { run(); });
        } catch (...) {
            // thread creation failed half way: the workers that did start wait on members of this object, and a
            // vector of joinable threads must not be destroyed — stop and join them before the exception leaves
            shutdown();
            throw;
        }
    }
    ~CopyPool() { shutdown(); }
    // synchronous: returns when all n bytes are in place.  min_slice: the smallest piece worth handing to another thread
    // (a large chunk of a long file: 2 MiB; a file of a few MiB, where the copy IS the call's critical path: 256 KiB)
    void copy(void* dst, const void* src, size_t n, size_t min_slice = kMinSlice) {
        run_sliced(Job{(char*)dst, (const char*)src, n, 0, kCopy, nullptr}, 1, min_slice);
    }
    // synchronous: dst[i] = base + src[i] for i < n
    void expand(uint64_t* dst, const uint32_t* src, size_t n, uint64_t base, size_t min_slice = kMinSlice) {
        run_sliced(Job{(char*)dst, (const char*)src, n, base, kWiden, nullptr}, 4, min_slice);
    }
    // synchronous: f(begin, end) over [0, n) in pieces of at least min_items (the host-pointer batch packs and unpacks
    // thousands of small files per group: one memcpy each, spread over the pool)
    void parallel_for(size_t n, size_t min_items, const std::function<void(size_t, size_t)>& f) {
        run_sliced(Job{nullptr, reinterpret_cast<const char*>(&f), n, 0, kFunc, nullptr}, 1, std::max<size_t>(min_items, 1));
    }
    // From here to the matching quiet(): idle workers poll for slices instead of sleeping on the condition variable — a
    // sleeping worker takes 20-60 us to start on a slice (futex wake + a core leaving its idle state), which is the whole
    // copy time of a few MiB.  Held for the duration of ONE ingest call (RAII: Busy); nests.
    // After the call has ended the workers go on polling for kLingerSeconds: a caller that reads file after file finds them
    // awake (a sleeping worker's first slice starts 20-60 us late — a quarter of a 4-MiB call).
    void busy() { spinners_.fetch_add(1, std::memory_order_acq_rel); cv_work_.notify_all(); }
    void quiet() {
        linger_until_.store(steady_seconds() + kLingerSeconds, std::memory_order_release);
        spinners_.fetch_sub(1, std::memory_order_acq_rel);
    }
    static constexpr double kLingerSeconds = 300e-6;
    static double steady_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    struct Busy {
        CopyPool* p;
        explicit Busy(CopyPool* pool) : p(pool) { if (p) p->busy(); }
        ~Busy() { if (p) p->quiet(); }
        Busy(const Busy&) = delete;
        Busy& operator=(const Busy&) = delete;
    };
    static constexpr size_t kMinSlice = 2u << 20;  // source bytes (512 KiB slices of a 4-MiB chunk of a LONG file: measured, no gain)

private:
    enum Kind : int { kCopy, kWiden, kFunc };
    struct Call {   // lives on the issuing call's frame until every slice has been executed
        size_t left = 0;  // slices still out (guarded by m_)
    };
    struct Job {
        char* dst;
        const char* src;
        size_t n;  // bytes (copy) or entries (expand)
        uint64_t base;  // expand: added to every offset
        Kind kind;
        Call* call;
    };
    static void execute(const Job& j) {
        if (j.kind == kWiden) expand_streaming((uint64_t*)j.dst, (const uint32_t*)j.src, j.n, j.base);
        else if (j.kind == kFunc) (*reinterpret_cast<const std::function<void(size_t, size_t)>*>(j.src))((size_t)j.base, (size_t)j.base + j.n);
        else copy_streaming(j.dst, j.src, j.n);
    }
    // unit = source bytes per item of n
    void run_sliced(Job whole, size_t unit, size_t min_slice) {
        const size_t bytes = whole.n * unit;
        const size_t floor_ = whole.kind == kFunc ? std::max<size_t>(min_slice, 1) : std::max<size_t>(min_slice, 4096);
        const size_t parts = std::min<size_t>(threads_.size() + 1, std::max<size_t>(1, bytes / floor_));
        Call call;
        if (parts <= 1) {
            execute(whole);
            return;
        }
        const size_t slice = whole.kind == kFunc ? (whole.n + parts - 1) / parts
                                                 : (((whole.n / parts) + 4095) & ~(size_t)4095);  // items; a multiple of 4096 keeps every slice aligned
        const size_t dst_unit = whole.kind == kWiden ? 8 : 1, src_unit = whole.kind == kWiden ? 4 : 1;
        {
            std::lock_guard<std::mutex> g(m_);
            if (jobs_.capacity() < jobs_.size() + parts) jobs_.reserve(jobs_.size() + parts);  // before anything is published
            for (size_t off = slice; off < whole.n; off += slice) {
                Job j = whole;
                if (whole.kind != kFunc) j.dst = whole.dst + off * dst_unit;
                j.n = std::min(slice, whole.n - off);
                if (whole.kind == kFunc) j.base = off;
                else j.src = whole.src + off * src_unit;
                j.call = &call;
                jobs_.push_back(j);
                ++call.left;
            }
            pending_.store(jobs_.size(), std::memory_order_release);
        }
        cv_work_.notify_all();  // (pollers see pending_ first; a worker that has just gone to sleep needs the notification)
        Job first = whole;
        first.n = std::min(slice, whole.n);
        execute(first);  // the calling thread takes the first slice
        std::unique_lock<std::mutex> g(m_);
        // ... and, rather than sleep while slices of its own call are still queued, more of them
        while (call.left != 0) {
            bool mine = false;
            Job j{};
            for (size_t q = jobs_.size(); q-- > 0;)
                if (jobs_[q].call == &call) {
                    j = jobs_[q];
                    jobs_.erase(jobs_.begin() + (std::ptrdiff_t)q);
                    pending_.store(jobs_.size(), std::memory_order_release);
                    mine = true;
                    break;
                }
            if (!mine) {
                if (spinners_.load(std::memory_order_acquire)) {  // the last slices are a few microseconds away: poll
                    g.unlock();
                    for (;;) {
                        for (int i = 0; i < 64; ++i) _mm_pause();
                        std::lock_guard<std::mutex> g2(m_);
                        if (call.left == 0) break;
                    }
                    g.lock();
                } else {
                    cv_done_.wait(g, [&call] { return call.left == 0; });
                }
                break;
            }
            g.unlock();
            execute(j);
            g.lock();
            --call.left;
        }
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            stop_flag_.store(true, std::memory_order_release);
        }
        cv_work_.notify_all();
        for (auto& t : threads_)
            if (t.joinable()) t.join();
    }
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                while (!stop_ && jobs_.empty()) {
                    const bool hot = spinners_.load(std::memory_order_acquire) != 0 ||
                                     steady_seconds() < linger_until_.load(std::memory_order_acquire);
                    if (hot) {
                        // an ingest call is running (or one has just ended): poll (without the lock) until a slice shows up,
                        // the call ends and the linger runs out, or the pool is stopped
                        g.unlock();
                        for (int spins = 0; pending_.load(std::memory_order_acquire) == 0 && !stop_flag_.load(std::memory_order_acquire); ++spins) {
                            for (int i = 0; i < 32; ++i) _mm_pause();
                            if ((spins & 15) == 15 && spinners_.load(std::memory_order_acquire) == 0 &&
                                steady_seconds() >= linger_until_.load(std::memory_order_acquire))
                                break;
                        }
                        g.lock();
                    } else {
                        cv_work_.wait(g);
                    }
                }
                if (jobs_.empty()) return;  // stop requested and nothing left
                j = jobs_.front();          // oldest first: the call that has waited longest
                jobs_.erase(jobs_.begin());
                pending_.store(jobs_.size(), std::memory_order_release);
            }
            execute(j);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--j.call->left == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Job> jobs_;
    std::atomic<size_t> pending_{0};     // jobs_.size(), readable without the lock (pollers)
    std::atomic<int> spinners_{0};       // ingest calls in progress: workers poll instead of sleeping
    std::atomic<double> linger_until_{0.0};  // ... and until then after the last one ended
    std::atomic<bool> stop_flag_{false};
    bool stop_ = false;
};

// A thread a context keeps for one role of the ingest pipeline (stager, expander, tape prefaulter): started when a call
// first needs it, then handed one task per call.  Round 4 created and joined two std::threads per call (~60-100 us, a
// third of a 4-MiB call).
class TaskThread {
public:
    TaskThread() : th_([this] { run(); }) {}
    ~TaskThread() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    TaskThread(const TaskThread&) = delete;
    TaskThread& operator=(const TaskThread&) = delete;
    // the task must not throw; everything it references has to stay alive until wait() has returned
    void post(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> g(m_);
            task_ = std::move(f);
            busy_ = true;
            posted_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !busy_; });
    }

private:
    void run() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            // a task has just ended: the next call's is probably microseconds away — poll for ~300 us before sleeping
            if (!stop_ && !busy_) {
                g.unlock();
                const double until = CopyPool::steady_seconds() + CopyPool::kLingerSeconds;
                for (int spins = 0; !posted_.load(std::memory_order_acquire); ++spins) {
                    for (int i = 0; i < 32; ++i) _mm_pause();
                    if ((spins & 15) == 15 && CopyPool::steady_seconds() >= until) break;
                }
                g.lock();
            }
            cv_.wait(g, [this] { return stop_ || busy_; });
            if (!busy_) return;
            posted_.store(false, std::memory_order_release);
            std::function<void()> f = std::move(task_);
            g.unlock();
            f();
            f = nullptr;
            g.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> task_;
    std::atomic<bool> posted_{false};  // busy_ went up (readable without the lock: the lingering poll)
    bool busy_ = false, stop_ = false;
    std::thread th_;  // last: started when everything above exists
};

}  // namespace csvsimd_host
