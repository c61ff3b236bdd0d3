// dev tool: what fresh host memory and page-cache reads cost on THIS host (round 5, csvsimd_create):
//   (a) first-touch of 512 MiB anonymous memory: touch / MADV_POPULATE_WRITE, 4-KiB / THP, 1..8 threads
//   (b) 1 GiB page-cache-hot file -> a (pinned, if a GPU is there) buffer: pread slices vs mmap + streaming copy, 1..8 threads
// build: hipcc -O2 -o hostmem hostmem.cpp -lpthread   run: ./hostmem /path/to/scratch_file
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>
#include <emmintrin.h>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void stream_copy(char* dst, const char* src, size_t n) {
    for (size_t i = 0; i + 64 <= n; i += 64) {
        __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a); _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c); _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    _mm_sfence();
}
template <class F> static double par(int threads, size_t n, F f) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([=] { size_t a = (n / threads * t) & ~(size_t)4095, b = t + 1 == threads ? n : (n / threads * (t + 1)) & ~(size_t)4095; f(a, b); });
    for (auto& t : th) t.join();
    return now() - t0;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* path = argc > 1 ? argv[1] : "/tmp/hostmem.bin";
    {
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char l[128] = "";
        if (f) { fgets(l, sizeof l, f); fclose(f); } printf("thp: %s", l);
    }
    const size_t n = 512ull << 20;
    for (int huge = 0; huge < 2; ++huge)
        for (int mode = 0; mode < 2; ++mode)
            for (int threads : {1, 2, 4, 8}) {
                char* raw = (char*)mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
                char* p = (char*)(((uintptr_t)raw + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
                if (huge && madvise(p, n, MADV_HUGEPAGE)) printf("MADV_HUGEPAGE: %s\n", strerror(errno));
                double dt = par(threads, n, [=](size_t a, size_t b) {
                    if (mode) { if (madvise(p + a, b - a, MADV_POPULATE_WRITE)) printf("populate: %s\n", strerror(errno)); }
                    else for (size_t i = a; i < b; i += 4096) p[i] = 1;
                });
                printf("fresh 512 MiB  %-8s thp=%d threads=%d: %7.1f ms = %6.2f GB/s\n", mode ? "populate" : "touch", huge, threads, dt * 1e3, n / dt / 1e9);
                munmap(raw, n + (2 << 20));
            }
    // (b) file -> buffer
    const size_t fn = 1ull << 30;
    {
        std::vector<char> blk(8 << 20);
        for (size_t i = 0; i < blk.size(); ++i) blk[i] = (char)(i * 2654435761u >> 13);
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fd < 0) { perror("open"); return 1; }
        for (size_t off = 0; off < fn; off += blk.size()) if (write(fd, blk.data(), blk.size()) != (ssize_t)blk.size()) { perror("write"); return 1; }
        close(fd);
    }
    char* dst = nullptr; bool pinned = false;
    int ndev = 0; if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc((void**)&dst, fn, hipHostMallocDefault) == hipSuccess) pinned = true;
    else { dst = (char*)malloc(fn); memset(dst, 1, fn); }
    printf("destination: %s\n", pinned ? "pinned (hipHostMalloc)" : "malloc (no GPU)");
    int fd = open(path, O_RDONLY);
    for (int rep = 0; rep < 2; ++rep)
        for (int threads : {1, 2, 4, 8, 12}) {
            double dt = par(threads, fn, [=](size_t a, size_t b) {
                for (size_t off = a; off < b;) { ssize_t r = pread(fd, dst + off, std::min<size_t>(b - off, 4 << 20), off); if (r <= 0) { perror("pread"); return; } off += r; }
            });
            printf("pread 1 GiB -> dst          threads=%2d: %7.1f ms = %6.2f GB/s\n", threads, dt * 1e3, fn / dt / 1e9);
        }
    for (int pop = 0; pop < 3; ++pop)
        for (int threads : {1, 4, 8, 12}) {
            double t0 = now();
            char* m = (char*)mmap(nullptr, fn, PROT_READ, MAP_PRIVATE | (pop == 1 ? MAP_POPULATE : 0), fd, 0);
            double t_map = now() - t0;
            double t_pop = 0;
            if (pop == 2) t_pop = par(threads, fn, [=](size_t a, size_t b) { if (madvise(m + a, b - a, MADV_POPULATE_READ)) printf("populate_read: %s\n", strerror(errno)); });
            double dt = par(threads, fn, [=](size_t a, size_t b) { stream_copy(dst + a, m + a, b - a); });
            printf("mmap(%s) %5.1f ms + populate %6.1f ms + stream copy threads=%2d: %7.1f ms = %6.2f GB/s (all: %.1f ms)\n",
                   pop == 1 ? "MAP_POPULATE" : "lazy", t_map * 1e3, t_pop * 1e3, threads, dt * 1e3, fn / dt / 1e9, (t_map + t_pop + dt) * 1e3);
            munmap(m, fn);
        }
    close(fd);
    unlink(path);
    return 0;
}
