// dev tool (round 5): what the host-side steps of a mid-size ingest call cost on THIS box.
//   enqueue cost of hipMemcpyAsync H2D (64 KiB / 1 MiB, pinned), hipEventRecord, hipStreamWaitEvent, a tiny kernel launch;
//   latency kernel-writes-flag-to-pinned-memory -> host poll sees it, vs hipStreamSynchronize, vs hipEventSynchronize;
//   H2D time of 256 KiB .. 8 MiB (events); a kernel READING pinned host memory directly (zero copy) at 1 .. 8 MiB.
// build: hipcc -O2 --offload-arch=gfx950 -o api_cost api_cost.cpp
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void flag_kernel(volatile unsigned long long* flag, unsigned long long v) {
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
// each workgroup reads a contiguous 256-KiB piece with 16-byte loads, sums into out (so the loads are not dropped)
__global__ __launch_bounds__(512) void read_kernel(const uint4* __restrict__ in, size_t n16, unsigned* out) {
    unsigned acc = 0;
    const size_t per = (256u << 10) / 16;
    for (size_t i = (size_t)blockIdx.x * per + threadIdx.x; i < (size_t)(blockIdx.x + 1) * per && i < n16; i += blockDim.x) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
template <class F> static double per_call_us(int reps, F f) {
    const double t0 = now();
    for (int i = 0; i < reps; ++i) f(i);
    return (now() - t0) / reps * 1e6;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(0));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *pin, *dev;
    const size_t cap = 64u << 20;
    CK(hipHostMalloc((void**)&pin, cap, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, cap));
    memset(pin, 1, cap);
    unsigned long long* flag;
    CK(hipHostMalloc((void**)&flag, 4096, hipHostMallocDefault));
    *flag = 0;
    unsigned long long* dflag;
    CK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
    hipEvent_t ev[64];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int* dint;
    CK(hipMalloc((void**)&dint, 64));
    // warm
    for (int i = 0; i < 50; ++i) { CK(hipMemcpyAsync(dev, pin, 1 << 20, hipMemcpyHostToDevice, s1)); empty_kernel<<<1, 64, 0, s2>>>(dint); }
    CK(hipDeviceSynchronize());
    for (size_t sz : {(size_t)64 << 10, (size_t)1 << 20}) {
        double us = per_call_us(200, [&](int i) { CK(hipMemcpyAsync(dev + (size_t)(i & 15) * sz, pin + (size_t)(i & 15) * sz, sz, hipMemcpyHostToDevice, s1)); });
        CK(hipStreamSynchronize(s1));
        printf("enqueue hipMemcpyAsync H2D %4zu KiB: %.2f us per call (200 back to back, includes any back-pressure)\n", sz >> 10, us);
    }
    {
        double us = per_call_us(500, [&](int i) { CK(hipEventRecord(ev[i & 63], s1)); });
        CK(hipStreamSynchronize(s1));
        printf("enqueue hipEventRecord: %.2f us\n", us);
        us = per_call_us(500, [&](int i) { CK(hipStreamWaitEvent(s2, ev[i & 63], 0)); });
        CK(hipStreamSynchronize(s2));
        printf("enqueue hipStreamWaitEvent: %.2f us\n", us);
        us = per_call_us(500, [&](int) { empty_kernel<<<1, 64, 0, s2>>>(dint); });
        CK(hipStreamSynchronize(s2));
        printf("enqueue tiny kernel: %.2f us\n", us);
        void* dp;
        us = per_call_us(500, [&](int) { CK(hipHostGetDevicePointer(&dp, pin, 0)); });
        printf("hipHostGetDevicePointer: %.2f us\n", us);
        us = per_call_us(200, [&](int) { (void)hipStreamQuery(s2); });
        printf("hipStreamQuery (idle stream): %.2f us\n", us);
    }
    // a chunk's worth of calls: memcpy + record + wait + 2 kernels
    {
        double us = per_call_us(100, [&](int i) {
            CK(hipMemcpyAsync(dev, pin, 256 << 10, hipMemcpyHostToDevice, s1));
            CK(hipEventRecord(ev[i & 63], s1));
            CK(hipStreamWaitEvent(s2, ev[i & 63], 0));
            empty_kernel<<<4, 512, 0, s2>>>(dint);
            empty_kernel<<<1, 256, 0, s2>>>(dint);
        });
        CK(hipDeviceSynchronize());
        printf("one chunk's five calls (256-KiB copy, record, wait, two launches): %.2f us\n", us);
    }
    // latency: launch -> flag visible to a polling host
    {
        std::vector<double> ts;
        for (int i = 1; i <= 200; ++i) {
            const double t0 = now();
            flag_kernel<<<1, 64, 0, s2>>>(dflag, (unsigned long long)i);
            const double t1 = now();
            while (*(volatile unsigned long long*)flag != (unsigned long long)i) _mm_pause();
            const double t2 = now();
            if (i > 20) ts.push_back((t2 - t0) * 1e6);
            (void)t1;
        }
        double mn = 1e9, sum = 0; for (double t : ts) { mn = std::min(mn, t); sum += t; }
        printf("launch tiny kernel -> its flag seen by a polling host: min %.2f avg %.2f us\n", mn, sum / ts.size());
        ts.clear();
        for (int i = 0; i < 200; ++i) {
            const double t0 = now();
            empty_kernel<<<1, 64, 0, s2>>>(dint);
            CK(hipStreamSynchronize(s2));
            if (i > 20) ts.push_back((now() - t0) * 1e6);
        }
        mn = 1e9; sum = 0; for (double t : ts) { mn = std::min(mn, t); sum += t; }
        printf("launch tiny kernel + hipStreamSynchronize: min %.2f avg %.2f us\n", mn, sum / ts.size());
        ts.clear();
        for (int i = 0; i < 200; ++i) {
            const double t0 = now();
            empty_kernel<<<1, 64, 0, s2>>>(dint);
            CK(hipEventRecord(ev[0], s2));
            CK(hipEventSynchronize(ev[0]));
            if (i > 20) ts.push_back((now() - t0) * 1e6);
        }
        mn = 1e9; sum = 0; for (double t : ts) { mn = std::min(mn, t); sum += t; }
        printf("launch tiny kernel + record + hipEventSynchronize: min %.2f avg %.2f us\n", mn, sum / ts.size());
    }
    // H2D by size: host wall time from enqueue to a flag kernel on the same stream being seen (best of 30)
    for (size_t sz : {(size_t)256 << 10, (size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20}) {
        double best = 1e9;
        static unsigned long long seq = 1000;
        for (int r = 0; r < 30; ++r) {
            const double t0 = now();
            CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1));
            flag_kernel<<<1, 64, 0, s1>>>(dflag, ++seq);
            while (*(volatile unsigned long long*)flag != seq) _mm_pause();
            best = std::min(best, (now() - t0) * 1e6);
        }
        printf("H2D %5zu KiB + flag kernel, enqueue -> seen: %.1f us  (%.1f GiB/s incl. everything)\n", sz >> 10, best, sz / (best * 1e-6) / (1u << 30));
    }
    // two copies on two streams vs one copy of the sum
    {
        static unsigned long long seq = 5000;
        for (size_t sz : {(size_t)1 << 20, (size_t)2 << 20}) {
            double best = 1e9;
            for (int r = 0; r < 30; ++r) {
                const double t0 = now();
                CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1));
                CK(hipMemcpyAsync(dev + sz, pin + sz, sz, hipMemcpyHostToDevice, s2));
                CK(hipEventRecord(ev[1], s2));
                CK(hipStreamWaitEvent(s1, ev[1], 0));
                flag_kernel<<<1, 64, 0, s1>>>(dflag, ++seq);
                while (*(volatile unsigned long long*)flag != seq) _mm_pause();
                best = std::min(best, (now() - t0) * 1e6);
            }
            printf("2 x %zu KiB on two streams + flag: %.1f us\n", sz >> 10, best);
        }
    }
    // zero-copy read of pinned memory by a kernel
    {
        uint4* dpin;
        CK(hipHostGetDevicePointer((void**)&dpin, pin, 0));
        unsigned* dout;
        CK(hipMalloc((void**)&dout, 64));
        static unsigned long long seq = 9000;
        for (size_t sz : {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
            double best = 1e9;
            const unsigned wgs = (unsigned)(sz / (256u << 10));
            for (int r = 0; r < 30; ++r) {
                const double t0 = now();
                read_kernel<<<wgs, 512, 0, s1>>>(dpin, sz / 16, dout);
                flag_kernel<<<1, 64, 0, s1>>>(dflag, ++seq);
                while (*(volatile unsigned long long*)flag != seq) _mm_pause();
                best = std::min(best, (now() - t0) * 1e6);
            }
            printf("kernel reads %5zu KiB of pinned host memory (%u workgroups) + flag, enqueue -> seen: %.1f us (%.1f GiB/s)\n", sz >> 10, wgs, best,
                   sz / (best * 1e-6) / (1u << 30));
        }
    }
    return 0;
}
