// capi.cpp — the C ABI of libcsvsimd_hip.so (include/csvsimd.h): context, stage-1 entry points,
// shard stitch, tape accessors, csv_simd::create().  Compiled with hipcc (HIP runtime API only;
// the kernels live in stage1_kernels.hip).  There is no CPU fallback anywhere in this file.
#include <fcntl.h>
#include <sched.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../host/csv_simd.hpp"
#include "../host/ingest_pool.hpp"
#include "abi_guard.h"
#include "csvsimd.h"
#include "stage1_kernels.h"

namespace {

thread_local std::string g_last_error;
}  // namespace
void csvsimd_set_last_error_noexcept(const char* what) noexcept {
    try {
        g_last_error = what ? what : "";
    } catch (...) {  // not even the message could be stored: the error code still tells the caller
    }
}
namespace {

int fail_hip(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return CSVSIMD_ERR_HIP;
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return fail_hip(_e, #expr);     \
    } while (0)

// The calling thread's current HIP device is the CALLER's state (torch.cuda.current_device() in a torch process): an
// entry point that needs another device current — a context's — switches for its own duration only and restores the
// caller's on every exit path.  Nothing happens when the device is already the right one (two thread-local reads).
class ScopedDevice {
public:
    explicit ScopedDevice(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != device) {
            err_ = hipSetDevice(device);
            if (err_ == hipSuccess) prev_ = cur;
        }
    }
    ~ScopedDevice() {
        if (prev_ >= 0) (void)hipSetDevice(prev_);
    }
    ScopedDevice(const ScopedDevice&) = delete;
    ScopedDevice& operator=(const ScopedDevice&) = delete;
    hipError_t error() const { return err_; }

private:
    int prev_ = -1;
    hipError_t err_ = hipSuccess;
};
#define WITH_DEVICE_OF(ctx_)                      \
    ScopedDevice scoped_device_((ctx_)->device);  \
    HIP_TRY(scoped_device_.error())

// (streaming copies, CopyPool, TaskThread: ../host/ingest_pool.hpp — plain C++, stress-tested under ThreadSanitizer)
using csvsimd_host::CopyPool;
using csvsimd_host::TaskThread;

int ingest_workers() {
    const char* e = getenv("CSVSIMD_INGEST_THREADS");
    if (e && *e) {
        const int n = atoi(e);
        return std::min(std::max(n, 1), 32) - 1;  // the calling thread is one of them
    }
    // Measured on the MI355X host (scripts/probe_ingest.py, 2 GiB, GiB/s PCIe-inclusive by copying threads):
    // 1: 20.5, 4: 41, 6: 45, 8: 50.4, 10: 47-50, 12: 41 (the container's CPU share is 16 of the host's 256 threads,
    // which neither hardware_concurrency nor the affinity mask reveals: both say 256) -> 8, fewer on small machines.
    unsigned cpus = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = (unsigned)CPU_COUNT(&set);
    return (int)std::min<unsigned>(std::max<unsigned>(cpus / 2, 2), 8) - 1;  // 2..8 copying threads, caller included
}

}  // namespace

struct csvsimd_ctx {
    int device = 0;
    void* scratch = nullptr;
    uint64_t scratch_bytes = 0;
    // entries per input byte of the data this context indexes, as far as the host knows (< 0: not known): what the
    // synchronous entry points and the ingest pipeline saw last, or what the caller said (csvsimd_ctx_hint_density).
    // Above kDenseThreshold an emitting launch of the reference dialect runs the instantiation whose emit path is built
    // for many entries per byte (stage1_kernels.hip: DENSE) — same tape, bit for bit; only the instruction mix differs.
    double density = -1.0;
    static constexpr double kDenseThreshold = 0.125;  // profiles/r04_density_threshold.txt: the two instantiations cross between 0.111 and 0.143
    hipStream_t last_stream = nullptr;  // stream of the most recent launch that used the scratch
    bool launched = false;
    uint32_t max_blocks = 0;
    int n_cus = 0;
    csvsimd_shard_result* d_result = nullptr;  // for the synchronous entry points
    void* d_small = nullptr;                   // 8 KiB: [0, 16) match / truncation counters, [16, 24) the columnar search's packed word (zero between calls), [64, 328) search needle, [512, 544)
                                               // status of the synchronous frequency count, [1024, 5120) field list of
                                               // csvsimd_chunk_to_columns_device
    void* h_small = nullptr;                   // 256 B pinned: where the synchronous consumers' few result words land
    uint64_t search_seq = 0;                   // calls of the columnar search (the number its kernel publishes)
    // host-buffer path (csvsimd_stage1_index): kSlots-slot pipeline; every slot is allocated when a call first needs it
    static constexpr uint64_t kChunk = 32ull << 20;  // bytes per slot
    static constexpr int kSlots = 4;
    hipStream_t pipe_stream = nullptr;         // kernels + result records
    hipStream_t in_stream = nullptr;           // H2D of input chunks (runs ahead of the kernels)
    hipStream_t in_stream2 = nullptr;          // ... of every other chunk (host-buffer pipeline; see stage1_index_host_body)
    hipEvent_t ev_in[kSlots] = {};             // chunk has landed in d_in[k]
    void* pin_in[kSlots] = {};                 // pinned staging of the input chunk
    void* d_in[kSlots] = {};
    uint64_t* d_tape[kSlots] = {};             // device tape of a chunk (grown on demand)
    uint64_t d_tape_entries[kSlots] = {};
    uint32_t* pin_out[kSlots] = {};            // pinned slot of a chunk's tape on its way back: 32-bit chunk-relative offsets
    uint64_t pin_out_entries[kSlots] = {};
    uint64_t slot_bytes[kSlots] = {};          // size of pin_in[k] / d_in[k]
    csvsimd_shard_result* d_res[kSlots] = {};
    // how a chunk's record reaches the host (round 5): the kernel that packs the chunk's tape into the pinned slot also
    // copies the record into pinned memory and then writes the chunk's sequence number next to it; the submitter polls
    // that word (a copy-out + event cost ~25 us of latency per chunk, the whole budget of a 1-MiB chunk)
    struct alignas(128) HostRecord {
        csvsimd_shard_result rec;
        volatile uint64_t seq;
    };
    HostRecord* h_res = nullptr;               // pinned, kSlots records
    uint32_t* d_pub = nullptr;                 // kSlots arrival counters of the publishing kernel (wrap to 0 by themselves)
    uint64_t pub_seq = 0;                      // last sequence number handed to a chunk
    std::unique_ptr<TaskThread> stager_thread, expander_thread;  // started by the first call that pipelines, kept
    // csvsimd_stage1_index_batch (many small host files in one call): a group's tapes and result records come back through
    // this pinned block, written by the batched launch itself
    void* pin_bout[kSlots] = {};
    size_t pin_bout_bytes[kSlots] = {};
    // small files (<= kSmallBytes) through the host-buffer entry point: ONE launch that reads the bytes from a pinned block
    // and writes tape and record into another, both mapped into the GPU's address space — no copy engine, no second stream
    static constexpr uint64_t kSmallBytes = 1ull << 20;
    void* pin_small_in = nullptr;             // kSmallBytes + 64
    void* pin_small_out = nullptr;            // [0, 64): the launch's result record; [64, ...): its tape (u64)
    uint64_t pin_small_out_entries = 0;
    std::unique_ptr<CopyPool> copier;         // host-side slices of the staging copies
    void* d_batch = nullptr;                  // csvsimd_stage1_index_batch_device_async: the buffers' table block
    size_t d_batch_bytes = 0;
    // small host -> device uploads of the asynchronous entry points (a batch's table, a field list): the caller's memory
    // is copied into one of two pinned blocks before the call returns, the transfer itself is stream ordered
    void* pin_up[2] = {nullptr, nullptr};
    size_t pin_up_bytes[2] = {0, 0};
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    bool up_pending[2] = {false, false};
    unsigned up_next = 0;
};

extern "C" {

const char* csvsimd_strerror(int code) {
    switch (code) {
        case CSVSIMD_OK: return "ok";
        case CSVSIMD_ERR_IO: return "io error";
        case CSVSIMD_ERR_MISSING_VALUE: return "Missing a value";
        case CSVSIMD_ERR_INVALID_STATE: return "Invalid state";
        case CSVSIMD_ERR_INVALID_CSV_FORMAT: return "Unsupported csv structure: likely variable number of fields";
        case CSVSIMD_ERR_INVALID_ARG: return "invalid argument";
        case CSVSIMD_ERR_TAPE_CAPACITY: return "tape capacity too small";
        case CSVSIMD_ERR_HIP: return "HIP runtime error";
        case CSVSIMD_ERR_NO_DEVICE: return "no HIP device";
        case CSVSIMD_ERR_INTERNAL: return "internal error (look-back spin bound)";
        case CSVSIMD_ERR_RCCL: return "RCCL unavailable or collective failed";
        default: return "unknown error";
    }
}

const char* csvsimd_last_error(void) { return g_last_error.c_str(); }

int csvsimd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint32_t csvsimd_abi_version(void) { return 5; }
uint32_t csvsimd_tile_bytes(void) { return CSVSIMD_TILE_BYTES; }

int csvsimd_ctx_create(int device, csvsimd_ctx** out) {
    return csvsimd_guarded([&]() -> int {
    if (!out) return CSVSIMD_ERR_INVALID_ARG;
    *out = nullptr;
    const int n = csvsimd_device_count();
    if (n <= 0) {
        g_last_error = "no HIP device visible: libcsvsimd_hip has no CPU fallback";
        return CSVSIMD_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return CSVSIMD_ERR_INVALID_ARG;
    ScopedDevice scoped_device_(device);  // the caller's current device is restored on every way out
    HIP_TRY(scoped_device_.error());
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::unique_ptr<csvsimd_ctx> ctx(new (std::nothrow) csvsimd_ctx);
    if (!ctx) return CSVSIMD_ERR_INVALID_STATE;
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int per_cu = csvsimd::stage1_max_blocks_per_cu();
    ctx->max_blocks = (uint32_t)(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * (uint32_t)per_cu;
    HIP_TRY(hipMalloc((void**)&ctx->d_result, sizeof(csvsimd_shard_result)));
    HIP_TRY(hipMalloc(&ctx->d_small, 8192));
    HIP_TRY(hipMemset(ctx->d_small, 0, 8192));  // (the columnar search's words and arrival counter start, and are left, at zero)
    HIP_TRY(hipHostMalloc(&ctx->h_small, 256, hipHostMallocDefault));
    memset(ctx->h_small, 0, 256);
    *out = ctx.release();
    const int rc = csvsimd_ctx_reserve(*out, 1ull << 30);
    if (rc != CSVSIMD_OK) {
        csvsimd_ctx_destroy(*out);
        *out = nullptr;
    }
    return rc;
    });
}

void csvsimd_ctx_destroy(csvsimd_ctx* ctx) {
    if (!ctx) return;
    ScopedDevice scoped_device_(ctx->device);
    (void)hipDeviceSynchronize();  // nothing of this context may still be in flight
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->d_result) (void)hipFree(ctx->d_result);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    for (int k = 0; k < 2; ++k) {
        if (ctx->pin_up[k]) (void)hipHostFree(ctx->pin_up[k]);
        if (ctx->ev_up[k]) (void)hipEventDestroy(ctx->ev_up[k]);
    }
    for (int k = 0; k < csvsimd_ctx::kSlots; ++k) {
        if (ctx->pin_in[k]) (void)hipHostFree(ctx->pin_in[k]);
        if (ctx->pin_out[k]) (void)hipHostFree(ctx->pin_out[k]);
        if (ctx->pin_bout[k]) (void)hipHostFree(ctx->pin_bout[k]);
        if (ctx->d_in[k]) (void)hipFree(ctx->d_in[k]);
        if (ctx->d_tape[k]) (void)hipFree(ctx->d_tape[k]);
        if (ctx->d_res[k]) (void)hipFree(ctx->d_res[k]);
        if (ctx->ev_in[k]) (void)hipEventDestroy(ctx->ev_in[k]);
    }
    if (ctx->pin_small_in) (void)hipHostFree(ctx->pin_small_in);
    if (ctx->pin_small_out) (void)hipHostFree(ctx->pin_small_out);
    if (ctx->in_stream) (void)hipStreamDestroy(ctx->in_stream);
    if (ctx->in_stream2) (void)hipStreamDestroy(ctx->in_stream2);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    if (ctx->d_pub) (void)hipFree(ctx->d_pub);
    if (ctx->pipe_stream) (void)hipStreamDestroy(ctx->pipe_stream);
    delete ctx;
}

int csvsimd_ctx_reserve(csvsimd_ctx* ctx, uint64_t max_len) {
    if (!ctx) return CSVSIMD_ERR_INVALID_ARG;
    const uint64_t need = csvsimd::Stage1Launch::scratch_bytes_for(max_len);
    if (need <= ctx->scratch_bytes) return CSVSIMD_OK;
    WITH_DEVICE_OF(ctx);
    // every launch that may still use the old block has to finish first.  The stream of the latest launch is remembered,
    // but the caller may have destroyed it since (a dangling handle is not a guaranteed error), so this rare path
    // (a context is asked for a larger shard than ever before) waits for the whole device
    if (ctx->launched) HIP_TRY(hipDeviceSynchronize());
    if (ctx->scratch) HIP_TRY(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->scratch, need));
    // the only time the block is ever cleared: control words 0, every descriptor "not published"; from
    // here on each launch leaves it ready for the next (stage1_kernels.hip: struct Control)
    HIP_TRY(hipMemset(ctx->scratch, 0, need));
    ctx->scratch_bytes = need;
    ctx->launched = false;
    return CSVSIMD_OK;
}

// Stream-ordered upload of a few KiB of the CALLER's (pageable, possibly temporary) memory: staged in a context-owned
// pinned block, so nothing of the caller's is referenced once this returns and the copy engine never reads pageable
// memory.  Two blocks alternate: the host may run two asynchronous calls ahead of the device.
static int ctx_upload(csvsimd_ctx* ctx, void* d_dst, const void* src, size_t n, hipStream_t s) {
    const unsigned k = ctx->up_next++ & 1u;
    if (!ctx->ev_up[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
    if (ctx->up_pending[k]) HIP_TRY(hipEventSynchronize(ctx->ev_up[k]));  // the block's previous upload has left it
    ctx->up_pending[k] = false;
    if (ctx->pin_up_bytes[k] < n) {
        if (ctx->pin_up[k]) HIP_TRY(hipHostFree(ctx->pin_up[k]));
        ctx->pin_up[k] = nullptr;
        ctx->pin_up_bytes[k] = 0;
        const size_t cap = std::max<size_t>(65536, n * 2);
        HIP_TRY(hipHostMalloc(&ctx->pin_up[k], cap, hipHostMallocDefault));
        ctx->pin_up_bytes[k] = cap;
    }
    memcpy(ctx->pin_up[k], src, n);
    HIP_TRY(hipMemcpyAsync(d_dst, ctx->pin_up[k], n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(ctx->ev_up[k], s));
    ctx->up_pending[k] = true;
    return CSVSIMD_OK;
}

// dialect == nullptr: the reference's hard-wired dialect (',' '"', no escape byte)
static int dialect_check(const csvsimd_dialect* d) {
    if (!d) return CSVSIMD_OK;
    const uint8_t dl = d->delimiter, q = d->quote, e = d->escape;
    if (dl == 0 || dl == 0x0a || dl == 0x0d) return CSVSIMD_ERR_INVALID_ARG;
    if (q && (q == dl || q == 0x0a || q == 0x0d)) return CSVSIMD_ERR_INVALID_ARG;
    if (e && (e == dl || e == q || e == 0x0a || e == 0x0d)) return CSVSIMD_ERR_INVALID_ARG;
    return CSVSIMD_OK;
}

static int stage1_async_impl(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const void* dbuf, uint64_t len,
                             uint64_t base_off, uint32_t in_quote_in, void* dtape, uint64_t tape_cap, void* d_result,
                             void* hip_stream, const uint32_t* d_state = nullptr,
                             const csvsimd_shard_result* d_chain = nullptr, bool short_launch = false) {
    if (!ctx || !d_result || (len && !dbuf) || (!dtape && tape_cap)) return CSVSIMD_ERR_INVALID_ARG;
    // a shard's entry count must fit the 39-bit field of a look-back word (288 GB of HBM is 2^38.1 bytes)
    if (len >= (1ull << 39)) return CSVSIMD_ERR_INVALID_ARG;
    if (((uintptr_t)dtape & 7) || ((uintptr_t)d_result & 15)) return CSVSIMD_ERR_INVALID_ARG;
    if (dialect_check(dialect) != CSVSIMD_OK || in_quote_in > CSVSIMD_ENTER_GUESS) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);  // hip_stream must belong to the context's device; the caller's current device may be another
    if (csvsimd::Stage1Launch::scratch_bytes_for(len) > ctx->scratch_bytes) {
        const int rc = csvsimd_ctx_reserve(ctx, len);  // allocates + synchronises: not capturable
        if (rc != CSVSIMD_OK) return rc;
    }
    csvsimd::Stage1Launch L;
    L.dbuf = dbuf;
    L.len = len;
    L.base_off = base_off;
    L.in_quote_in = in_quote_in;  // 0, 1 or CSVSIMD_ENTER_GUESS
    L.dtape = dtape;
    L.tape_cap = tape_cap;
    L.d_result = (csvsimd_shard_result*)d_result;
    L.bind_scratch(ctx->scratch);
    L.max_blocks = ctx->max_blocks;
    L.d_state = d_state;
    L.d_chain = d_chain;
    // (the dense instantiation exists for emitting launches without an escape byte: the reference dialect, another
    // delimiter / quote byte)
    L.dense = (!dialect || !dialect->escape) && dtape && ctx->density > csvsimd_ctx::kDenseThreshold;
    // The host-buffer paths' launches over a few MiB (a small file, a chunk of a mid-size one) run the 64-KiB-tile geometry
    // whatever the density: such a launch is all fill and drain (~21 us for 1 ... 8 MiB in the default geometry), and the
    // shorter tiles shorten exactly that (a 4-MiB call: 178 -> 166 us, 2 MiB: 126 -> 115: profiles/r05_midsize_after.json).
    if (short_launch && len <= (4ull << 20) && (!dialect || !dialect->escape) && dtape) L.dense = true;
    if (dialect) {
        L.delimiter = dialect->delimiter;
        L.quote = dialect->quote;
        L.escape = dialect->escape;
        L.escape_in = dialect->escape_in ? 1u : 0u;
    }
    ctx->last_stream = (hipStream_t)hip_stream;
    ctx->launched = true;
    HIP_TRY(csvsimd::launch_stage1(L, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

int csvsimd_stage1_index_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, uint64_t base_off,
                                      uint32_t in_quote_in, void* dtape, uint64_t tape_cap, void* d_result,
                                      void* hip_stream) {
    return stage1_async_impl(ctx, nullptr, dbuf, len, base_off, in_quote_in, dtape, tape_cap, d_result, hip_stream);
}

int csvsimd_stage1_reemit_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, uint64_t base_off,
                                       const void* d_stitch, void* dtape, uint64_t tape_cap, void* d_result,
                                       void* hip_stream) {
    if (!d_stitch || ((uintptr_t)d_stitch & 7)) return CSVSIMD_ERR_INVALID_ARG;
    static_assert(offsetof(csvsimd_stitch, in_quote_in) == 0, "the re-emit launch reads the stitch record's first word");
    return stage1_async_impl(ctx, nullptr, dbuf, len, base_off, 1, dtape, tape_cap, d_result, hip_stream,
                             (const uint32_t*)d_stitch);
}

int csvsimd_stitch_shards_device_async(const void* d_results, uint32_t n_shards, uint32_t rank,
                                       uint32_t file_in_quote_in, void* d_stitch, void* hip_stream) {
    if (!d_results || !d_stitch || rank >= n_shards || ((uintptr_t)d_results & 7) || ((uintptr_t)d_stitch & 7))
        return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    HIP_TRY(csvsimd::launch_stitch(d_results, n_shards, rank, file_in_quote_in, d_stitch, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

/* ---- K independent buffers in ONE persistent launch ------------------------------------------------------------ */
int csvsimd_stage1_index_batch_device_async(csvsimd_ctx* ctx, const csvsimd_batch_item* items, uint32_t n_items,
                                            void* d_results, void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !items || n_items == 0 || n_items > 65536 || !d_results || ((uintptr_t)d_results & 15))
        return CSVSIMD_ERR_INVALID_ARG;
    // one block: [n items x 64 B | n first tiles x 4 B, padded to 64 | n totals x 8 B (zero)]
    const size_t off_first = (size_t)n_items * sizeof(csvsimd::BatchItemHost);
    const size_t off_tot = off_first + (((size_t)n_items * 4 + 63) & ~(size_t)63);
    const size_t block = off_tot + (size_t)n_items * 8;
    std::vector<unsigned char> host(block, 0);
    csvsimd::BatchItemHost* const table = reinterpret_cast<csvsimd::BatchItemHost*>(host.data());
    uint32_t* const firsts = reinterpret_cast<uint32_t*>(host.data() + off_first);
    // delimiter-dense data (as far as this context knows): the batch runs the dense instantiation, whose tiles are 64 KiB
    const bool dense = ctx->density > csvsimd_ctx::kDenseThreshold;
    const uint64_t tile_bytes = dense ? CSVSIMD_MIN_TILE_BYTES : CSVSIMD_TILE_BYTES;
    uint64_t tiles = 0;
    for (uint32_t i = 0; i < n_items; ++i) {
        const csvsimd_batch_item& it = items[i];
        if ((it.len && !it.dbuf) || (!it.dtape && it.tape_cap) || ((uintptr_t)it.dtape & 7) || it.in_quote_in > 1 ||
            it.len >= (1ull << 39))
            return CSVSIMD_ERR_INVALID_ARG;
        const uintptr_t addr = (uintptr_t)it.dbuf;
        csvsimd::BatchItemHost& t = table[i];
        t.abase = (const void*)(addr & ~(uintptr_t)15);
        t.lo = addr & 15;
        t.hi = t.lo + it.len;
        t.base_off = it.base_off;
        t.tape = it.dtape;
        t.tape_cap = it.dtape ? it.tape_cap : 0;
        t.first_tile = firsts[i] = (uint32_t)tiles;
        t.in_quote_in = it.in_quote_in;
        t.reserved = 0;
        tiles += it.len ? (t.hi + tile_bytes - 1) / tile_bytes : 0;
        if (tiles >= (1ull << 31)) return CSVSIMD_ERR_INVALID_ARG;
    }
    WITH_DEVICE_OF(ctx);
    // scratch: one descriptor word per tile of the whole batch (allocates + synchronises only when it has to grow)
    int rc = csvsimd_ctx_reserve(ctx, tiles * tile_bytes);
    if (rc != CSVSIMD_OK) return rc;
    if (ctx->d_batch_bytes < block) {
        if (ctx->launched) HIP_TRY(hipDeviceSynchronize());  // an earlier batch may still be reading the old table
        if (ctx->d_batch) HIP_TRY(hipFree(ctx->d_batch));
        ctx->d_batch = nullptr;
        ctx->d_batch_bytes = 0;
        const size_t cap = std::max<size_t>(8192, block * 2);
        HIP_TRY(hipMalloc(&ctx->d_batch, cap));
        ctx->d_batch_bytes = cap;
    }
    hipStream_t s = (hipStream_t)hip_stream;
    // staged in the context's pinned block; the transfer is ordered on the stream behind whatever batch is still
    // running from the device table
    rc = ctx_upload(ctx, ctx->d_batch, host.data(), block, s);
    if (rc != CSVSIMD_OK) return rc;
    csvsimd::Stage1Launch L;
    L.bind_scratch(ctx->scratch);
    ctx->last_stream = s;
    ctx->launched = true;
    HIP_TRY((dense ? csvsimd_dense::launch_stage1_batch_dense : csvsimd::launch_stage1_batch)(
        ctx->d_batch, (char*)ctx->d_batch + off_first, (char*)ctx->d_batch + off_tot, n_items, (uint32_t)tiles,
        (csvsimd_shard_result*)d_results, ctx->scratch, L.scratch_desc, ctx->max_blocks, s));
    return CSVSIMD_OK;
    });
}

int csvsimd_dialect_init(csvsimd_dialect* d) {
    if (!d) return CSVSIMD_ERR_INVALID_ARG;
    memset(d, 0, sizeof(*d));
    d->delimiter = ',';
    d->quote = '"';
    return CSVSIMD_OK;
}

int csvsimd_stage1_index_device_dialect_async(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const void* dbuf,
                                              uint64_t len, uint64_t base_off, uint32_t in_quote_in, void* dtape,
                                              uint64_t tape_cap, void* d_result, void* hip_stream) {
    if (!dialect) return CSVSIMD_ERR_INVALID_ARG;
    return stage1_async_impl(ctx, dialect, dbuf, len, base_off, in_quote_in, dtape, tape_cap, d_result, hip_stream);
}

int csvsimd_stage1_index_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, uint64_t base_off,
                                uint32_t in_quote_in, void* dtape, uint64_t tape_cap,
                                csvsimd_shard_result* result, void* hip_stream) {
    if (!ctx || !result) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    // A context that knows nothing about its data yet (no earlier synchronous call, no ingest, no hint) looks before it
    // chooses the instantiation: sixteen 64-KiB windows of the buffer, bytes equal to ',', CR or LF (text_kernels.hip).
    // ~20 us, once per context; buffers below 8 MiB are not worth it (the instantiations differ by microseconds there).
    // The asynchronous entry point never does this (it may not synchronise): it runs the default geometry until told or
    // until a synchronous call / an ingest on the same context has seen the data.
    if (ctx->density < 0 && dbuf && dtape && len >= (8ull << 20)) {
        HIP_TRY(csvsimd::launch_density_sample(dbuf, len, ',', ctx->d_small, (hipStream_t)hip_stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_small, ctx->d_small, 16, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
        const uint64_t* s2 = (const uint64_t*)ctx->h_small;
        if (s2[1]) ctx->density = (double)s2[0] / (double)s2[1];
    }
    int rc = csvsimd_stage1_index_device_async(ctx, dbuf, len, base_off, in_quote_in, dtape, tape_cap,
                                               ctx->d_result, hip_stream);
    if (rc != CSVSIMD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(result, ctx->d_result, sizeof(*result), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    if (result->error) {
        g_last_error = "stage1 kernel: look-back spin bound hit";
        return CSVSIMD_ERR_INTERNAL;
    }
    // what this data looks like, for the next launch's choice of instantiation (a guessed entering state may have counted
    // the wrong hypothesis: the larger of the two is what the text holds)
    if (len >= (1u << 16))
        ctx->density = (double)std::max(result->count_enter_outside, result->count_enter_inside) / (double)len;
    if (dtape && result->count > tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
}

int csvsimd_ctx_hint_density(csvsimd_ctx* ctx, uint64_t entries, uint64_t bytes) {
    if (!ctx) return CSVSIMD_ERR_INVALID_ARG;
    ctx->density = bytes ? (double)entries / (double)bytes : -1.0;
    return CSVSIMD_OK;
}

int csvsimd_ctx_limit_workgroups(csvsimd_ctx* ctx, uint32_t n) {
    if (!ctx) return CSVSIMD_ERR_INVALID_ARG;
    const uint32_t full = (uint32_t)ctx->n_cus * (uint32_t)csvsimd::stage1_max_blocks_per_cu();
    ctx->max_blocks = n ? std::min(n, full) : full;
    return CSVSIMD_OK;
}

const char* csvsimd_ctx_kernel_name(const csvsimd_ctx* ctx, const csvsimd_dialect* dialect) {
    if (!ctx) return "";
    const bool dense = ctx->density > csvsimd_ctx::kDenseThreshold;
    if (!dialect || (dialect->delimiter == ',' && dialect->quote == '"' && !dialect->escape))
        return csvsimd::stage1_kernel_name(true, 0, dense);
    if (!dialect->escape && dense) return csvsimd::stage1_kernel_name(true, 1, true);
    return csvsimd_stage1_kernel_name(1, dialect);
}

int csvsimd_stage1_bound(uint64_t len, uint64_t* max_entries) {
    if (!max_entries) return CSVSIMD_ERR_INVALID_ARG;
    *max_entries = len + 1;  // every byte structural + the sentinel
    return CSVSIMD_OK;
}

// Host-buffer drop-in for reader::read (ingest).  The file is streamed through the GPU in chunks
// over up to four slots on private streams (H2D, two taking turns; kernels): a stager thread copies chunks into pinned
// slots ahead of the H2D copies, the caller enqueues copy + kernels per chunk and reads records two chunks late, an
// expander thread widens the offsets that came back into the caller's tape (stage1_index_host_body).  This path is
// PCIe bound by construction; the HBM-resident entry points are the timed ones.  Slots are allocated when first needed.
static int pipe_setup(csvsimd_ctx* ctx, int slots = 2, uint64_t slot_bytes = csvsimd_ctx::kChunk) {
    // everything is created when it is first needed and kept: a previous attempt may have failed half way (out of
    // memory), a small file needs small slots, a large one all four at full size
    slots = std::min(std::max(slots, 1), csvsimd_ctx::kSlots);
    slot_bytes = std::min<uint64_t>(std::max<uint64_t>(slot_bytes, 1u << 16), csvsimd_ctx::kChunk);
    if (!ctx->copier) ctx->copier.reset(new CopyPool(ingest_workers()));
    if (!ctx->pipe_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->pipe_stream, hipStreamNonBlocking));
    if (!ctx->in_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->in_stream, hipStreamNonBlocking));
    if (!ctx->in_stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->in_stream2, hipStreamNonBlocking));
    if (!ctx->h_res) {
        HIP_TRY(hipHostMalloc((void**)&ctx->h_res, csvsimd_ctx::kSlots * sizeof(csvsimd_ctx::HostRecord), hipHostMallocDefault));
        memset((void*)ctx->h_res, 0, csvsimd_ctx::kSlots * sizeof(csvsimd_ctx::HostRecord));
    }
    if (!ctx->d_pub) {
        HIP_TRY(hipMalloc((void**)&ctx->d_pub, csvsimd_ctx::kSlots * 64));  // one counter per slot, a line apart
        HIP_TRY(hipMemset(ctx->d_pub, 0, csvsimd_ctx::kSlots * 64));
    }
    for (int k = 0; k < slots; ++k) {
        if (ctx->slot_bytes[k] < slot_bytes) {
            // (nothing of an earlier call is in flight: every host entry point drains its streams before it returns)
            if (ctx->pin_in[k]) HIP_TRY(hipHostFree(ctx->pin_in[k]));
            ctx->pin_in[k] = nullptr;
            if (ctx->d_in[k]) HIP_TRY(hipFree(ctx->d_in[k]));
            ctx->d_in[k] = nullptr;
            ctx->slot_bytes[k] = 0;
            // three size classes — 2, 8, 32 MiB — so that a caller whose files grow re-pins a slot twice at most (pinning
            // costs ~1 ms per 4 MiB; round 4 re-pinned for every new maximum)
            const uint64_t want = slot_bytes <= (2u << 20) ? (2u << 20) : slot_bytes <= (8u << 20) ? (8u << 20) : csvsimd_ctx::kChunk;
            HIP_TRY(hipHostMalloc(&ctx->pin_in[k], want, hipHostMallocDefault));
            HIP_TRY(hipMalloc(&ctx->d_in[k], want));
            ctx->slot_bytes[k] = want;
        }
        if (!ctx->d_res[k]) HIP_TRY(hipMalloc((void**)&ctx->d_res[k], sizeof(csvsimd_shard_result)));
        if (!ctx->ev_in[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_in[k], hipEventDisableTiming));
    }
    return CSVSIMD_OK;
}
// a slot's device tape and its pinned way back, both for `entries` entries; everything that touches them runs on
// pipe_stream (the stage-1 launch writes the device tape, the narrow kernel reads it and writes the pinned slot)
static int pipe_ensure_tape(csvsimd_ctx* ctx, int k, uint64_t entries) {
    if (ctx->d_tape_entries[k] >= entries && ctx->pin_out_entries[k] >= entries) return CSVSIMD_OK;
    HIP_TRY(hipStreamSynchronize(ctx->pipe_stream));
    if (ctx->d_tape_entries[k] < entries) {
        if (ctx->d_tape[k]) HIP_TRY(hipFree(ctx->d_tape[k]));
        ctx->d_tape[k] = nullptr;
        ctx->d_tape_entries[k] = 0;
        HIP_TRY(hipMalloc((void**)&ctx->d_tape[k], entries * 8));
        ctx->d_tape_entries[k] = entries;
    }
    if (ctx->pin_out_entries[k] < entries) {
        if (ctx->pin_out[k]) HIP_TRY(hipHostFree(ctx->pin_out[k]));
        ctx->pin_out[k] = nullptr;
        ctx->pin_out_entries[k] = 0;
        HIP_TRY(hipHostMalloc((void**)&ctx->pin_out[k], entries * 4 + 16, hipHostMallocDefault));
        ctx->pin_out_entries[k] = entries;
    }
    return CSVSIMD_OK;
}

static int stage1_index_host_body(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf, uint64_t len,
                                  uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out);

static int stage1_index_host_impl(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf, uint64_t len,
                                  uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out) {
    const int rc = stage1_index_host_body(ctx, dialect, buf, len, tape, tape_cap, tape_len, in_quote_out);
    if (rc != CSVSIMD_OK && rc != CSVSIMD_ERR_TAPE_CAPACITY && ctx && ctx->pipe_stream) {
        // an error exit in mid-pipeline leaves copies and kernels in flight on the pinned slots: drain the
        // streams so the next call (or the caller freeing `buf` / `tape`) cannot race them
        const std::string keep = g_last_error;
        ScopedDevice scoped_device_(ctx->device);
        if (ctx->in_stream) (void)hipStreamSynchronize(ctx->in_stream);
        if (ctx->in_stream2) (void)hipStreamSynchronize(ctx->in_stream2);
        (void)hipStreamSynchronize(ctx->pipe_stream);
        (void)hipGetLastError();
        g_last_error = keep;
    }
    return rc;
}

// The chunk plan of the ingest pipeline: chunk i = [cuts[i], cuts[i + 1]).  The slots hold up to 32 MiB.  A call's wall time
// is   staging of the FIRST chunk  +  the H2D copies of all chunks (the link: 53 GiB/s, ~8 us between copies)  +  kernel,
// way back and expansion of the LAST chunk — so the plan starts small, doubles up to full slots, and halves down again at
// the end:  len / 32 (1 ... 4 MiB)  x2 x2 ...  32 MiB ... 32 MiB  ... x1/2 x1/2  len / 16 (1 ... 8 MiB).
// 2 GiB: 4, 8, 16, 32 x 62, 16, 8 MiB.  32 MiB: 1, 2, 4, 8, 3, 8, 4, 2.  4 MiB: 1, 2, 1.  No chunk below 1 MiB: a copy has a
// fixed cost of ~18 us that only another copy's transfer can hide (1 MiB crosses the link in 18 us), and ENQUEUEING a copy
// below ~1 MiB blocks the caller (64 KiB: 44 us per call, 1 MiB: 1.4 us — profiles/r05_api_cost.txt).
// Round 4 cut a file below 128 MiB into four equal chunks of >= 4 MiB (a 32-MiB file waited 120 us for the staging of its
// first 8 MiB and 100 us for the way back of its last 8 MiB: 1.03 ms against 0.59 ms of link time) because a chunk cost the
// submitter ~100 us then; with the record published by the packing kernel and polled it is ~20 us (five calls).
// No stub: a remainder shorter than a quarter of the chunk before it is folded into that chunk.
static std::vector<uint64_t> ingest_chunk_plan(uint64_t len, uint64_t uniform_override) {
    constexpr uint64_t kMiB = 1ull << 20, kMax = csvsimd_ctx::kChunk;
    std::vector<uint64_t> cuts;
    cuts.push_back(0);
    if (len == 0) return cuts;
    if (uniform_override) {
        const uint64_t uniform = uniform_override, least = std::min<uint64_t>(4 * kMiB, uniform);
        for (uint64_t off = 0; off < len;) {
            const uint64_t rem = len - off;
            uint64_t sz = std::min(uniform, rem);
            if (rem > sz && rem - sz < least)  // a stub would be left over: one chunk if the slot holds it, else two halves
                sz = rem <= kMax ? rem : (((rem / 2) + kMiB - 1) >> 20) << 20;
            off += sz;
            cuts.push_back(off);
        }
        return cuts;
    }
    auto round64k = [](uint64_t v) { return (v + 65535) & ~(uint64_t)65535; };
    const uint64_t first = std::min(4 * kMiB, std::max(1 * kMiB, round64k(len / 32)));
    const uint64_t last = std::min(8 * kMiB, std::max(1 * kMiB, round64k(len / 16)));
    std::vector<uint64_t> front, back;  // sizes from the file's start / from its end
    uint64_t rem = len, f = std::min(first, kMax), b = std::min(last, kMax);
    for (bool at_front = true; rem; at_front = !at_front) {
        uint64_t& next = at_front ? f : b;
        std::vector<uint64_t>& side = at_front ? front : back;
        uint64_t sz = std::min(next, rem);
        // no stub (a remainder below a quarter of this chunk, or below 512 KiB: enqueueing a short copy blocks the caller): fold
        // it, or cut what is left in two
        if (rem - sz < std::max<uint64_t>(sz / 4, 512u << 10)) sz = rem <= kMax ? rem : round64k(rem / 2);
        side.push_back(sz);
        rem -= sz;
        next = std::min(next * 2, kMax);
    }
    uint64_t off = 0;
    for (uint64_t sz : front) cuts.push_back(off += sz);
    for (size_t i = back.size(); i-- > 0;) cuts.push_back(off += back[i]);
    return cuts;
}

extern "C" int csvsimd_ingest_chunk_plan(uint64_t len, uint64_t* cuts, uint64_t cap, uint64_t* n_cuts) {
    return csvsimd_guarded([&]() -> int {
    if (!n_cuts || (cap && !cuts)) return CSVSIMD_ERR_INVALID_ARG;
    const std::vector<uint64_t> plan = ingest_chunk_plan(len, 0);
    *n_cuts = plan.size();
    if (plan.size() > cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    std::copy(plan.begin(), plan.end(), cuts);
    return CSVSIMD_OK;
    });
}

// Phase times of the most recent csvsimd_stage1_index call on this thread (csvsimd_ingest_last_phases): where a call's
// wall time went, per thread of the pipeline.
namespace {
thread_local csvsimd_ingest_phases g_ingest_phases = {};
}
extern "C" int csvsimd_ingest_last_phases(csvsimd_ingest_phases* out) {
    if (!out) return CSVSIMD_ERR_INVALID_ARG;
    *out = g_ingest_phases;
    return CSVSIMD_OK;
}

// A file of at most kSmallBytes (the reference's own inputs are 96 to 623 bytes, res/*.csv): the pipeline above would
// spend its time in set-up.  Here: memcpy into a pinned block, ONE stage-1 launch whose input, tape and result record
// all live in pinned host memory that the GPU addresses directly, one wait, memcpy of the entries.  Nothing is copied
// by an engine, nothing crosses streams.  (PCIe reads at ~36 GiB/s: from a megabyte up the staged pipeline wins.)
static int stage1_index_host_small(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf, uint64_t len,
                                   uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out) {
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (!ctx->pipe_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->pipe_stream, hipStreamNonBlocking));
    if (!ctx->pin_small_in) HIP_TRY(hipHostMalloc(&ctx->pin_small_in, csvsimd_ctx::kSmallBytes + 64, hipHostMallocDefault));
    auto ensure_out = [&](uint64_t entries) -> int {
        if (ctx->pin_small_out && ctx->pin_small_out_entries >= entries) return CSVSIMD_OK;
        if (ctx->pin_small_out) HIP_TRY(hipHostFree(ctx->pin_small_out));
        ctx->pin_small_out = nullptr;
        ctx->pin_small_out_entries = 0;
        const uint64_t want = std::max<uint64_t>(entries, 8192);
        HIP_TRY(hipHostMalloc(&ctx->pin_small_out, 64 + want * 8, hipHostMallocDefault));
        ctx->pin_small_out_entries = want;
        return CSVSIMD_OK;
    };
    // first guess: an entry per 4 bytes (what fits is kept for the next call: a caller that reads many small files
    // allocates once)
    int rc = ensure_out(tape ? std::max<uint64_t>(len / 4, 64) : 0);
    if (rc != CSVSIMD_OK) return rc;
    if (len) memcpy(ctx->pin_small_in, buf, len);
    csvsimd_shard_result r;
    for (int attempt = 0;; ++attempt) {
        void *d_in = nullptr, *d_out = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d_in, ctx->pin_small_in, 0));
        HIP_TRY(hipHostGetDevicePointer(&d_out, ctx->pin_small_out, 0));
        const uint64_t cap = tape ? ctx->pin_small_out_entries : 0;
        rc = stage1_async_impl(ctx, dialect, d_in, len, 0, 0, tape ? (char*)d_out + 64 : nullptr, cap, d_out, ctx->pipe_stream, nullptr,
                               nullptr, /*short_launch=*/true);
        if (rc != CSVSIMD_OK) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->pipe_stream));
        r = *reinterpret_cast<const csvsimd_shard_result*>(ctx->pin_small_out);
        if (r.error) { g_last_error = "stage1 kernel: look-back spin bound hit"; return CSVSIMD_ERR_INTERNAL; }
        if (!tape || r.count <= cap) break;
        if (attempt) return CSVSIMD_ERR_INTERNAL;
        rc = ensure_out(r.count);  // denser than one entry per 4 bytes: exact capacity, once more
        if (rc != CSVSIMD_OK) return rc;
    }
    const uint64_t n = 1 + r.count;
    if (tape && tape_cap >= 1) {
        tape[0] = 0;  // src/reader.rs:216
        const uint64_t ncopy = std::min<uint64_t>(tape_cap - 1, r.count);
        if (ncopy) memcpy(tape + 1, (const char*)ctx->pin_small_out + 64, ncopy * 8);
    }
    g_ingest_phases = csvsimd_ingest_phases{};
    g_ingest_phases.bytes = len;
    g_ingest_phases.chunks = len ? 1 : 0;
    g_ingest_phases.host_threads = 1;
    g_ingest_phases.wall = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin;
    *tape_len = n;
    if (in_quote_out) *in_quote_out = r.in_quote_out;
    if (tape && n > tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
}

static int stage1_index_host_body(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf, uint64_t len,
                                  uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out) {
    if (!ctx || (len && !buf) || (!tape && tape_cap) || !tape_len) return CSVSIMD_ERR_INVALID_ARG;
    if (dialect_check(dialect) != CSVSIMD_OK) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    if (len <= csvsimd_ctx::kSmallBytes && !getenv("CSVSIMD_INGEST_CHUNK_MIB"))
        return stage1_index_host_small(ctx, dialect, buf, len, tape, tape_cap, tape_len, in_quote_out);
    constexpr uint64_t kMiB = 1ull << 20, kMax = csvsimd_ctx::kChunk;
    constexpr int S = csvsimd_ctx::kSlots;
    uint64_t uniform = 0;
    if (const char* e = getenv("CSVSIMD_INGEST_CHUNK_MIB")) {
        const uint64_t v = (uint64_t)atoi(e) << 20;
        if (v >= kMiB && v <= kMax) uniform = v;
    }
    // Entries per input byte, as the latest chunk whose record was read had them (a first guess until then): sizes the grid
    // of the kernel that writes a chunk's offsets into its pinned slot.  Those writes share the PCIe link's upstream
    // direction with the read requests of the H2D copies; a full-width grid dumps a chunk's 4 MiB in one burst and the
    // copies behind it stall (measured on 2 GiB of the 64-column corpus: 49.8 GiB/s with 256 workgroups, 50.4 with one,
    // 51.5 with no tape at all).  One workgroup per 2 MiB of offsets keeps up with the link many times over.
    double entries_per_byte = 1.0 / 32.0;
    // a context that has not seen its data yet: sixteen 4-KiB windows of the caller's buffer, bytes equal to the delimiter,
    // CR or LF — the first chunk's launch then already runs the instantiation the file's density asks for
    if (ctx->density < 0 && len >= (8ull << 20)) {
        const uint8_t dl = dialect ? dialect->delimiter : (uint8_t)',';
        uint64_t hits = 0, seen = 0;
        for (int wdw = 0; wdw < 16; ++wdw) {
            const uint8_t* p = buf + ((len - 4096) / 15) * (uint64_t)wdw;
            for (int i = 0; i < 4096; ++i) hits += (p[i] == dl) | (p[i] == 0x0a) | (p[i] == 0x0d);
            seen += 4096;
        }
        ctx->density = (double)hits / (double)seen;
        entries_per_byte = std::max(ctx->density, 1.0 / 256.0);
    }
    constexpr int h2d_streams = 2;
    const std::vector<uint64_t> cuts = ingest_chunk_plan(len, uniform);  // chunk i = [cuts[i], cuts[i + 1])
    const uint64_t nchunks = cuts.size() - 1;
    uint64_t largest = 0;
    for (uint64_t i = 0; i < nchunks; ++i) largest = std::max(largest, cuts[i + 1] - cuts[i]);
    int rc = pipe_setup(ctx, (int)std::min<uint64_t>(std::max<uint64_t>(nchunks, 1), S), largest);
    if (rc != CSVSIMD_OK) return rc;
    hipStream_t st = ctx->pipe_stream;
    // (Measured and rejected, round 5: letting the stage-1 kernel read a chunk straight from its pinned slot — no H2D copy, no
    // event.  A plain streaming kernel reads pinned memory at the link's rate (4 MiB in 81 us against 98 us for copy +
    // kernel: profiles/r05_api_cost.txt), but this kernel's loads are the transposing ones — a wave instruction takes 16 bytes
    // from each of 64 different 64-byte segments — and host memory is not cached on the GPU side: every segment crosses the
    // link four times.  32 MiB: 1 341 us against 947 us with the copies; 4 MiB: no difference: profiles/r05_midsize_zero_copy.txt.)
    // (Measured and rejected as well: a chunk's way in as a KERNEL that reads the pinned slot with contiguous 16-byte wave loads
    // instead of a hipMemcpyAsync with its ~18 us of fixed cost — alone such a kernel reads pinned memory at the link's rate,
    // inside the pipeline, next to the staging copies and the stage-1 launches, it made 22 GiB/s: 4 MiB 262 us against 225,
    // 32 MiB 1 429 against 956, the 10 000-file batch 1.92 ms against 1.36: profiles/r05_midsize_blit.txt.)
    // a file of a few MiB: the host-side copies ARE the critical path — slices of 128 KiB, workers polling for them
    const size_t min_slice = len <= (64ull << 20) ? (128u << 10) : CopyPool::kMinSlice;
    CopyPool::Busy busy(len <= (256ull << 20) ? ctx->copier.get() : nullptr);

    // How a chunk's tape reaches the host: a small kernel right behind the stage-1 launch packs the chunk's entries into
    // 32-bit chunk-relative offsets and writes them straight into the slot's pinned host buffer (narrow_tape_kernel,
    // text_kernels.hip): half the bytes cross PCIe, as the kernel's own posted writes, and no copy engine is involved.
    // Measured on the pool's two-socket hosts (scripts/probe_ingest3.py, 2 GiB): with the tape copied back by D2H
    // copies the H2D copies of the following chunks are served one after the other with them (49 ms = 39 ms of H2D +
    // 10 ms of D2H; the count-only call: 39 ms); with the stage-1 kernel writing its u64 tape into the pinned slot
    // itself, 42 ms.  The same kernel publishes the chunk's record (text_kernels.hip): the host polls a pinned word.
    uint64_t n = 1;  // entries so far, sentinel included
    if (tape && tape_cap >= 1) tape[0] = 0;  // src/reader.rs:216

    // The loop-carried values of the reference (inside_str, array_idx: src/reader.rs:217-218) between chunks: the
    // entering state travels ON THE DEVICE — the launch of chunk i + 1 reads in_quote_out (and escape_out) from chunk i's
    // result record when it starts (KernelArgs::chain) — so chunk i + 1 is enqueued before the host has seen chunk i's
    // record, and nothing on the GPU ever waits for the host.  The host reads the records kLag chunks late, only to learn
    // how many entries came back and where they belong in the caller's tape.
    //
    // Three host threads (a plan of one or two chunks runs everything on the caller's):
    //   stager     user buffer -> pinned slot, up to S - 1 chunks ahead of the H2D copies (sliced over the copy pool); chunk 0 is
    //              staged by the caller itself while the stager thread wakes up
    //   submitter  (the caller's thread) H2D copy, stage-1 launch, packing + publishing kernel; polls records kLag behind;
    //              expands the LAST chunk itself (no hand-over on the way out)
    //   expander   pinned 32-bit offsets -> the caller's tape (sliced over the copy pool)
    // The two threads belong to the context and sleep between calls (round 4 created and joined them per call).
    struct Slot {
        uint64_t chunk = 0, cap = 0, seq = 0;
        uint64_t at = 0, ncopy = 0, base = 0;  // entries waiting in pin_out[k] for the caller's tape (written by the submitter
                                               // before `finished` passes the chunk, read by the expander after)
    } slot[S];
    struct Shared {
        std::mutex m;
        std::condition_variable cv;
        std::atomic<uint64_t> staged{0};    // chunks whose bytes are in their pinned slot: all below this number are — counted by
                                            // whoever stages chunks 1 .. (the stager thread, or the caller when it does everything)
        std::atomic<bool> staged0{false};   // ... and chunk 0, which the CALLER stages while the stager thread is on chunk 1
                                            // already: the two finish in either order (a hot stager with a short chunk 1 was
                                            // first, its count was overwritten with 1, and the call waited for ever: found by
                                            // scripts/fuzz_gpu.py's host-batch mode, round 5)
        std::atomic<uint64_t> h2d{0};       // chunks whose H2D copy has been enqueued (ev_in recorded)
        std::atomic<uint64_t> finished{0};  // chunks whose record has been read (slot[].at / ncopy / base valid)
        std::atomic<uint64_t> expanded{0};  // chunks whose entries have left their pinned slot
        std::atomic<bool> abort{false};
        int err = CSVSIMD_OK;
        std::string msg;
    } sh;
    // a thread of the pipeline waits for another: poll for a while (the other is microseconds away in a file of a few MiB),
    // then sleep on the condition variable (every update of the counters notifies under the mutex)
    auto await = [&sh](auto&& ready) {
        for (int i = 0; i < 4000; ++i) {
            if (ready()) return;
            for (int p = 0; p < 8; ++p) _mm_pause();
        }
        std::unique_lock<std::mutex> g(sh.m);
        sh.cv.wait(g, ready);
    };
    auto bump = [&sh](std::atomic<uint64_t>& c, uint64_t v) {
        {
            std::lock_guard<std::mutex> g(sh.m);
            c.store(v, std::memory_order_release);
        }
        sh.cv.notify_all();
    };
    uint32_t host_inq = 0, host_esc = dialect ? dialect->escape_in : 0;  // the state after the last chunk whose record was read
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_stage = 0, t_stage_wait = 0, t_expand = 0, t_wait_staged = 0, t_wait_record = 0, t_wait_expanded = 0, t_submit = 0;
    double t_stage0 = 0, t_expand_last = 0;  // the caller's own share of staging / expanding (first / last chunk)
#define CSVSIMD_TIMED(acc, stmt) do { const double t0_ = now(); stmt; acc += now() - t0_; } while (0)
    auto fail = [&](int code, const std::string& msg) {
        {
            std::lock_guard<std::mutex> g(sh.m);
            if (sh.err == CSVSIMD_OK) { sh.err = code; sh.msg = msg; }
            sh.abort.store(true, std::memory_order_release);
        }
        sh.cv.notify_all();
    };

    // ---- stager: chunk j's bytes into pin_in[j % S] -----------------------------------------------------------------------
    auto stage_one = [&](uint64_t j, double& t_copy, double& t_wait) -> bool {
        const int k = (int)(j % S);
        {
            const double t0 = now();
            await([&] { return sh.abort.load(std::memory_order_acquire) || j < (uint64_t)S || sh.h2d.load(std::memory_order_acquire) + S > j; });  // chunk j - S has been enqueued
            t_wait += now() - t0;
            if (sh.abort.load(std::memory_order_acquire)) return false;
        }
        if (j >= (uint64_t)S) {  // ... and has left the staging slot
            const double t0 = now();
            const hipError_t e = hipEventSynchronize(ctx->ev_in[k]);
            t_wait += now() - t0;
            if (e != hipSuccess) { fail(CSVSIMD_ERR_HIP, std::string("hipEventSynchronize(ev_in): ") + hipGetErrorString(e)); return false; }
        }
        CSVSIMD_TIMED(t_copy, ctx->copier->copy(ctx->pin_in[k], buf + cuts[j], cuts[j + 1] - cuts[j], min_slice));
        if (j == 0) {
            {
                std::lock_guard<std::mutex> g(sh.m);
                sh.staged0.store(true, std::memory_order_release);
            }
            sh.cv.notify_all();
        } else {
            bump(sh.staged, j + 1);
        }
        return true;
    };
    // ---- expander: chunk j's 32-bit offsets -> the caller's tape ------------------------------------------------------------
    auto expand_one = [&](uint64_t j, double& t_copy) -> bool {
        const int k = (int)(j % S);
        await([&] { return sh.abort.load(std::memory_order_acquire) || sh.finished.load(std::memory_order_acquire) > j; });
        if (sh.abort.load(std::memory_order_acquire)) return false;
        const Slot info = slot[k];  // (written before `finished` passed j, not touched again before `expanded` passes it)
        if (info.ncopy) CSVSIMD_TIMED(t_copy, ctx->copier->expand(tape + info.at, ctx->pin_out[k], info.ncopy, info.base, min_slice));
        bump(sh.expanded, j + 1);
        return true;
    };

    // enqueues the kernels of the chunk in slot k: stage 1, then the kernel that packs the tape into the pinned slot and
    // publishes the record under a fresh sequence number
    auto launch = [&](int k, bool chained) -> int {
        const uint64_t i = slot[k].chunk, off = cuts[i], clen = cuts[i + 1] - off;
        csvsimd_dialect dia;
        if (dialect) { dia = *dialect; dia.escape_in = (uint8_t)host_esc; }
        int rc_ = stage1_async_impl(ctx, dialect ? &dia : nullptr, ctx->d_in[k], clen, off, host_inq,
                                    tape ? ctx->d_tape[k] : nullptr, slot[k].cap, ctx->d_res[k], st, nullptr,
                                    chained ? ctx->d_res[(k + S - 1) % S] : nullptr, /*short_launch=*/true);
        if (rc_ != CSVSIMD_OK) return rc_;
        void *out_dev = nullptr, *rec_dev = nullptr;
        if (tape) HIP_TRY(hipHostGetDevicePointer(&out_dev, ctx->pin_out[k], 0));
        HIP_TRY(hipHostGetDevicePointer(&rec_dev, (void*)ctx->h_res, 0));
        rec_dev = (char*)rec_dev + (size_t)k * sizeof(csvsimd_ctx::HostRecord);
        const double out_bytes = tape ? std::min<double>(entries_per_byte * (double)clen, (double)slot[k].cap) * 4.0 : 0.0;
        // one workgroup per 2 MiB of offsets for a full-size chunk of a long file (a burst of writes stalls the copies behind it,
        // above); the chunks of a file of a few MiB — and the ramp of a long one — have little behind them to protect, and a
        // lone workgroup needs 20-150 us to push 128 KiB - 1 MiB over the link (the GPU-side timeline of a 4-MiB and a 32-MiB
        // call, profiles/r05_midsize_timeline.txt: these kernels, one after the other on the stream, were the calls' critical
        // path, not the copies: 597 of a 32-MiB call's 936 us): one workgroup per 16 KiB there
        const double per_wg = clen <= (16ull << 20) ? (double)(16u << 10) : (double)(2u << 20);
        const int wgs = (int)std::min<double>(std::max(1.0, std::ceil(out_bytes / per_wg)), (double)ctx->n_cus);
        slot[k].seq = ++ctx->pub_seq;
        HIP_TRY(csvsimd::launch_narrow_tape(ctx->d_tape[k], ctx->d_res[k], slot[k].cap, off, out_dev, wgs, st, rec_dev, slot[k].seq,
                                            (char*)ctx->d_pub + 64 * k));
        return CSVSIMD_OK;
    };
    // waits for the record of the chunk in slot k: polls the sequence word its publishing kernel writes last.  Every
    // ~250 us it asks the stream whether it is still busy: a stream that has drained (or failed) without the word is an error,
    // never an endless wait.
    auto await_record = [&](int k, csvsimd_shard_result& r) -> int {
        const volatile uint64_t* seq = &ctx->h_res[k].seq;
        const uint64_t want = slot[k].seq;
        for (double t_check = now() + 250e-6;;) {
            if (*seq == want) break;
            for (int p = 0; p < 16; ++p) _mm_pause();
            if (now() < t_check) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {  // everything enqueued has run: the word must be there
                if (*seq == want) break;
                g_last_error = "ingest: a chunk's record was not published";
                return CSVSIMD_ERR_INTERNAL;
            }
            if (q != hipErrorNotReady) return fail_hip(q, "hipStreamQuery(pipe_stream)");
            t_check = now() + 250e-6;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(&r, (const void*)&ctx->h_res[k].rec, sizeof r);
        return CSVSIMD_OK;
    };
    // reads the record of chunk j (kLag behind the launches) and re-runs the chunk if its tape did not fit
    auto finish = [&](uint64_t j) -> int {
        const int k = (int)(j % S);
        csvsimd_shard_result r;
        {
            int rc_ = CSVSIMD_OK;
            CSVSIMD_TIMED(t_wait_record, rc_ = await_record(k, r));
            if (rc_ != CSVSIMD_OK) return rc_;
        }
        if (r.error) { g_last_error = "stage1 kernel: look-back spin bound hit"; return CSVSIMD_ERR_INTERNAL; }
        if (tape && r.count > slot[k].cap) {
            // denser than guessed (first guess: one entry per 4 bytes): exact capacity, run the chunk again.  Its input is
            // still in the slot (the slot's next chunk is S - kLag iterations away), its entering state is the host's:
            // everything before it has been read.  The chunks enqueued behind it meanwhile chained from this chunk's first
            // record, whose in_quote_out / escape_out do not depend on the capacity — and the re-run writes the same.
            int rc_ = pipe_ensure_tape(ctx, k, r.count);  // waits for pipe_stream
            if (rc_ != CSVSIMD_OK) return rc_;
            slot[k].cap = std::min(ctx->d_tape_entries[k], ctx->pin_out_entries[k]);
            entries_per_byte = (double)r.count / (double)std::max<uint64_t>(1, cuts[j + 1] - cuts[j]);  // now known exactly
            rc_ = launch(k, false);
            if (rc_ != CSVSIMD_OK) return rc_;
            rc_ = await_record(k, r);  // the second pass's own record: its error flag and count are what count
            if (rc_ != CSVSIMD_OK) return rc_;
            if (r.error) { g_last_error = "stage1 kernel: look-back spin bound hit"; return CSVSIMD_ERR_INTERNAL; }
            if (r.count > slot[k].cap) return CSVSIMD_ERR_INTERNAL;
        }
        slot[k].ncopy = 0;
        if (tape && n < tape_cap) {
            slot[k].ncopy = std::min<uint64_t>(tape_cap - n, r.count);
            slot[k].at = n;
            slot[k].base = cuts[j];
        }
        bump(sh.finished, j + 1);
        n += r.count;
        if (cuts[j + 1] > cuts[j]) {
            entries_per_byte = (double)r.count / (double)(cuts[j + 1] - cuts[j]);
            ctx->density = entries_per_byte;  // the following chunks' launches choose their instantiation by it
        }
        host_inq = r.in_quote_out;
        host_esc = r.escape_out;
        return CSVSIMD_OK;
    };
    // the submitter's share of chunk i: H2D copy, kernels
    auto submit = [&](uint64_t i) -> int {
        const int k = (int)(i % S);
        const uint64_t clen = cuts[i + 1] - cuts[i];
        {
            const double t0 = now();
            await([&] {
                return sh.abort.load(std::memory_order_acquire) ||
                       (i == 0 ? sh.staged0.load(std::memory_order_acquire) : sh.staged.load(std::memory_order_acquire) > i);
            });
            t_wait_staged += now() - t0;
            if (sh.abort.load(std::memory_order_acquire)) return sh.err != CSVSIMD_OK ? sh.err : CSVSIMD_ERR_INTERNAL;
        }
        const double t0 = now();
        // Two copy streams take turns: a copy's set-up and completion signalling (~30 us, measured as the difference between
        // 68 chunked copies and one copy of the same 2 GiB) overlap the other stream's transfer instead of idling the link.
        // d_in[k] is free: the record of the slot's previous chunk (i - S) was read before this call (S > kLag).
        hipStream_t cs = (h2d_streams == 2 && (i & 1)) ? ctx->in_stream2 : ctx->in_stream;
        HIP_TRY(hipMemcpyAsync(ctx->d_in[k], ctx->pin_in[k], clen, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipEventRecord(ctx->ev_in[k], cs));
        bump(sh.h2d, i + 1);
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_in[k], 0));
        t_submit += now() - t0;
        if (i >= (uint64_t)S) {  // the slot's previous chunk (i - S): its offsets must have left pin_out[k]
            const double t1 = now();
            await([&] { return sh.abort.load(std::memory_order_acquire) || sh.expanded.load(std::memory_order_acquire) + S > i; });
            t_wait_expanded += now() - t1;
            if (sh.abort.load(std::memory_order_acquire)) return sh.err != CSVSIMD_OK ? sh.err : CSVSIMD_ERR_INTERNAL;
        }
        const double t2 = now();
        slot[k].chunk = i;
        slot[k].cap = 0;
        if (tape) {
            // first guess: one entry per 4 bytes of the SLOT (sized once per size class, not per chunk)
            int rc_ = pipe_ensure_tape(ctx, k, std::max<uint64_t>(std::max(clen, ctx->slot_bytes[k]) / 4, 4096));
            if (rc_ != CSVSIMD_OK) return rc_;
            slot[k].cap = std::min(ctx->d_tape_entries[k], ctx->pin_out_entries[k]);
        }
        const int rc_ = launch(k, i > 0);
        t_submit += now() - t2;
        return rc_;
    };

    constexpr uint64_t kLag = 2;  // records are read two chunks behind the launches: two H2D copies stay queued meanwhile
    static_assert(kLag < (uint64_t)S, "a slot's input buffer is reused once the record of its previous chunk has been read");
    const bool threaded = nchunks >= 3;
    if (threaded) {
        if (!ctx->stager_thread) ctx->stager_thread.reset(new TaskThread);
        if (tape && !ctx->expander_thread) ctx->expander_thread.reset(new TaskThread);
        struct Joiner {  // whatever happens below (an error return, an exception), both tasks have ended before the frame goes
            Shared& sh;
            TaskThread *a, *b;
            ~Joiner() {
                {
                    std::lock_guard<std::mutex> g(sh.m);
                    sh.abort.store(true, std::memory_order_release);  // (a completed run: both loops have ended, nobody is listening)
                }
                sh.cv.notify_all();
                if (a) a->wait();
                if (b) b->wait();
            }
        } joiner{sh, nullptr, nullptr};
        const int dev = ctx->device;
        joiner.a = ctx->stager_thread.get();
        ctx->stager_thread->post([&, dev] {
            if (hipSetDevice(dev) != hipSuccess) { fail(CSVSIMD_ERR_HIP, "hipSetDevice (stager)"); return; }
            try {
                for (uint64_t j = 1; j < nchunks; ++j)  // chunk 0 is the caller's
                    if (!stage_one(j, t_stage, t_stage_wait)) return;
            } catch (...) { fail(CSVSIMD_ERR_INVALID_STATE, "ingest stager: exception"); }
        });
        if (tape) {
            joiner.b = ctx->expander_thread.get();
            ctx->expander_thread->post([&] {
                try {
                    for (uint64_t j = 0; j + 1 < nchunks; ++j)  // the last chunk is the caller's
                        if (!expand_one(j, t_expand)) return;
                } catch (...) { fail(CSVSIMD_ERR_INVALID_STATE, "ingest expander: exception"); }
            });
        } else {
            bump(sh.expanded, nchunks);  // count only: nothing ever waits in a pinned slot
        }
        double t_none = 0;
        rc = stage_one(0, t_stage0, t_none) ? CSVSIMD_OK : CSVSIMD_ERR_INTERNAL;
        for (uint64_t i = 0; i < nchunks && rc == CSVSIMD_OK; ++i) {
            rc = submit(i);
            if (rc == CSVSIMD_OK && i >= kLag) rc = finish(i - kLag);
        }
        for (uint64_t j = nchunks > kLag ? nchunks - kLag : 0; j < nchunks && rc == CSVSIMD_OK; ++j) rc = finish(j);
        if (rc == CSVSIMD_OK && tape) {
            const double t0 = now();
            await([&] { return sh.abort.load(std::memory_order_acquire) || sh.expanded.load(std::memory_order_acquire) + 1 >= nchunks; });
            t_wait_expanded += now() - t0;
            if (!sh.abort.load(std::memory_order_acquire) && !expand_one(nchunks - 1, t_expand_last)) rc = CSVSIMD_ERR_INTERNAL;
        }
        {
            std::lock_guard<std::mutex> g(sh.m);
            if (sh.err != CSVSIMD_OK) {  // a worker's failure is the call's (the submitter may only have seen `abort`)
                rc = sh.err;
                g_last_error = sh.msg;
            }
        }
        if (rc != CSVSIMD_OK) {
            const std::string keep = g_last_error;
            fail(rc, keep);  // stops the workers; the Joiner waits for them
            g_last_error = keep;
            return rc;
        }
    } else {
        // a plan of one or two chunks (a uniform override): the same steps in turn on the caller's thread
        if (!tape) bump(sh.expanded, nchunks);
        for (uint64_t i = 0; i < nchunks; ++i) {
            if (!stage_one(i, t_stage, t_stage_wait)) { g_last_error = sh.msg; return sh.err; }
            rc = submit(i);
            if (rc != CSVSIMD_OK) return rc;
            if (i >= kLag) {
                rc = finish(i - kLag);
                if (rc != CSVSIMD_OK) return rc;
                if (tape && !expand_one(i - kLag, t_expand)) return sh.err;
            }
        }
        for (uint64_t j = nchunks > kLag ? nchunks - kLag : 0; j < nchunks; ++j) {
            rc = finish(j);
            if (rc != CSVSIMD_OK) return rc;
            if (tape && !expand_one(j, t_expand)) return sh.err;
        }
    }
#undef CSVSIMD_TIMED
    g_ingest_phases = csvsimd_ingest_phases{len, nchunks, threaded ? 3u : 1u, 0u, now() - t_begin, t_stage + t_stage0, t_stage_wait,
                                            t_expand + t_expand_last, t_submit, t_wait_staged, t_wait_record, t_wait_expanded};
    *tape_len = n;
    if (in_quote_out) *in_quote_out = host_inq;
    if (tape && n > tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
}

int csvsimd_stage1_index(csvsimd_ctx* ctx, const uint8_t* buf, uint64_t len, uint64_t* tape, uint64_t tape_cap,
                         uint64_t* tape_len, uint32_t* in_quote_out) {
    return csvsimd_guarded([&]() -> int {
    return stage1_index_host_impl(ctx, nullptr, buf, len, tape, tape_cap, tape_len, in_quote_out);
    });
}

int csvsimd_stage1_index_dialect(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf, uint64_t len,
                                 uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out) {
    return csvsimd_guarded([&]() -> int {
    if (!dialect) return CSVSIMD_ERR_INVALID_ARG;
    return stage1_index_host_impl(ctx, dialect, buf, len, tape, tape_cap, tape_len, in_quote_out);
    });
}

/* ---- many small HOST files in one call ------------------------------------------------------------------------------
 * The reference's unit of work is a file (csv_simd::create, src/lib.rs:61-74) and its own inputs are 96 to 623 bytes
 * (its res directory): through csvsimd_stage1_index a 300-byte file costs 30 us — a launch and a wait — where one CPU core needs
 * 1.3 us.  Here n files share that cost: they are packed into pinned groups (a stager thread, the copy pool), each group
 * crosses PCIe as ONE copy that also carries the group's buffer table, is indexed by ONE batched launch (stage1_kernel<...,
 * BATCH>: every file is a buffer of its own, entered outside a string, look-backs stop at file boundaries) whose tapes and
 * result records land in pinned memory as the kernel's own stores, and an expander thread hands every file its tape.
 * Groups ramp up (256 KiB, 512 KiB ... 4 MiB) for the same reason chunks do.  A file above kBatchItemMax, and a file denser
 * than its share of the group's tape block (an entry per 4 bytes), goes through csvsimd_stage1_index's own path afterwards. */
namespace {
constexpr uint64_t kBatchItemMax = 1ull << 20;       // larger files are not worth packing: they fill a pipeline by themselves
constexpr uint64_t kBatchGroupBytes = 4ull << 20;    // packed bytes per group (measured: 2 MiB 1.30-1.41 ms, 4 MiB 1.17-1.27, 7.5 MiB 1.50-1.55 for 10 000 files of 3.7 KiB)
constexpr uint32_t kBatchGroupItems = 2048;          // files per group (a file is at least one tile of ticket space: the default scratch holds 16 384 64-KiB tiles)
struct BatchGroup {
    uint32_t first = 0, count = 0;   // items [first, first + count) of the packed order
    uint64_t in_bytes = 0;           // packed input incl. table
    uint64_t off_table = 0, off_first = 0, off_tot = 0;
    uint64_t out_bytes = 0;          // results + tapes
    uint32_t tiles = 0;
};
}  // namespace

static int stage1_index_batch_body(csvsimd_ctx* ctx, csvsimd_host_batch_item* items, uint32_t n_items) {
    if (!ctx || (n_items && !items)) return CSVSIMD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_items; ++i) {
        csvsimd_host_batch_item& it = items[i];
        if ((it.len && !it.buf) || (!it.tape && it.tape_cap)) return CSVSIMD_ERR_INVALID_ARG;
        it.tape_len = 0;
        it.in_quote_out = 0;
        it.status = CSVSIMD_ERR_INVALID_STATE;  // until its group has come back
    }
    WITH_DEVICE_OF(ctx);
    constexpr int S = csvsimd_ctx::kSlots;
    // ---- the plan: which files are packed (in their order), where each one sits in its group -------------------------------
    struct Placed {
        uint32_t item;
        uint32_t in_off;     // in the group's input block (64-byte aligned)
        uint32_t out_cap;    // entries its share of the group's tape block holds
        uint64_t out_off;    // of its tape in the group's output block (bytes, 128-byte aligned)
    };
    // Which geometry?  A file is at least one tile of the launch's ticket space, and a tile costs its workgroup a full count
    // phase whatever it holds: 1 100 files of 3.7 KiB (a 4-MiB group) took the default geometry's 256-KiB tiles 90-100 us —
    // as long as the group's copy (the call's GPU-side timeline, profiles/r05_small_files_timeline.txt) — so batches of
    // small files run the 64-KiB-tile (dense) instantiation, which is the same tape for any density.
    uint64_t packed_bytes = 0, packed_files = 0;
    for (uint32_t i = 0; i < n_items; ++i)
        if (items[i].len <= kBatchItemMax) { packed_bytes += items[i].len; ++packed_files; }
    const bool small_tiles = packed_files && packed_bytes / packed_files <= (128u << 10);
    const uint64_t tile_bytes = small_tiles ? CSVSIMD_MIN_TILE_BYTES : CSVSIMD_TILE_BYTES;
    std::vector<Placed> placed;
    std::vector<uint32_t> alone;  // items that take the single-file path
    std::vector<BatchGroup> groups;
    placed.reserve(n_items);
    {
        BatchGroup g;
        uint64_t in_cur = 0, out_cur = 0, target = 256u << 10;
        auto close_group = [&] {
            if (!g.count) return;
            g.off_table = (in_cur + 63) & ~(uint64_t)63;
            g.off_first = g.off_table + (uint64_t)g.count * sizeof(csvsimd::BatchItemHost);
            g.off_tot = g.off_first + (((uint64_t)g.count * 4 + 63) & ~(uint64_t)63);
            g.in_bytes = g.off_tot + (uint64_t)g.count * 8;
            // output block: [count result records | tapes]; the tape offsets were laid out behind a records area of
            // kBatchGroupItems records, so that they do not depend on the group's final count
            g.out_bytes = out_cur;
            groups.push_back(g);
            target = std::min<uint64_t>(target * 2, kBatchGroupBytes);
            g = BatchGroup{};
            g.first = (uint32_t)placed.size();
            in_cur = 0;
            out_cur = 0;
        };
        for (uint32_t i = 0; i < n_items; ++i) {
            const csvsimd_host_batch_item& it = items[i];
            if (it.len > kBatchItemMax) { alone.push_back(i); continue; }
            const uint64_t in_need = (it.len + 63) & ~(uint64_t)63;
            const uint64_t want_cap = it.tape ? std::min<uint64_t>(it.tape_cap ? it.tape_cap - 1 : 0, it.len / 4 + 16) : 0;
            const uint64_t out_need = (want_cap * 8 + 127) & ~(uint64_t)127;
            if (g.count && (in_cur + in_need > target || g.count >= kBatchGroupItems)) close_group();
            if (!g.count) out_cur = (uint64_t)kBatchGroupItems * sizeof(csvsimd_shard_result);
            placed.push_back(Placed{i, (uint32_t)in_cur, (uint32_t)want_cap, out_cur});
            in_cur += in_need;
            out_cur += out_need;
            g.tiles += it.len ? (uint32_t)((it.len + tile_bytes - 1) / tile_bytes) : 0u;
            ++g.count;
        }
        close_group();
    }
    const uint64_t ngroups = groups.size();
    int any_capacity = 0;
    if (ngroups) {
        uint64_t max_in = 0, max_out = 0;
        uint32_t max_tiles = 0;
        for (const BatchGroup& g : groups) {
            max_in = std::max(max_in, g.in_bytes);
            max_out = std::max(max_out, g.out_bytes);
            max_tiles = std::max(max_tiles, g.tiles);
        }
        int rc = pipe_setup(ctx, (int)std::min<uint64_t>(ngroups, S), max_in);
        if (rc != CSVSIMD_OK) return rc;
        rc = csvsimd_ctx_reserve(ctx, (uint64_t)max_tiles * tile_bytes);
        if (rc != CSVSIMD_OK) return rc;
        for (int k = 0; k < (int)std::min<uint64_t>(ngroups, S); ++k)
            if (ctx->pin_bout_bytes[k] < max_out) {
                if (ctx->pin_bout[k]) HIP_TRY(hipHostFree(ctx->pin_bout[k]));
                ctx->pin_bout[k] = nullptr;
                ctx->pin_bout_bytes[k] = 0;
                const size_t want = std::max<size_t>(max_out, (size_t)kBatchGroupItems * sizeof(csvsimd_shard_result) + (2u << 20));
                HIP_TRY(hipHostMalloc(&ctx->pin_bout[k], want, hipHostMallocDefault));
                ctx->pin_bout_bytes[k] = want;
            }
        hipStream_t st = ctx->pipe_stream;
        void* bout_dev[S] = {};
        void* rec_dev_base = nullptr;
        for (int k = 0; k < (int)std::min<uint64_t>(ngroups, S); ++k) HIP_TRY(hipHostGetDevicePointer(&bout_dev[k], ctx->pin_bout[k], 0));
        HIP_TRY(hipHostGetDevicePointer(&rec_dev_base, (void*)ctx->h_res, 0));
        CopyPool::Busy busy(ctx->copier.get());

        struct Shared {
            std::mutex m;
            std::condition_variable cv;
            std::atomic<uint64_t> staged{0}, h2d{0}, finished{0}, expanded{0};
            std::atomic<bool> staged0{false};  // (group 0 is the caller's, groups 1 .. the stager's: see stage1_index_host_body)
            std::atomic<bool> abort{false};
            int err = CSVSIMD_OK;
            std::string msg;
        } sh;
        auto await = [&sh](auto&& ready) {
            for (int i = 0; i < 4000; ++i) {
                if (ready()) return;
                for (int p = 0; p < 8; ++p) _mm_pause();
            }
            std::unique_lock<std::mutex> g(sh.m);
            sh.cv.wait(g, ready);
        };
        auto bump = [&sh](std::atomic<uint64_t>& c, uint64_t v) {
            {
                std::lock_guard<std::mutex> g(sh.m);
                c.store(v, std::memory_order_release);
            }
            sh.cv.notify_all();
        };
        auto fail = [&](int code, const std::string& msg) {
            {
                std::lock_guard<std::mutex> g(sh.m);
                if (sh.err == CSVSIMD_OK) { sh.err = code; sh.msg = msg; }
                sh.abort.store(true, std::memory_order_release);
            }
            sh.cv.notify_all();
        };
        auto aborted = [&sh] { return sh.abort.load(std::memory_order_acquire); };
        uint64_t seqs[S] = {};
        std::atomic<int> capacity_seen{0};

        // ---- stager: group j's files and its buffer table into pin_in[j % S] -------------------------------------------------
        auto stage_one = [&](uint64_t j) -> bool {
            const int k = (int)(j % S);
            await([&] { return aborted() || j < (uint64_t)S || sh.h2d.load(std::memory_order_acquire) + S > j; });
            if (aborted()) return false;
            if (j >= (uint64_t)S) {
                const hipError_t e = hipEventSynchronize(ctx->ev_in[k]);  // the slot's previous group has left the staging block
                if (e != hipSuccess) { fail(CSVSIMD_ERR_HIP, std::string("hipEventSynchronize(ev_in): ") + hipGetErrorString(e)); return false; }
            }
            const BatchGroup& g = groups[j];
            char* const base = (char*)ctx->pin_in[k];
            const char* const d_in = (const char*)ctx->d_in[k];
            char* const d_out = (char*)bout_dev[k];
            csvsimd::BatchItemHost* const table = reinterpret_cast<csvsimd::BatchItemHost*>(base + g.off_table);
            uint32_t* const firsts = reinterpret_cast<uint32_t*>(base + g.off_first);
            memset(base + g.off_tot, 0, (size_t)g.count * 8);
            // first tiles: a prefix sum over the group (cheap, serial), then the copies and table lines in parallel
            uint32_t tiles = 0;
            for (uint32_t q = 0; q < g.count; ++q) {
                firsts[q] = tiles;
                const uint64_t len = items[placed[g.first + q].item].len;
                tiles += len ? (uint32_t)((len + tile_bytes - 1) / tile_bytes) : 0u;
            }
            const std::function<void(size_t, size_t)> pack = [&](size_t a, size_t b) {
                for (size_t q = a; q < b; ++q) {
                    const Placed& pl = placed[g.first + q];
                    const csvsimd_host_batch_item& it = items[pl.item];
                    if (it.len) memcpy(base + pl.in_off, it.buf, it.len);
                    csvsimd::BatchItemHost& t = table[q];
                    t.abase = d_in + pl.in_off;  // 64-byte aligned
                    t.lo = 0;
                    t.hi = it.len;
                    t.base_off = 0;              // offsets relative to the file's first byte, like csvsimd_stage1_index
                    t.tape = pl.out_cap ? d_out + pl.out_off : nullptr;
                    t.tape_cap = pl.out_cap;
                    t.first_tile = firsts[q];
                    t.in_quote_in = 0;
                    t.reserved = 0;
                }
            };
            ctx->copier->parallel_for(g.count, 64, pack);
            if (j == 0) {
                {
                    std::lock_guard<std::mutex> g_(sh.m);
                    sh.staged0.store(true, std::memory_order_release);
                }
                sh.cv.notify_all();
            } else {
                bump(sh.staged, j + 1);
            }
            return true;
        };
        // ---- expander: every file of group j gets its tape and its outputs ---------------------------------------------------
        auto expand_one = [&](uint64_t j) -> bool {
            const int k = (int)(j % S);
            await([&] { return aborted() || sh.finished.load(std::memory_order_acquire) > j; });
            if (aborted()) return false;
            const BatchGroup& g = groups[j];
            const char* const out = (const char*)ctx->pin_bout[k];
            const csvsimd_shard_result* const recs = reinterpret_cast<const csvsimd_shard_result*>(out);
            const std::function<void(size_t, size_t)> unpack = [&](size_t a, size_t b) {
                for (size_t q = a; q < b; ++q) {
                    const Placed& pl = placed[g.first + q];
                    csvsimd_host_batch_item& it = items[pl.item];
                    const csvsimd_shard_result& r = recs[q];
                    it.tape_len = r.count + 1;
                    it.in_quote_out = r.in_quote_out;
                    if (r.error) { it.status = CSVSIMD_ERR_INTERNAL; continue; }
                    if (it.tape && it.tape_cap) it.tape[0] = 0;  // src/reader.rs:216
                    const uint64_t room = it.tape_cap ? it.tape_cap - 1 : 0;
                    if (it.tape && r.count > pl.out_cap && pl.out_cap < room) { it.status = CSVSIMD_ERR_TAPE_CAPACITY + 1000; continue; }  // denser than its share: alone, later
                    const uint64_t ncopy = it.tape ? std::min<uint64_t>(r.count, room) : 0;
                    if (ncopy) memcpy(it.tape + 1, out + pl.out_off, ncopy * 8);
                    it.status = (it.tape && r.count + 1 > it.tape_cap) ? CSVSIMD_ERR_TAPE_CAPACITY : CSVSIMD_OK;
                    if (it.status != CSVSIMD_OK) capacity_seen.store(1, std::memory_order_relaxed);
                }
            };
            ctx->copier->parallel_for(g.count, 64, unpack);
            bump(sh.expanded, j + 1);
            return true;
        };
        auto submit = [&](uint64_t j) -> int {
            const int k = (int)(j % S);
            const BatchGroup& g = groups[j];
            await([&] { return aborted() || (j == 0 ? sh.staged0.load(std::memory_order_acquire) : sh.staged.load(std::memory_order_acquire) > j); });
            if (aborted()) return sh.err != CSVSIMD_OK ? sh.err : CSVSIMD_ERR_INTERNAL;
            hipStream_t cs = (j & 1) ? ctx->in_stream2 : ctx->in_stream;
            HIP_TRY(hipMemcpyAsync(ctx->d_in[k], ctx->pin_in[k], g.in_bytes, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipEventRecord(ctx->ev_in[k], cs));
            bump(sh.h2d, j + 1);
            HIP_TRY(hipStreamWaitEvent(st, ctx->ev_in[k], 0));
            if (j >= (uint64_t)S) {  // the slot's previous group must have left pin_bout[k]
                await([&] { return aborted() || sh.expanded.load(std::memory_order_acquire) + S > j; });
                if (aborted()) return sh.err != CSVSIMD_OK ? sh.err : CSVSIMD_ERR_INTERNAL;
            }
            csvsimd::Stage1Launch L;
            L.bind_scratch(ctx->scratch);
            ctx->last_stream = st;
            ctx->launched = true;
            char* const d_in = (char*)ctx->d_in[k];
            HIP_TRY((small_tiles ? csvsimd_dense::launch_stage1_batch_dense : csvsimd::launch_stage1_batch)(
                d_in + g.off_table, d_in + g.off_first, d_in + g.off_tot, g.count, g.tiles, (csvsimd_shard_result*)bout_dev[k],
                ctx->scratch, L.scratch_desc, ctx->max_blocks, st));
            seqs[k] = ++ctx->pub_seq;
            // the publisher: (no tape to pack) copies record 0 next to the sequence word, which is what the host polls
            HIP_TRY(csvsimd::launch_narrow_tape(nullptr, bout_dev[k], 0, 0, nullptr, 1, st,
                                                (char*)rec_dev_base + (size_t)k * sizeof(csvsimd_ctx::HostRecord), seqs[k],
                                                (char*)ctx->d_pub + 64 * k));
            return CSVSIMD_OK;
        };
        auto finish = [&](uint64_t j) -> int {
            const int k = (int)(j % S);
            const volatile uint64_t* seq = &ctx->h_res[k].seq;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            for (double t_check = now() + 250e-6;;) {
                if (*seq == seqs[k]) break;
                for (int p = 0; p < 16; ++p) _mm_pause();
                if (now() < t_check) continue;
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    if (*seq == seqs[k]) break;
                    g_last_error = "batch ingest: a group's records were not published";
                    return CSVSIMD_ERR_INTERNAL;
                }
                if (q != hipErrorNotReady) return fail_hip(q, "hipStreamQuery(pipe_stream)");
                t_check = now() + 250e-6;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            bump(sh.finished, j + 1);
            return CSVSIMD_OK;
        };

        constexpr uint64_t kLag = 2;
        if (!ctx->stager_thread) ctx->stager_thread.reset(new TaskThread);
        if (!ctx->expander_thread) ctx->expander_thread.reset(new TaskThread);
        struct Joiner {
            Shared& sh;
            TaskThread *a, *b;
            ~Joiner() {
                {
                    std::lock_guard<std::mutex> g(sh.m);
                    sh.abort.store(true, std::memory_order_release);
                }
                sh.cv.notify_all();
                if (a) a->wait();
                if (b) b->wait();
            }
        } joiner{sh, nullptr, nullptr};
        const int dev = ctx->device;
        const bool threaded = ngroups >= 2;
        rc = CSVSIMD_OK;
        if (threaded) {
            joiner.a = ctx->stager_thread.get();
            ctx->stager_thread->post([&, dev] {
                if (hipSetDevice(dev) != hipSuccess) { fail(CSVSIMD_ERR_HIP, "hipSetDevice (stager)"); return; }
                try {
                    for (uint64_t j = 1; j < ngroups; ++j)
                        if (!stage_one(j)) return;
                } catch (...) { fail(CSVSIMD_ERR_INVALID_STATE, "batch stager: exception"); }
            });
            joiner.b = ctx->expander_thread.get();
            ctx->expander_thread->post([&] {
                try {
                    for (uint64_t j = 0; j + 1 < ngroups; ++j)
                        if (!expand_one(j)) return;
                } catch (...) { fail(CSVSIMD_ERR_INVALID_STATE, "batch expander: exception"); }
            });
        }
        if (!stage_one(0)) rc = CSVSIMD_ERR_INTERNAL;
        for (uint64_t j = 0; j < ngroups && rc == CSVSIMD_OK; ++j) {
            if (!threaded && j > 0 && !stage_one(j)) { rc = CSVSIMD_ERR_INTERNAL; break; }
            rc = submit(j);
            if (rc == CSVSIMD_OK && j >= kLag) {
                rc = finish(j - kLag);
                if (rc == CSVSIMD_OK && !threaded && !expand_one(j - kLag)) rc = CSVSIMD_ERR_INTERNAL;
            }
        }
        for (uint64_t j = ngroups > kLag ? ngroups - kLag : 0; j < ngroups && rc == CSVSIMD_OK; ++j) {
            rc = finish(j);
            if (rc == CSVSIMD_OK && !threaded && !expand_one(j)) rc = CSVSIMD_ERR_INTERNAL;
        }
        if (rc == CSVSIMD_OK && threaded) {
            await([&] { return aborted() || sh.expanded.load(std::memory_order_acquire) + 1 >= ngroups; });
            if (!aborted() && !expand_one(ngroups - 1)) rc = CSVSIMD_ERR_INTERNAL;
        }
        {
            std::lock_guard<std::mutex> g(sh.m);
            if (sh.err != CSVSIMD_OK) {
                rc = sh.err;
                g_last_error = sh.msg;
            }
        }
        if (rc != CSVSIMD_OK) {
            const std::string keep = g_last_error;
            fail(rc, keep);
            g_last_error = keep;
            return rc;
        }
        any_capacity = capacity_seen.load();
    }
    // ---- the files that go alone: too large to pack, or denser than an entry per 4 bytes ---------------------------------------
    for (const Placed& pl : placed)
        if (items[pl.item].status == CSVSIMD_ERR_TAPE_CAPACITY + 1000) alone.push_back(pl.item);
    for (uint32_t i : alone) {
        csvsimd_host_batch_item& it = items[i];
        uint64_t n = 0;
        uint32_t q = 0;
        const int rc = stage1_index_host_body(ctx, nullptr, it.buf, it.len, it.tape, it.tape_cap, &n, &q);
        it.tape_len = n;
        it.in_quote_out = q;
        it.status = rc;
        if (rc == CSVSIMD_ERR_TAPE_CAPACITY) any_capacity = 1;
        else if (rc != CSVSIMD_OK) return rc;
    }
    for (uint32_t i = 0; i < n_items; ++i)
        if (items[i].status == CSVSIMD_ERR_INTERNAL) {
            g_last_error = "stage1 kernel: look-back spin bound hit";
            return CSVSIMD_ERR_INTERNAL;
        }
    return any_capacity ? CSVSIMD_ERR_TAPE_CAPACITY : CSVSIMD_OK;
}

int csvsimd_stage1_index_batch(csvsimd_ctx* ctx, csvsimd_host_batch_item* items, uint32_t n_items) {
    return csvsimd_guarded([&]() -> int {
    const int rc = stage1_index_batch_body(ctx, items, n_items);
    if (rc != CSVSIMD_OK && rc != CSVSIMD_ERR_TAPE_CAPACITY && ctx && ctx->pipe_stream) {
        const std::string keep = g_last_error;  // (as stage1_index_host_impl: nothing may be left in flight on the pinned slots)
        ScopedDevice scoped_device_(ctx->device);
        if (ctx->in_stream) (void)hipStreamSynchronize(ctx->in_stream);
        if (ctx->in_stream2) (void)hipStreamSynchronize(ctx->in_stream2);
        (void)hipStreamSynchronize(ctx->pipe_stream);
        (void)hipGetLastError();
        g_last_error = keep;
    }
    return rc;
    });
}

/* ---- one host file -> G GPUs of this process ----------------------------------------------------------------------- */

int csvsimd_multi_shard_range(uint64_t len, uint32_t n_shards, uint32_t i, uint64_t* begin, uint64_t* end) {
    if (!begin || !end || n_shards == 0 || i >= n_shards) return CSVSIMD_ERR_INVALID_ARG;
    // contiguous byte ranges, interior cuts rounded down to 64 bytes (any cut is legal: the semantics are
    // blocking-independent; 64 keeps every shard's loads aligned when the file's device copy is)
    auto cut = [&](uint32_t k) -> uint64_t {
        if (k == 0) return 0;
        if (k >= n_shards) return len;
        return (uint64_t)((unsigned __int128)len * k / n_shards) & ~(uint64_t)63;
    };
    *begin = cut(i);
    *end = cut(i + 1);
    return CSVSIMD_OK;
}

namespace {
// one shard's share of csvsimd_stage1_index_multi, on the calling thread: stream the bytes into the device buffer
// (two pinned staging slots, the context's own H2D stream), then the first stage-1 pass and its record
int multi_feed_and_index(csvsimd_multi_shard* sh, const uint8_t* buf, uint32_t in_quote_in) {
    csvsimd_ctx* ctx = sh->ctx;
    WITH_DEVICE_OF(ctx);  // (shard 0 runs on the CALLER's thread: its current device is put back)
    int rc = pipe_setup(ctx);
    if (rc != CSVSIMD_OK) return rc;
    const uint64_t n = sh->end - sh->begin;
    rc = csvsimd_ctx_reserve(ctx, n);
    if (rc != CSVSIMD_OK) return rc;
    constexpr uint64_t kChunk = csvsimd_ctx::kChunk;
    for (uint64_t off = 0, i = 0; off < n; off += kChunk, ++i) {
        const int k = (int)(i & 1);
        const uint64_t clen = std::min<uint64_t>(kChunk, n - off);
        if (i >= 2) HIP_TRY(hipEventSynchronize(ctx->ev_in[k]));  // the slot's previous chunk has left the staging buffer
        ctx->copier->copy(ctx->pin_in[k], buf + sh->begin + off, clen);
        HIP_TRY(hipMemcpyAsync((char*)sh->dbuf + off, ctx->pin_in[k], clen, hipMemcpyHostToDevice, ctx->in_stream));
        HIP_TRY(hipEventRecord(ctx->ev_in[k], ctx->in_stream));
    }
    // the kernel follows the last copy on the device (no host wait in between)
    if (n) HIP_TRY(hipStreamWaitEvent(ctx->pipe_stream, ctx->ev_in[((n - 1) / kChunk) & 1], 0));
    rc = csvsimd_stage1_index_device_async(ctx, sh->dbuf, n, sh->begin, in_quote_in, sh->dtape, sh->tape_cap, ctx->d_result,
                                           ctx->pipe_stream);
    if (rc != CSVSIMD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(&sh->result, ctx->d_result, sizeof(sh->result), hipMemcpyDeviceToHost, ctx->pipe_stream));
    HIP_TRY(hipStreamSynchronize(ctx->in_stream));
    HIP_TRY(hipStreamSynchronize(ctx->pipe_stream));
    return CSVSIMD_OK;
}
}  // namespace

int csvsimd_stage1_index_multi(const uint8_t* buf, uint64_t len, csvsimd_multi_shard* shards, uint32_t n_shards,
                               uint32_t file_in_quote_in) {
    return csvsimd_guarded([&]() -> int {
    if ((len && !buf) || !shards || n_shards == 0 || n_shards > 64) return CSVSIMD_ERR_INVALID_ARG;
    for (uint32_t g = 0; g < n_shards; ++g) {
        csvsimd_multi_shard& sh = shards[g];
        int rc = csvsimd_multi_shard_range(len, n_shards, g, &sh.begin, &sh.end);
        if (rc != CSVSIMD_OK) return rc;
        if (!sh.ctx || (sh.end > sh.begin && !sh.dbuf) || (!sh.dtape && sh.tape_cap) || ((uintptr_t)sh.dtape & 7))
            return CSVSIMD_ERR_INVALID_ARG;
        for (uint32_t h = 0; h < g; ++h)
            if (shards[h].ctx == sh.ctx) return CSVSIMD_ERR_INVALID_ARG;  // one context per shard (they run concurrently)
    }
    // one host thread per shard: each owns its device's H2D stream and staging slots; rank 0 knows how the file
    // starts, every other shard lets the kernel choose its entering state from its own first tiles (CSVSIMD_ENTER_GUESS)
    std::vector<int> rcs(n_shards, CSVSIMD_OK);
    std::vector<std::string> errs(n_shards);
    {
        std::vector<std::thread> th;
        th.reserve(n_shards);
        try {
            for (uint32_t g = 1; g < n_shards; ++g)
                th.emplace_back([&, g] {
                    rcs[g] = csvsimd_guarded([&]() -> int { return multi_feed_and_index(&shards[g], buf, CSVSIMD_ENTER_GUESS); });
                    if (rcs[g] != CSVSIMD_OK) errs[g] = csvsimd_last_error();
                });
        } catch (...) {  // a thread could not be created: the ones that did start are joined before the exception leaves
            for (auto& t : th) t.join();
            throw;
        }
        rcs[0] = csvsimd_guarded([&]() -> int { return multi_feed_and_index(&shards[0], buf, file_in_quote_in ? 1u : 0u); });
        if (rcs[0] != CSVSIMD_OK) errs[0] = csvsimd_last_error();
        for (auto& t : th) t.join();
    }
    for (uint32_t g = 0; g < n_shards; ++g)
        if (rcs[g] != CSVSIMD_OK) {
            g_last_error = errs[g];
            return rcs[g];
        }
    // the exchange: G records of 64 bytes are already on the host; stitch, and re-index only the shards whose first
    // pass ran under another entering state than the true one (README.md:24)
    std::vector<csvsimd_shard_result> recs(n_shards);
    for (uint32_t g = 0; g < n_shards; ++g) recs[g] = shards[g].result;
    for (uint32_t g = 0; g < n_shards; ++g) {
        csvsimd_multi_shard& sh = shards[g];
        int rc = csvsimd_stitch_shards(recs.data(), n_shards, g, file_in_quote_in, &sh.stitch);
        if (rc != CSVSIMD_OK) return rc;
        if (sh.stitch.error) { g_last_error = "stage1 kernel: look-back spin bound hit"; return CSVSIMD_ERR_INTERNAL; }
        if (sh.stitch.reemit) {
            WITH_DEVICE_OF(sh.ctx);
            rc = csvsimd_stage1_index_device(sh.ctx, sh.dbuf, sh.end - sh.begin, sh.begin, sh.stitch.in_quote_in, sh.dtape,
                                             sh.tape_cap, &sh.result, sh.ctx->pipe_stream);
            if (rc != CSVSIMD_OK && rc != CSVSIMD_ERR_TAPE_CAPACITY) return rc;
        }
    }
    for (uint32_t g = 0; g < n_shards; ++g)
        if (shards[g].dtape && shards[g].result.count > shards[g].tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
    });
}

int csvsimd_stitch_shards(const csvsimd_shard_result* results, uint32_t n_shards, uint32_t rank,
                          uint32_t file_in_quote_in, csvsimd_stitch* out) {
    if (!results || !out || rank >= n_shards) return CSVSIMD_ERR_INVALID_ARG;
    uint32_t state = file_in_quote_in ? 1u : 0u, err = 0;
    uint64_t idx = 1;  // the sentinel occupies global index 0
    for (uint32_t i = 0; i < n_shards; ++i) {
        const uint64_t cnt = state ? results[i].count_enter_inside : results[i].count_enter_outside;
        if (i == rank) {
            out->in_quote_in = state;
            out->count = cnt;
            out->tape_index_base = idx;
            out->reemit = (results[i].in_quote_in_used & 1u) != state ? 1u : 0u;
        }
        idx += cnt;
        state ^= results[i].quote_parity & 1u;
        err |= results[i].error;
    }
    out->in_quote_final = state;
    out->total_entries = idx;
    out->error = err ? 1u : 0u;
    return CSVSIMD_OK;
}

/* ---- tape --------------------------------------------------------------------------------- */

}  // extern "C"

// The index a tape owns (csvsimd_create): anonymous memory straight from the kernel — untouched capacity costs nothing
// (round 4's std::vector zero-filled an entry per 8 bytes of the file up front: 2 GiB of memset and page faults in front of
// a 40 ms pipeline, and 140 ms of munmap behind it), pages are faulted in by whoever writes them first (the expander's
// slices, in parallel), 2-MiB pages where the kernel hands them out.
struct IndexBlock {
    uint64_t* p = nullptr;
    size_t bytes = 0;
    IndexBlock() = default;
    IndexBlock(const IndexBlock&) = delete;
    IndexBlock& operator=(const IndexBlock&) = delete;
    ~IndexBlock() { release(); }
    void release() {
        if (p) munmap(p, bytes);
        p = nullptr;
        bytes = 0;
    }
    bool reserve(uint64_t entries) {
        release();
        const size_t want = (((size_t)entries * 8) + ((2u << 20) - 1)) & ~(size_t)((2u << 20) - 1);
        void* m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) return false;
        (void)madvise(m, want, MADV_HUGEPAGE);  // advice only: without THP the block is faulted in 4-KiB pages
        p = (uint64_t*)m;
        bytes = want;
        return true;
    }
    uint64_t capacity() const { return bytes / 8; }
    void shrink_to(uint64_t entries) {  // gives the untouched tail of the reservation back
        const size_t keep = std::max<size_t>(4096, (((size_t)entries * 8) + 4095) & ~(size_t)4095);
        if (p && keep < bytes && mremap(p, bytes, keep, 0) != MAP_FAILED) bytes = keep;
    }
};

struct csvsimd_tape {
    csv_simd::Tape tape;
    // owned storage when built by csvsimd_create
    void* map = nullptr;
    uint64_t map_len = 0;
    IndexBlock owned_index;
    csvsimd_tape() = default;
    csvsimd_tape(const csvsimd_tape&) = delete;
    csvsimd_tape& operator=(const csvsimd_tape&) = delete;
    ~csvsimd_tape() {
        if (map) munmap(map, map_len);  // also on every error path of csvsimd_create
    }
};

extern "C" {

int csvsimd_tape_create(const uint8_t* bytes, uint64_t len, const uint64_t* index, uint64_t index_len,
                        csvsimd_tape** out) {
    return csvsimd_guarded([&]() -> int {
    if (!bytes || !index || !out) return CSVSIMD_ERR_INVALID_ARG;
    *out = nullptr;
    csv_simd::Header h;
    csv_simd::StructureError e = csv_simd::Header::create(bytes, len, h);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    std::unique_ptr<csvsimd_tape> t(new (std::nothrow) csvsimd_tape);
    if (!t) return CSVSIMD_ERR_INVALID_STATE;
    e = csv_simd::Tape::from_core(bytes, len, csv_simd::StructureIndex{index, index_len}, std::move(h), t->tape);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    *out = t.release();
    return CSVSIMD_OK;
    });
}

void csvsimd_tape_destroy(csvsimd_tape* t) { delete t; }

// a NULL tape reads as an empty one (0 fields, 0 records) — never a crash across the FFI
uint32_t csvsimd_tape_field_cnt(const csvsimd_tape* t) { return t ? t->tape.header.field_cnt : 0; }
uint32_t csvsimd_tape_record_cnt(const csvsimd_tape* t) { return t ? t->tape.record_cnt_ : 0; }
uint64_t csvsimd_tape_record_jump_size(const csvsimd_tape* t) { return t ? t->tape.record_jump_size_ : 0; }
uint32_t csvsimd_tape_record_offset(const csvsimd_tape* t) { return t ? t->tape.header.record_offset : 0; }
int csvsimd_tape_new_line(const csvsimd_tape* t) {
    if (!t) return CSVSIMD_ERR_INVALID_ARG;
    return t->tape.header.new_line == csv_simd::NewLine::CRLF ? CSVSIMD_NEWLINE_CRLF : CSVSIMD_NEWLINE_LF;
}

int64_t csvsimd_tape_header_name(const csvsimd_tape* t, uint32_t i, char* dst, uint64_t cap) {
    if (!t || i >= t->tape.header.header.size()) return CSVSIMD_ERR_INVALID_ARG;
    const std::string& s = t->tape.header.header[i];
    if (dst && cap) memcpy(dst, s.data(), std::min<uint64_t>(cap, s.size()));
    return (int64_t)s.size();
}

int csvsimd_tape_seek_record(const csvsimd_tape* t, uint32_t record_idx, uint64_t* begin, uint64_t* end, int* found) {
    return csvsimd_guarded([&]() -> int {
    if (!t || !begin || !end || !found) return CSVSIMD_ERR_INVALID_ARG;
    std::optional<std::pair<uint64_t, uint64_t>> span;
    const auto e = csv_simd::RecordSource<csv_simd::Tape>::seek_record(t->tape, record_idx, span);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    *found = span ? 1 : 0;
    if (span) { *begin = span->first; *end = span->second; }
    return CSVSIMD_OK;
    });
}

int csvsimd_tape_seek_field(const csvsimd_tape* t, uint32_t record_idx, uint32_t field_idx, uint64_t* begin,
                            uint64_t* end, int* found) {
    return csvsimd_guarded([&]() -> int {
    if (!t || !begin || !end || !found) return CSVSIMD_ERR_INVALID_ARG;
    std::optional<std::pair<uint64_t, uint64_t>> span;
    const auto e = csv_simd::RecordSource<csv_simd::Tape>::seek_field(t->tape, record_idx, field_idx, span);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    *found = span ? 1 : 0;
    if (span) { *begin = span->first; *end = span->second; }
    return CSVSIMD_OK;
    });
}

int csvsimd_boundaries(uint32_t task_size, uint8_t job_count, csvsimd_boundary* out, uint32_t* n_out) {
    return csvsimd_guarded([&]() -> int {
    if (!out || !n_out) return CSVSIMD_ERR_INVALID_ARG;
    const auto b = csv_simd::boundaries(task_size, job_count);
    if (!b) return CSVSIMD_ERR_INVALID_STATE;
    *n_out = (uint32_t)b->size();
    for (size_t i = 0; i < b->size(); ++i) out[i] = csvsimd_boundary{(*b)[i].start, (*b)[i].len};
    return CSVSIMD_OK;
    });
}

int csvsimd_tape_chunks(const csvsimd_tape* t, uint8_t num, csvsimd_chunk* out, uint32_t* n_out) {
    return csvsimd_guarded([&]() -> int {
    if (!t || !out || !n_out) return CSVSIMD_ERR_INVALID_ARG;
    std::vector<csv_simd::Chunk> c;
    const auto e = t->tape.chunks(num, c);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    *n_out = (uint32_t)c.size();
    for (size_t i = 0; i < c.size(); ++i) out[i] = csvsimd_chunk{c[i].id, c[i].start, c[i].end, c[i].record_cnt};
    return CSVSIMD_OK;
    });
}

const uint64_t* csvsimd_tape_index(const csvsimd_tape* t, uint64_t* index_len) {
    if (index_len) *index_len = t ? t->tape.index().len() : 0;
    return t ? t->tape.index().data : nullptr;
}
const uint8_t* csvsimd_tape_bytes(const csvsimd_tape* t, uint64_t* len) {
    if (len) *len = t ? t->tape.data_len() : 0;
    return t ? t->tape.data_bytes() : nullptr;
}

// csv_simd::create (src/lib.rs:61-74): open, mmap, Header::new, reader::read (GPU), tape.
// The mapping is the tape's data (TapeCore owns the Mmap, src/tape.rs:303) AND what stage 1 reads: the stager's slices touch
// the page-cache pages of a chunk first, in parallel, a few chunks ahead of the link.  Measured (round 5, 2 GiB, page cache
// hot, `profiles/r05_file_ingest.txt`): 40.3 ms, the same as the same bytes in a caller's buffer (40.0); pread by slices
// straight into the pinned slots instead — no page of the mapping touched — 58-61 ms (the kernel's copy reads the
// destination lines for ownership and runs at 65-85 GB/s over eight threads, the streaming copy out of the mapping at
// 90-138: profiles/r05_hostmem_probe.txt).  What round 4 lost was not in the pipeline: 385 ms for the same file, 345 of them
// in a std::vector zero-filled to an entry per 8 bytes of the file in front of it and its munmap behind it.
int csvsimd_create(csvsimd_ctx* ctx, const char* filename, csvsimd_tape** out) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !filename || !out) return CSVSIMD_ERR_INVALID_ARG;
    *out = nullptr;
    const int fd = open(filename, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { g_last_error = std::string("open: ") + strerror(errno); return CSVSIMD_ERR_IO; }
    struct FdCloser {
        int fd;
        ~FdCloser() { close(fd); }
    } closer{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) return CSVSIMD_ERR_IO;
    const uint64_t len = (uint64_t)st.st_size;
    if (len == 0) return CSVSIMD_ERR_IO;  // memmap refuses empty files too
    void* map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { g_last_error = std::string("mmap: ") + strerror(errno); return CSVSIMD_ERR_IO; }
    std::unique_ptr<csvsimd_tape> t(new (std::nothrow) csvsimd_tape);
    if (!t) { munmap(map, len); return CSVSIMD_ERR_INVALID_STATE; }
    t->map = map;
    t->map_len = len;
    const uint8_t* bytes = (const uint8_t*)map;
    csv_simd::Header h;
    csv_simd::StructureError e = csv_simd::Header::create(bytes, len, h);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    // one pass with a capacity guess (an entry per 8 bytes: address space only); one exact retry if the file is denser
    uint64_t n = 0;
    if (!t->owned_index.reserve(len / 8 + 64)) { g_last_error = "mmap (index)"; return CSVSIMD_ERR_INVALID_STATE; }
    int rc = stage1_index_host_impl(ctx, nullptr, bytes, len, t->owned_index.p, t->owned_index.capacity(), &n, nullptr);
    if (rc == CSVSIMD_ERR_TAPE_CAPACITY) {
        if (!t->owned_index.reserve(n)) { g_last_error = "mmap (index)"; return CSVSIMD_ERR_INVALID_STATE; }
        rc = stage1_index_host_impl(ctx, nullptr, bytes, len, t->owned_index.p, t->owned_index.capacity(), &n, nullptr);
    }
    if (rc != CSVSIMD_OK) return rc;
    t->owned_index.shrink_to(n);
    e = csv_simd::Tape::from_core(bytes, len, csv_simd::StructureIndex{t->owned_index.p, n}, std::move(h), t->tape);
    if (e != csv_simd::StructureError::Ok) return (int)e;
    *out = t.release();
    return CSVSIMD_OK;
    });
}

/* ---- utilities ------------------------------------------------------------------------------ */

int csvsimd_synth_fill_device(void* dbuf, uint64_t file_off, uint64_t len, uint32_t cols, uint32_t width,
                              uint64_t seed, uint32_t quote_pct, void* hip_stream) {
    if ((len && !dbuf) || cols == 0 || width == 0 || ((uintptr_t)dbuf & 3)) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    HIP_TRY(csvsimd::launch_synth(dbuf, file_off, len, cols, width, seed, quote_pct, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

int csvsimd_tape_checksum_device(const void* dtape, uint64_t n, uint64_t first_index, void* d_out,
                                 void* hip_stream) {
    if ((n && !dtape) || !d_out) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    HIP_TRY(csvsimd::launch_checksum(dtape, n, first_index, d_out, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

// row_size / record count of a tape (TapeCore::init, src/tape.rs:315-347); CSVSIMD_OK or the reference's error
static int tape_shape(uint64_t index_len, uint32_t field_cnt, int new_line, uint64_t* row_size, uint64_t* record_cnt) {
    if (field_cnt == 0) return CSVSIMD_ERR_INVALID_ARG;
    *row_size = (uint64_t)field_cnt + (new_line == CSVSIMD_NEWLINE_CRLF ? 1 : 0);  // = record jump size
    if (index_len == 0) return CSVSIMD_ERR_INVALID_STATE;
    if ((index_len - 1) % *row_size != 0) return CSVSIMD_ERR_INVALID_CSV_FORMAT;
    *record_cnt = (index_len - 1) / *row_size;  // includes the header row
    return CSVSIMD_OK;
}

// a chunk must consist of whole rows of this tape (Tape::chunks, src/tape.rs:95-140, produces nothing else)
static int chunk_rows(const csvsimd_chunk* c, uint64_t index_len, uint64_t row_size, uint64_t* n_rows) {
    if (!c || c->end < c->start || c->start % row_size || c->end % row_size || c->end > index_len - 1 || c->start == 0)
        return CSVSIMD_ERR_INVALID_ARG;  // start == 0 would be the header row, whose first field the tape cannot address
    *n_rows = (c->end - c->start) / row_size;
    return CSVSIMD_OK;
}

int csvsimd_tape_field_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                    uint32_t field_idx, uint64_t first_record, uint64_t n_records, void* d_begin,
                                    void* d_end, uint64_t* n_valid, void* hip_stream) {
    if (!dindex || !n_valid || (n_records && (!d_begin || !d_end))) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    uint64_t row_size = 0, record_cnt = 0;
    const int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc != CSVSIMD_OK) return rc;
    *n_valid = 0;
    // seek_field: Ok(None) when record_idx + 1 >= record_cnt or field_idx >= field_cnt
    if (field_idx >= field_cnt || first_record + 1 >= record_cnt) return CSVSIMD_OK;
    const uint64_t n = std::min<uint64_t>(n_records, record_cnt - 1 - first_record);
    HIP_TRY(csvsimd::launch_chunk_spans(dindex, (first_record + 1) * row_size, row_size, field_idx, 1, n, d_begin, d_end,
                                        (hipStream_t)hip_stream));
    *n_valid = n;
    return CSVSIMD_OK;
}

int csvsimd_tape_record_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                     uint64_t first_record, uint64_t n_records, void* d_begin, void* d_end,
                                     uint64_t* n_valid, void* hip_stream) {
    if (!dindex || !n_valid || (n_records && (!d_begin || !d_end))) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    uint64_t row_size = 0, record_cnt = 0;
    const int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc != CSVSIMD_OK) return rc;
    *n_valid = 0;
    if (first_record + 1 >= record_cnt) return CSVSIMD_OK;  // seek_record: Ok(None)
    const uint64_t n = std::min<uint64_t>(n_records, record_cnt - 1 - first_record);
    HIP_TRY(csvsimd::launch_chunk_spans(dindex, (first_record + 1) * row_size, row_size, 0, field_cnt, n, d_begin, d_end,
                                        (hipStream_t)hip_stream));
    *n_valid = n;
    return CSVSIMD_OK;
}

int csvsimd_chunk_field_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                     const csvsimd_chunk* chunk, uint32_t field_idx, void* d_begin, void* d_end,
                                     uint64_t* n_records, void* hip_stream) {
    if (!dindex || !n_records) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    uint64_t row_size = 0, record_cnt = 0, n = 0;
    int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc == CSVSIMD_OK) rc = chunk_rows(chunk, index_len, row_size, &n);
    if (rc != CSVSIMD_OK) return rc;
    if (field_idx >= field_cnt || (n && (!d_begin || !d_end))) return CSVSIMD_ERR_INVALID_ARG;
    HIP_TRY(csvsimd::launch_chunk_spans(dindex, chunk->start, row_size, field_idx, 1, n, d_begin, d_end,
                                        (hipStream_t)hip_stream));
    *n_records = n;
    return CSVSIMD_OK;
}

int csvsimd_gather_fields_device(const void* dbytes, uint64_t bytes_len, const void* d_begin, const void* d_end,
                                 uint64_t n_records, void* d_dst, uint32_t stride, void* d_len, void* hip_stream) {
    if ((n_records && (!dbytes || !d_begin || !d_end || !d_dst)) || stride == 0) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    HIP_TRY(csvsimd::launch_gather_fields(dbytes, bytes_len, d_begin, d_end, n_records, d_dst, stride, d_len,
                                          (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

// scratch of csvsimd_column_frequency_device: [64 B: longest field | chunk map | lengths n x 4 | the count's own scratch | its
// status 64 B | column n x stride]
namespace {
struct FreqLayout {
    uint64_t off_map, off_len, off_cf, off_status, off_col, total;
};
FreqLayout freq_layout(uint64_t n, uint32_t n_chunks, uint64_t stride) {
    auto up = [](uint64_t v) { return (v + 255) & ~(uint64_t)255; };
    FreqLayout L;
    L.off_map = 64;
    L.off_len = up(L.off_map + (uint64_t)n_chunks * sizeof(csvsimd::FreqRowMap));
    L.off_cf = up(L.off_len + n * 4);
    L.off_status = up(L.off_cf + csvsimd::colfreq_scratch_bytes(n));
    L.off_col = up(L.off_status + 64);  // the only part whose size depends on the stride: last
    L.total = L.off_col + n * stride;
    return L;
}
uint64_t stride_for(uint64_t max_field_bytes) { return std::max<uint64_t>(16, (max_field_bytes + 15) & ~(uint64_t)15); }
}  // namespace

uint64_t csvsimd_column_frequency_scratch_bytes(uint64_t n_records, uint32_t n_chunks, uint64_t max_field_bytes) {
    return freq_layout(n_records, n_chunks ? n_chunks : 1, stride_for(max_field_bytes)).total;
}

int csvsimd_column_frequency_device(csvsimd_ctx* ctx, const void* dbytes, const void* dindex, uint64_t index_len,
                                    uint32_t field_cnt, int new_line, const csvsimd_chunk* chunks, uint32_t n_chunks,
                                    uint32_t field_idx, void* d_scratch, uint64_t scratch_bytes, void* d_entries,
                                    uint64_t entries_cap, csvsimd_freq_status* status, void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !dbytes || !dindex || !chunks || !n_chunks || !d_scratch || !status || (entries_cap && !d_entries))
        return CSVSIMD_ERR_INVALID_ARG;
    if (((uintptr_t)d_scratch & 255) || ((uintptr_t)d_entries & 7)) return CSVSIMD_ERR_INVALID_ARG;
    uint64_t row_size = 0, record_cnt = 0;
    int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc != CSVSIMD_OK) return rc;
    if (field_idx >= field_cnt) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    std::vector<uint64_t> rows(n_chunks);
    using RowMap = csvsimd::FreqRowMap;
    std::vector<RowMap> map(n_chunks);
    uint64_t n = 0;
    for (uint32_t i = 0; i < n_chunks; ++i) {
        if ((rc = chunk_rows(&chunks[i], index_len, row_size, &rows[i])) != CSVSIMD_OK) return rc;
        map[i] = RowMap{n, chunks[i].start / row_size - 1, chunks[i].start};  // seek_field numbering: the header row is not a record
        n += rows[i];
    }
    if (n > (1ull << 28)) return CSVSIMD_ERR_INVALID_ARG;  // (see csvsimd_columnar_frequency_device_async: at most 2^28 records per count)
    static_assert(sizeof(csvsimd_freq_status) == 32 && sizeof(csvsimd_freq_entry) == 32, "layouts shared with the kernels");
    *status = csvsimd_freq_status{n, 0, 0, 0};
    if (n == 0) return CSVSIMD_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    // The column is gathered at the LARGEST stride the scratch holds (the caller sized it for the longest field it expects,
    // csvsimd_column_frequency_scratch_bytes), without asking the device first how long the longest field really is: the
    // gather reports that on the side, and the count reports records that did not fit.  One synchronisation, at the end
    // (rounds 1-3 and the first version of this path waited for the longest field before they gathered).
    const FreqLayout L0 = freq_layout(n, n_chunks, 0);
    if (scratch_bytes < L0.total + n * 16) return CSVSIMD_ERR_TAPE_CAPACITY;  // not even 16-byte rows (status->max_field_bytes stays 0)
    // ... but never wider than the documented 4 096 bytes: a caller that hands over one large arena for everything would
    // otherwise have every record gathered and hashed as a row of KiB or MiB
    const uint64_t stride = std::min<uint64_t>(((scratch_bytes - L0.total) / n) & ~(uint64_t)15, 4096);
    const FreqLayout L = freq_layout(n, n_chunks, stride);
    char* const base = (char*)d_scratch;
    HIP_TRY(hipMemsetAsync(base, 0, 64, s));
    rc = ctx_upload(ctx, base + L.off_map, map.data(), map.size() * sizeof(RowMap), s);
    if (rc != CSVSIMD_OK) return rc;
    for (uint32_t i = 0; i < n_chunks; ++i)
        HIP_TRY(csvsimd::launch_gather_column(dbytes, dindex, index_len, chunks[i].start, row_size, field_idx, rows[i],
                                              base + L.off_col + map[i].row0 * stride, (uint32_t)stride,
                                              base + L.off_len + map[i].row0 * 4, base, s));
    // the count's second pass writes the entries in their final form: record id through the chunk map, text span from the tape
    const csvsimd::FreqWideOut wide = {(const uint64_t*)dindex, row_size, field_idx, n_chunks, (const RowMap*)(base + L.off_map)};
    HIP_TRY(csvsimd::launch_colfreq(base + L.off_col, base + L.off_len, n, (uint32_t)stride, 0, base + L.off_cf, d_entries,
                                    entries_cap, base + L.off_status, ctx->n_cus, s, &wide));
    HIP_TRY(hipMemcpyAsync((char*)ctx->h_small + 64, base + L.off_status, sizeof(csvsimd_colfreq_status), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ctx->h_small, base, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const csvsimd_colfreq_status cs = *reinterpret_cast<const csvsimd_colfreq_status*>((const char*)ctx->h_small + 64);
    status->max_field_bytes = *reinterpret_cast<const volatile uint64_t*>(ctx->h_small);
    if (cs.truncated) return CSVSIMD_ERR_TAPE_CAPACITY;  // fields longer than the scratch allows: size it from status->max_field_bytes
    status->n_distinct = cs.n_distinct;
    status->overflow = cs.overflow;
    if (cs.overflow) return CSVSIMD_ERR_TAPE_CAPACITY;
    if (cs.n_distinct > entries_cap) return CSVSIMD_ERR_TAPE_CAPACITY;  // status->n_distinct = the size needed
    return CSVSIMD_OK;
    });
}

int csvsimd_column_search_device(csvsimd_ctx* ctx, const void* dbytes, uint64_t bytes_len, const void* dindex, uint64_t index_len,
                                 uint32_t field_cnt, int new_line, const csvsimd_chunk* chunk, uint32_t field_idx,
                                 const void* needle, uint32_t needle_len, int mode, void* d_bitmap, uint64_t* n_matches,
                                 void* hip_stream) {
    if (!ctx || !dbytes || !dindex || !n_matches || (needle_len && !needle) || needle_len > 256) return CSVSIMD_ERR_INVALID_ARG;
    if (mode != CSVSIMD_SEARCH_EQUALS && mode != CSVSIMD_SEARCH_STARTS_WITH && mode != CSVSIMD_SEARCH_CONTAINS)
        return CSVSIMD_ERR_INVALID_ARG;
    uint64_t row_size = 0, record_cnt = 0, n = 0;
    int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc == CSVSIMD_OK) rc = chunk_rows(chunk, index_len, row_size, &n);
    if (rc != CSVSIMD_OK) return rc;
    if (field_idx >= field_cnt || (n && (!d_bitmap || ((uintptr_t)d_bitmap & 7)))) return CSVSIMD_ERR_INVALID_ARG;
    *n_matches = 0;
    if (n == 0) return CSVSIMD_OK;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    // the needle and the match counter live in the context's small device block: [0, 8) counter, [64, 64 + 264) needle
    HIP_TRY(hipMemsetAsync(ctx->d_small, 0, 8, s));
    if (needle_len) HIP_TRY(hipMemcpyAsync((char*)ctx->d_small + 64, needle, needle_len, hipMemcpyHostToDevice, s));
    HIP_TRY(csvsimd::launch_search(dbytes, bytes_len, dindex, chunk->start, row_size, n, field_idx, (char*)ctx->d_small + 64, needle_len,
                                   mode, d_bitmap, ctx->d_small, s));
    HIP_TRY(hipMemcpyAsync(n_matches, ctx->d_small, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return CSVSIMD_OK;
}

uint64_t csvsimd_bitmap_select_scratch_bytes(uint64_t n_rows) { return ((n_rows + 4095) / 4096 + 2) * 8; }

int csvsimd_bitmap_select_device(const void* d_bitmap, uint64_t n_rows, uint64_t first_record, void* d_scratch, void* d_out,
                                 uint64_t out_cap, uint64_t* n_out, void* hip_stream) {
    if (!n_out || (n_rows && (!d_bitmap || !d_scratch)) || (out_cap && !d_out) || ((uintptr_t)d_scratch & 7))
        return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    *n_out = 0;
    if (n_rows == 0) return CSVSIMD_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    const uint64_t n_blocks = (n_rows + 4095) / 4096;
    uint64_t* const d_total = (uint64_t*)d_scratch + n_blocks;  // behind the per-block counts
    HIP_TRY(csvsimd::launch_bitmap_select(d_bitmap, n_rows, first_record, d_scratch, d_out, out_cap, d_total, s));
    HIP_TRY(hipMemcpyAsync(n_out, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return *n_out > out_cap ? CSVSIMD_ERR_TAPE_CAPACITY : CSVSIMD_OK;
}

/* ---- columnar consumers (columnar_kernels.hip) ------------------------------------------------------------------ */

int csvsimd_chunk_to_columns_device(csvsimd_ctx* ctx, const void* dbytes, uint64_t bytes_len, const void* dindex,
                                    uint64_t index_len, uint32_t field_cnt, int new_line, const csvsimd_chunk* chunk,
                                    const uint32_t* fields, uint32_t n_fields, void* d_cols, uint32_t stride, void* d_lens,
                                    uint64_t* n_records, void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !dbytes || !dindex || !n_records) return CSVSIMD_ERR_INVALID_ARG;
    uint64_t row_size = 0, record_cnt = 0, n = 0;
    int rc = tape_shape(index_len, field_cnt, new_line, &row_size, &record_cnt);
    if (rc == CSVSIMD_OK) rc = chunk_rows(chunk, index_len, row_size, &n);
    if (rc != CSVSIMD_OK) return rc;
    if (!fields && n_fields == 0) n_fields = field_cnt;  // all columns
    if (n_fields == 0 || n_fields > 1024 || stride == 0 || (stride & 15u) || stride > 4096) return CSVSIMD_ERR_INVALID_ARG;
    if (n && (!d_cols || ((uintptr_t)d_cols & 15) || ((uintptr_t)d_lens & 3))) return CSVSIMD_ERR_INVALID_ARG;
    if (n >= 0xffffffffull) return CSVSIMD_ERR_INVALID_ARG;  // record ids of a chunk are 32-bit (Tape.record_cnt, src/tape.rs:76)
    if (fields)
        for (uint32_t i = 0; i < n_fields; ++i)
            if (fields[i] >= field_cnt) return CSVSIMD_ERR_INVALID_ARG;
    if (!fields && n_fields > field_cnt) return CSVSIMD_ERR_INVALID_ARG;
    *n_records = n;
    if (n == 0) return CSVSIMD_OK;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    void* d_fields = nullptr;
    if (fields) {
        d_fields = (char*)ctx->d_small + 1024;
        rc = ctx_upload(ctx, d_fields, fields, n_fields * sizeof(uint32_t), s);
        if (rc != CSVSIMD_OK) return rc;
    }
    // rows staged per workgroup step: what fits the window on average, with a fifth of it to spare for longer rows
    const uint64_t avg_row = std::max<uint64_t>(1, bytes_len / std::max<uint64_t>(1, record_cnt));
    const uint64_t r = std::max<uint64_t>(1, (uint64_t)csvsimd::to_columns_window_bytes() * 4 / 5 / avg_row);
    HIP_TRY(csvsimd::launch_to_columns(dbytes, bytes_len, dindex, chunk->start, row_size, n, d_fields, n_fields, d_cols,
                                       stride, d_lens, (uint32_t)std::min<uint64_t>(r, 4096), ctx->n_cus, s));
    return CSVSIMD_OK;
    });
}

uint64_t csvsimd_columnar_frequency_scratch_bytes(uint64_t n_records) { return csvsimd::colfreq_scratch_bytes(n_records); }

int csvsimd_columnar_frequency_device_async(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                            uint32_t stride, uint64_t first_record, void* d_scratch, uint64_t scratch_bytes,
                                            void* d_entries, uint64_t entries_cap, void* d_status, void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !d_scratch || !d_status || (n_records && !d_col) || (entries_cap && !d_entries)) return CSVSIMD_ERR_INVALID_ARG;
    if (((uintptr_t)d_scratch & 15) || ((uintptr_t)d_entries & 7) || ((uintptr_t)d_status & 7)) return CSVSIMD_ERR_INVALID_ARG;
    // n_records: one call counts up to 2^28 records (268 M).  The second pass cuts the hash space into at most 4 096
    // partitions and a partition's tuples are merged 6 144 at a time, every round re-reading the partition's tuple run: all
    // distinct, 2^28 records are 11 rounds per partition; towards the 2^32 the record ids allow the re-reads would grow
    // quadratically.  Larger columns: count slices of <= 2^28 records and merge the entry lists.
    if (stride == 0 || (stride & 15u) || stride > 4096 || ((uintptr_t)d_col & 15) || ((uintptr_t)d_len & 3) ||
        n_records > (1ull << 28))
        return CSVSIMD_ERR_INVALID_ARG;
    if (scratch_bytes < csvsimd::colfreq_scratch_bytes(n_records)) return CSVSIMD_ERR_TAPE_CAPACITY;
    static_assert(sizeof(csvsimd_colfreq_status) == 32 && sizeof(csvsimd_colfreq_entry) == 16, "layouts shared with the kernels");
    WITH_DEVICE_OF(ctx);
    HIP_TRY(csvsimd::launch_colfreq(d_col, d_len, n_records, stride, first_record, d_scratch, d_entries, entries_cap, d_status,
                                    ctx->n_cus, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
    });
}

int csvsimd_columnar_frequency_device(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                      uint32_t stride, uint64_t first_record, void* d_scratch, uint64_t scratch_bytes,
                                      void* d_entries, uint64_t entries_cap, csvsimd_colfreq_status* status,
                                      void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !status) return CSVSIMD_ERR_INVALID_ARG;
    // the status record of the synchronous form lives in the context's small device block ([512, 544): nothing else uses it)
    void* const d_status = (char*)ctx->d_small + 512;
    const int rc = csvsimd_columnar_frequency_device_async(ctx, d_col, d_len, n_records, stride, first_record, d_scratch,
                                                           scratch_bytes, d_entries, entries_cap, d_status, hip_stream);
    if (rc != CSVSIMD_OK) return rc;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(hipMemcpyAsync((char*)ctx->h_small + 128, d_status, sizeof(*status), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(status, (const char*)ctx->h_small + 128, sizeof(*status));
    if (status->overflow) return CSVSIMD_ERR_TAPE_CAPACITY;   // more values with equal hash bits than a partition's table holds
    if (status->truncated) return CSVSIMD_ERR_TAPE_CAPACITY;  // values longer than the stride: counts would merge them
    if (status->n_distinct > entries_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
    });
}

int csvsimd_columnar_search_device(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                   uint32_t stride, const void* needle, uint32_t needle_len, int mode, void* d_bitmap,
                                   uint64_t* n_matches, void* hip_stream) {
    if (!ctx || !n_matches || (n_records && (!d_col || !d_bitmap)) || (needle_len && !needle) || needle_len > 256)
        return CSVSIMD_ERR_INVALID_ARG;
    if (mode != CSVSIMD_SEARCH_EQUALS && mode != CSVSIMD_SEARCH_STARTS_WITH && mode != CSVSIMD_SEARCH_CONTAINS)
        return CSVSIMD_ERR_INVALID_ARG;
    if (stride == 0 || (stride & 15u) || stride > 4096 || ((uintptr_t)d_col & 15) || ((uintptr_t)d_len & 3) ||
        ((uintptr_t)d_bitmap & 7))
        return CSVSIMD_ERR_INVALID_ARG;
    *n_matches = 0;
    if (n_records == 0) return CSVSIMD_OK;
    if (n_records >> 36) return CSVSIMD_ERR_INVALID_ARG;  // (the kernel's packed result word counts matches in 36 bits)
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    // One launch and the wait (round 5; before: a memset, the needle's copy from pageable memory, the launch, a 16-byte copy back
    // and the wait — ~25 us around the kernel): the needle rides in the kernel's arguments, the kernel's last workgroup writes
    // one word — this call's number, "a record was longer than the stride", the matches — to pinned memory and leaves its
    // device word at zero.
    volatile uint64_t* const rec = reinterpret_cast<volatile uint64_t*>((char*)ctx->h_small + 192);
    void* rec_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&rec_dev, (void*)rec, 0));
    const uint64_t seq = (++ctx->search_seq) & 0xffffu;
    HIP_TRY(csvsimd::launch_colsearch(d_col, d_len, n_records, stride, needle, needle_len, mode, d_bitmap, (char*)ctx->d_small + 16,
                                      rec_dev, seq, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint64_t w = *rec;
    if ((w >> 48) != seq) {
        g_last_error = "columnar search: the result was not published";
        return CSVSIMD_ERR_INTERNAL;
    }
    *n_matches = w & ((1ull << 47) - 1);
    return ((w >> 47) & 1) ? CSVSIMD_ERR_TAPE_CAPACITY : CSVSIMD_OK;  // truncated records: the match is only known for their first `stride` bytes
}

int csvsimd_trim_spans_device(const void* dbytes, void* d_begin, void* d_end, uint64_t n_records, uint32_t flags,
                              uint8_t quote, void* hip_stream) {
    if (n_records && (!dbytes || !d_begin || !d_end)) return CSVSIMD_ERR_INVALID_ARG;
    if (flags & ~(CSVSIMD_TRIM_SPACE | CSVSIMD_TRIM_QUOTES)) return CSVSIMD_ERR_INVALID_ARG;
    if (csvsimd_device_count() <= 0) return CSVSIMD_ERR_NO_DEVICE;
    HIP_TRY(csvsimd::launch_trim_spans(dbytes, d_begin, d_end, n_records, flags, quote, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

int csvsimd_utf8_validate_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* d_result,
                                       void* hip_stream) {
    if (!ctx || !d_result || (len && !dbuf) || ((uintptr_t)d_result & 7)) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    HIP_TRY(csvsimd::launch_utf8_validate(dbuf, len, d_result, ctx->n_cus, (hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

int csvsimd_utf8_validate_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, csvsimd_utf8_result* result,
                                 void* hip_stream) {
    if (!ctx || !result) return CSVSIMD_ERR_INVALID_ARG;
    static_assert(sizeof(csvsimd_utf8_result) <= sizeof(csvsimd_shard_result), "reuses the context's result slot");
    WITH_DEVICE_OF(ctx);
    const int rc = csvsimd_utf8_validate_device_async(ctx, dbuf, len, ctx->d_result, hip_stream);
    if (rc != CSVSIMD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(result, ctx->d_result, sizeof(*result), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    return CSVSIMD_OK;
}

int csvsimd_selftest_device(int device) {
    const int n = csvsimd_device_count();
    if (n <= 0) return CSVSIMD_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return CSVSIMD_ERR_INVALID_ARG;
    ScopedDevice scoped_device_(device);
    HIP_TRY(scoped_device_.error());
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 8));
    HIP_TRY(hipMemset(d, 0, 8));
    hipError_t e = csvsimd::launch_selftest(d, nullptr);
    uint32_t h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "selftest");
    if (h[0] || h[1]) {
        char buf[96];
        snprintf(buf, sizeof buf, "wavefront self-test failed: bits 0x%x / 0x%x", h[0], h[1]);
        g_last_error = buf;
        return CSVSIMD_ERR_INTERNAL;
    }
    return CSVSIMD_OK;
}

namespace {
// owns a batch of timing events: destroyed on every exit path
struct EventBatch {
    std::vector<hipEvent_t> ev;
    ~EventBatch() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    hipError_t create(size_t n) {
        ev.assign(n, nullptr);
        for (auto& e : ev) {
            const hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }
};
}  // namespace

int csvsimd_stage1_time_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* dtape, uint64_t tape_cap,
                               void* d_result, void* hip_stream, int warmup, int iters, float* avg_ms) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !avg_ms || iters <= 0 || iters > 4096 || !d_result || !len) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    int rc = csvsimd_ctx_reserve(ctx, len);
    if (rc != CSVSIMD_OK) return rc;
    csvsimd::Stage1Launch L;
    L.dbuf = dbuf;
    L.len = len;
    L.base_off = 0;
    L.in_quote_in = 0;
    L.dtape = dtape;
    L.tape_cap = tape_cap;
    L.d_result = (csvsimd_shard_result*)d_result;
    L.bind_scratch(ctx->scratch);
    L.max_blocks = ctx->max_blocks;
    L.dense = dtape && ctx->density > csvsimd_ctx::kDenseThreshold;
#ifdef CSVSIMD_DEV_PROBES
    // development builds only (libcsvsimd_probes.so for scripts/probe*.py): the product library has neither
    // these hooks nor the kernel instantiations behind them
    if (const char* dbg = getenv("CSVSIMD_PROBE_MODE")) {
        L.debug_mode = atoi(dbg);
        if (const char* mb = getenv("CSVSIMD_PROBE_BLOCKS_PER_CU")) {
            const int n = atoi(mb);
            if (n >= 1 && n <= 16) L.max_blocks = (uint32_t)ctx->n_cus * (uint32_t)n;
        }
    }
    if (const char* dia = getenv("CSVSIMD_PROBE_DIALECT")) {  // "delimiter,quote,escape" as decimal bytes
        unsigned d = ',', q = '"', e = 0;
        if (sscanf(dia, "%u,%u,%u", &d, &q, &e) >= 1) {
            L.delimiter = (uint8_t)d;
            L.quote = (uint8_t)q;
            L.escape = (uint8_t)e;
        }
    }
    if (getenv("CSVSIMD_PROBE_NO_HASHED_DIALECT")) L.allow_hashed_dialect = false;  // escape dialects: force the direct compares
    if (const char* e = getenv("CSVSIMD_PROBE_CU_TOKEN")) L.pace_cu_token = atoi(e);
    if (const char* e = getenv("CSVSIMD_PROBE_EMIT_DELAY")) L.pace_emit_delay = atoi(e);
    if (const char* e = getenv("CSVSIMD_PROBE_COUNT_PRIO")) L.pace_count_prio = atoi(e);
    if (L.debug_mode == 8) HIP_TRY(hipMemsetAsync(L.scratch_prof, 0, 17 * 8, s));
    uint64_t* d_trace = nullptr;
    const uint64_t trace_tiles = (len + 127 + CSVSIMD_TILE_BYTES - 1) / CSVSIMD_TILE_BYTES + 1;
    if (L.debug_mode == 40 || L.debug_mode == 56) {  // per-tile timeline (scripts/trace_tiles.py): its own buffer, slots as in CSVSIMD_TRACE
        HIP_TRY(hipMalloc((void**)&d_trace, (32 + trace_tiles * 16) * 8));   // two 8-slot records per tile
        HIP_TRY(hipMemset(d_trace, 0, (32 + trace_tiles * 16) * 8));
        L.scratch_prof = d_trace;
    }
#endif
    ctx->last_stream = s;
    ctx->launched = true;
    for (int i = 0; i < warmup; ++i) HIP_TRY(csvsimd::launch_stage1(L, s));
#ifndef CSVSIMD_DEV_PROBES
    if (warmup > 0 && dtape) {
        // the timed launches run what a caller's launches run once the context has seen this data: the warm-up's record
        // says how dense it is (csvsimd_stage1_index_device learns it the same way)
        csvsimd_shard_result* const hr = (csvsimd_shard_result*)ctx->h_small;
        HIP_TRY(hipMemcpyAsync(hr, d_result, sizeof(*hr), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (!hr->error && len >= (1u << 16)) ctx->density = (double)hr->count / (double)len;
        L.dense = ctx->density > csvsimd_ctx::kDenseThreshold;
        for (int i = 0; i < std::min(warmup, 4); ++i) HIP_TRY(csvsimd::launch_stage1(L, s));
    }
#endif
    // one event pair per launch, recorded on the launch stream right around the stage-1 kernel (a launch
    // IS that one kernel)
    EventBatch eb;
    HIP_TRY(eb.create(2 * (size_t)iters));
    for (int i = 0; i < iters; ++i) {
        L.ev_begin = eb.ev[2 * i];
        L.ev_end = eb.ev[2 * i + 1];
        HIP_TRY(csvsimd::launch_stage1(L, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    double total = 0;
    for (int i = 0; i < iters; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, eb.ev[2 * i], eb.ev[2 * i + 1]));
        total += ms;
    }
    *avg_ms = (float)(total / iters);
#ifdef CSVSIMD_DEV_PROBES
    if (d_trace) {
        std::vector<uint64_t> h(32 + trace_tiles * 16);   // the second records start 8 * num_tiles words in
        HIP_TRY(hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        const char* path = getenv("CSVSIMD_PROBE_TRACE");
        if (FILE* f = fopen(path ? path : "/tmp/csvsimd_trace.bin", "wb")) {
            fwrite(h.data() + 32, 8, trace_tiles * 16, f);
            fclose(f);
        }
    }
    if (L.debug_mode == 8) {  // print the per-phase stamps (summed over warm-up + timed launches)
        uint64_t h[17];
        HIP_TRY(hipMemcpy(h, L.scratch_prof, sizeof h, hipMemcpyDeviceToHost));
        const double nwg = h[16] ? (double)h[16] : 1.0;
        fprintf(stderr, "PROF %llu workgroup runs over %d launches (normalising by the former)\n",
                (unsigned long long)h[16], warmup + iters);
        static const char* names[6] = {"ticket+barrier T", "count phase", "barrier A", "publish+resolve", "barrier B", "emit"};
        for (int wv = 0; wv < 2; ++wv)
            for (int k = 0; k < 6; ++k)
                fprintf(stderr, "PROF wave%d %-16s %.2f us per workgroup (sum over its tiles)\n", wv, names[k],
                        (double)h[wv * 8 + k] / nwg / 100.0);
    }
#endif
    return CSVSIMD_OK;
    });
}

const char* csvsimd_stage1_kernel_name(int emit, const csvsimd_dialect* dialect) {
    int d = 0;
    if (dialect) d = dialect->escape ? 2 : (dialect->delimiter != ',' || dialect->quote != '"') ? 1 : 0;
    csvsimd::DialectHash dh;
    if (d == 2 && csvsimd::dialect_hash(dialect->delimiter, dialect->quote, dialect->escape, dh)) d = 3;
    return csvsimd::stage1_kernel_name(emit != 0, d);
}

uint32_t csvsimd_build_has_probes(void) {
#ifdef CSVSIMD_DEV_PROBES
    return 1;
#else
    return 0;
#endif
}

int csvsimd_hbm_probe_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* dout, int write_div,
                             int blocks_per_cu, void* hip_stream, int warmup, int iters, float* avg_ms) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !dbuf || !avg_ms || iters <= 0 || iters > 4096 || ((uintptr_t)dbuf & 15)) return CSVSIMD_ERR_INVALID_ARG;
    if ((write_div != 0 && write_div != 4 && write_div != 25) || blocks_per_cu < 1 || blocks_per_cu > 8)
        return CSVSIMD_ERR_INVALID_ARG;
    if (!dout || ((uintptr_t)dout & 15) || len < 131072) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    int rc = csvsimd_ctx_reserve(ctx, 1 << 20);  // the probe's ticket pair lives in the context's control block
    ctx->last_stream = s;
    ctx->launched = true;
    if (rc != CSVSIMD_OK) return rc;
    const uint32_t blocks = (uint32_t)ctx->n_cus * (uint32_t)blocks_per_cu;  // 4-wave workgroups: 4 per CU = the stage-1 kernel's 16 waves
    for (int i = 0; i < warmup; ++i)
        HIP_TRY(csvsimd::launch_hbm_probe(dbuf, len, dout, write_div, ctx->scratch, blocks, s));
    EventBatch eb;
    HIP_TRY(eb.create(2));
    HIP_TRY(hipEventRecord(eb.ev[0], s));
    for (int i = 0; i < iters; ++i)
        HIP_TRY(csvsimd::launch_hbm_probe(dbuf, len, dout, write_div, ctx->scratch, blocks, s));
    HIP_TRY(hipEventRecord(eb.ev[1], s));
    HIP_TRY(hipEventSynchronize(eb.ev[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, eb.ev[0], eb.ev[1]));
    *avg_ms = ms / (float)iters;
    return CSVSIMD_OK;
    });
}

int csvsimd_copy_probe_device(csvsimd_ctx* ctx, const void* dsrc, void* ddst, uint64_t len, int mode, void* hip_stream,
                              int warmup, int iters, float* avg_ms) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !dsrc || !ddst || !avg_ms || iters <= 0 || iters > 4096 || len < 16) return CSVSIMD_ERR_INVALID_ARG;
    if (mode < 0 || mode > 2 || ((uintptr_t)dsrc & 15) || ((uintptr_t)ddst & 15)) return CSVSIMD_ERR_INVALID_ARG;
    WITH_DEVICE_OF(ctx);
    hipStream_t s = (hipStream_t)hip_stream;
    for (int i = 0; i < warmup; ++i) HIP_TRY(csvsimd::launch_copy_probe(dsrc, ddst, len, mode, s));
    EventBatch eb;
    HIP_TRY(eb.create(2));
    HIP_TRY(hipEventRecord(eb.ev[0], s));
    for (int i = 0; i < iters; ++i) HIP_TRY(csvsimd::launch_copy_probe(dsrc, ddst, len, mode, s));
    HIP_TRY(hipEventRecord(eb.ev[1], s));
    HIP_TRY(hipEventSynchronize(eb.ev[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, eb.ev[0], eb.ev[1]));
    *avg_ms = ms / (float)iters;
    return CSVSIMD_OK;
    });
}

}  // extern "C"
