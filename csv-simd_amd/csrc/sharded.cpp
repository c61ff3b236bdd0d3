// sharded.cpp — native multi-GPU stage 1: one rank per GPU, one RCCL all-gather over xGMI.
//
// The C++ twin of csv-simd_amd/sharded.py for hosts without torch (a Rust or C++ caller).  New
// relative to the reference (single-threaded; its README.md:24 lists "splitting work without first
// knowing record breaks" as a TODO): the two values the reference carries between 64-byte blocks
// (`inside_str`, `array_idx`, src/reader.rs:217-218) are carried between GPUs by ONE all-gather of
// the 64-byte csvsimd_shard_result per rank, then the stitch on the device (stage1_kernels.hip:
// stitch_kernel; csvsimd_stitch_shards is its host twin).
//
// RCCL is resolved at run time (dlopen): libcsvsimd_hip.so itself has no link-time dependency on it,
// and inside a PyTorch process it binds to the librccl that torch already loaded.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "abi_guard.h"
#include "csvsimd.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)  // prefer a copy that is already mapped (torch's)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.handle) { api.error = "librccl.so not found"; return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather)
            api.error = "librccl.so lacks an expected symbol";
    });
    return api;
}

// the caller's current HIP device is put back on every way out (capi.cpp: ScopedDevice)
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct csvsimd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    csvsimd_shard_result* d_mine = nullptr;  // device: this rank's record (speculative, then final)
    csvsimd_shard_result* d_all = nullptr;   // device: world records, rank order
    csvsimd_stitch* d_stitch = nullptr;      // device: this rank's stitch (written by stitch_kernel)
    // pinned host block: [0] = final record of this rank, [1] = a record with only the error flag set
    // (what a rank whose local launch failed contributes to the all-gather), then the stitch
    csvsimd_shard_result* h_rec = nullptr;
    csvsimd_stitch* h_stitch = nullptr;
};

extern "C" {

int csvsimd_comm_unique_id(uint8_t id[CSVSIMD_COMM_ID_BYTES]) {
    return csvsimd_guarded([&]() -> int {
    static_assert(CSVSIMD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size must match RCCL's");
    if (!id) return CSVSIMD_ERR_INVALID_ARG;
    RcclApi& api = rccl();
    if (!api.error.empty()) return CSVSIMD_ERR_RCCL;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return CSVSIMD_ERR_RCCL;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return CSVSIMD_OK;
    });
}

int csvsimd_comm_create(const uint8_t id[CSVSIMD_COMM_ID_BYTES], int rank, int world, int device, csvsimd_comm** out) {
    return csvsimd_guarded([&]() -> int {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return CSVSIMD_ERR_INVALID_ARG;
    *out = nullptr;
    RcclApi& api = rccl();
    if (!api.error.empty()) return CSVSIMD_ERR_RCCL;
    DeviceRestore restore_;
    if (hipSetDevice(device) != hipSuccess) return CSVSIMD_ERR_HIP;
    csvsimd_comm* c = new (std::nothrow) csvsimd_comm;
    if (!c) return CSVSIMD_ERR_INVALID_STATE;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    bool ok = api.CommInitRank(&c->comm, world, u, rank) == ncclSuccess;
    ok = ok && hipMalloc((void**)&c->d_mine, sizeof(csvsimd_shard_result)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_all, sizeof(csvsimd_shard_result) * (size_t)world) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_stitch, sizeof(csvsimd_stitch)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_rec, 2 * sizeof(csvsimd_shard_result), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_stitch, sizeof(csvsimd_stitch), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        csvsimd_comm_destroy(c);
        return CSVSIMD_ERR_RCCL;
    }
    memset(c->h_rec, 0, 2 * sizeof(csvsimd_shard_result));
    c->h_rec[1].error = 1;
    *out = c;
    return CSVSIMD_OK;
    });
}

void csvsimd_comm_destroy(csvsimd_comm* c) {
    if (!c) return;
    DeviceRestore restore_;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->d_mine) (void)hipFree(c->d_mine);
    if (c->d_all) (void)hipFree(c->d_all);
    if (c->d_stitch) (void)hipFree(c->d_stitch);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_stitch) (void)hipHostFree(c->h_stitch);
    delete c;
}

// One sharded step for this rank, entirely stream-ordered on the device:
//   speculative pass (entering state: known on rank 0, the kernel's own guess elsewhere) -> ONE all-gather of the
//   64-byte result records over xGMI -> stitch kernel (one lane: this rank's true entering state, tape index base,
//   totals, "re-emit" flag, in device memory) -> re-emit launch that reads state and flag from device memory and
//   returns at once unless the flag is set -> the final record and the stitch travel to the host; ONE synchronisation at the
//   very end.  Nothing in between waits for the host, so the whole step can overlap the next one's
//   speculative pass or be captured into a hipGraph.
// Every rank reaches the collective whatever happens locally: arguments are checked and the scratch
// is reserved before anything is enqueued, and a rank whose own launch fails contributes a record
// with the error flag set, so all ranks return an error instead of hanging in the all-gather.
int csvsimd_stage1_index_sharded(csvsimd_ctx* ctx, csvsimd_comm* c, const void* dbuf, uint64_t len, uint64_t base_off,
                                 uint32_t file_in_quote_in, void* dtape, uint64_t tape_cap,
                                 csvsimd_shard_result* result, csvsimd_stitch* stitch, void* hip_stream) {
    return csvsimd_guarded([&]() -> int {
    if (!ctx || !c || !result || !stitch) return CSVSIMD_ERR_INVALID_ARG;
    DeviceRestore restore_;
    if (hipSetDevice(c->device) != hipSuccess) return CSVSIMD_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;
    int local = csvsimd_ctx_reserve(ctx, len);  // may allocate: never inside the stream-ordered part
    if (local == CSVSIMD_OK)
        // rank 0 knows how the file starts; every other rank lets the kernel choose the entering state its first
        // tile speaks for (CSVSIMD_ENTER_GUESS) — the stitch tells who chose wrong, and only those re-emit
        local = csvsimd_stage1_index_device_async(ctx, dbuf, len, base_off,
                                                  c->rank == 0 ? (file_in_quote_in ? 1u : 0u) : CSVSIMD_ENTER_GUESS,
                                                  dtape, tape_cap, c->d_mine, st);
    if (local != CSVSIMD_OK &&
        hipMemcpyAsync(c->d_mine, &c->h_rec[1], sizeof(csvsimd_shard_result), hipMemcpyHostToDevice, st) != hipSuccess)
        return CSVSIMD_ERR_HIP;  // cannot even tell the peers: nothing more to be done here
    if (rccl().AllGather(c->d_mine, c->d_all, sizeof(csvsimd_shard_result), ncclUint8, c->comm, st) != ncclSuccess)
        return CSVSIMD_ERR_RCCL;
    int rc = csvsimd_stitch_shards_device_async(c->d_all, (uint32_t)c->world, (uint32_t)c->rank, file_in_quote_in,
                                                c->d_stitch, st);
    if (rc == CSVSIMD_OK && local == CSVSIMD_OK)
        rc = csvsimd_stage1_reemit_device_async(ctx, dbuf, len, base_off, c->d_stitch, dtape, tape_cap, c->d_mine, st);
    if (hipMemcpyAsync(&c->h_rec[0], c->d_mine, sizeof(csvsimd_shard_result), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(c->h_stitch, c->d_stitch, sizeof(csvsimd_stitch), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return CSVSIMD_ERR_HIP;
    if (local != CSVSIMD_OK) return local;
    if (rc != CSVSIMD_OK) return rc;
    *result = c->h_rec[0];
    *stitch = *c->h_stitch;
    if (stitch->error || result->error) return CSVSIMD_ERR_INTERNAL;  // some rank's pass failed: no rank has a tape
    if (dtape && result->count > tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
    });
}

}  // extern "C"
