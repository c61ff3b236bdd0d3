// sharded.cpp — native multi-GPU stage 1: one rank per GPU, one RCCL all-gather over xGMI.
//
// The C++ twin of csv-simd_amd/sharded.py for hosts without torch (a Rust or C++ caller).  New
// relative to the reference (single-threaded; its README.md:24 lists "splitting work without first
// knowing record breaks" as a TODO): the two values the reference carries between 64-byte blocks
// (`inside_str`, `array_idx`, src/reader.rs:217-218) are carried between GPUs by ONE all-gather of
// the 64-byte csvsimd_shard_result per rank, then csvsimd_stitch_shards.
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

}  // namespace

struct csvsimd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    csvsimd_shard_result* d_mine = nullptr;  // device: this rank's record
    csvsimd_shard_result* d_all = nullptr;   // device: world records, rank order
    csvsimd_shard_result* h_all = nullptr;   // pinned host copy
};

extern "C" {

int csvsimd_comm_unique_id(uint8_t id[CSVSIMD_COMM_ID_BYTES]) {
    static_assert(CSVSIMD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size must match RCCL's");
    if (!id) return CSVSIMD_ERR_INVALID_ARG;
    RcclApi& api = rccl();
    if (!api.error.empty()) return CSVSIMD_ERR_RCCL;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return CSVSIMD_ERR_RCCL;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return CSVSIMD_OK;
}

int csvsimd_comm_create(const uint8_t id[CSVSIMD_COMM_ID_BYTES], int rank, int world, int device, csvsimd_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return CSVSIMD_ERR_INVALID_ARG;
    *out = nullptr;
    RcclApi& api = rccl();
    if (!api.error.empty()) return CSVSIMD_ERR_RCCL;
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
    ok = ok && hipHostMalloc((void**)&c->h_all, sizeof(csvsimd_shard_result) * (size_t)world, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        csvsimd_comm_destroy(c);
        return CSVSIMD_ERR_RCCL;
    }
    *out = c;
    return CSVSIMD_OK;
}

void csvsimd_comm_destroy(csvsimd_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->d_mine) (void)hipFree(c->d_mine);
    if (c->d_all) (void)hipFree(c->d_all);
    if (c->h_all) (void)hipHostFree(c->h_all);
    delete c;
}

// One sharded step for this rank: speculative pass (entered outside a string) -> all-gather of the
// result records, device to device -> one copy to the host (the step's only synchronisation) ->
// stitch -> re-emit only if this shard really starts inside a quoted string.
int csvsimd_stage1_index_sharded(csvsimd_ctx* ctx, csvsimd_comm* c, const void* dbuf, uint64_t len, uint64_t base_off,
                                 uint32_t file_in_quote_in, void* dtape, uint64_t tape_cap,
                                 csvsimd_shard_result* result, csvsimd_stitch* stitch, void* hip_stream) {
    if (!ctx || !c || !result || !stitch) return CSVSIMD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = csvsimd_stage1_index_device_async(ctx, dbuf, len, base_off, 0, dtape, tape_cap, c->d_mine, st);
    if (rc != CSVSIMD_OK) return rc;
    if (rccl().AllGather(c->d_mine, c->d_all, sizeof(csvsimd_shard_result), ncclUint8, c->comm, st) != ncclSuccess)
        return CSVSIMD_ERR_RCCL;
    if (hipMemcpyAsync(c->h_all, c->d_all, sizeof(csvsimd_shard_result) * (size_t)c->world, hipMemcpyDeviceToHost, st) !=
            hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return CSVSIMD_ERR_HIP;
    for (int i = 0; i < c->world; ++i)
        if (c->h_all[i].error) return CSVSIMD_ERR_INTERNAL;
    rc = csvsimd_stitch_shards(c->h_all, (uint32_t)c->world, (uint32_t)c->rank, file_in_quote_in, stitch);
    if (rc != CSVSIMD_OK) return rc;
    *result = c->h_all[c->rank];
    if (stitch->in_quote_in) {  // the speculation was wrong for this shard: emit again, for real
        rc = csvsimd_stage1_index_device(ctx, dbuf, len, base_off, 1, dtape, tape_cap, result, st);
        if (rc != CSVSIMD_OK && rc != CSVSIMD_ERR_TAPE_CAPACITY) return rc;
    }
    if (dtape && result->count > tape_cap) return CSVSIMD_ERR_TAPE_CAPACITY;
    return CSVSIMD_OK;
}

}  // extern "C"
