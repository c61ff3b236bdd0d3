// stage1_kernels.h — internal C++ interface between the C ABI (capi.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csvsimd.h"

// tile = CSVSIMD_COMPUTE_WAVES waves x CSVSIMD_ROUNDS rounds x 4 KiB.  Keep the wave count a multiple
// of 4: the dispatcher was measured to reserve ceil(waves / 4) slots on EVERY SIMD (5-wave
// workgroups: 2 per CU instead of 4).
#ifndef CSVSIMD_ROUNDS
#define CSVSIMD_ROUNDS 8
#endif
#ifndef CSVSIMD_COMPUTE_WAVES
#define CSVSIMD_COMPUTE_WAVES 8
#endif
// register budget: minimum waves per SIMD the stage-1 kernel is compiled for (0 = let hipcc choose)
#ifndef CSVSIMD_WAVES_PER_EU
#define CSVSIMD_WAVES_PER_EU 4
#endif
#define CSVSIMD_TILE_BYTES (CSVSIMD_COMPUTE_WAVES * CSVSIMD_ROUNDS * 4096)
#define CSVSIMD_MIN_TILE_BYTES (CSVSIMD_COMPUTE_WAVES * 2 * 4096) /* the dense instantiation: 2 rounds (stage1_dense.hip) */

namespace csvsimd {

// scratch block layout (zeroed ONCE, when it is allocated; every launch leaves it ready for the next):
//   [0, 64)            control block (stage1_kernels.hip: struct Control): ticket, epoch, done/total, ...
//   [64, 528)          development builds only: per-phase timing slots
//   [528, 8720)        one u32 count-phase token per physical CU (XCC id x HW_ID cu/sh/se bits); 0 = free
//   [8720, ...)        one u64 look-back descriptor per tile, tagged with the launch epoch
#define CSVSIMD_SCRATCH_CTL_BYTES 64
#define CSVSIMD_SCRATCH_TOKEN_OFFSET 528            /* 8 x 256 u32: one count-phase token per physical CU */
#define CSVSIMD_SCRATCH_DESC_OFFSET (528 + 8192)
struct Stage1Launch {
    const void* dbuf;
    uint64_t len;
    uint64_t base_off;
    uint32_t in_quote_in;
    void* dtape;
    uint64_t tape_cap;
    csvsimd_shard_result* d_result;
    void* scratch_base;
    uint64_t* scratch_desc;
    uint64_t* scratch_prof;
    uint32_t max_blocks;
    // sharded re-emit: device word with the shard's true entering state; the launch is a no-op unless it is 1
    const uint32_t* d_state = nullptr;
    // chunked ingest: device result record of the previous chunk; entering state (and escape_in) are read from it on the device
    const csvsimd_shard_result* d_chain = nullptr;
    // optional: recorded immediately around the stage-1 kernel itself (bench roofline leg)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int debug_mode = 0;  // development probes (-DCSVSIMD_DEV_PROBES builds only; see stage1_kernel's DBG)
    int pace_emit_delay = -1, pace_count_prio = -1;  // -1 = chosen from the launch size (launch_stage1)
    int pace_cu_token = -1;      // 1 = the two workgroups of a CU take turns in the count phase
    // dialect extension (csvsimd_dialect): the defaults are the reference's hard-wired dialect
    uint8_t delimiter = ',', quote = '"', escape = 0;
    uint32_t escape_in = 0;
    bool allow_hashed_dialect = true;  // escape dialects: use the hashed LUT classification when the bytes allow it
    bool dense = false;                // reference dialect, emitting launch: the instantiation with the test-free emit path

    static uint64_t scratch_bytes_for(uint64_t len) {
        // one descriptor word per tile of the SMALLEST geometry a launch may choose (the dense instantiation's 64-KiB
        // tiles: 8 bytes per 64 KiB of input); + 1 tile: an unaligned dbuf shifts the data by up to 127 bytes
        const uint64_t tiles = (len + 127 + CSVSIMD_MIN_TILE_BYTES - 1) / CSVSIMD_MIN_TILE_BYTES + 1;
        return CSVSIMD_SCRATCH_DESC_OFFSET + 8 * tiles + 16;
    }
    void bind_scratch(void* base) {
        scratch_base = base;
        scratch_prof = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(base) + CSVSIMD_SCRATCH_CTL_BYTES);
        scratch_desc = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(base) + CSVSIMD_SCRATCH_DESC_OFFSET);
    }
};

// hashed classification tables of an escape dialect (stage1_kernels.hip: classify_dword_h); false = no collision-free hash
struct DialectHash {
    uint32_t sh1, sh2, lut_lo, lut_hi, cls_lo, cls_hi;
};
bool dialect_hash(uint32_t delim, uint32_t quote, uint32_t esc, DialectHash& h);
hipError_t launch_stage1(const Stage1Launch& L, hipStream_t stream);
}  // namespace csvsimd
namespace csvsimd_dense {
hipError_t launch_stage1_dense(const csvsimd::Stage1Launch& L, hipStream_t stream);     // stage1_dense.hip
hipError_t launch_stage1_dense_d1(const csvsimd::Stage1Launch& L, hipStream_t stream);  // stage1_dense_d1.hip (another delimiter / quote byte)
hipError_t launch_stage1_batch_dense(void* d_items, void* d_first_tiles, void* d_tots, uint32_t n_items, uint32_t total_tiles,
                                     csvsimd_shard_result* d_results, void* scratch_base, uint64_t* scratch_desc,
                                     uint32_t max_blocks, hipStream_t stream);  // stage1_dense_batch.hip
}
namespace csvsimd {
hipError_t launch_synth(void* dbuf, uint64_t file_off, uint64_t len, uint32_t cols, uint32_t width,
                        uint64_t seed, uint32_t quote_pct, hipStream_t stream);
hipError_t launch_checksum(const void* dtape, uint64_t n, uint64_t first_index, void* d_out,
                           hipStream_t stream);
hipError_t launch_selftest(uint32_t* d_out, hipStream_t stream);
hipError_t launch_hbm_probe(const void* din, uint64_t len, void* dout, int write_div, void* scratch_base,
                            uint32_t blocks, hipStream_t stream);
// plain copy yardstick: mode 0 = hipMemcpyDtoDAsync, 1 = one 16-byte element per thread, 2 = the same, non-temporal
hipError_t launch_copy_probe(const void* din, void* dout, uint64_t len, int mode, hipStream_t stream);
// csvsimd_stitch_shards on the device (one wave), so the sharded step needs no host round trip
hipError_t launch_stitch(const void* d_results, uint32_t n_shards, uint32_t rank, uint32_t file_in_quote_in,
                         void* d_stitch, hipStream_t stream);
// one persistent launch over n_items independent buffers; d_items = device table of 64-byte lines laid out as
// stage1_kernels.hip's BatchItem {abase, lo, hi, base_off, tape, tape_cap, first_tile, in_quote_in, 0}; d_first_tiles =
// the first_tile fields again, compact (u32 each); d_tots = n_items zeroed u64 counters
struct BatchItemHost {
    const void* abase;
    uint64_t lo, hi, base_off;
    void* tape;
    uint64_t tape_cap;
    uint32_t first_tile, in_quote_in;
    uint64_t reserved;
};
static_assert(sizeof(BatchItemHost) == 64, "mirrors the device-side BatchItem");
hipError_t launch_stage1_batch(void* d_items, void* d_first_tiles, void* d_tots, uint32_t n_items, uint32_t total_tiles,
                               csvsimd_shard_result* d_results, void* scratch_base, uint64_t* scratch_desc,
                               uint32_t max_blocks, hipStream_t stream);
const char* stage1_kernel_name(bool emit, int dialect, bool dense = false);
// consumer_kernels.hip: consumers of a finished, device-resident tape
hipError_t launch_chunk_spans(const void* dindex, uint64_t first_key, uint64_t jump, uint32_t field, uint32_t fields,
                              uint64_t n_rows, void* d_begin, void* d_end, hipStream_t stream, void* d_longest = nullptr);
hipError_t launch_gather_fields(const void* dbytes, uint64_t bytes_len, const void* d_begin, const void* d_end,
                                uint64_t n_records, void* d_dst, uint32_t stride, void* d_len, hipStream_t stream);
// field `field` of a run of whole rows -> fixed-stride rows + lengths, straight from the tape; *d_longest = max(it, longest field)
hipError_t launch_gather_column(const void* dbytes, const void* dindex, uint64_t index_len, uint64_t first_key, uint64_t jump,
                                uint32_t field, uint64_t n_rows, void* d_dst, uint32_t stride, void* d_len, void* d_longest,
                                hipStream_t stream);
hipError_t launch_search(const void* dbytes, uint64_t bytes_len, const void* dindex, uint64_t first_key, uint64_t jump, uint64_t n_rows,
                         uint32_t field, const void* d_needle, uint32_t needle_len, int mode, void* d_bitmap, void* d_count,
                         hipStream_t stream);
hipError_t launch_bitmap_select(const void* d_bitmap, uint64_t n_rows, uint64_t first_row, void* d_block_scratch,
                                void* d_out, uint64_t out_cap, void* d_total, hipStream_t stream);
// columnar_kernels.hip: row-major CSV + tape -> columns in one pass; frequency count and search on a column
hipError_t launch_to_columns(const void* dbytes, uint64_t bytes_len, const void* dindex, uint64_t first_key, uint64_t jump,
                             uint64_t n_rows, const void* d_fields, uint32_t n_fields, void* d_cols, uint32_t stride,
                             void* d_lens, uint32_t rows_per_block, int n_cus, hipStream_t stream);
uint32_t to_columns_window_bytes();
// exact frequency count on a column: two launches (three on long columns), no device-memory table (columnar_kernels.hip)
uint64_t colfreq_scratch_bytes(uint64_t n_rows);
// csvsimd_column_frequency_device: the column was gathered from chunks of the row-major file; the entries then leave as
// csvsimd_freq_entry {record id, text span, count} — row -> record through the chunk map, span from the tape itself
struct FreqRowMap {  // one per chunk, ascending row0: rows [row0, next row0) of the column are records first_record + ...,
    uint64_t row0, first_record, first_key;  // whose field sits at tape key first_key + (row - row0) * jump + field
};
struct FreqWideOut {
    const uint64_t* index;   // the tape WITH its sentinel
    uint64_t jump;
    uint32_t field, n_chunks;
    const FreqRowMap* map;   // device
};
hipError_t launch_colfreq(const void* d_col, const void* d_len, uint64_t n_rows, uint32_t stride, uint64_t first_record,
                          void* d_scratch, void* d_entries, uint64_t entries_cap, void* d_status, int n_cus, hipStream_t stream,
                          const FreqWideOut* wide = nullptr);
// (the needle is HOST memory: it travels in the kernel's arguments; the last workgroup to arrive writes seq << 48 | truncated << 47 |
// matches to the pinned word at h_pub_dev — its device address — and leaves the device word d_acc at zero, as it found it)
hipError_t launch_colsearch(const void* d_col, const void* d_len, uint64_t n_rows, uint32_t stride, const void* needle_host,
                            uint32_t needle_len, int mode, void* d_bitmap, void* d_acc, void* h_pub_dev, uint64_t seq,
                            hipStream_t stream);
// text_kernels.hip
hipError_t launch_utf8_validate(const void* dbuf, uint64_t len, void* d_result, int n_cus, hipStream_t stream);
hipError_t launch_trim_spans(const void* dbytes, void* d_begin, void* d_end, uint64_t n, uint32_t flags,
                             uint32_t quote, hipStream_t stream);
// ingest: a chunk's tape (u64, device) -> 32-bit chunk-relative offsets in a pinned host slot; count read on the device
// h_rec_dev != nullptr: the launch also publishes the record (64 bytes, then `seq` in the 8 bytes behind it) to that
// device-visible host address; d_arrivals = a zeroed u32 the launch leaves zeroed
hipError_t launch_narrow_tape(const void* d_tape, const void* d_result, uint64_t cap, uint64_t base, void* d_out, int workgroups,
                              hipStream_t stream, void* h_rec_dev = nullptr, uint64_t seq = 0, void* d_arrivals = nullptr);
// 16 x 64 KiB of the buffer: d_out2[0] += bytes equal to the delimiter, CR or LF, d_out2[1] += bytes looked at (both zeroed first)
hipError_t launch_density_sample(const void* dbuf, uint64_t len, uint32_t delimiter, void* d_out2, hipStream_t stream);
int stage1_max_blocks_per_cu();

}  // namespace csvsimd
