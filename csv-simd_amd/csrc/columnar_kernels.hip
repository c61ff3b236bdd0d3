// columnar_kernels.hip — gfx950 kernels that turn a finished, device-resident tape + the row-major CSV bytes into
// COLUMNS, and the reference's two stated uses of the tape on top of them.
//
// "use the result to run frequency counts, and function search" (reference design_notes_1.md:1-4) over
// `Chunk {start, end, record_cnt}` (src/tape.rs:12-19, 95-140).  The per-column consumers of consumer_kernels.hip
// re-read the row-major file once per column: one 32-byte field of a 528-byte row touches one or two 64-byte sectors
// plus a 16-byte slice of tape per record (measured round 2: 1.8-3.7x the algorithmic bytes for the gather, 9-18x for
// the frequency count's verification pass).  Here a chunk's bytes and tape are read ONCE:
//
//   to_columns_kernel   a workgroup stages a run of whole rows — their bytes (coalesced 16-byte loads) and their slice
//                       of the tape (as 32-bit offsets) — in LDS, then writes every requested field of every staged row
//                       to its column: column c, record i -> cols[(c * n_rows + i) * stride ...), truncated to `stride`,
//                       zero padded; consecutive lanes write consecutive 16-byte pieces of one column, so a wave store
//                       is 1 KiB contiguous.  Field arithmetic is RecordSource::seek_field's
//                       (src/record_source.rs:106-140): field f of row r = bytes[index[r*jump+f] + 1 .. index[r*jump+f+1]).
//   colfreq_*           exact frequency count of one column of the columnar copy: contiguous 16-byte loads, the key
//                       hashed from registers, a hash table whose slots POINT at a representative record (tag + record
//                       id in one 64-bit word): a slot is the value's only if the bytes are equal, so there are no hash
//                       collisions to detect afterwards, no verification pass and no retry; one returning atomic per
//                       new value.
//   colsearch_kernel    equals / starts-with / contains over a column of the columnar copy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_kernels.h"

namespace csvsimd {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4cu __attribute__((ext_vector_type(4), aligned(1)));

// ---------------------------------------------------------------------------------------------
// row-major -> columnar
// ---------------------------------------------------------------------------------------------
#ifndef CSVSIMD_COLWIN_BYTES
#define CSVSIMD_COLWIN_BYTES 16384
#endif
#ifndef CSVSIMD_COLWIN_ENTRIES
#define CSVSIMD_COLWIN_ENTRIES 2048
#endif
static constexpr u32 kWinBytes = CSVSIMD_COLWIN_BYTES;      // bytes of whole rows a workgroup stages per step
static constexpr u32 kWinEntries = CSVSIMD_COLWIN_ENTRIES;  // tape entries (+ 1) it stages with them, as 16-bit offsets
static_assert(kWinBytes + 16 < 65536, "staged tape entries are 16-bit offsets into the window");
// LDS per workgroup: 16 KiB + 4 KiB -> seven workgroups (28 waves) per CU, each at another point of its
// load -> stage -> write cycle: that overlap is what keeps HBM busy (a workgroup alone is a chain of dependent steps).
// Measured (scripts/ab_columns.py, 16x32 1 GiB -> 16 columns, kernel ms): windows of 32 / 24 / 16 / 12 KiB:
// 0.67 / 0.48 / 0.43 / 0.46 — more resident workgroups win until a run of rows becomes too short to write long segments.

struct ToColumnsArgs {
    const uint8_t* bytes;
    u64 bytes_len;
    const u64* index;   // the tape WITH its sentinel
    u64 first_key;      // index key of the first row's field 0 (= chunk.start)
    u64 jump;           // record_jump_size
    u64 n_rows;
    const u32* fields;  // device: the n_fields requested field ids; nullptr = 0 .. n_fields - 1
    u32 n_fields;
    uint8_t* cols;      // n_fields x n_rows x stride
    u32 stride;         // multiple of 16
    u32* lens;          // n_fields x n_rows untruncated lengths, or nullptr
    u32 rows_per_block;
};

// 16 bytes of v keep their first `nvalid` bytes (0..16), the rest become zero
__device__ __forceinline__ u32x4c keep_first_bytes(u32x4c v, u32 nvalid) {
    u32 m[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
        const u32 vb = nvalid > 4 * j ? nvalid - 4 * j : 0u;
        m[j] = vb >= 4 ? 0xffffffffu : ((1u << (8u * vb)) - 1u);
    }
    v.x &= m[0];
    v.y &= m[1];
    v.z &= m[2];
    v.w &= m[3];
    return v;
}

// q / d for q < 2^24 (d >= 1): float estimate + one correction either way
__device__ __forceinline__ u32 div_small(u32 q, u32 d, float inv_d) {
    u32 c = (u32)((float)q * inv_d);
    if (c * d > q) --c;
    else if ((c + 1) * d <= q) ++c;
    return c;
}

__global__ __launch_bounds__(256) void to_columns_kernel(const ToColumnsArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[kWinBytes + 32];  // + the 5th dword of the last unaligned read
    __shared__ unsigned short s_ent[kWinEntries + 2];
    const u32 t = threadIdx.x;
    const u32 spr = a.stride >> 4;  // 16-byte pieces per output row
    const float inv_spr = 1.0f / (float)spr;
    const int spr_shift = (spr & (spr - 1u)) == 0u ? (int)__builtin_ctz(spr) : -1;
    const u64 n_blocks = (a.n_rows + a.rows_per_block - 1) / a.rows_per_block;
    const uintptr_t base = (uintptr_t)a.bytes;
    // the two tape entries that bound a run of rows: every lane loads both (one broadcast transaction each), and the
    // NEXT run's pair is requested before this run is processed, so no step of the loop waits for them alone
    u64 blk = blockIdx.x;
    u64 e0 = 0, e1 = 0;
    if (blk < n_blocks) {
        const u64 r0 = blk * a.rows_per_block;
        const u64 rb = a.n_rows - r0 < a.rows_per_block ? a.n_rows - r0 : a.rows_per_block;
        e0 = a.index[a.first_key + r0 * a.jump];
        e1 = a.index[a.first_key + (r0 + rb) * a.jump];
    }
    for (; blk < n_blocks; blk += gridDim.x) {
        const u64 r0 = blk * a.rows_per_block;
        const u32 rb = (u32)(a.n_rows - r0 < a.rows_per_block ? a.n_rows - r0 : a.rows_per_block);
        const u64 k0 = a.first_key + r0 * a.jump;
        const u64 n_ent = (u64)rb * a.jump + 1;
        const u64 cur0 = e0, cur1 = e1;  // the line end before the first staged row / of the last one
        const u64 nblk = blk + gridDim.x;
        if (nblk < n_blocks) {
            const u64 nr0 = nblk * a.rows_per_block;
            const u64 nrb = a.n_rows - nr0 < a.rows_per_block ? a.n_rows - nr0 : a.rows_per_block;
            e0 = a.index[a.first_key + nr0 * a.jump];
            e1 = a.index[a.first_key + (nr0 + nrb) * a.jump];
        }
        // staged bytes: from the 16-byte line that holds the first row's first byte up to the last row's line end
        const uintptr_t a0 = (base + cur0 + 1) & ~(uintptr_t)15;
        const int64_t off0 = (int64_t)(a0 - base);  // file offset of the window's byte 0 (>= -15)
        const u64 extent = cur1 > cur0 && cur1 <= a.bytes_len ? (u64)((int64_t)cur1 - off0) : 0;
        const bool fast = n_ent <= kWinEntries + 1 && extent <= kWinBytes && extent > 0;
        const u32 per_col = rb * spr;
        const float inv_per_col = 1.0f / (float)per_col;
        const u32 items = a.n_fields * per_col;
        if (fast) {
            for (u32 i = t; i < (u32)n_ent; i += 256) s_ent[i] = (unsigned short)((int64_t)a.index[k0 + i] - off0);
            const u32 n16 = ((u32)extent + 15) >> 4;
            const u32x4c* src = reinterpret_cast<const u32x4c*>(a0);
            for (u32 i = t; i < n16; i += 256) reinterpret_cast<u32x4c*>(s_bytes)[i] = __builtin_nontemporal_load(src + i);
            __syncthreads();
            // item q = (column c, piece j of the column's rb x spr pieces): q advances by 256 per step, so (c, j) is
            // carried along instead of divided out — a step crosses a handful of columns at most
            u32 c = div_small(t, per_col, inv_per_col), j = t - c * per_col;
            for (u32 q = t; q < items; q += 256, j += 256) {
                while (j >= per_col) { j -= per_col; ++c; }
                u32 r, k;
                if (spr_shift >= 0) { r = j >> spr_shift; k = j & (spr - 1u); }   // (wave-uniform: the stride is a power of two)
                else { r = div_small(j, spr, inv_spr); k = j - r * spr; }
                const u32 f = a.fields ? a.fields[c] : c;
                // the first entry of the window is the line end BEFORE the first row: offset -1 if that row starts the
                // window's first 16-byte line, hence the 16-bit wrap-around of "+ 1"
                u32 lo = (u32)(unsigned short)(s_ent[r * (u32)a.jump + f] + 1u), hi = s_ent[r * (u32)a.jump + f + 1];
                hi = hi < (u32)extent ? hi : (u32)extent;  // a corrupt tape must not read outside the window
                lo = lo < hi ? lo : hi;
                const u32 len = hi - lo;
                const u32 pos = lo + 16 * k;
                const u32 nvalid = len > 16 * k ? (len - 16 * k < 16 ? len - 16 * k : 16u) : 0u;
                u32x4c v = {0, 0, 0, 0};
                if (nvalid) {
                    // 16 bytes at an arbitrary LDS byte offset: five aligned dwords, funnel-shifted
                    const u32* w = reinterpret_cast<const u32*>(s_bytes) + (pos >> 2);
                    const u32 d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
                    const u32 sh = pos & 3u;
                    v.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
                    v.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    v.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
                    v.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
                    if (nvalid < 16u) v = keep_first_bytes(v, nvalid);
                }
                const u64 row = (u64)c * a.n_rows + r0 + r;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4c*>(a.cols + row * a.stride + 16 * k));
                if (k == 0 && a.lens) a.lens[row] = len;
            }
            __syncthreads();  // the window is rewritten by the next run of rows
        } else {
            // a run of rows that does not fit the window (very long rows, or thousands of columns): the same result
            // straight from global memory, 16 bytes at any alignment per step
            for (u32 q = t; q < items; q += 256) {
                const u32 c = q / per_col, j = q - c * per_col;
                const u32 r = j / spr, k = j - r * spr;
                const u32 f = a.fields ? a.fields[c] : c;
                const u64 key = k0 + (u64)r * a.jump + f;
                u64 lo = a.index[key] + 1, hi = a.index[key + 1];
                hi = hi < a.bytes_len ? hi : a.bytes_len;
                lo = lo < hi ? lo : hi;
                const u64 len = hi - lo;
                const u64 pos = lo + 16 * k;
                const u32 nvalid = len > 16 * k ? (u32)(len - 16 * k < 16 ? len - 16 * k : 16) : 0u;
                u32x4c v = {0, 0, 0, 0};
                if (nvalid) {
                    if (pos + 16 <= a.bytes_len) {
                        v = *reinterpret_cast<const u32x4cu*>(a.bytes + pos);
                    } else {  // the last 15 bytes of the buffer: never read past it
                        uint8_t tmp[16];
                        for (u32 b = 0; b < 16; ++b) tmp[b] = pos + b < a.bytes_len ? a.bytes[pos + b] : (uint8_t)0;
                        v = *reinterpret_cast<const u32x4c*>(tmp);
                    }
                    v = keep_first_bytes(v, nvalid);
                }
                const u64 row = (u64)c * a.n_rows + r0 + r;
                *reinterpret_cast<u32x4c*>(a.cols + row * a.stride + 16 * k) = v;
                if (k == 0 && a.lens) a.lens[row] = (u32)(len > 0xffffffffull ? 0xffffffffull : len);
            }
        }
    }
}

hipError_t launch_to_columns(const void* dbytes, u64 bytes_len, const void* dindex, u64 first_key, u64 jump, u64 n_rows,
                             const void* d_fields, u32 n_fields, void* d_cols, u32 stride, void* d_lens, u32 rows_per_block,
                             int n_cus, hipStream_t stream) {
    if (n_rows == 0 || n_fields == 0) return hipSuccess;
    ToColumnsArgs a;
    a.bytes = (const uint8_t*)dbytes;
    a.bytes_len = bytes_len;
    a.index = (const u64*)dindex;
    a.first_key = first_key;
    a.jump = jump;
    a.n_rows = n_rows;
    a.fields = (const u32*)d_fields;
    a.n_fields = n_fields;
    a.cols = (uint8_t*)d_cols;
    a.stride = stride;
    a.lens = (u32*)d_lens;
    // rows per step: as many as fit the window on average (the kernel itself checks every run and falls back); a
    // multiple of the rows that make up whole 128-byte lines of a column, so that segments of neighbouring steps do
    // not share lines
    u32 r = rows_per_block;
    const u32 jcap = kWinEntries / (u32)(jump < kWinEntries ? jump : kWinEntries);
    if (r > jcap) r = jcap;
    u32 line_rows = 1;
    while ((line_rows * stride) % 128u) line_rows <<= 1;  // stride is a multiple of 16: at most 8
    if (r > line_rows) r -= r % line_rows;
    while ((u64)n_fields * r * (stride >> 4) >= (1ull << 24) && r > 1) r >>= 1;  // the kernel's index math is exact below 2^24 items
    if (r < 1) r = 1;
    a.rows_per_block = r;
    const u64 n_blocks = (n_rows + r - 1) / r;
    const u64 cap = (u64)(n_cus > 0 ? n_cus : 256) * 28;  // 7 resident workgroups per CU (LDS), four rounds of them
    hipLaunchKernelGGL(to_columns_kernel, dim3((u32)(n_blocks < cap ? n_blocks : cap)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

u32 to_columns_window_bytes() { return kWinBytes; }

// ---------------------------------------------------------------------------------------------
// a column of the columnar copy: record i = col[i * stride .. + min(len[i], stride)), zero padded
// ---------------------------------------------------------------------------------------------
struct ColView {
    const uint8_t* col;
    const u32* len;  // nullptr: every value is its whole zero-padded row (fixed-width keys)
    u64 n_rows;
    u32 stride;  // multiple of 16
};

__device__ __forceinline__ u64 cmix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// hash of record i's padded bytes + length, 16 bytes per step from aligned, coalesced loads
__device__ __forceinline__ u64 hash_row(const ColView& c, u64 i, u32 len) {
    const u32x4c* p = reinterpret_cast<const u32x4c*>(c.col + i * c.stride);
    u64 h = 0x243F6A8885A308D3ull ^ ((u64)len * 0x9E3779B97F4A7C15ull);
    for (u32 k = 0; k < (c.stride >> 4); ++k) {
        const u32x4c v = p[k];
        h = cmix64(h ^ (((u64)v.y << 32) | v.x)) + 0x9E3779B97F4A7C15ull;
        h = cmix64(h ^ (((u64)v.w << 32) | v.z));
    }
    return h;
}
__device__ __forceinline__ bool rows_equal(const ColView& c, u64 i, u64 j, u32 len_i) {
    if (c.len && c.len[j] != len_i) return false;
    const u32x4c* p = reinterpret_cast<const u32x4c*>(c.col + i * c.stride);
    const u32x4c* q = reinterpret_cast<const u32x4c*>(c.col + j * c.stride);
    for (u32 k = 0; k < (c.stride >> 4); ++k) {
        const u32x4c x = p[k], y = q[k];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// exact frequency count on a column.  Table slot (16 bytes):
//   key    : tag (upper 32 bits of the hash) << 32 | representative record + 1        0 = empty
//   extra  : records holding the value beyond the one that claimed the slot
//   first  : ~(smallest record id among the records that found the slot taken), 0 = none
// A record looks at a taken slot's representative only if the tags agree, and the slot is its value's only if the
// bytes are equal: two values with the same hash simply occupy two slots.  One returning atomic (the claim) per new
// value; a workgroup first aggregates in LDS so that a column of few distinct values does not hammer a handful of
// global slots.
// ---------------------------------------------------------------------------------------------
struct ColFreqSlot {
    u64 key;
    u32 extra;
    u32 first_inv;
};
struct ColFreqStatus {  // == csvsimd_colfreq_status
    u64 n_records, n_distinct, truncated, overflow;
};
static constexpr u32 kCfLds = 1024;

__device__ __forceinline__ bool colfreq_global_insert(const ColView& c, ColFreqSlot* table, u64 mask, u64 h, u32 rep,
                                                      u32 len_rep, u32 count, u32 first) {
    const u64 mine = (h & 0xffffffff00000000ull) | ((u64)rep + 1);
    u64 s = cmix64(h) & mask;
    for (u64 probes = 0; probes <= mask; ++probes, s = (s + 1) & mask) {
        u64 old = __hip_atomic_load(&table[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) old = atomicCAS((unsigned long long*)&table[s].key, 0ull, (unsigned long long)mine);
        if (old == 0) {  // claimed: this record represents the value
            if (count > 1) atomicAdd(&table[s].extra, count - 1);
            if (first != rep) atomicMax(&table[s].first_inv, ~first);
            return true;
        }
        if ((old >> 32) == (mine >> 32) && rows_equal(c, rep, (u32)old - 1u, len_rep)) {
            atomicAdd(&table[s].extra, count);
            atomicMax(&table[s].first_inv, ~first);
            return true;
        }
    }
    return false;
}

static constexpr u32 kCfInsertThreads = 1024;  // one LDS table, one flush per 8 192 records: a column of few values sends every
                                               // workgroup to the same global slots, and accesses to one line retire one by one
__global__ __launch_bounds__(kCfInsertThreads) void colfreq_insert_kernel(const ColView c, ColFreqSlot* __restrict__ table, u64 mask,
                                                             ColFreqStatus* __restrict__ status) {
    __shared__ u64 s_key[kCfLds];    // tag << 32 | (record - r0 of this workgroup) + 1
    __shared__ u32 s_count[kCfLds];
    __shared__ u32 s_first[kCfLds];
    __shared__ u32 s_fill;
    for (u32 k = threadIdx.x; k < kCfLds; k += blockDim.x) {
        s_key[k] = 0;
        s_count[k] = 0;
        s_first[k] = 0xffffffffu;
    }
    if (threadIdx.x == 0) s_fill = 0;
    __syncthreads();
    u32 overflow = 0, truncated = 0;
    const u64 per = (c.n_rows + gridDim.x - 1) / gridDim.x;  // contiguous slab of records per workgroup
    const u64 r0 = (u64)blockIdx.x * per, r1 = r0 + per < c.n_rows ? r0 + per : c.n_rows;
    for (u64 i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const u32 len = c.len ? c.len[i] : c.stride;
        if (len > c.stride) ++truncated;
        const u64 h = hash_row(c, i, len);
        const u64 mine = (h & 0xffffffff00000000ull) | (u64)((u32)(i - r0) + 1u);
        bool done = false;
        // a column of many distinct values fills the LDS table with its first rows; from then on new values only find
        // full probe sequences there, so the probing is limited to a look at the home slot
        const int max_probes = s_fill < kCfLds * 3 / 4 ? 8 : 1;
        u32 s = (u32)h & (kCfLds - 1);
        for (int p = 0; p < max_probes && !done; ++p, s = (s + 1) & (kCfLds - 1)) {
            u64 old = s_key[s];
            if (old == 0) old = atomicCAS((unsigned long long*)&s_key[s], 0ull, (unsigned long long)mine);
            if (old == 0) {
                atomicAdd(&s_fill, 1u);
                atomicAdd(&s_count[s], 1u);
                atomicMin(&s_first[s], (u32)i);
                done = true;
            } else if ((old >> 32) == (mine >> 32) && rows_equal(c, i, r0 + ((u32)old - 1u), len)) {
                atomicAdd(&s_count[s], 1u);
                atomicMin(&s_first[s], (u32)i);
                done = true;
            }
        }
        if (!done && !colfreq_global_insert(c, table, mask, h, (u32)i, len, 1u, (u32)i)) ++overflow;
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < kCfLds; k += blockDim.x) {
        const u64 key = s_key[k];
        if (!key) continue;
        const u32 rep = (u32)(r0 + ((u32)key - 1u));
        const u32 len = c.len ? c.len[rep] : c.stride;
        // the tag is the upper half of the hash, the global slot comes from the whole hash: recompute it
        const u64 h = hash_row(c, rep, len);
        if (!colfreq_global_insert(c, table, mask, h, rep, len, s_count[k], s_first[k])) ++overflow;
    }
    if (overflow) atomicAdd((unsigned long long*)&status->overflow, (unsigned long long)overflow);
    if (truncated) atomicAdd((unsigned long long*)&status->truncated, (unsigned long long)truncated);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long*)&status->n_records, (unsigned long long)c.n_rows);
}

struct ColFreqEntry {  // == csvsimd_colfreq_entry
    u64 first_record, count;
};
// occupied slots -> dense entries; one output reservation per workgroup (a returning atomic on one word retires at
// ~90 per us chip-wide)
static constexpr u32 kCfPerThread = 8;
static constexpr u32 kCfCompactThreads = 1024;  // 8192 slots per reservation: 512 of them for a 4 M-slot table (2048: 23 us of atomics alone)
__global__ __launch_bounds__(kCfCompactThreads) void colfreq_compact_kernel(const ColFreqSlot* __restrict__ table, u64 slots,
                                                              u64 first_record, ColFreqEntry* __restrict__ out, u64 out_cap,
                                                              ColFreqStatus* __restrict__ status) {
    __shared__ u32 s_wave[kCfCompactThreads / 64];
    __shared__ u64 s_base;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u64 chunk = (u64)blockDim.x * kCfPerThread;
    for (u64 c0 = (u64)blockIdx.x * chunk; c0 < slots; c0 += (u64)gridDim.x * chunk) {
        u64 masks[kCfPerThread];
        u32 mine = 0, wave_total = 0;
#pragma unroll
        for (u32 j = 0; j < kCfPerThread; ++j) {
            const u64 sl = c0 + (u64)j * blockDim.x + threadIdx.x;
            const bool used = sl < slots && table[sl].key != 0;
            masks[j] = __ballot(used);
            mine |= (used ? 1u : 0u) << j;
            wave_total += (u32)__builtin_popcountll(masks[j]);
        }
        if (lane == 0) s_wave[w] = wave_total;
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 tot = 0;
            for (u32 k = 0; k < kCfCompactThreads / 64; ++k) tot += s_wave[k];
            s_base = tot ? atomicAdd((unsigned long long*)&status->n_distinct, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        u64 at = s_base;
        for (u32 k = 0; k < w; ++k) at += s_wave[k];
#pragma unroll
        for (u32 j = 0; j < kCfPerThread; ++j) {
            if ((mine >> j) & 1u) {
                const u64 o = at + (u64)__builtin_popcountll(masks[j] & ((1ull << lane) - 1ull));
                if (o < out_cap) {
                    const ColFreqSlot sl = table[c0 + (u64)j * blockDim.x + threadIdx.x];
                    u32 first = (u32)sl.key - 1u;
                    if (sl.first_inv && ~sl.first_inv < first) first = ~sl.first_inv;
                    out[o] = ColFreqEntry{first_record + first, (u64)sl.extra + 1};
                }
            }
            at += (u64)__builtin_popcountll(masks[j]);
        }
        __syncthreads();
    }
}

static u32 cgrid_for(u64 items, u32 per_block, u32 cap) {
    u64 blocks = (items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    return (u32)(blocks > cap ? cap : blocks);
}

hipError_t launch_colfreq_insert(const void* d_col, const void* d_len, u64 n_rows, u32 stride, void* d_table, u64 slots,
                                 void* d_status, int n_cus, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const ColView c = {(const uint8_t*)d_col, (const u32*)d_len, n_rows, stride};
    const u32 grid = cgrid_for(n_rows, 8 * kCfInsertThreads, (u32)(n_cus > 0 ? n_cus : 256) * 2);
    hipLaunchKernelGGL(colfreq_insert_kernel, dim3(grid), dim3(kCfInsertThreads), 0, stream, c, (ColFreqSlot*)d_table, slots - 1,
                       (ColFreqStatus*)d_status);
    return hipGetLastError();
}

hipError_t launch_colfreq_compact(const void* d_table, u64 slots, u64 first_record, void* d_out, u64 out_cap,
                                  void* d_status, hipStream_t stream) {
    hipLaunchKernelGGL(colfreq_compact_kernel, dim3(cgrid_for(slots, kCfCompactThreads * kCfPerThread, 4096)),
                       dim3(kCfCompactThreads), 0, stream,
                       (const ColFreqSlot*)d_table, slots, first_record, (ColFreqEntry*)d_out, out_cap,
                       (ColFreqStatus*)d_status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// search on a column: bit i of the bitmap = record i matches; the definitions of consumer_kernels.hip's search_kernel
// (==, bytes.startswith, `needle in field`), one lane per record, the record's bytes contiguous and 16-byte aligned.
// ---------------------------------------------------------------------------------------------
static constexpr u32 kColMaxNeedle = 256;
__device__ __forceinline__ u64 col_load8(const uint8_t* row, u32 at, u32 stride) {  // 8 bytes of the padded row from `at`
    u64 v = 0;
    if (at + 8 <= stride) {
        const u32* p = reinterpret_cast<const u32*>(row + (at & ~3u));
        const u32 sh = at & 3u;
        // sh != 0 and at + 8 <= stride (a multiple of 4) imply that the third dword still belongs to the row
        const u32 d0 = p[0], d1 = p[1], d2 = sh ? p[2] : 0u;
        v = ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
    } else {
        for (u32 j = 0; j < 8 && at + j < stride; ++j) v |= (u64)row[at + j] << (8 * j);
    }
    return v;
}

__global__ __launch_bounds__(256) void colsearch_kernel(const ColView c, const uint8_t* __restrict__ needle, u32 m, int mode,
                                                        u64* __restrict__ bitmap, u64* __restrict__ count,
                                                        u64* __restrict__ truncated) {
    __shared__ u64 s_needle[kColMaxNeedle / 8 + 1];
    for (u32 k = threadIdx.x; k < kColMaxNeedle / 8 + 1; k += blockDim.x) {
        u64 w = 0;
        for (u32 j = 0; j < 8; ++j)
            if (8 * k + j < m) w |= (u64)needle[8 * k + j] << (8 * j);
        s_needle[k] = w;
    }
    __syncthreads();
    const u64 n_words = (c.n_rows + 63) / 64;
    const u32 lane = threadIdx.x & 63u;
    const u32 head = m < 8 ? m : 8;
    const u64 head_mask = head == 8 ? ~0ull : ((1ull << (8 * head)) - 1ull);
    const u64 needle0 = s_needle[0];
    u32 hits = 0, trunc = 0;
    for (u64 word = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; word < n_words;
         word += ((u64)gridDim.x * blockDim.x) >> 6) {
        const u64 i = word * 64 + lane;
        bool match = false;
        if (i < c.n_rows) {
            const u32 full = c.len ? c.len[i] : c.stride;
            if (full > c.stride) ++trunc;
            const u32 n = full < c.stride ? full : c.stride;
            const uint8_t* row = c.col + i * c.stride;
            if (mode != 2) {
                match = mode == 0 ? n == m : n >= m;
                for (u32 k = 0; 8 * k < m && match; ++k) {
                    const u32 left = m - 8 * k;
                    const u64 mask = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
                    match = ((col_load8(row, 8 * k, c.stride) ^ s_needle[k]) & mask) == 0;
                }
            } else if (m == 0) {
                match = true;
            } else if (n >= m) {
                const u32 last = n - m;  // last start position
                u64 cur = col_load8(row, 0, c.stride);
                for (u32 base = 0; base <= last && !match; base += 8) {
                    const u64 nxt = col_load8(row, base + 8, c.stride);
#pragma unroll
                    for (u32 j = 0; j < 8; ++j) {
                        const u64 view = j ? (cur >> (8 * j)) | (nxt << (64 - 8 * j)) : cur;
                        if (base + j <= last && ((view ^ needle0) & head_mask) == 0) {
                            bool ok = true;
                            for (u32 k = 1; 8 * k < m && ok; ++k) {
                                const u32 left = m - 8 * k;
                                const u64 mask = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
                                ok = ((col_load8(row, base + j + 8 * k, c.stride) ^ s_needle[k]) & mask) == 0;
                            }
                            match = match || ok;
                        }
                    }
                    cur = nxt;
                }
            }
        }
        const u64 bits = __ballot(match);
        if (lane == 0) {
            bitmap[word] = bits;
            hits += (u32)__builtin_popcountll(bits);
        }
    }
    if (lane == 0 && hits) atomicAdd((unsigned long long*)count, (unsigned long long)hits);
    if (trunc) atomicAdd((unsigned long long*)truncated, (unsigned long long)trunc);
}

hipError_t launch_colsearch(const void* d_col, const void* d_len, u64 n_rows, u32 stride, const void* d_needle,
                            u32 needle_len, int mode, void* d_bitmap, void* d_count, void* d_truncated, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const ColView c = {(const uint8_t*)d_col, (const u32*)d_len, n_rows, stride};
    hipLaunchKernelGGL(colsearch_kernel, dim3(cgrid_for(n_rows, 256, 8192)), dim3(256), 0, stream, c,
                       (const uint8_t*)d_needle, needle_len, mode, (u64*)d_bitmap, (u64*)d_count, (u64*)d_truncated);
    return hipGetLastError();
}

}  // namespace csvsimd
