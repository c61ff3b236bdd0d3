// consumer_kernels.hip — gfx950 kernels that CONSUME a finished, device-resident tape.
//
// The reference's stated goal for the tape is "use the result to run frequency counts, and function search"
// (design_notes_1.md:1-4), with `Chunk {start, end, record_cnt}` as "atomic representation of how to utilize the
// tape in a parallel-processing context" (src/tape.rs:12-19; produced by Tape::chunks, src/tape.rs:95-140).  The
// reference itself stops at the chunk list (single-threaded seek_field, src/record_source.rs:106-140); what is
// here is the device side of that plan, driven by the same chunk records:
//
//   * field spans of a column of a chunk           (bulk seek_field)
//   * gather of a column into fixed-stride rows    (16-byte loads at any alignment)
//   * frequency count of a column                  (exact: hash table + byte-for-byte verification pass)
//   * search in a column: equals / starts-with / contains  -> bitmap of records + count, bitmap -> record ids
//
// Index arithmetic is seek_field's: row r of the file (0 = header) owns index keys [r * jump, (r + 1) * jump),
// field f of it is bytes[index[r * jump + f] + 1 .. index[r * jump + f + 1]).  A chunk covers keys
// [chunk.start, chunk.end), i.e. chunk.record_cnt rows from row chunk.start / jump (chunk 0 starts after the
// header: src/tape.rs:116-122).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_kernels.h"

namespace csvsimd {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));  // 16 bytes at any address: ONE global_load_dwordx4
typedef uint64_t u64u __attribute__((aligned(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One column of one chunk (or of any run of whole rows): everything a consumer kernel needs to find its fields.
struct Column {
    const uint8_t* bytes;
    const u64* index;  // the tape WITH its sentinel
    u64 first_key;     // index key of the first row's field 0 (= chunk.start)
    u64 jump;          // record_jump_size
    u64 n_rows;
    u64 first_row;     // row number of the first row in the file (for record ids): first_key / jump
    u32 field;
};

__device__ __forceinline__ void field_span(const Column& c, u64 i, u64& b, u64& e) {
    const u64 k = c.first_key + i * c.jump + c.field;
    b = c.index[k] + 1;
    e = c.index[k + 1];
    if (e < b) e = b;  // cannot happen on a well-formed tape; keeps a corrupt one from running backwards
}

// ---------------------------------------------------------------------------------------------
// bulk seek_field / seek_record over a run of rows
// ---------------------------------------------------------------------------------------------
// longest: optional; the longest span seen is folded into *longest (one atomic per wave that has something to say)
__global__ void chunk_spans_kernel(const u64* __restrict__ index, u64 first_key, u64 jump, u32 field, u32 fields, u64 n_rows,
                                   u64* __restrict__ begin, u64* __restrict__ end, u64* __restrict__ longest) {
    u64 m = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = first_key + i * jump + field;
        const u64 b = index[k] + 1, e = index[k + fields];
        begin[i] = b;
        end[i] = e;
        m = e > b && e - b > m ? e - b : m;
    }
    if (longest) {
        for (int d = 32; d >= 1; d >>= 1) {
            const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(m >> 32), d) << 32) | (u32)__shfl_xor((int)(u32)m, d);
            m = o > m ? o : m;
        }
        // an atomic on ONE word retires at ~90 per microsecond chip-wide (31 k waves: 0.35 ms, measured): a wave only
        // sends one if the word it sees does not already say as much — on a column of similar lengths almost none do
        if ((threadIdx.x & 63u) == 0 &&
            m > __hip_atomic_load((const unsigned long long*)longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax((unsigned long long*)longest, (unsigned long long)m);
    }
}

// ---------------------------------------------------------------------------------------------
// gather: text of each span -> row i of dst (stride bytes, truncated, zero padded); len[i] = untruncated length.
// A group of 2^gshift lanes per record (as many as the row has 16-byte pieces, at most 16: with 16 lanes on a 32-byte
// row, rounds 1-3, seven lanes in eight had nothing to do), each lane moves 16 bytes per step with one unaligned 16-byte
// load and one 16-byte store (stride % 16 == 0 and an aligned dst) — round 1 moved single bytes.
// ---------------------------------------------------------------------------------------------
__global__ void gather_fields_kernel(const uint8_t* __restrict__ bytes, u64 bytes_len, const u64* __restrict__ begin,
                                     const u64* __restrict__ end, u64 n_records, uint8_t* __restrict__ dst, u32 stride,
                                     u32* __restrict__ len, int wide, u32 gshift) {
    const u32 lanes = 1u << gshift;
    const u32 sub = threadIdx.x & (lanes - 1u);
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> gshift; i < n_records;
         i += ((u64)gridDim.x * blockDim.x) >> gshift) {
        const u64 b = begin[i], e = end[i];
        const u64 n = e > b ? e - b : 0;
        if (sub == 0 && len) len[i] = (u32)(n > 0xffffffffull ? 0xffffffffull : n);
        uint8_t* const row = dst + i * stride;
        if (wide) {
            for (u32 k = sub * 16u; k < stride; k += lanes * 16u) {
                u32x4 v = {0, 0, 0, 0};
                if (k < n) {
                    if (b + k + 16 <= bytes_len) {
                        v = *reinterpret_cast<const u32x4u*>(bytes + b + k);
                    } else {  // the last 15 bytes of the buffer: never read past it
                        uint8_t t[16];
                        for (u32 j = 0; j < 16; ++j) t[j] = b + k + j < bytes_len ? bytes[b + k + j] : (uint8_t)0;
                        v = *reinterpret_cast<const u32x4*>(t);
                    }
                    if (n - k < 16) {  // zero the bytes past the field's end
                        const u32 keep = (u32)(n - k);
                        const u32 full = keep >> 2, part = keep & 3u;
                        const u32 m = part ? (0xffffffffu >> (32u - 8u * part)) : 0u;
                        v.x = full > 0 ? v.x : (full == 0 ? v.x & m : 0u);
                        v.y = full > 1 ? v.y : (full == 1 ? v.y & m : 0u);
                        v.z = full > 2 ? v.z : (full == 2 ? v.z & m : 0u);
                        v.w = full > 3 ? v.w : (full == 3 ? v.w & m : 0u);
                    }
                }
                *reinterpret_cast<u32x4*>(row + k) = v;
            }
        } else {
            for (u32 k = sub; k < stride; k += lanes) row[k] = k < n ? bytes[b + k] : (uint8_t)0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// spans + gather in one kernel, straight from the tape (csvsimd_column_frequency_device): field `field` of rows
// [0, n_rows) of a chunk -> rows of dst, lengths -> len, the longest length seen -> *longest.  The file's end is taken
// from the tape itself (its last entry is the last structural byte: the buffer is at least that long + 1).
// ---------------------------------------------------------------------------------------------
__global__ void gather_column_kernel(const uint8_t* __restrict__ bytes, const u64* __restrict__ index, u64 index_len, u64 first_key,
                                     u64 jump, u32 field, u64 n_rows, uint8_t* __restrict__ dst, u32 stride, u32* __restrict__ len,
                                     u64* __restrict__ longest, u32 gshift) {
    const u32 lanes = 1u << gshift;
    const u32 sub = threadIdx.x & (lanes - 1u);
    const u64 bytes_len = index[index_len - 1] + 1;
    u64 m = 0;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> gshift; i < n_rows;
         i += ((u64)gridDim.x * blockDim.x) >> gshift) {
        const u64 k0 = first_key + i * jump + field;
        const u64 b = index[k0] + 1, e = index[k0 + 1];
        const u64 n = e > b ? e - b : 0;
        m = n > m ? n : m;
        if (sub == 0) len[i] = (u32)(n > 0xffffffffull ? 0xffffffffull : n);
        uint8_t* const row = dst + i * stride;
        for (u32 k = sub * 16u; k < stride; k += lanes * 16u) {
            u32x4 v = {0, 0, 0, 0};
            if (k < n) {
                if (b + k + 16 <= bytes_len) {
                    v = *reinterpret_cast<const u32x4u*>(bytes + b + k);
                } else {  // the last 15 bytes of the buffer: never read past it
                    uint8_t t[16];
                    for (u32 j = 0; j < 16; ++j) t[j] = b + k + j < bytes_len ? bytes[b + k + j] : (uint8_t)0;
                    v = *reinterpret_cast<const u32x4*>(t);
                }
                if (n - k < 16) {  // zero the bytes past the field's end
                    const u32 keep = (u32)(n - k);
                    const u32 full = keep >> 2, part = keep & 3u;
                    const u32 mk = part ? (0xffffffffu >> (32u - 8u * part)) : 0u;
                    v.x = full > 0 ? v.x : (full == 0 ? v.x & mk : 0u);
                    v.y = full > 1 ? v.y : (full == 1 ? v.y & mk : 0u);
                    v.z = full > 2 ? v.z : (full == 2 ? v.z & mk : 0u);
                    v.w = full > 3 ? v.w : (full == 3 ? v.w & mk : 0u);
                }
            }
            *reinterpret_cast<u32x4*>(row + k) = v;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(m >> 32), d) << 32) | (u32)__shfl_xor((int)(u32)m, d);
        m = o > m ? o : m;
    }
    // (one atomic per wave that has something new to say: see chunk_spans_kernel)
    if ((threadIdx.x & 63u) == 0 && m > __hip_atomic_load((const unsigned long long*)longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax((unsigned long long*)longest, (unsigned long long)m);
}

// ---------------------------------------------------------------------------------------------
// frequency count of a column of the ROW-MAJOR file (csvsimd_column_frequency_device).  Rounds 1-3 kept a second
// counting algorithm here (64-bit hashes in a 32-byte-slot device table + a verification pass over every record:
// 11-22 x the column's bytes in traffic).  Now the column is gathered once into fixed-stride rows (the two kernels
// above) and counted by the ONE implementation of columnar_kernels.hip; what is left here is the glue:
//   chunk_spans_kernel   (above) also reports the longest field of the column: the gather's stride
// and the count's second pass writes csvsimd_freq_entry {record id, text span, count} itself (FreqWideOut).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// search: one bit per row of the chunk (bit i of word i / 64 = row i matches) + the number of matches.
//   mode 0 = field == needle, 1 = field starts with needle, 2 = field contains needle
// One lane per row; the needle sits in LDS.  "contains" is the plain quadratic scan with a first-byte filter
// (fields are tens of bytes): bytes.find's definition, not its algorithm.
// ---------------------------------------------------------------------------------------------
static constexpr u32 kMaxNeedle = 256;

// 8 bytes at p, never touching a byte at or past `limit` (the end of the file buffer): bytes past it read as 0
__device__ __forceinline__ u64 load8_guarded(const uint8_t* p, const uint8_t* limit) {
    if (p + 8 <= limit) return *reinterpret_cast<const u64u*>(p);
    u64 v = 0;
    for (u32 j = 0; j < 8 && p + j < limit; ++j) v |= (u64)p[j] << (8 * j);
    return v;
}

__global__ __launch_bounds__(256) void search_kernel(const Column c, const uint8_t* __restrict__ bytes_end,
                                                     const uint8_t* __restrict__ needle, u32 m, int mode,
                                                     u64* __restrict__ bitmap, u64* __restrict__ count) {
    __shared__ u64 s_needle[kMaxNeedle / 8 + 1];  // the needle as little-endian words, zero padded
    for (u32 k = threadIdx.x; k < kMaxNeedle / 8 + 1; k += blockDim.x) {
        u64 w = 0;
        for (u32 j = 0; j < 8; ++j)
            if (8 * k + j < m) w |= (u64)needle[8 * k + j] << (8 * j);
        s_needle[k] = w;
    }
    __syncthreads();
    const u64 n_words = (c.n_rows + 63) / 64;
    const u32 lane = threadIdx.x & 63u;
    const u32 head = m < 8 ? m : 8;                                   // bytes of the needle in its first word
    const u64 head_mask = head == 8 ? ~0ull : ((1ull << (8 * head)) - 1ull);
    const u64 needle0 = s_needle[0];
    u32 hits = 0;
    for (u64 word = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; word < n_words;
         word += ((u64)gridDim.x * blockDim.x) >> 6) {
        const u64 i = word * 64 + lane;
        bool match = false;
        if (i < c.n_rows) {
            u64 b, e;
            field_span(c, i, b, e);
            const u64 n = e - b;
            const uint8_t* p = c.bytes + b;
            if (mode != 2) {
                // equals / starts with: the first m bytes, a word at a time
                match = mode == 0 ? n == m : n >= m;
                for (u32 k = 0; 8 * k < m && match; ++k) {
                    const u32 left = m - 8 * k;
                    const u64 mask = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
                    match = ((load8_guarded(p + 8 * k, bytes_end) ^ s_needle[k]) & mask) == 0;
                }
            } else if (m == 0) {
                match = true;
            } else if (n >= m) {
                // contains: every start position `at` sees the 8 bytes from it as a view of two loaded words — one
                // load per 8 positions instead of one per byte; needles longer than 8 bytes verify the rest on a hit
                const u64 last = n - m;  // last start position
                u64 cur = load8_guarded(p, bytes_end);
                for (u64 base = 0; base <= last && !match; base += 8) {
                    const u64 nxt = load8_guarded(p + base + 8, bytes_end);
#pragma unroll
                    for (u32 j = 0; j < 8; ++j) {
                        const u64 view = j ? (cur >> (8 * j)) | (nxt << (64 - 8 * j)) : cur;
                        if (base + j <= last && ((view ^ needle0) & head_mask) == 0) {
                            bool ok = true;
                            for (u32 k = 1; 8 * k < m && ok; ++k) {
                                const u32 left = m - 8 * k;
                                const u64 mask = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
                                ok = ((load8_guarded(p + base + j + 8 * k, bytes_end) ^ s_needle[k]) & mask) == 0;
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
}

// bitmap -> ascending record ids.  Two small passes keep the order without any global atomics on the output:
//   pass 1: popcount per 4096-bit block -> block_counts;  (host-free) pass 2: one workgroup scans the block
//   counts;  pass 3: every block writes its ids from its base.
// bits at positions >= n_rows in the last word are not rows: a caller's bitmap may hold anything there
__device__ __forceinline__ u64 bitmap_word(const u64* __restrict__ bitmap, u64 w, u64 n_words, u64 n_rows) {
    if (w >= n_words) return 0;
    u64 bits = bitmap[w];
    if (w == n_words - 1 && (n_rows & 63u)) bits &= (1ull << (n_rows & 63u)) - 1ull;
    return bits;
}
__global__ void bitmap_block_count_kernel(const u64* __restrict__ bitmap, u64 n_words, u64 n_rows, u64* __restrict__ block_counts) {
    const u64 blk = blockIdx.x;
    u32 c = 0;
    for (u32 k = threadIdx.x; k < 64; k += blockDim.x) c += (u32)__builtin_popcountll(bitmap_word(bitmap, blk * 64 + k, n_words, n_rows));
    for (int d = 32; d >= 1; d >>= 1) c += (u32)__shfl_xor((int)c, d);
    if (threadIdx.x == 0) block_counts[blk] = c;
}
__global__ __launch_bounds__(1024) void bitmap_scan_kernel(u64* __restrict__ block_counts, u64 n_blocks, u64* __restrict__ total) {
    __shared__ u64 s_part[1024];
    // each thread owns a contiguous run of blocks
    const u64 per = (n_blocks + blockDim.x - 1) / blockDim.x;
    const u64 a = (u64)threadIdx.x * per, b = a + per < n_blocks ? a + per : n_blocks;
    u64 sum = 0;
    for (u64 k = a; k < b; ++k) sum += block_counts[k];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (u32 k = 0; k < blockDim.x; ++k) {
            const u64 v = s_part[k];
            s_part[k] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    u64 run = s_part[threadIdx.x];
    for (u64 k = a; k < b; ++k) {
        const u64 v = block_counts[k];
        block_counts[k] = run;  // exclusive prefix
        run += v;
    }
}
__global__ void bitmap_select_kernel(const u64* __restrict__ bitmap, u64 n_words, u64 n_rows, const u64* __restrict__ block_base,
                                     u64 first_row, u64* __restrict__ out, u64 out_cap) {
    // one wave per 4096-bit block: lane l owns word l of the block
    const u64 blk = blockIdx.x;
    const u32 lane = threadIdx.x;
    const u64 w = blk * 64 + lane;
    u64 bits = bitmap_word(bitmap, w, n_words, n_rows);
    u32 c = (u32)__builtin_popcountll(bits);
    u32 incl = c;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 v = (u32)__shfl_up((int)incl, d);
        if (lane >= (u32)d) incl += v;
    }
    u64 at = block_base[blk] + (incl - c);
    while (bits) {
        const u32 b = (u32)__builtin_ctzll(bits);
        bits &= bits - 1;
        if (at < out_cap) out[at] = first_row + w * 64 + b;
        ++at;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static u32 grid_for(u64 items, u32 per_block, u32 cap) {
    u64 blocks = (items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    return (u32)(blocks > cap ? cap : blocks);
}

hipError_t launch_chunk_spans(const void* dindex, u64 first_key, u64 jump, u32 field, u32 fields, u64 n_rows, void* d_begin,
                              void* d_end, hipStream_t stream, void* d_longest) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_spans_kernel, dim3(grid_for(n_rows, 256, 4096)), dim3(256), 0, stream, (const u64*)dindex,
                       first_key, jump, field, fields, n_rows, (u64*)d_begin, (u64*)d_end, (u64*)d_longest);
    return hipGetLastError();
}

hipError_t launch_gather_fields(const void* dbytes, u64 bytes_len, const void* d_begin, const void* d_end, u64 n_records,
                                void* d_dst, u32 stride, void* d_len, hipStream_t stream) {
    if (n_records == 0 || stride == 0) return hipSuccess;
    const int wide = (stride % 16 == 0) && (((uintptr_t)d_dst & 15) == 0);
    u32 gshift = 0;  // lanes per record: one per 16-byte piece of the row (per byte on the narrow path), a power of two <= 16
    while (gshift < 4 && (1u << gshift) < (wide ? (stride + 15) / 16 : stride)) ++gshift;
    hipLaunchKernelGGL(gather_fields_kernel, dim3(grid_for(n_records << gshift, 256, 16384)), dim3(256), 0, stream,
                       (const uint8_t*)dbytes, bytes_len, (const u64*)d_begin, (const u64*)d_end, n_records, (uint8_t*)d_dst,
                       stride, (u32*)d_len, wide, gshift);
    return hipGetLastError();
}

static Column make_column(const void* dbytes, const void* dindex, u64 first_key, u64 jump, u64 n_rows, u32 field) {
    Column c;
    c.bytes = (const uint8_t*)dbytes;
    c.index = (const u64*)dindex;
    c.first_key = first_key;
    c.jump = jump;
    c.n_rows = n_rows;
    c.first_row = first_key / jump;
    c.field = field;
    return c;
}

hipError_t launch_gather_column(const void* dbytes, const void* dindex, u64 index_len, u64 first_key, u64 jump, u32 field, u64 n_rows,
                                void* d_dst, u32 stride, void* d_len, void* d_longest, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    u32 gshift = 0;  // lanes per record: one per 16-byte piece of the row, a power of two <= 16
    while (gshift < 4 && (1u << gshift) < (stride + 15) / 16) ++gshift;
    hipLaunchKernelGGL(gather_column_kernel, dim3(grid_for(n_rows << gshift, 256, 16384)), dim3(256), 0, stream,
                       (const uint8_t*)dbytes, (const u64*)dindex, index_len, first_key, jump, field, n_rows, (uint8_t*)d_dst, stride,
                       (u32*)d_len, (u64*)d_longest, gshift);
    return hipGetLastError();
}

hipError_t launch_search(const void* dbytes, u64 bytes_len, const void* dindex, u64 first_key, u64 jump, u64 n_rows, u32 field,
                         const void* d_needle, u32 needle_len, int mode, void* d_bitmap, void* d_count, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const Column c = make_column(dbytes, dindex, first_key, jump, n_rows, field);
    hipLaunchKernelGGL(search_kernel, dim3(grid_for(n_rows, 256, 8192)), dim3(256), 0, stream, c,
                       (const uint8_t*)dbytes + bytes_len, (const uint8_t*)d_needle, needle_len, mode, (u64*)d_bitmap,
                       (u64*)d_count);
    return hipGetLastError();
}

hipError_t launch_bitmap_select(const void* d_bitmap, u64 n_rows, u64 first_row, void* d_block_scratch, void* d_out,
                                u64 out_cap, void* d_total, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const u64 n_words = (n_rows + 63) / 64;
    const u64 n_blocks = (n_words + 63) / 64;
    if (n_blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bitmap_block_count_kernel, dim3((u32)n_blocks), dim3(64), 0, stream, (const u64*)d_bitmap, n_words,
                       n_rows, (u64*)d_block_scratch);
    hipLaunchKernelGGL(bitmap_scan_kernel, dim3(1), dim3(1024), 0, stream, (u64*)d_block_scratch, n_blocks, (u64*)d_total);
    hipLaunchKernelGGL(bitmap_select_kernel, dim3((u32)n_blocks), dim3(64), 0, stream, (const u64*)d_bitmap, n_words, n_rows,
                       (const u64*)d_block_scratch, first_row, (u64*)d_out, out_cap);
    return hipGetLastError();
}

}  // namespace csvsimd
