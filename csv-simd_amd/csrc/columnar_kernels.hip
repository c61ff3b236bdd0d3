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
#include <string.h>

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

// The row hash.  Its quality only decides how evenly slots and partitions fill (values are equal when their BYTES are); its
// cost is paid once per record by pass 1: the 64-bit finalisers used first (nine 64 x 64-bit multiplies per 32-byte row = 33
// quarter-rate instructions) were ~5 us of a 23-us kernel.  This one folds 32 x 32 -> 64-bit products (one v_mad_u64_u32 per
// eight bytes, the scheme of wyhash's 32-bit variant): two words of state, the row xored in eight bytes at a time, each step
// replacing the state by the two halves of (s0 ^ k0) * (s1 ^ k1).  The low bits of a product are poorly mixed, so the
// result's halves are s0 ^ s1 after two and after three closing steps.
// The old halves are folded back in (round 5): a bare product is absorbing — a row whose first dword made
// s0 ^ 0x53c5ca59 zero zeroed the whole state, whatever its next four bytes held, so a crafted (or binary) column could put
// more than a table's worth of DISTINCT values on one 64-bit hash and make the count fail with `overflow`, every time.
__device__ __forceinline__ void cf_mix(u32& s0, u32& s1) {
    const u64 c = (u64)(s0 ^ 0x53c5ca59u) * (u64)(s1 ^ 0x74743c1bu);
    const u32 o0 = s0;
    s0 = (u32)c + s1;
    s1 = (u32)(c >> 32) ^ o0;
}
__device__ __forceinline__ void cf_absorb(u32& s0, u32& s1, const u32x4c v) {
    s0 ^= v.x;
    s1 ^= v.y;
    cf_mix(s0, s1);
    s0 ^= v.z;
    s1 ^= v.w;
    cf_mix(s0, s1);
}
__device__ __forceinline__ u64 cf_close(u32 s0, u32 s1) {
    cf_mix(s0, s1);
    cf_mix(s0, s1);
    const u32 lo = s0 ^ s1;
    cf_mix(s0, s1);
    return ((u64)(s0 ^ s1) << 32) | lo;
}
// hash of record i's padded bytes + length, 16 bytes per step from aligned, coalesced loads
__device__ __forceinline__ u64 hash_row(const ColView& c, u64 i, u32 len) {
    const u32x4c* p = reinterpret_cast<const u32x4c*>(c.col + i * c.stride);
    u32 s0 = 0x243F6A88u ^ len, s1 = 0x85A308D3u;
    for (u32 k = 0; k < (c.stride >> 4); ++k) cf_absorb(s0, s1, p[k]);
    return cf_close(s0, s1);
}
// the same function of a row that is already in registers (strides of 16 and 32 bytes: b is ignored for 16)
__device__ __forceinline__ u64 hash_regs(u32 stride, u32 len, const u32x4c a, const u32x4c b) {
    u32 s0 = 0x243F6A88u ^ len, s1 = 0x85A308D3u;
    cf_absorb(s0, s1, a);
    if (stride > 16) cf_absorb(s0, s1, b);
    return cf_close(s0, s1);
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
// exact frequency count on a column, in two passes and WITHOUT a global hash table (round 4).
//
// Rounds 2-3 inserted every value into an open-addressing table in device memory: a 64-MiB table cleared on every
// call, one returning atomic per new value on a random 16-byte slot (each a whole line through the memory-side
// atomic path: 3.8-6.1 x the algorithmic bytes, profiles/r03_pmc_consumers.json), a scan of all 4 Mi slots to find
// the 100 that were used.  Here no record ever touches a shared word in device memory:
//
//   pass 1 (colfreq_partition_kernel)  a workgroup owns a slab of 8 192 consecutive records.  It hashes every record,
//       aggregates repeated values in an LDS table (a column of few distinct values collapses to a handful of tuples
//       per workgroup), and writes one TUPLE (first record, count, 32 hash bits) per surviving value into its own block
//       of the scratch, sorted by PARTITION (hash bits 40..) — a counting sort in LDS, so the block is written once,
//       in place, and the per-partition offsets go to a small table.
//   pass 2 (colfreq_reduce_kernel)     a workgroup owns a partition: it gathers that partition's run from every block
//       (the offsets table says where), merges the tuples in an LDS table that is big enough for all of them (values are
//       equal only if hash bits AND bytes are: a representative record is compared) and writes the partition's entries
//       (first record, count) straight to the output, reserving its range with ONE atomic per workgroup.
//
// Nothing is cleared per call (the status words are written, not accumulated, or zeroed by pass 1), nothing is scanned
// that was not written, and the call is two launches on the caller's stream: asynchronous and capturable.  (Round 5: from
// 4 Mi records on a third launch goes first — colfreq_stream_kernel below, which counts long columns of few values in one
// streaming pass per CU and hands the shares it cannot count to pass 1.)
// Algorithmic bytes: the column and its lengths read once, 16 bytes per distinct value written; the tuples add 12 bytes
// written + read per value that survives pass 1's aggregation.
// ---------------------------------------------------------------------------------------------
// rows i and j equal (lengths and bytes)?  Both lengths and both rows are requested at once: one trip to memory instead
// of three dependent ones (length of i, length of j, rows) — pass 2 calls this once per tuple of a column of few values.
__device__ __forceinline__ bool rows_equal_eager(const ColView& c, u64 i, u64 j) {
    if (c.stride > 32) return rows_equal(c, i, j, c.len ? c.len[i] : c.stride);
    const u32x4c* const p = reinterpret_cast<const u32x4c*>(c.col + i * c.stride);
    const u32x4c* const q = reinterpret_cast<const u32x4c*>(c.col + j * c.stride);
    const bool two = c.stride > 16;
    const u32 li = c.len ? c.len[i] : 0u, lj = c.len ? c.len[j] : 0u;
    const u32x4c x0 = p[0], y0 = q[0];
    const u32x4c x1 = two ? p[1] : u32x4c{0, 0, 0, 0}, y1 = two ? q[1] : u32x4c{0, 0, 0, 0};
    return li == lj && x0.x == y0.x && x0.y == y0.y && x0.z == y0.z && x0.w == y0.w && x1.x == y1.x && x1.y == y1.y &&
           x1.z == y1.z && x1.w == y1.w;
}
struct ColFreqStatus {  // == csvsimd_colfreq_status
    u64 n_records, n_distinct, truncated, overflow;
};
struct ColFreqEntry {  // == csvsimd_colfreq_entry
    u64 first_record, count;
};
#ifndef CSVSIMD_CF_SLAB
#define CSVSIMD_CF_SLAB 8192
#endif
static constexpr u32 kCfSlab = CSVSIMD_CF_SLAB;  // records per pass-1 workgroup
static constexpr u32 kCfPerThread = 8;       // records per pass-1 thread
static constexpr u32 kCfThreads1 = kCfSlab / kCfPerThread;
static constexpr u32 kCfLds = kCfThreads1;   // pass-1 LDS table slots (one per thread when the table goes out)
static constexpr u32 kCfMaxParts = 4096;
static constexpr u32 kCfCap2 = 8192;         // pass-2 LDS table slots.  (4 096 slots and twice the partitions, two workgroups per
                                             // CU instead of one: 20-25 % slower on every cardinality measured)
static constexpr u32 kCfRound2 = 6144;       // tuples a pass-2 round may be asked to hold (distinct values <= tuples)
#ifndef CSVSIMD_CF_THREADS2
#define CSVSIMD_CF_THREADS2 1024
#endif
static constexpr u32 kCfThreads2 = CSVSIMD_CF_THREADS2;
static constexpr u32 kCfBatch2 = 1;         // tuples a pass-2 thread has in flight (4 and 8 were measured: see the loop)
static constexpr u32 kCfTupleWords = 3;      // {first record, count, low 32 hash bits}
static constexpr u32 kCfTickets = 16;        // pass 2's ticket counters, a 128-byte line each
static constexpr u32 kCfTicketBytes = kCfTickets * 128;
static constexpr u32 kCfMaxShares = 1024;    // colfreq_stream_kernel's workgroups (one per CU)
static constexpr u32 kCfShareFlagBytes = kCfMaxShares * 4;

struct ColFreqGeom {
    u32 slabs;       // W: pass-1 workgroups = tuple blocks
    u32 parts;       // P: partitions (a power of two)
    u64 offs_bytes;  // u16 offs[P + 2][W]: row p = where partition p starts in every block, row P = the block's tuple
                     // count, row P + 1 = records of the slab longer than the stride
    u64 bytes;       // the whole scratch: [offs | tuples: W blocks of kCfSlab x 12 bytes | pass 2's tickets | the shares' flags]
};
static ColFreqGeom colfreq_geom(u64 n_rows) {
    ColFreqGeom g;
    g.slabs = (u32)((n_rows + kCfSlab - 1) / kCfSlab);
    if (g.slabs == 0) g.slabs = 1;
    u64 want = (n_rows + 4095) / 4096;  // ~4 096 records per partition: a pass-2 table is half full when all are distinct
    u32 p = 1;
    while (p < want && p < kCfMaxParts) p <<= 1;
    g.parts = p;
    g.offs_bytes = (((u64)(g.parts + 2) * g.slabs * 2) + 255) & ~255ull;
    g.bytes = g.offs_bytes + (u64)g.slabs * kCfSlab * kCfTupleWords * 4 + kCfTicketBytes + kCfShareFlagBytes;
    return g;
}
u64 colfreq_scratch_bytes(u64 n_rows) { return colfreq_geom(n_rows).bytes; }

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)v, d);
        if ((threadIdx.x & 63u) >= (u32)d) v += o;
    }
    return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (u32)__shfl_xor((int)v, d);
    return v;
}
__device__ __forceinline__ u32 cf_part(u64 h, u32 parts) { return (u32)(h >> 40) & (parts - 1u); }

// dev builds only (make EXTRA=-DCSVSIMD_CF_TRACE): thread 0 of every workgroup leaves s_memrealtime stamps (100 MHz) at
// the phase boundaries of both passes; scripts/r04_cf_trace.py reads them back.  The product build has none of this.
#ifdef CSVSIMD_CF_TRACE
__device__ u64 g_cf_trace[2][4096][8];
#define CF_STAMP(pass, wg, idx)                                                          \
    if (threadIdx.x == 0 && (wg) < 4096) {                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                   \
        g_cf_trace[pass][wg][idx] = __builtin_amdgcn_s_memrealtime();                    \
    }
#else
#define CF_STAMP(pass, wg, idx)
#endif

// ROWS_IN_LDS (strides up to 32 bytes): the table keeps a copy of every slot's representative row.  A record that meets
// its value in the table compares against LDS instead of gathering the representative's row from memory — on a column
// of few values that gather (64 lanes, ~64 different lines per load) was 12 of the kernel's 34 us (ablation, round 4).
// The thread's own eight rows stay in registers from the hashing on: reading them again for the comparison was a trip
// to the L2 per record, one after the other (the L1 has long moved on: a slab is 256 KiB) — 13 of the remaining 30 us
// (phase stamps, profiles/r04_cf_trace.txt).
template <bool ROWS_IN_LDS>
__global__ __launch_bounds__(kCfThreads1, 4) void colfreq_partition_kernel(const ColView c, unsigned short* __restrict__ offs,
                                                                       u32* __restrict__ tuples, u32 parts, u32 slabs,
                                                                       ColFreqStatus* __restrict__ status, u32* __restrict__ ticket,
                                                                       const u32* __restrict__ share_done, u32 per_share) {
    __shared__ u64 s_key[kCfLds];    // hash bits 32..63 << 32 | (record - r0) + 1; 0 = empty
    __shared__ u32 s_count[kCfLds];
    __shared__ u32 s_first[kCfLds];  // smallest (record - r0) holding the slot's value
    __shared__ u32 s_hlo[kCfLds];    // hash bits 0..31 of the slot's value (written by the claimer, read after the barrier)
    __shared__ u32 s_rlen[kCfLds];   // ROWS_IN_LDS: the representative's length; bit 31 = its row is in s_buf (set last)
    __shared__ u32 s_hist[kCfMaxParts];
    __shared__ u32 s_scan[kCfThreads1];
    __shared__ u32 s_fill, s_trunc;
    // phase A: the representatives' rows (kCfLds x 32 bytes); phase B: this block's tuples, sorted, before they leave
    __shared__ __attribute__((aligned(16))) u32 s_buf[kCfSlab * kCfTupleWords];
    static_assert(kCfSlab * kCfTupleWords * 4 >= kCfLds * 32, "the tuple staging doubles as the row cache");
    const u32 t = threadIdx.x;
    // one slab per workgroup — or, behind colfreq_stream_kernel (long columns), one workgroup per CU that walks the slabs and
    // skips those whose share the streaming kernel has counted already (4 096 workgroups of 140 KiB that only look at a flag
    // took 8.5 us to pass through the CUs)
    for (u32 w = blockIdx.x; w < slabs; w += gridDim.x) {
    if (share_done && share_done[w / per_share]) continue;
    CF_STAMP(0, w, 0)
    for (u32 k = t; k < kCfLds; k += kCfThreads1) {
        s_key[k] = 0;
        s_count[k] = 0;
        s_first[k] = 0xffffffffu;
        s_rlen[k] = 0;
    }
    for (u32 k = t; k < parts; k += kCfThreads1) s_hist[k] = 0;
    if (t == 0) {
        s_fill = 0;
        s_trunc = 0;
        if (w == 0 && !share_done) {  // pass 2 (the next launch on this stream) adds to the one and may set the other
            status->n_distinct = 0;
            status->overflow = 0;
            for (u32 k = 0; k < kCfTickets; ++k) ticket[k * 32] = 0;
        }
    }
    __syncthreads();
    const u64 r0 = (u64)w * kCfSlab;
    const u32 nrec = (u32)(c.n_rows - r0 < kCfSlab ? c.n_rows - r0 : kCfSlab);
    // slot s: bytes 0..15 in s_rows[s], bytes 16..31 in s_rows[kCfLds + s] (two arrays of 16-byte elements: a wave's
    // reads of random slots meet half as many bank conflicts as with 32-byte elements)
    u32x4c* const s_rows = reinterpret_cast<u32x4c*>(s_buf);
    // ---- phase A: hash every record; repeated values meet in the LDS table -----------------------------------------
    u64 hs[kCfPerThread];
    u32 single = 0, trunc = 0;  // bit j: record j of this thread goes out as its own tuple
    // a batch of this thread's rows is requested before the first one is used: independent load -> hash chains instead
    // of round trips one after the other (the probing below is a chain of its own).  ROWS_IN_LDS keeps the batch's rows in
    // registers for the comparisons, so a batch is four records (eight would not fit a 1 024-thread workgroup's 128).
    constexpr u32 kBatch = ROWS_IN_LDS ? 4 : kCfPerThread;
#pragma unroll
    for (u32 jb = 0; jb < kCfPerThread; jb += kBatch) {
        u32 lens[kBatch];
        u32x4c ra[kBatch], rb[kBatch];  // ROWS_IN_LDS: the rows themselves
#pragma unroll
        for (u32 jj = 0; jj < kBatch; ++jj) {
            const u32 j = jb + jj;
            const u32 li = j * kCfThreads1 + t;  // consecutive lanes, consecutive records: coalesced loads
            hs[j] = 0;
            lens[jj] = 0;
            ra[jj] = rb[jj] = u32x4c{0, 0, 0, 0};
            if (li < nrec) {
                const u64 i = r0 + li;
                const u32 len = c.len ? c.len[i] : c.stride;
                lens[jj] = len;
                if (len > c.stride) ++trunc;
                if (ROWS_IN_LDS) {
                    const u32x4c* const p0 = reinterpret_cast<const u32x4c*>(c.col + i * c.stride);
                    ra[jj] = p0[0];
                    if (c.stride > 16) rb[jj] = p0[1];
                    hs[j] = hash_regs(c.stride, len, ra[jj], rb[jj]);
                } else {
                    hs[j] = hash_row(c, i, len);
                }
            }
        }
        if (jb == 0) { CF_STAMP(0, w, 1) }
        if (jb == kBatch) { CF_STAMP(0, w, 7) }
#pragma unroll
        for (u32 jj = 0; jj < kBatch; ++jj) {
            const u32 j = jb + jj;
            const u32 li = j * kCfThreads1 + t;
            if (li >= nrec) continue;
            const u64 i = r0 + li;
            const u32 len = lens[jj];
            const u64 h = hs[j];
            const u64 mine = (h & 0xffffffff00000000ull) | (u64)(li + 1u);
            bool done = false;
            // a column of many distinct values fills the table with its first rows; from then on new values only find
            // full probe sequences, so the probing is limited to a look at the home slot
            const int max_probes = s_fill < kCfLds * 3 / 4 ? 8 : 1;
            u32 s = (u32)h & (kCfLds - 1);
            for (int p = 0; p < max_probes && !done; ++p, s = (s + 1) & (kCfLds - 1)) {
                u64 old = s_key[s];
                // (key and flag are requested together; the row is read after the flag, and a wave's LDS operations
                // execute in order: a flag that is up means the row behind it is complete)
                u32 rl = ROWS_IN_LDS ? __hip_atomic_load(&s_rlen[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                if (old == 0) old = atomicCAS((unsigned long long*)&s_key[s], 0ull, (unsigned long long)mine);
                if (old == 0) {
                    s_hlo[s] = (u32)h;
                    atomicAdd(&s_fill, 1u);
                    atomicAdd(&s_count[s], 1u);
                    atomicMin(&s_first[s], li);
                    if (ROWS_IN_LDS) {
                        // this record's row becomes the slot's copy; the flag goes up behind it (release at workgroup
                        // scope: LDS operations of a wave complete in order)
                        s_rows[s] = ra[jj];
                        s_rows[kCfLds + s] = rb[jj];
                        __hip_atomic_store(&s_rlen[s], len | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    done = true;
                } else if ((old >> 32) == (mine >> 32)) {
                    bool eq;
                    if (ROWS_IN_LDS && (rl >> 31)) {
                        asm volatile("" ::: "memory");
                        const u32x4c a0 = ra[jj], a1 = rb[jj];
                        const u32x4c b0 = s_rows[s], b1 = s_rows[kCfLds + s];
                        eq = (rl & 0x7fffffffu) == len && a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w &&
                             a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w;
                    } else {
                        eq = rows_equal(c, i, r0 + ((u32)old - 1u), len);  // (the copy is not there yet, or strides > 32)
                    }
                    if (eq) {
                        atomicAdd(&s_count[s], 1u);
                        // (a thread meets its records in ascending order and so does the workgroup, roughly: after the
                        // first round the slot's smallest record is rarely beaten, and a read is cheaper than a contended
                        // atomic)
                        if (li < s_first[s]) atomicMin(&s_first[s], li);
                        done = true;
                    }
                }
            }
            if (!done) single |= 1u << j;
        }
        if (jb == 0) { CF_STAMP(0, w, 6) }
    }
    CF_STAMP(0, w, 2)
    if (trunc) atomicAdd(&s_trunc, trunc);
    __syncthreads();
    CF_STAMP(0, w, 3)
    // ---- phase B: counting sort of this workgroup's tuples by partition ----------------------------------------------
    // every tuple takes a rank within its partition (one returning LDS atomic), the histogram is scanned, and the tuple
    // goes to slot prefix[partition] + rank of the staging buffer, which then leaves as one contiguous, coalesced write
    // (scattered straight to memory the 12-byte tuples were 64 different lines per wave store: 12 of the kernel's 35 us
    // on a column of distinct values)
    u32 rank[kCfPerThread + 1], part[kCfPerThread + 1];
#pragma unroll
    for (u32 j = 0; j < kCfPerThread; ++j) {
        part[j] = 0;
        rank[j] = 0;
        if ((single >> j) & 1u) {
            part[j] = cf_part(hs[j], parts);
            rank[j] = atomicAdd(&s_hist[part[j]], 1u);
        }
    }
    const bool own_slot = t < kCfLds && s_key[t] != 0;  // (kCfLds == kCfThreads1: one table slot per thread)
    part[kCfPerThread] = rank[kCfPerThread] = 0;
    if (own_slot) {
        part[kCfPerThread] = cf_part(s_key[t], parts);  // (the key's upper half = hash bits 32..63)
        rank[kCfPerThread] = atomicAdd(&s_hist[part[kCfPerThread]], 1u);
    }
    __syncthreads();
    // exclusive scan of s_hist[0, parts): thread t owns bins [t * per, (t + 1) * per)
    const u32 per = (parts + kCfThreads1 - 1) / kCfThreads1;  // <= 4
    u32 local = 0;
    for (u32 k = 0; k < per; ++k) {
        const u32 bin = t * per + k;
        if (bin < parts) local += s_hist[bin];
    }
    const u32 incl = wave_incl_scan_u32(local);
    if ((t & 63u) == 63u) s_scan[t >> 6] = incl;
    __syncthreads();
    u32 run = incl - local;  // exclusive prefix of this thread's first bin: within the wave, + the waves before it
    for (u32 k = 0; k < (t >> 6); ++k) run += s_scan[k];
    u32 total = 0;
    for (u32 k = 0; k < kCfThreads1 / 64; ++k) total += s_scan[k];
    for (u32 k = 0; k < per; ++k) {
        const u32 bin = t * per + k;
        if (bin < parts) {
            const u32 cnt = s_hist[bin];
            s_hist[bin] = run;  // the bin's start within the block
            offs[(u64)bin * slabs + w] = (unsigned short)run;
            run += cnt;
        }
    }
    if (t == 0) {
        offs[(u64)parts * slabs + w] = (unsigned short)total;             // tuples in this block (<= 8 192)
        offs[(u64)(parts + 1) * slabs + w] = (unsigned short)s_trunc;     // (<= 8 192 as well)
    }
    __syncthreads();  // (the row copies in s_buf are no longer read: phase A ended two barriers ago)
#pragma unroll
    for (u32 j = 0; j < kCfPerThread; ++j) {
        if ((single >> j) & 1u) {
            u32* const q = s_buf + (s_hist[part[j]] + rank[j]) * kCfTupleWords;
            q[0] = (u32)r0 + j * kCfThreads1 + t;
            q[1] = 1u;
            q[2] = (u32)hs[j];
        }
    }
    if (own_slot) {
        u32* const q = s_buf + (s_hist[part[kCfPerThread]] + rank[kCfPerThread]) * kCfTupleWords;
        q[0] = (u32)r0 + s_first[t];  // the smallest record holding the value stands for it from here on
        q[1] = s_count[t];
        q[2] = s_hlo[t];
    }
    __syncthreads();
    CF_STAMP(0, w, 4)
    u32* const block = tuples + (u64)w * kCfSlab * kCfTupleWords;  // 96 KiB apart: 16-byte aligned
    const u32 words = total * kCfTupleWords;
    for (u32 k = 4 * t; k < words; k += 4 * kCfThreads1) {
        if (k + 4 <= words) {
            *reinterpret_cast<u32x4c*>(block + k) = *reinterpret_cast<const u32x4c*>(s_buf + k);
        } else {
            for (u32 q = k; q < words; ++q) block[q] = s_buf[q];
        }
    }
    CF_STAMP(0, w, 5)
    __syncthreads();  // (the next slab clears the tables)
    }
}

// ---- long columns of few values (round 5) ---------------------------------------------------------------------------------
// At 32 Mi records the kernel above runs sixteen times per CU, each time a chain of phases behind barriers (load, probe, sort,
// write: 25 us per slab, the memory pipe idle in most of them) and leaves the same hundred tuples per slab for pass 2.  This
// kernel is what a categorical column needs instead: ONE workgroup per CU walks a SHARE of `per_share` consecutive slabs with
// one table that it keeps, two batches of rows per thread in flight while a third is counted, and no barrier until the share
// ends; the table then leaves as the share's tuples, in the block of the share's first slab and in the same sorted form (the
// other slabs' runs are empty), so pass 2 does not know the difference.  A value that finds no slot ends the attempt: the
// workgroup leaves its share "not done" and colfreq_partition_kernel — launched behind this kernel, its workgroups skip
// finished shares — counts those slabs the general way.  A column of distinct values costs every workgroup one batch before
// it gives up.
// The table: 3 072 slots in groups of four (rows of 16 bytes: 5 120).  A slot is a 32-bit key (12 hash bits | the number of a
// representative record within the share, 20 bits; all ones = empty), a count, the representative's length (bit 31: its row
// has arrived) and a copy of its row.  A value's probe sequence starts at its group: ONE 16-byte LDS read fetches the group's
// keys (the next group's too if the first is full of other tags), the first with the record's tag names the slot, row and
// length are compared from LDS.  Anything else — a new value, one that sits further along, a second key with the same tag, a
// row still on its way — goes slot by slot from the group's start; the table takes 2 304 values (three quarters of its slots;
// 3 840 for rows of 16 bytes).  The smallest record of a value ends up as its representative (a 32-bit minimum on the key:
// same tag, smaller record), the tuple's hash bits are recomputed from the row copy when the table leaves.
// Measured at 32 Mi records x 32 bytes: 100 values 0.50 -> 0.24 ms, 1 000 values 0.29, 2 000 0.79 (general passes: 1.9-2.2); the
// kernel's bare stream — no hash, no table — is 0.225 ms with per-lane row loads.  A version that compared against the
// representative's row in the COLUMN instead of LDS, with 16 384 slots: 2 000 values 0.65 ms, 5 000 1.0, 10 000 2.7 — a gather
// per record from the L2, whose working set is a share's representatives x 32 CUs (profiles/r05_colfreq_stream_variants.txt).
// groups of four slots: 768 for rows of 32 bytes, 1 280 for rows of 16 (a slot is 12 bytes + the row: ~132 and ~140 KiB of LDS)
template <u32 STRIDE> struct CfStreamTable { static constexpr u32 kGroups = STRIDE > 16 ? 768u : 1280u; };
static constexpr u32 kCfStreamThreads = 1024;
static constexpr u32 kCfStreamBatch = 2;  // rows per thread and batch (1 and 3: the same and 2 % slower)
static constexpr u32 kCfStreamStep = kCfStreamThreads * kCfStreamBatch;  // records per workgroup and step
static constexpr u32 kCfStreamEmpty = 0xffffffffu;
static constexpr u32 kCfStreamRecBits = 20;
static constexpr u32 kCfStreamMaxShare = (1u << kCfStreamRecBits) - 1u;  // records per share (the empty key is tag and record all ones)
template <u32 STRIDE>
__global__ __launch_bounds__(kCfStreamThreads) void colfreq_stream_kernel(const ColView c, unsigned short* __restrict__ offs,
                                                                        u32* __restrict__ tuples, u32 parts, u32 slabs, u32 per_share,
                                                                        ColFreqStatus* __restrict__ status, u32* __restrict__ ticket,
                                                                        u32* __restrict__ share_done) {
    constexpr u32 kCfStreamGroups = CfStreamTable<STRIDE>::kGroups, kCfStreamSlots = kCfStreamGroups * 4;
    constexpr u32 kRowParts = STRIDE > 16 ? 2u : 1u;  // 16-byte pieces of a row
    __shared__ __attribute__((aligned(16))) u32 s_key[kCfStreamSlots];
    __shared__ u32 s_count[kCfStreamSlots];
    __shared__ u32 s_rlen[kCfStreamSlots];
    __shared__ u32 s_hist[kCfMaxParts];
    __shared__ u32 s_scan[kCfStreamThreads / 64];
    __shared__ u32 s_fill, s_trunc, s_fail;
    // the representatives' rows (bytes 0..15 of slot s in s_rows[s], 16..31 in s_rows[slots + s]); at the end: the tuples
    __shared__ __attribute__((aligned(16))) u32x4c s_rows[kRowParts * kCfStreamSlots];
    static_assert(sizeof(u32x4c) * kRowParts * kCfStreamSlots >= kCfStreamSlots * kCfTupleWords * 4, "the row cache doubles as the tuple staging");
    static_assert(kCfStreamSlots % kCfStreamThreads == 0, "the table leaves as whole slots per thread");
    const u32 t = threadIdx.x, w = blockIdx.x;
    for (u32 k = t; k < kCfStreamSlots; k += kCfStreamThreads) {
        s_key[k] = kCfStreamEmpty;
        s_count[k] = 0;
        s_rlen[k] = 0;
    }
    for (u32 k = t; k < parts; k += kCfStreamThreads) s_hist[k] = 0;
    if (t == 0) {
        s_fill = 0;
        s_trunc = 0;
        s_fail = 0;
        if (w == 0) {  // pass 2 adds to the one and may set the other
            status->n_distinct = 0;
            status->overflow = 0;
            for (u32 k = 0; k < kCfTickets; ++k) ticket[k * 32] = 0;
        }
    }
    __syncthreads();
    const u32 w0 = w * per_share;  // the share's first slab
    const u64 r0 = (u64)w0 * kCfSlab;
    const u64 share_records = (u64)per_share * kCfSlab;
    const u32 nrec = (u32)(c.n_rows - r0 < share_records ? c.n_rows - r0 : share_records);
    const u32* const lens = c.len ? c.len : reinterpret_cast<const u32*>(c.col);  // (no lengths: read one word of the column, use the stride)
    // (loads without branches around them — a record past the share's end is fetched from its last record's address and not
    // looked at — so that the compiler can wait for the oldest batch alone while the younger ones stay in flight)
    struct Rows {
        u32x4c a[kCfStreamBatch], b[kCfStreamBatch];
        u32 len[kCfStreamBatch];
    };
    // Rows of 32 bytes are fetched the way colsearch_small_kernel fetches them: a wave's two loads are 16 bytes per lane over the
    // CONTIGUOUS 2 KiB of its 64 records (a lane fetching its own row asks for 16 bytes every 32: 4.8 instead of 5.2 TB/s), a
    // lane then holds piece `lane` of records 0..31 and piece `lane` of records 32..63 of the group, neighbours swap one of them
    // by DPP, and lane l counts record (l >> 1) + 32 * (l & 1) of the group.
    constexpr bool kSwap = STRIDE == 32;
    const u32 lane = t & 63u;
    const u32 mine_in_wave = kSwap ? (lane >> 1) + ((lane & 1u) << 5) : lane;
    const u32 tt = (t & ~63u) + mine_in_wave;  // this thread's record within a step's run of 1 024
    auto request = [&](Rows& n, u32 base) {
        base = base < nrec ? base : 0u;
#pragma unroll
        for (u32 jj = 0; jj < kCfStreamBatch; ++jj) {
            u32 li = base + jj * kCfStreamThreads + tt;
            li = li < nrec ? li : nrec - 1u;
            const u64 i = r0 + li;
            if (kSwap) {
                const u32x4c* const p = reinterpret_cast<const u32x4c*>(c.col + r0 * STRIDE);
                const u32 last_piece = 2u * nrec - 1u;
                const u32 q0 = 2u * (base + jj * kCfStreamThreads + (t & ~63u)) + lane, q1 = q0 + 64u;
                n.a[jj] = __builtin_nontemporal_load(p + (q0 < last_piece ? q0 : last_piece));
                n.b[jj] = __builtin_nontemporal_load(p + (q1 < last_piece ? q1 : last_piece));
            } else {
                const u32x4c* const p0 = reinterpret_cast<const u32x4c*>(c.col + i * STRIDE);
                n.a[jj] = __builtin_nontemporal_load(p0);
                n.b[jj] = STRIDE > 16 ? __builtin_nontemporal_load(p0 + 1) : u32x4c{0, 0, 0, 0};
            }
            const u32 l = __builtin_nontemporal_load(lens + (c.len ? i : 0));  // (no lengths: every lane reads the same word)
            n.len[jj] = c.len ? l : STRIDE;
        }
    };
    u32 trunc = 0;
    auto count = [&](const Rows& r, const u32 base) {
        u32 home[kCfStreamBatch], mine[kCfStreamBatch];
        u32x4c kq[kCfStreamBatch];
        u32x4c A[kCfStreamBatch], B[kCfStreamBatch];  // this thread's rows
#pragma unroll
        for (u32 jj = 0; jj < kCfStreamBatch; ++jj) {
            A[jj] = r.a[jj];
            B[jj] = r.b[jj];
            if (kSwap) {
                // the half my neighbour needs: an even lane gives away its piece of records 32..63, an odd lane its piece of 0..31
                const bool odd = (lane & 1u) != 0;
                const u32x4c give = odd ? r.a[jj] : r.b[jj];
                u32x4c got;
                got.x = (u32)__builtin_amdgcn_mov_dpp((int)give.x, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
                got.y = (u32)__builtin_amdgcn_mov_dpp((int)give.y, 0xb1, 0xf, 0xf, true);
                got.z = (u32)__builtin_amdgcn_mov_dpp((int)give.z, 0xb1, 0xf, 0xf, true);
                got.w = (u32)__builtin_amdgcn_mov_dpp((int)give.w, 0xb1, 0xf, 0xf, true);
                A[jj] = odd ? got : r.a[jj];
                B[jj] = odd ? r.b[jj] : got;
            }
        }
#pragma unroll
        for (u32 jj = 0; jj < kCfStreamBatch; ++jj) {  // the batch's groups of keys are requested together
            const u32 li = base + jj * kCfStreamThreads + tt;
            const u64 h = hash_regs(STRIDE, r.len[jj], A[jj], B[jj]);
            home[jj] = (u32)(((u64)(u32)h * kCfStreamGroups) >> 32) * 4;
            mine[jj] = ((u32)(h >> 52) << kCfStreamRecBits) | li;
            kq[jj] = *reinterpret_cast<const u32x4c*>(&s_key[home[jj]]);
        }
#pragma unroll
        for (u32 jj = 0; jj < kCfStreamBatch; ++jj) {
            const u32 li = base + jj * kCfStreamThreads + tt;
            if (li >= nrec) continue;
            const u32 len = r.len[jj];
            if (len > STRIDE) ++trunc;
            // the first of the group's keys with this record's tag (an empty key's tag bits are all ones, and so may a record's
            // be: the empty key itself is excluded)
            const u32 tag = mine[jj] >> kCfStreamRecBits;
            u32x4c q = kq[jj];
            u32 g0 = home[jj];
            bool m0 = (q.x >> kCfStreamRecBits) == tag && q.x != kCfStreamEmpty;
            bool m1 = (q.y >> kCfStreamRecBits) == tag && q.y != kCfStreamEmpty;
            bool m2 = (q.z >> kCfStreamRecBits) == tag && q.z != kCfStreamEmpty;
            bool m3 = (q.w >> kCfStreamRecBits) == tag && q.w != kCfStreamEmpty;
            if (!(m0 || m1 || m2 || m3) && q.x != kCfStreamEmpty && q.y != kCfStreamEmpty && q.z != kCfStreamEmpty && q.w != kCfStreamEmpty) {
                // no key with this tag in a group that is FULL: the value, if the table has it, sits further along — nearly always
                // in the next group (with 1 000 values in the table 2 % of them sit past their group, and a wave in which one
                // lane walks slot by slot waits for it: 72 % of the waves did; 1 000 values 0.377 -> 0.302 ms, 1 400: 0.54 ->
                // 0.39).  An empty slot in the group means the value is not in the table at all (nothing is ever removed).
                g0 = g0 + 4 < kCfStreamSlots ? g0 + 4 : 0u;
                q = *reinterpret_cast<const u32x4c*>(&s_key[g0]);
                m0 = (q.x >> kCfStreamRecBits) == tag && q.x != kCfStreamEmpty;
                m1 = (q.y >> kCfStreamRecBits) == tag && q.y != kCfStreamEmpty;
                m2 = (q.z >> kCfStreamRecBits) == tag && q.z != kCfStreamEmpty;
                m3 = (q.w >> kCfStreamRecBits) == tag && q.w != kCfStreamEmpty;
            }
            if (m0 || m1 || m2 || m3) {
                const u32 at = g0 + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u);
                const u32 k0 = m0 ? q.x : m1 ? q.y : m2 ? q.z : q.w;
                // (the row is read BEHIND the flag and a wave's LDS operations execute in order: a flag that is up means the row
                // read after it is complete)
                const u32 rl0 = __hip_atomic_load(&s_rlen[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                const u32x4c b0 = s_rows[at], b1 = STRIDE > 16 ? s_rows[kCfStreamSlots + at] : u32x4c{0, 0, 0, 0};
                u32 diff = (rl0 ^ (len | 0x80000000u)) | (A[jj].x ^ b0.x) | (A[jj].y ^ b0.y) | (A[jj].z ^ b0.z) | (A[jj].w ^ b0.w);
                if (STRIDE > 16) diff |= (B[jj].x ^ b1.x) | (B[jj].y ^ b1.y) | (B[jj].z ^ b1.z) | (B[jj].w ^ b1.w);
                if (diff == 0) {
                    atomicAdd(&s_count[at], 1u);
                    if (mine[jj] < k0) atomicMin(&s_key[at], mine[jj]);  // (same tag: the smaller record becomes the representative)
                    continue;
                }
            }
            // slot by slot from the group's start
            // (an empty slot ends the search: the value is new.  New values are taken while the table is less than three quarters
            // full, however long the probe sequence: a share either fits or gives up when its 2 305th value shows up — for a
            // column of many values within its first few thousand records.  With a limit on the probes instead, a share of 2 000
            // values gave up at its END: 2.4 ms, the whole pass wasted.  Measured, ms at 32 Mi records: 1 400 values 0.40, 1 800
            // 0.64, 2 000 0.79, 2 200 0.94, against 1.9-2.0 for the general passes.)
            bool done = false;
            const bool room = __hip_atomic_load(&s_fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kCfStreamSlots * 3 / 4;
            u32 s = home[jj];
            for (u32 p = 0; p < kCfStreamSlots && !done; ++p, s = s + 1 < kCfStreamSlots ? s + 1 : 0) {
                u32 old = s_key[s];
                const u32 rl = __hip_atomic_load(&s_rlen[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == kCfStreamEmpty && !room) break;
                if (old == kCfStreamEmpty) old = atomicCAS(&s_key[s], kCfStreamEmpty, mine[jj]);
                if (old == kCfStreamEmpty) {
                    atomicAdd(&s_fill, 1u);
                    atomicAdd(&s_count[s], 1u);
                    s_rows[s] = A[jj];
                    if (STRIDE > 16) s_rows[kCfStreamSlots + s] = B[jj];
                    __hip_atomic_store(&s_rlen[s], len | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = true;
                } else if ((old >> kCfStreamRecBits) == tag) {
                    bool eq;
                    if (rl >> 31) {
                        asm volatile("" ::: "memory");
                        const u32x4c c0 = s_rows[s], c1 = STRIDE > 16 ? s_rows[kCfStreamSlots + s] : u32x4c{0, 0, 0, 0};
                        eq = (rl & 0x7fffffffu) == len && A[jj].x == c0.x && A[jj].y == c0.y && A[jj].z == c0.z && A[jj].w == c0.w &&
                             (STRIDE <= 16 || (B[jj].x == c1.x && B[jj].y == c1.y && B[jj].z == c1.z && B[jj].w == c1.w));
                    } else {
                        eq = rows_equal(c, r0 + li, r0 + (old & kCfStreamMaxShare), len);  // (the slot's copy is not there yet)
                    }
                    if (eq) {
                        atomicAdd(&s_count[s], 1u);
                        if (mine[jj] < old) atomicMin(&s_key[s], mine[jj]);
                        done = true;
                    }
                }
            }
            if (!done) __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // TWO batches are in flight while a third is counted
    Rows n0, n1;
    request(n0, 0);
    request(n1, kCfStreamStep);
    for (u32 base = 0; base < nrec; base += 2 * kCfStreamStep) {
        if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        {
            const Rows r = n0;
            request(n0, base + 2 * kCfStreamStep);
            count(r, base);
        }
        if (base + kCfStreamStep < nrec) {
            const Rows r = n1;
            request(n1, base + 3 * kCfStreamStep);
            count(r, base + kCfStreamStep);
        }
    }
    if (trunc) atomicAdd(&s_trunc, trunc);
    __syncthreads();
    if (s_fail) {
        if (t == 0) share_done[w] = 0;
        return;
    }
    // ---- the table leaves as the share's tuples, sorted by partition (the counting sort of colfreq_partition_kernel); a slot's
    // hash is that of its row copy -------------------------------------------------------------------------------------------
    constexpr u32 kOwn = kCfStreamSlots / kCfStreamThreads;
    u32 hlo[kOwn], place[kOwn];  // place: partition << 16 | rank within it; all ones = no tuple
#pragma unroll
    for (u32 j = 0; j < kOwn; ++j) {
        const u32 slot = j * kCfStreamThreads + t;
        place[j] = 0xffffffffu;
        hlo[j] = 0;
        if (s_key[slot] != kCfStreamEmpty) {
            const u64 h = hash_regs(STRIDE, s_rlen[slot] & 0x7fffffffu, s_rows[slot], STRIDE > 16 ? s_rows[kCfStreamSlots + slot] : u32x4c{0, 0, 0, 0});
            const u32 part = cf_part(h, parts);
            hlo[j] = (u32)h;
            place[j] = (part << 16) | atomicAdd(&s_hist[part], 1u);
        }
    }
    __syncthreads();
    const u32 per = (parts + kCfStreamThreads - 1) / kCfStreamThreads;  // <= 4
    u32 local = 0;
    for (u32 k = 0; k < per; ++k) {
        const u32 bin = t * per + k;
        if (bin < parts) local += s_hist[bin];
    }
    const u32 incl = wave_incl_scan_u32(local);
    if ((t & 63u) == 63u) s_scan[t >> 6] = incl;
    __syncthreads();
    u32 run = incl - local;
    for (u32 k = 0; k < (t >> 6); ++k) run += s_scan[k];
    u32 total = 0;
    for (u32 k = 0; k < kCfStreamThreads / 64; ++k) total += s_scan[k];
    const u32 share_slabs = slabs - w0 < per_share ? slabs - w0 : per_share;
    for (u32 k = 0; k < per; ++k) {
        const u32 bin = t * per + k;
        if (bin < parts) {
            const u32 cnt = s_hist[bin];
            s_hist[bin] = run;  // the bin's start within the block
            unsigned short* const o = offs + (u64)bin * slabs + w0;
            o[0] = (unsigned short)run;
            for (u32 q = 1; q < share_slabs; ++q) o[q] = 0;  // the share's other slabs: empty runs
            run += cnt;
        }
    }
    // rows `parts` (tuples in the block) and `parts + 1` (records longer than the stride: at most a slab's worth per entry)
    for (u32 q = t; q < share_slabs; q += kCfStreamThreads) {
        offs[(u64)parts * slabs + w0 + q] = (unsigned short)(q == 0 ? total : 0u);
        const u32 before = q * kCfSlab;
        const u32 left = s_trunc > before ? s_trunc - before : 0u;
        offs[(u64)(parts + 1) * slabs + w0 + q] = (unsigned short)(left < kCfSlab ? left : kCfSlab);
    }
    __syncthreads();  // (the rows in s_rows are no longer read; every thread has its slots' partition starts)
    u32* const s_buf = reinterpret_cast<u32*>(s_rows);
#pragma unroll
    for (u32 j = 0; j < kOwn; ++j) {
        if (place[j] != 0xffffffffu) {
            const u32 slot = j * kCfStreamThreads + t;
            u32* const q = s_buf + (s_hist[place[j] >> 16] + (place[j] & 0xffffu)) * kCfTupleWords;
            q[0] = (u32)r0 + (s_key[slot] & kCfStreamMaxShare);
            q[1] = s_count[slot];
            q[2] = hlo[j];
        }
    }
    __syncthreads();
    u32* const block = tuples + (u64)w0 * kCfSlab * kCfTupleWords;
    const u32 words = total * kCfTupleWords;
    for (u32 k = 4 * t; k < words; k += 4 * kCfStreamThreads) {
        if (k + 4 <= words) {
            *reinterpret_cast<u32x4c*>(block + k) = *reinterpret_cast<const u32x4c*>(s_buf + k);
        } else {
            for (u32 q = k; q < words; ++q) block[q] = s_buf[q];
        }
    }
    if (t == 0) share_done[w] = 1;
}

static constexpr u32 kCfGroup = 2 * kCfThreads2;  // blocks whose runs a pass-2 workgroup lines up at a time (two per thread)
struct ColFreqWideEntry {  // == csvsimd_freq_entry
    u64 first_record, begin, end, count;
};
template <bool WIDE>
__global__ __launch_bounds__(kCfThreads2) void colfreq_reduce_kernel(const ColView c, const unsigned short* __restrict__ offs,
                                                                    const u32* __restrict__ tuples, u32 parts, u32 slabs,
                                                                    u64 first_record, ColFreqEntry* __restrict__ out, u64 out_cap,
                                                                    ColFreqStatus* __restrict__ status, const FreqWideOut wide,
                                                                    u32* __restrict__ ticket, u32 groups) {
    __shared__ u64 s_key[kCfCap2];    // low 32 hash bits << 32 | representative record + 1; 0 = empty
    __shared__ u32 s_count[kCfCap2];
    __shared__ u32 s_first[kCfCap2];
    __shared__ unsigned short s_beg[kCfGroup];  // where the partition's run starts in block w0 + k
    __shared__ u32 s_pre[kCfGroup + 1];         // tuples of the partition in blocks w0 .. w0 + k - 1
    __shared__ u32 s_wave[kCfThreads2 / 64];
    __shared__ u32 s_total, s_trsum, s_overflow, s_next;
    __shared__ u64 s_base;
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    // A workgroup's first partition is its index; the ones after that come from a ticket (zeroed by pass 1).  A column of
    // few values leaves most partitions empty (a microsecond each) and a few with seven microseconds of dependent steps:
    // with a fixed p += gridDim.x walk the call waited for the workgroups that drew two of the latter.  The ticket for
    // the next partition is requested before the work on this one and read after it: no trip is added.  There are
    // `groups` counters (workgroup b draws from counter b % groups the partitions congruent to it; gridDim.x is a multiple
    // of `groups`): 256 draws on ONE word took 3 us to serve, and loads return behind an atomic issued before them.
    const u32 grp = blockIdx.x % groups;
    // ---- once per workgroup: how many tuples pass 1 left in all (row `parts` of the table) --------------------------------
    // With few tuples (a column of few values: a handful per block) most partitions are empty and the rest hold a few
    // hundred tuples each; `span` neighbouring partitions — neighbours in every block's sorted run as well — are then
    // merged as one, so that every workgroup has exactly one to do instead of two or more one after the other.
    // Workgroup 0 adds up the blocks' counts of records longer than the stride on the same trip.
    u32 span = 1;
    if (parts > gridDim.x || blockIdx.x == 0) {
        if (t == 0) { s_total = 0; s_trsum = 0; }
        __syncthreads();
        u32 mine = 0, tr = 0;
        for (u32 w = t; w < slabs; w += kCfThreads2) {
            mine += offs[(u64)parts * slabs + w];
            if (blockIdx.x == 0) tr += offs[(u64)(parts + 1) * slabs + w];
        }
        mine = wave_sum_u32(mine);
        tr = wave_sum_u32(tr);
        if (lane == 0 && mine) atomicAdd(&s_total, mine);
        if (lane == 0 && tr) atomicAdd(&s_trsum, tr);
        __syncthreads();
        if (parts % gridDim.x == 0 && (u64)s_total <= (u64)gridDim.x * (kCfRound2 / 2)) span = parts / gridDim.x;
        if (blockIdx.x == 0 && t == 0) {
            status->n_records = c.n_rows;
            status->truncated = s_trsum;
        }
        __syncthreads();  // (s_total is used again below)
    }
    const u32 nparts = parts / span;
    const bool draws = nparts > gridDim.x;
    const bool one_group = slabs <= kCfGroup;
    u32 p = blockIdx.x;
    while (p < nparts) {
        // the partition's run in block w: [offs[pa][w], offs[pb][w]) (a block's run for partition q ends where its run for
        // q + 1 starts; the last partition's ends at the block's tuple count — row `parts` of the table)
        const u32 pa = p * span, pb = pa + span;
        if (p == blockIdx.x) { CF_STAMP(1, p, 0) }
        // line the runs of a group of kCfGroup blocks up: s_pre = exclusive prefix of their lengths (two blocks per thread, a
        // wave scan, sixteen wave totals), so that tuple k of the group is found by a search in LDS and EVERY thread has a
        // tuple to work on — a run is one tuple when the column has few values, sixteen when all differ.  Returns the
        // group's tuple count.
        auto line_up = [&](const u32 w0, const u32 g) -> u32 {
            u32 len2[2];
#pragma unroll
            for (u32 j = 0; j < 2; ++j) {
                const u32 k = 2 * t + j;
                len2[j] = 0;
                if (k < g) {
                    const u32 b = offs[(u64)pa * slabs + w0 + k];
                    len2[j] = (u32)offs[(u64)pb * slabs + w0 + k] - b;
                    s_beg[k] = (unsigned short)b;
                }
            }
            const u32 incl = wave_incl_scan_u32(len2[0] + len2[1]);
            if (lane == 63) s_wave[wv] = incl;
            __syncthreads();  // (also: a cleared table is complete, the previous group's s_pre / s_beg are no longer read)
            u32 before = 0;
            for (u32 k = 0; k < wv; ++k) before += s_wave[k];
            const u32 excl = before + incl - (len2[0] + len2[1]);
            if (2 * t < g) s_pre[2 * t] = excl;
            if (2 * t + 1 < g) s_pre[2 * t + 1] = excl + len2[0];
            if (t == kCfThreads2 - 1) s_pre[g] = before + incl;  // (threads past g hold zeros: the last thread's inclusive sum is the total)
            __syncthreads();
            return s_pre[g];
        };
        u32 total;
        if (one_group) {
            // up to kCfGroup blocks (16.7 M records): the line-up is the count — one trip to the table per partition
            if (t == 0) s_overflow = 0;
            total = line_up(0, slabs);
        } else {
            if (t == 0) { s_total = 0; s_overflow = 0; }
            __syncthreads();
            u32 mine = 0;
            for (u32 w = t; w < slabs; w += kCfThreads2)
                mine += (u32)offs[(u64)pb * slabs + w] - (u32)offs[(u64)pa * slabs + w];
            mine = wave_sum_u32(mine);
            if (lane == 0 && mine) atomicAdd(&s_total, mine);
            __syncthreads();
            total = s_total;
        }
        u32 drawn = 0;
        if (draws && t == 0) drawn = atomicAdd(ticket + grp * 32, 1u);  // (behind this partition's first loads)
        if (p == blockIdx.x) { CF_STAMP(1, p, 1) }
        const u32 rounds = total ? (total + kCfRound2 - 1) / kCfRound2 : 0;  // a skewed or huge partition: several passes over its tuples
        // the table is as large as this partition needs: >= 2 slots per tuple, a power of two (a partition of a column
        // of few values holds a few hundred tuples: clearing and scanning 1 024 slots instead of 8 192)
        u32 cap = kCfThreads2;
        while (cap < kCfCap2 && cap < 2 * total) cap <<= 1;
        for (u32 r = 0; r < rounds; ++r) {
            for (u32 k = t; k < cap; k += kCfThreads2) {
                s_key[k] = 0;
                s_count[k] = 0;
                s_first[k] = 0xffffffffu;
            }
            if (one_group) __syncthreads();
            for (u32 w0 = 0; w0 < slabs; w0 += kCfGroup) {
                const u32 g = slabs - w0 < kCfGroup ? slabs - w0 : kCfGroup;
                const u32 tg = one_group ? total : line_up(w0, g);
                if (p == blockIdx.x && w0 == 0 && r == 0) { CF_STAMP(1, p, 2) }
                // (kCfBatch2 tuples per thread can be looked up and requested before the first one is merged.  Measured at 32 Mi
                // distinct records, where a partition's 8 192 tuples take 62 us: 4 or 8 in flight 1.52-1.56 ms against 1.44 with
                // one — the dependent trips of ONE tuple are not what a partition waits for; the product keeps one)
                for (u32 k0 = t; k0 < tg; k0 += kCfBatch2 * kCfThreads2) {
                    u32 rec[kCfBatch2], cnt[kCfBatch2], h32[kCfBatch2];
#pragma unroll
                    for (u32 j = 0; j < kCfBatch2; ++j) {
                        const u32 k = k0 + j * kCfThreads2;
                        rec[j] = cnt[j] = h32[j] = 0;  // (cnt == 0: no tuple)
                        if (k < tg) {
                            u32 lo = 0, hi = g;  // the block whose run holds tuple k: the last one with s_pre <= k
                            while (hi - lo > 1) {
                                const u32 mid = (lo + hi) >> 1;
                                if (s_pre[mid] <= k) lo = mid; else hi = mid;
                            }
                            const u32* const q = tuples + ((u64)(w0 + lo) * kCfSlab + s_beg[lo] + (k - s_pre[lo])) * kCfTupleWords;
                            rec[j] = q[0], cnt[j] = q[1], h32[j] = q[2];
                        }
                    }
#pragma unroll
                    for (u32 j = 0; j < kCfBatch2; ++j) {
                        if (cnt[j] == 0) continue;
                        if (rounds > 1 && ((h32[j] >> 13) % rounds) != r) continue;
                        const u64 key = ((u64)h32[j] << 32) | ((u64)rec[j] + 1);
                        u32 s = h32[j] & (cap - 1);
                        bool done = false;
                        for (u32 probes = 0; probes < cap && !done; ++probes, s = (s + 1) & (cap - 1)) {
                            u64 old = s_key[s];
                            if (old == 0) old = atomicCAS((unsigned long long*)&s_key[s], 0ull, (unsigned long long)key);
                            // (lengths and rows are read only when hash bits meet: a partition's records lie all over the
                            // column, and a 4-byte read per tuple from a random place was a sector of traffic per distinct value)
                            if (old == 0 || ((old >> 32) == h32[j] && rows_equal_eager(c, rec[j], (u32)old - 1u))) {
                                atomicAdd(&s_count[s], cnt[j]);
                                atomicMin(&s_first[s], rec[j]);
                                done = true;
                            }
                        }
                        if (!done) s_overflow = 1;  // more distinct values with these hash bits than a table holds
                    }
                }
                __syncthreads();
                if (p == blockIdx.x && w0 == 0 && r == 0) { CF_STAMP(1, p, 3) }
            }
            // occupied slots -> entries; ONE reservation per workgroup and round
            u32 used[kCfCap2 / kCfThreads2], wave_total = 0;
#pragma unroll
            for (u32 j = 0; j < kCfCap2 / kCfThreads2; ++j) {
                used[j] = j * kCfThreads2 < cap && s_key[j * kCfThreads2 + t] != 0 ? 1u : 0u;
                wave_total += (u32)__builtin_popcountll(__ballot(used[j] != 0));
            }
            if (lane == 0) s_wave[wv] = wave_total;
            __syncthreads();
            if (t == 0) {
                u32 tot = 0;
                for (u32 k = 0; k < kCfThreads2 / 64; ++k) tot += s_wave[k];
                s_base = tot ? atomicAdd((unsigned long long*)&status->n_distinct, (unsigned long long)tot) : 0ull;
            }
            __syncthreads();
            u64 at = s_base;
            if (p == blockIdx.x && r == 0) { CF_STAMP(1, p, 4) }
            for (u32 k = 0; k < wv; ++k) at += s_wave[k];
#pragma unroll
            for (u32 j = 0; j < kCfCap2 / kCfThreads2; ++j) {
                const u64 m = __ballot(used[j] != 0);
                if (used[j]) {
                    const u64 o = at + (u64)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    const u32 slot = j * kCfThreads2 + t;
                    if (o < out_cap) {
                        if (WIDE) {
                            const u64 row = s_first[slot];
                            u32 lo = 0, hi = wide.n_chunks;  // the last chunk whose row0 <= row
                            while (hi - lo > 1) {
                                const u32 mid = (lo + hi) >> 1;
                                if (wide.map[mid].row0 <= row) lo = mid; else hi = mid;
                            }
                            const u64 d = row - wide.map[lo].row0, key = wide.map[lo].first_key + d * wide.jump + wide.field;
                            const u64 b = wide.index[key] + 1, e = wide.index[key + 1];  // two neighbouring tape entries
                            reinterpret_cast<ColFreqWideEntry*>(out)[o] =
                                ColFreqWideEntry{wide.map[lo].first_record + d, b, e > b ? e : b, (u64)s_count[slot]};
                        } else {
                            out[o] = ColFreqEntry{first_record + s_first[slot], (u64)s_count[slot]};
                        }
                    }
                }
                at += (u64)__builtin_popcountll(m);
            }
            __syncthreads();
        }
        if (p == blockIdx.x) { CF_STAMP(1, p, 5) }
        if (t == 0) {
            if (s_overflow) status->overflow = 1;
            s_next = draws ? gridDim.x + grp + groups * drawn : nparts;
        }
        __syncthreads();
        p = s_next;
    }
    CF_STAMP(1, blockIdx.x, 6)
}

#ifdef CSVSIMD_CF_TRACE
}  // namespace csvsimd
extern "C" int csvsimd_dev_cf_trace(uint64_t* out, int clear) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(csvsimd::g_cf_trace), sizeof(uint64_t) * 2 * 4096 * 8);
    if (e == hipSuccess && clear) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(csvsimd::g_cf_trace));
        if (e == hipSuccess) e = hipMemset(p, 0, sizeof(uint64_t) * 2 * 4096 * 8);
    }
    return (int)e;
}
namespace csvsimd {
#endif
static u32 cgrid_for(u64 items, u32 per_block, u32 cap) {
    u64 blocks = (items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    return (u32)(blocks > cap ? cap : blocks);
}

hipError_t launch_colfreq(const void* d_col, const void* d_len, u64 n_rows, u32 stride, u64 first_record, void* d_scratch,
                          void* d_entries, u64 entries_cap, void* d_status, int n_cus, hipStream_t stream, const FreqWideOut* wide) {
    ColFreqStatus* const status = (ColFreqStatus*)d_status;
    if (n_rows == 0) return hipMemsetAsync(status, 0, sizeof(ColFreqStatus), stream);
    const ColFreqGeom g = colfreq_geom(n_rows);
    const ColView c = {(const uint8_t*)d_col, (const u32*)d_len, n_rows, stride};
    unsigned short* const offs = (unsigned short*)d_scratch;
    u32* const tuples = (u32*)((char*)d_scratch + g.offs_bytes);
    u32* const ticket = (u32*)((char*)d_scratch + g.bytes - kCfTicketBytes - kCfShareFlagBytes);
    u32* const share_flags = (u32*)((char*)d_scratch + g.bytes - kCfShareFlagBytes);
    const u32 cus = (u32)(n_cus > 0 ? n_cus : 256);
    // two rounds of slabs per CU and more (4 Mi records): first the streaming attempt, one workgroup per CU over a share of slabs
    const bool stream_first = (stride == 16 || stride == 32) && g.slabs >= 2 * cus && cus <= kCfMaxShares &&
                              (u64)((g.slabs + cus - 1) / cus) * kCfSlab <= kCfStreamMaxShare;  // (a share's records number 20 bits)
    const u32 per_share = stream_first ? (g.slabs + cus - 1) / cus : 1u;
    if (stream_first) {
        const u32 shares = (g.slabs + per_share - 1) / per_share;
        if (stride == 32)
            hipLaunchKernelGGL(colfreq_stream_kernel<32>, dim3(shares), dim3(kCfStreamThreads), 0, stream, c, offs, tuples, g.parts,
                               g.slabs, per_share, status, ticket, share_flags);
        else
            hipLaunchKernelGGL(colfreq_stream_kernel<16>, dim3(shares), dim3(kCfStreamThreads), 0, stream, c, offs, tuples, g.parts,
                               g.slabs, per_share, status, ticket, share_flags);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
    }
    const u32* const done = stream_first ? share_flags : nullptr;
    const u32 grid1 = stream_first ? cus : g.slabs;
    if (stride <= 32)
        hipLaunchKernelGGL(colfreq_partition_kernel<true>, dim3(grid1), dim3(kCfThreads1), 0, stream, c, offs, tuples, g.parts,
                           g.slabs, status, ticket, done, per_share);
    else
        hipLaunchKernelGGL(colfreq_partition_kernel<false>, dim3(grid1), dim3(kCfThreads1), 0, stream, c, offs, tuples, g.parts,
                           g.slabs, status, ticket, done, per_share);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const u32 cap = (u32)(n_cus > 0 ? n_cus : 256);  // one 140-KiB workgroup per CU
    const u32 grid2 = g.parts < cap ? g.parts : cap;
    const u32 groups = grid2 % kCfTickets == 0 ? kCfTickets : 1u;
    if (wide)
        hipLaunchKernelGGL(colfreq_reduce_kernel<true>, dim3(grid2), dim3(kCfThreads2), 0, stream, c, offs,
                           tuples, g.parts, g.slabs, first_record, (ColFreqEntry*)d_entries, entries_cap, status, *wide, ticket, groups);
    else
        hipLaunchKernelGGL(colfreq_reduce_kernel<false>, dim3(grid2), dim3(kCfThreads2), 0, stream, c, offs,
                           tuples, g.parts, g.slabs, first_record, (ColFreqEntry*)d_entries, entries_cap, status, FreqWideOut{}, ticket, groups);
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

// The needle travels in the kernel's arguments and the result comes back through pinned host memory (round 5): the synchronous
// call used to be a memset, a copy of the needle from pageable memory, the launch, a 16-byte copy back and a wait — ~25 us
// around a 190-us kernel at 1 GiB, and most of the call at 73 MB.  Now it is the launch and the wait.  Every workgroup adds ONE
// 64-bit word to a device word — its matches (36 bits), whether it saw a record longer than the stride (a count of such
// workgroups, 14 bits) and an arrival (14 bits) — so the workgroup whose addition returns "all others have arrived" holds the
// totals in that one return value: no fence, no second trip.  It stores them, with the call's number, as one word in pinned host
// memory and zeroes the device word for the next call.  (A version in which every thread fenced before its workgroup's arrival
// — release at agent scope = a write-back of the L2 — cost 170 us per call.)
struct ColNeedle {
    u64 w[kColMaxNeedle / 8];  // the needle's bytes, little endian, zero padded
};
struct ColSearchOut {
    u64* acc;    // device: the packed word (zero between calls)
    u64* h_pub;  // pinned host memory (device address): seq << 48 | truncated << 47 | matches
    u64 seq;
};
static constexpr u32 kCsHitBits = 36, kCsTruncBits = 14;  // + 14 bits of arrivals (at most 8 192 workgroups)
__device__ __forceinline__ void colsearch_finish(const ColSearchOut o, u32* s_acc, u32 lane, u32 hits, u32 trunc) {
    if (lane == 0 && hits) atomicAdd(&s_acc[0], hits);
    if (trunc) atomicAdd(&s_acc[1], trunc);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 mine = (1ull << (kCsHitBits + kCsTruncBits)) | ((u64)(s_acc[1] ? 1u : 0u) << kCsHitBits) | (u64)s_acc[0];
        const u64 old = atomicAdd((unsigned long long*)o.acc, (unsigned long long)mine);
        if ((old >> (kCsHitBits + kCsTruncBits)) == gridDim.x - 1) {  // the last to arrive: old + mine = everybody's
            const u64 total = old + mine;
            const u64 matches = total & ((1ull << kCsHitBits) - 1), tr = (total >> kCsHitBits) & ((1ull << kCsTruncBits) - 1);
            __hip_atomic_store(o.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.h_pub, (o.seq << 48) | ((u64)(tr ? 1u : 0u) << 47) | matches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(256) void colsearch_kernel(const ColView c, const ColNeedle needle, u32 m, int mode,
                                                        u64* __restrict__ bitmap, const ColSearchOut out) {
    __shared__ u64 s_needle[kColMaxNeedle / 8 + 1];
    __shared__ u32 s_acc[2];  // the workgroup's matches and records longer than the stride
    if (threadIdx.x == 0) {  // (constant indices: the words come straight from the argument segment; indexed by the thread
                             // the compiler would copy the struct to scratch memory first)
#pragma unroll
        for (u32 k = 0; k < kColMaxNeedle / 8; ++k) s_needle[k] = needle.w[k];
        s_needle[kColMaxNeedle / 8] = 0;
        s_acc[0] = s_acc[1] = 0;
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
    colsearch_finish(out, s_acc, lane, hits, trunc);
}

// ---- rows of 16 or 32 bytes (round 5) ----------------------------------------------------------------------------------------
// The kernel above reads a row eight bytes at a time, three dword loads per step and lane: on a 1-GiB column it is bound by
// the number of load instructions, not by HBM (profiles/r05_consumers_1GiB_before.json).  Here a lane fetches its whole row
// with one or two 16-byte loads — the NEXT batch of rows is requested before this one is searched — and searches it in
// registers: `contains` finds the start positions whose first three bytes are the needle's, four positions per dword and a
// handful of instructions (prefix_candidates), and only those — one position in 17 576 on random lower-case text — are
// compared in full, from the row's cache line.  Same results, bit for bit (tests/test_gpu_columnar.py: every mode, needle
// length and alignment against Python's ==, startswith, in).
// Start positions of a row (D dwords + a zero dword) whose first PREFIX bytes are the needle's: bit p of the result = position
// p.  Per dword and prefix byte: one xor with the broadcast byte and one v_lerp_u8 — lerp(y, 0xff, 0) = (y + 255) >> 1 per byte
// has bit 7 set exactly where y != 0 (the classification's trick, stage1_kernels.hip: no carry crosses a byte) — the shifted
// views of the row by v_alignbyte; the three "differs" words are or-ed and inverted in one three-input bit operation, and the
// four flags of a dword are gathered into the position mask by ONE v_dot4 (flags of 0x80 times weights 1, 2, 4, 8 — the odd
// dwords use 16 ... 128 — summed into an accumulator per pair of dwords).  11 VALU per dword for a three-byte prefix; the
// first version of this kernel spent 24 (64-bit SWAR zero-byte tests, a position limit per dword).
template <u32 D, u32 PREFIX>
__device__ __forceinline__ u32 prefix_candidates(const u32 (&d)[D + 1], u32 b0, u32 b1, u32 b2) {
    u32 acc[D / 2];
#pragma unroll
    for (u32 k = 0; k < D; ++k) {
        u32 nz = __builtin_amdgcn_lerp(d[k] ^ b0, 0xffffffffu, 0u);
        if (PREFIX >= 2) nz |= __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(d[k + 1], d[k], 1) ^ b1, 0xffffffffu, 0u);
        if (PREFIX >= 3) nz |= __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(d[k + 1], d[k], 2) ^ b2, 0xffffffffu, 0u);
        const u32 z = ~nz & 0x80808080u;  // bit 7 of every position whose prefix bytes all agree
        acc[k >> 1] = __builtin_amdgcn_udot4(z, (k & 1u) ? 0x80402010u : 0x08040201u, (k & 1u) ? acc[k >> 1] : 0u, false);
    }
    u32 c = 0;
#pragma unroll
    for (u32 j = 0; j < D / 2; ++j) c |= (acc[j] >> 7) << (8 * j);  // (sum of 0x80 * weight = the pair's eight flags << 7)
    return c;
}
template <u32 STRIDE, bool COALESCED>
__global__ __launch_bounds__(256, 2) void colsearch_small_kernel(const ColView c, const ColNeedle needle, u32 m, int mode,
                                                                 u64* __restrict__ bitmap, const ColSearchOut out) {
    static_assert(STRIDE == 16 || STRIDE == 32, "rows that fit one or two 16-byte loads");
    constexpr u32 D = STRIDE / 4;  // dwords per row
    __shared__ u64 s_needle[kColMaxNeedle / 8 + 1];
    __shared__ u32 s_acc[2];  // the workgroup's matches and records longer than the stride
    if (threadIdx.x == 0) {  // (constant indices: the words come straight from the argument segment; indexed by the thread
                             // the compiler would copy the struct to scratch memory first)
#pragma unroll
        for (u32 k = 0; k < kColMaxNeedle / 8; ++k) s_needle[k] = needle.w[k];
        s_needle[kColMaxNeedle / 8] = 0;
        s_acc[0] = s_acc[1] = 0;
    }
    __syncthreads();
    const u64 n_words = (c.n_rows + 63) / 64;
    const u32 lane = threadIdx.x & 63u;
    u32 nd[D], nmask[D];  // the needle's first STRIDE bytes and which of them exist
#pragma unroll
    for (u32 k = 0; k < D; ++k) {
        nd[k] = (u32)(s_needle[k >> 1] >> (32 * (k & 1)));
        nmask[k] = m >= 4 * k + 4 ? ~0u : (m > 4 * k ? (1u << (8 * (m - 4 * k))) - 1u : 0u);
    }
    // the needle's first three bytes, each broadcast to a dword: the filter of `contains`
    const u32 b0 = (nd[0] & 0xffu) * 0x01010101u, b1 = ((nd[0] >> 8) & 0xffu) * 0x01010101u, b2 = ((nd[0] >> 16) & 0xffu) * 0x01010101u;
    u32 hits = 0, trunc = 0;
    const u64 step = ((u64)gridDim.x * blockDim.x) >> 6;
    u64 word = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // The batch in flight: 64 rows.  Two ways to fetch 32-byte rows:
    //   COALESCED  a wave's loads are 16 bytes per lane over CONTIGUOUS kibibytes (two instructions cover the 2 KiB of 64 rows).
    //              A lane then holds two HALVES — piece `lane` of rows 0..31 (row lane / 2, half lane & 1) and piece `lane` of
    //              rows 32..63 — neighbours swap one of them (DPP), the even lanes end up with rows 0..31, the odd lanes with
    //              rows 32..63 (row_of_lane), and the ballot's bits are un-interleaved.  equals / starts-with, which are all
    //              memory: 0.213 ms for 1 GiB = 63 % of 8 TB/s (a lane fetching its own row: 0.230 = 58 %).
    //   otherwise  a lane fetches its own row (two loads 32 bytes apart from its neighbour's: 4.9 TB/s): rows of 16 bytes.  With
    //              the first version's three-byte filter and one batch in flight `contains` was faster this way (0.265 against
    //              0.29 ms: the swap, the selects and the bit shuffle cost more than the better stream gave); with the four-byte
    //              compare and two batches in flight it is 6 % slower (0.302 against 0.286 on one box), so rows of 32 bytes
    //              take the contiguous loads in every mode (profiles/r05_colsearch_variants.txt).
    constexpr bool kSwap = COALESCED && STRIDE == 32;
    const u32 row_of_lane = kSwap ? (lane >> 1) + ((lane & 1u) << 5) : lane;
    // (no branch around a load: rows past the end are fetched from the last row's address and never looked at, a missing lengths
    // array reads ONE word of the column instead — every lane the same: indexed by the row it was 4 bytes per row of extra
    // traffic, 12.5 % on 32-byte rows, seen in the request counters — and the value is replaced; with loads under conditions the compiler can only wait for
    // ALL outstanding loads, `s_waitcnt vmcnt(0)`, and the second batch in flight would be waited for with the first)
    const u64 last_row = c.n_rows - 1;
    const u32* const lens = c.len ? c.len : reinterpret_cast<const u32*>(c.col);
    auto fetch = [&](u64 wd, u32x4c& va, u32x4c& vb, u32& vlen) {
        if (wd >= n_words) wd = n_words - 1;
        const u64 i0 = wd * 64;
        if (kSwap) {
            const u64 last_piece = c.n_rows * 2 - 1;  // 16-byte pieces of the column
            const u32x4c* const p = reinterpret_cast<const u32x4c*>(c.col);
            const u64 q0 = i0 * 2 + lane, q1 = q0 + 64;
            va = __builtin_nontemporal_load(p + (q0 < last_piece ? q0 : last_piece));
            vb = __builtin_nontemporal_load(p + (q1 < last_piece ? q1 : last_piece));
        } else {
            const u64 r = i0 + lane < last_row ? i0 + lane : last_row;
            const u32x4c* const p = reinterpret_cast<const u32x4c*>(c.col + r * STRIDE);
            va = __builtin_nontemporal_load(p);
            if (STRIDE == 32) vb = __builtin_nontemporal_load(p + 1);
        }
        const u64 rl = i0 + row_of_lane < last_row ? i0 + row_of_lane : last_row;
        const u32 l = __builtin_nontemporal_load(lens + (c.len ? rl : 0));  // (no lengths: every lane reads the same word)
        vlen = c.len ? l : STRIDE;
    };
    auto search = [&](const u64 word, const u32x4c ra, const u32x4c rb, const u32 full) {
        const u64 i = word * 64 + row_of_lane;
        u32 d[D + 1];
        if (kSwap) {
            // the half my neighbour needs: an even lane gives away its piece of rows 32..63, an odd lane its piece of rows 0..31
            const bool odd = (lane & 1u) != 0;
            const u32x4c give = odd ? ra : rb;
            u32x4c got;
            got.x = (u32)__builtin_amdgcn_mov_dpp((int)give.x, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
            got.y = (u32)__builtin_amdgcn_mov_dpp((int)give.y, 0xb1, 0xf, 0xf, true);
            got.z = (u32)__builtin_amdgcn_mov_dpp((int)give.z, 0xb1, 0xf, 0xf, true);
            got.w = (u32)__builtin_amdgcn_mov_dpp((int)give.w, 0xb1, 0xf, 0xf, true);
            const u32x4c lo = odd ? got : ra, hi = odd ? rb : got;
            d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w;
            d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
        } else {
            d[0] = ra.x; d[1] = ra.y; d[2] = ra.z; d[3] = ra.w;
            if (STRIDE == 32) { d[4] = rb.x; d[5] = rb.y; d[6] = rb.z; d[7] = rb.w; }
        }
        d[D] = 0;
        bool match = false;
        if (i < c.n_rows) {
            if (full > STRIDE) ++trunc;
            const u32 n = full < STRIDE ? full : STRIDE;
            if (mode != 2) {
                match = m <= STRIDE && (mode == 0 ? n == m : n >= m);
                u32 diff = 0;
#pragma unroll
                for (u32 k = 0; k < D; ++k) diff |= (d[k] ^ nd[k]) & nmask[k];
                match = match && diff == 0;
            } else if (m == 0) {
                match = true;
            } else if (n >= m) {
                const u32 last = n - m;  // last start position
                if (m >= 4) {
                    // Needles of four bytes and more: is there a start position whose first FOUR bytes are the needle's?  The answer
                    // is per row, not per position: the rare row that has one (one position in 456 976 on random lower-case text;
                    // a hit in the zero padding only costs the look) is searched position by position from its cache line.
                    // (First as one v_alignbyte + one 32-bit compare per start position, the compares' lane masks or-ed on the
                    // scalar side — 50 VALU + 28 SALU per row: 0.284 / 0.270 ms for 6 / 12-byte needles where the instruction
                    // below takes 0.255 / 0.246.)
                    // v_mqsad_u32_u8: the sums of absolute differences between the needle's first four bytes and the four-byte
                    // windows at byte offsets 0..3 of an 8-byte source, in ONE instruction — a sum of zero is a window equal to
                    // them (bytes the instruction masks out only add candidates).  Eight of them cover a 32-byte row; the
                    // smallest sum decides.
                    u32 best = ~0u;
#pragma unroll
                    for (u32 k = 0; k < D; ++k) {
                        const u64 src = (u64)d[k] | ((u64)d[k + 1] << 32);
                        const u32x4c r = __builtin_amdgcn_mqsad_u32_u8(src, nd[0], u32x4c{0, 0, 0, 0});
                        best = min(best, min(min(r.x, r.y), min(r.z, r.w)));
                    }
                    const bool candidate = best == 0;
                    if (candidate) {
                        const uint8_t* const row = c.col + i * STRIDE;
                        for (u32 pos = 0; pos <= last && !match; ++pos) {
                            bool ok = true;
                            for (u32 q = 0; 8 * q < m && ok; ++q) {
                                const u32 left = m - 8 * q;
                                const u64 mask = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
                                ok = ((col_load8(row, pos + 8 * q, STRIDE) ^ s_needle[q]) & mask) == 0;
                            }
                            match = ok;
                        }
                    }
                } else {
                // start positions whose first m (<= 3) bytes are the needle's (prefix_candidates: exact), limited to the positions
                // a match may start at: 0 .. last: needles of up to three bytes are decided by that alone.
                const u32 cand = (m == 1   ? prefix_candidates<D, 1>(d, b0, b1, b2)
                                  : m == 2 ? prefix_candidates<D, 2>(d, b0, b1, b2)
                                           : prefix_candidates<D, 3>(d, b0, b1, b2)) &
                                 (last >= 31u ? ~0u : ((2u << last) - 1u));
                match = cand != 0;
                }
            }
        }
        u64 bits = __ballot(match);
        if (kSwap) {
            // bit 2k = row k, bit 2k + 1 = row 32 + k: the even bits, packed, are the word's low half, the odd bits its high half
            auto pack_even = [](u64 x) -> u64 {
                x &= 0x5555555555555555ull;
                x = (x | (x >> 1)) & 0x3333333333333333ull;
                x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
                x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
                x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
                x = (x | (x >> 16)) & 0x00000000ffffffffull;
                return x;
            };
            bits = pack_even(bits) | (pack_even(bits >> 1) << 32);
        }
        if (lane == 0) {
            bitmap[word] = bits;
            hits += (u32)__builtin_popcountll(bits);
        }
    };
    // TWO batches of rows are in flight per wave while a third is searched: with one, a wave's 2 KiB were requested a
    // search ahead of their use and the column streamed at the rate of (resident waves x 2 KiB) per memory latency.
    constexpr u32 kDepth = 2;
    u32x4c va[kDepth], vb[kDepth];
    u32 vlen[kDepth];
#pragma unroll
    for (u32 j = 0; j < kDepth; ++j) {
        va[j] = vb[j] = u32x4c{0, 0, 0, 0};
        vlen[j] = STRIDE;
        fetch(word + j * step, va[j], vb[j], vlen[j]);
    }
    for (; word < n_words; word += kDepth * step) {
#pragma unroll
        for (u32 j = 0; j < kDepth; ++j) {
            if (word + j * step < n_words) {
                const u32x4c ra = va[j], rb = vb[j];
                const u32 full = vlen[j];
                fetch(word + (kDepth + j) * step, va[j], vb[j], vlen[j]);
                search(word + j * step, ra, rb, full);
            }
        }
    }
    colsearch_finish(out, s_acc, lane, hits, trunc);
}

// ---- rows of 32 bytes without the swap ---------------------------------------------------------------------------------------
// The same contiguous fetch (a lane holds piece `lane` of the batch's rows 0..31 — half lane & 1 of row lane >> 1 — and piece
// `lane` of rows 32..63), but every lane searches the HALVES it fetched: equals / starts-with compare a half with the needle's
// half (a row matches if both of its lanes say so: the ballot's neighbouring bits and-ed), `contains` with needles of up to
// three bytes looks for the needle STARTING in the lane's half — a first half borrows the first dword of its neighbour's second
// half for the windows that cross over (one DPP move), a second half is followed by the row's end — and a row matches if either
// lane found it.  (Longer needles — a filter on four bytes, then the rare exact search — were slower this way, 0.273 against
// 0.246 ms, and take colsearch_small_kernel<32, true>.)  The swap's eight DPP moves and eight selects per row are gone; the arithmetic of these kernels ADDS to their time (a wave asks for
// its next rows when it has finished with the previous ones).
__global__ __launch_bounds__(256, 2) void colsearch32_kernel(const ColView c, const ColNeedle needle, u32 m, int mode,
                                                             u64* __restrict__ bitmap, const ColSearchOut out) {
    constexpr u32 STRIDE = 32;
    __shared__ u64 s_needle[kColMaxNeedle / 8 + 1];
    __shared__ u32 s_acc[2];  // the workgroup's matches and records longer than the stride
    if (threadIdx.x == 0) {  // (constant indices: the words come straight from the argument segment; indexed by the thread
                             // the compiler would copy the struct to scratch memory first)
#pragma unroll
        for (u32 k = 0; k < kColMaxNeedle / 8; ++k) s_needle[k] = needle.w[k];
        s_needle[kColMaxNeedle / 8] = 0;
        s_acc[0] = s_acc[1] = 0;
    }
    __syncthreads();
    const u64 n_words = (c.n_rows + 63) / 64;
    const u32 lane = threadIdx.x & 63u;
    const u32 half = lane & 1u;
    // this lane's half of the needle's first 32 bytes and which of its bytes exist; the needle's first four bytes and first
    // three bytes broadcast (the filters of `contains`)
    u32 nd[4], nmask[4];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 kk = 4 * half + k;
        nd[k] = (u32)(s_needle[kk >> 1] >> (32 * (kk & 1)));
        nmask[k] = m >= 4 * kk + 4 ? ~0u : (m > 4 * kk ? (1u << (8 * (m - 4 * kk))) - 1u : 0u);
    }
    const u32 n4 = (u32)s_needle[0];  // (needles of up to three bytes: the launch sends longer ones of `contains` elsewhere)
    const u32 b0 = (n4 & 0xffu) * 0x01010101u, b1 = ((n4 >> 8) & 0xffu) * 0x01010101u, b2 = ((n4 >> 16) & 0xffu) * 0x01010101u;
    u32 hits = 0, trunc = 0;
    const u64 step = ((u64)gridDim.x * blockDim.x) >> 6;
    u64 word = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 last_row = c.n_rows - 1, last_piece = c.n_rows * 2 - 1;
    const u32* const lens = c.len ? c.len : reinterpret_cast<const u32*>(c.col);
    struct Batch {
        u32x4c a, b;  // piece `lane` of rows 0..31 and of rows 32..63
        u32 la, lb;   // those rows' lengths
    };
    auto fetch = [&](u64 wd, Batch& v) {  // (no branch around a load: see colsearch_small_kernel)
        if (wd >= n_words) wd = n_words - 1;
        const u64 i0 = wd * 64;
        const u32x4c* const p = reinterpret_cast<const u32x4c*>(c.col);
        const u64 q0 = i0 * 2 + lane, q1 = q0 + 64;
        v.a = __builtin_nontemporal_load(p + (q0 < last_piece ? q0 : last_piece));
        v.b = __builtin_nontemporal_load(p + (q1 < last_piece ? q1 : last_piece));
        const u64 ra = i0 + (lane >> 1), rb = ra + 32;
        const u32 la = __builtin_nontemporal_load(lens + (c.len ? (ra < last_row ? ra : last_row) : 0));
        const u32 lb = __builtin_nontemporal_load(lens + (c.len ? (rb < last_row ? rb : last_row) : 0));
        v.la = c.len ? la : STRIDE;
        v.lb = c.len ? lb : STRIDE;
    };
    // one half: does it (mode 0 / 1) equal its half of the needle, or (mode 2, needles of up to three bytes) hold the start of
    // the needle?
    auto look = [&](const u32x4c v, const u32 nx, const u32 full, const bool valid) -> bool {
        if (!valid) return false;
        const u32 n = full < STRIDE ? full : STRIDE;
        if (mode != 2) {
            const u32 diff = ((v.x ^ nd[0]) & nmask[0]) | ((v.y ^ nd[1]) & nmask[1]) | ((v.z ^ nd[2]) & nmask[2]) | ((v.w ^ nd[3]) & nmask[3]);
            return m <= STRIDE && (mode == 0 ? n == m : n >= m) && diff == 0;
        }
        if (m == 0) return true;
        if (n < m) return false;
        const u32 last = n - m;  // last start position of the row
        // the dword behind this half: the neighbour's first (a first half), nothing (a second half: the row ends)
        const u32 d[5] = {v.x, v.y, v.z, v.w, half ? 0u : nx};
        if (last < 16u * half) return false;  // no window may start in this half
        const u32 lim = last - 16u * half;  // start positions 0 .. lim of this half
        const u32 cand = (m == 1 ? prefix_candidates<4, 1>(d, b0, b1, b2) : m == 2 ? prefix_candidates<4, 2>(d, b0, b1, b2)
                                                                                   : prefix_candidates<4, 3>(d, b0, b1, b2)) &
                         (lim >= 15u ? 0xffffu : ((2u << lim) - 1u));
        return cand != 0;
    };
    auto pack_even = [](u64 x) -> u64 {  // the even bits of x, packed
        x &= 0x5555555555555555ull;
        x = (x | (x >> 1)) & 0x3333333333333333ull;
        x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
        x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
        x = (x | (x >> 16)) & 0x00000000ffffffffull;
        return x;
    };
    const bool both = mode != 2;  // a row matches if BOTH halves do (equals / starts-with) or if EITHER does (`contains`)
    auto search = [&](const u64 wd, const Batch& v) {
        const u64 ia = wd * 64 + (lane >> 1), ib = ia + 32;
        // (the neighbour's first dword, fetched while every lane is still active)
        const u32 nxa = (u32)__builtin_amdgcn_mov_dpp((int)v.a.x, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
        const u32 nxb = (u32)__builtin_amdgcn_mov_dpp((int)v.b.x, 0xb1, 0xf, 0xf, true);
        const bool fa = look(v.a, nxa, v.la, ia < c.n_rows), fb = look(v.b, nxb, v.lb, ib < c.n_rows);
        if (half == 0) trunc += (ia < c.n_rows && v.la > STRIDE ? 1u : 0u) + (ib < c.n_rows && v.lb > STRIDE ? 1u : 0u);
        u64 xa = __ballot(fa), xb = __ballot(fb);  // bits 2r and 2r + 1: the halves of row r (of row 32 + r)
        xa = both ? xa & (xa >> 1) : xa | (xa >> 1);
        xb = both ? xb & (xb >> 1) : xb | (xb >> 1);
        const u64 bits = pack_even(xa) | (pack_even(xb) << 32);
        if (lane == 0) {
            bitmap[wd] = bits;
            hits += (u32)__builtin_popcountll(bits);
        }
    };
    constexpr u32 kDepth = 2;  // batches in flight per wave
    Batch v[kDepth];
#pragma unroll
    for (u32 j = 0; j < kDepth; ++j) fetch(word + j * step, v[j]);
    for (; word < n_words; word += kDepth * step) {
#pragma unroll
        for (u32 j = 0; j < kDepth; ++j) {
            if (word + j * step < n_words) {
                const Batch cur = v[j];
                fetch(word + (kDepth + j) * step, v[j]);
                search(word + j * step, cur);
            }
        }
    }
    colsearch_finish(out, s_acc, lane, hits, trunc);
}

hipError_t launch_colsearch(const void* d_col, const void* d_len, u64 n_rows, u32 stride, const void* needle_host,
                            u32 needle_len, int mode, void* d_bitmap, void* d_acc, void* h_pub_dev, u64 seq, hipStream_t stream) {
    if (n_rows == 0 || needle_len > kColMaxNeedle) return n_rows == 0 ? hipSuccess : hipErrorInvalidValue;
    const ColView c = {(const uint8_t*)d_col, (const u32*)d_len, n_rows, stride};
    ColNeedle nd;
    memset(&nd, 0, sizeof nd);
    if (needle_len) memcpy(&nd, needle_host, needle_len);  // (little endian host: byte k of the needle = byte k of the words)
    const ColSearchOut out = {(u64*)d_acc, (u64*)h_pub_dev, seq & 0xffffu};
    // rows of 16 / 32 bytes and a needle that fits the row: the register-resident search.  A persistent-sized grid (4
    // workgroups per CU) whose waves walk the column with two batches of loads in flight each.
    const bool small = (stride == 16 || stride == 32) && needle_len <= stride;
    // workgroups per CU x 256 CUs.  Measured on one box, 1 GiB of 32-byte rows, equals / contains in ms, (batches in flight,
    // workgroups per CU): (1, 8) 0.236 / 0.297, (1, 6) 0.222 / 0.283, (2, 8) 0.233 / 0.294, (2, 6) 0.224 / 0.285, (2, 4) 0.215 / 0.286,
    // (2, 3) 0.230 / 0.340, (2, 2) 0.276 / 0.380, (3, 4) 0.229 / 0.309, (4, 4) 0.220 / 0.295 — fewer, longer streams until the
    // arithmetic of `contains` runs out of waves (profiles/r05_colsearch_variants.txt)
    constexpr u32 kSmallGrid = 256 * 4;
    // rows of 32 bytes: halves searched where they were fetched — except `contains` with needles of four bytes and more, which is
    // faster on whole rows (0.257 / 0.244 against 0.277 / 0.265 ms for 6 / 12 bytes; equals 0.215 -> 0.194-0.204, one-byte
    // `contains` 0.240 -> 0.221: profiles/r05_colsearch_variants.txt)
    if (small && stride == 32 && !(mode == 2 && needle_len >= 4))
        hipLaunchKernelGGL(colsearch32_kernel, dim3(cgrid_for(n_rows, 256, kSmallGrid)), dim3(256), 0, stream, c,
                           nd, needle_len, mode, (u64*)d_bitmap, out);
    else if (small && stride == 32)  // (`contains` too is 6 % faster with the contiguous loads once two batches are in flight)
        hipLaunchKernelGGL((colsearch_small_kernel<32, true>), dim3(cgrid_for(n_rows, 256, kSmallGrid)), dim3(256), 0, stream, c,
                           nd, needle_len, mode, (u64*)d_bitmap, out);
    else if (small)
        hipLaunchKernelGGL((colsearch_small_kernel<16, false>), dim3(cgrid_for(n_rows, 256, kSmallGrid)), dim3(256), 0, stream, c,
                           nd, needle_len, mode, (u64*)d_bitmap, out);
    else
    hipLaunchKernelGGL(colsearch_kernel, dim3(cgrid_for(n_rows, 256, 8192)), dim3(256), 0, stream, c,
                       nd, needle_len, mode, (u64*)d_bitmap, out);
    return hipGetLastError();
}

}  // namespace csvsimd
