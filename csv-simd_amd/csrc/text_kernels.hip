// text_kernels.hip — gfx950 kernels for the dialect-coverage row of SURVEY.md §8f (rank 4) that sit
// next to stage 1 rather than in it: UTF-8 validation of the raw bytes and space / quote trimming of
// field spans.  None of this is executed by the reference: its UTF-8 checker is dead code
// (src/avx/utf8check.rs, never called from reader::read) and "trim" is a todo in its class-table
// legend (src/stage1.rs:41-48, class 4 = 0x20).  Both stay OFF the default stage-1 path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_kernels.h"

namespace csvsimd {

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// UTF-8 validation (RFC 3629: shortest form, no surrogates, <= U+10FFFF, no truncated tail).
//
// result[0] = offset of the first byte that does not start / continue a well-formed sequence
// (the position Python's bytes.decode reports as UnicodeDecodeError.start), UINT64_MAX if valid.
//
// Pass 1 (utf8_scan_kernel, the streaming pass): branch-free byte-parallel rules that only look
// BACKWARDS, so a wave needs just the dword before its block as halo and carries no state:
//   * continuation expected  (byte i-1 >= C0, or i-2 >= E0, or i-3 >= F0)  XOR  byte i is 10xxxxxx
//   * byte i is C0, C1 or F5..FF
//   * byte i-1 is E0 / ED / F0 / F4 and byte i is outside that lead's narrowed second-byte range
// Every violation lies within 3 bytes after the start of the first ill-formed sequence and there
// are no false positives, so the minimum flagged offset p satisfies  start <= p <= start + 3.
// Shape: 16 bytes per lane, 4 KiB per wave per iteration in flight; a block that is pure ASCII and
// does not follow a pending lead (the normal CSV case) costs 5 VALU per 16 bytes and streams at the
// HBM read rate; any other block runs the rules as SWAR on dwords (flags live in bit 7 of each byte,
// byte-shifted views via v_alignbyte), ~45 VALU per 4 bytes — or the first rule alone when a 1-KiB chunk
// holds none of the bytes the others are about (see utf8_basic below).
// Pass 2 (utf8_refine_kernel, one thread): backs up from p to a sequence start (<= 7 bytes) and
// decodes forward sequentially to the exact offset.
// ---------------------------------------------------------------------------------------------
struct Utf8Range {
    const uint8_t* abase;  // 128-byte aligned
    u64 lo, hi;            // valid bytes are abase[lo, hi)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// flag words: only bit 7 of every byte is meaningful
struct Utf8Flags {
    u32 ge_c0, ge_e0, ge_f0;      // lead of length >= 2 / >= 3 / >= 4 (or an invalid F8..FF)
    u32 is_e0, is_ed, is_f0, is_f4;
};

__device__ __forceinline__ u32 prev_bytes(u32 cur, u32 prev, int n) {
    // byte i of the result = byte i - n of the stream (prev holds the four bytes before cur)
    return __builtin_amdgcn_alignbyte(cur, prev, 4 - n);
}

// The part of the rules that text without "special" bytes needs: continuation expected XOR continuation found, plus the
// two bytes that are never valid as 2-byte leads (C0, C1).  The other rules only ever fire on or right behind a byte from
// {E0, ED, F0..FF} (overlong 3- and 4-byte forms, surrogates, > U+10FFFF, F8+).  A 1-KiB wave chunk without any — Latin,
// Greek, Cyrillic, Arabic, Hebrew, CJK text — is checked with the basic rule alone: ~28 VALU per dword instead of ~55; one with
// E0 / ED leads but no byte >= F0 (Devanagari, Thai, Hangul) adds those two leads' second-byte test (~5 more); only a chunk with
// F0..FF bytes (emoji, or garbage) runs everything.  What is computed to find that out is used by the full rules as well.
struct Utf8Basic {
    u32 ge_c0, ge_e0, ge_f0, cont;  // bit 7 of every byte
    u32 c0c1;  // bit 7 of the LOWEST byte that is C0 or C1 is exact (bytes above one may be flagged too: same dword, the
               // minimum offset wins)
    u32 is_e0, is_ed;  // 11100000, 11101101: with ge_f0, the bytes that call for the narrowed rules
};
__device__ __forceinline__ Utf8Basic utf8_basic(u32 x) {
    const u32 s1 = x << 1, s2 = x << 2, s3 = x << 3;
    Utf8Basic f;
    f.ge_c0 = x & s1;
    f.ge_e0 = f.ge_c0 & s2;
    f.ge_f0 = f.ge_e0 & s3;
    f.cont = x & ~s1;
    const u32 c = (x & 0xFEFEFEFEu) ^ 0xC0C0C0C0u;
    f.c0c1 = (c - 0x01010101u) & ~c;
    const u32 s4 = x << 4, s5 = x << 5, s6 = x << 6, s7 = x << 7;
    const u32 e_lead = f.ge_e0 & ~s3;
    f.is_e0 = (e_lead & ~s4) & (~s5 & ~s6 & ~s7);
    f.is_ed = (e_lead & s4) & (s5 & ~s6 & s7);
    return f;
}
// middle tier: E0 / ED leads but no byte >= F0 in the chunk (Thai, Devanagari, Hangul text): the basic rule plus those two
// leads' narrowed second byte (E0 -> A0..BF: bit 5 set; ED -> 80..9F: bit 5 clear)
__device__ __forceinline__ u32 utf8_e0ed_errors(u32 x, const Utf8Basic& f, const Utf8Basic& pf) {
    const u32 s2 = x << 2;
    return ((prev_bytes(f.is_e0, pf.is_e0, 1) & ~s2) | (prev_bytes(f.is_ed, pf.is_ed, 1) & s2)) & 0x80808080u;
}
__device__ __forceinline__ u32 utf8_basic_errors(const Utf8Basic& f, const Utf8Basic& pf) {
    const u32 must = prev_bytes(f.ge_c0, pf.ge_c0, 1) | prev_bytes(f.ge_e0, pf.ge_e0, 2) | prev_bytes(f.ge_f0, pf.ge_f0, 3);
    return ((must ^ f.cont) | f.c0c1) & 0x80808080u;
}

// all rules on dword x (its basic flags in b); returns its error flags given the flags of the dword before it
__device__ __forceinline__ u32 utf8_swar(u32 x, const Utf8Basic& b, Utf8Flags& pf) {
    const u32 s2 = x << 2, s3 = x << 3, s4 = x << 4, s5 = x << 5, s6 = x << 6, s7 = x << 7;
    Utf8Flags f;
    f.ge_c0 = b.ge_c0;                // 11xxxxxx
    f.ge_e0 = b.ge_e0;                // 111xxxxx
    f.ge_f0 = b.ge_f0;                // 1111xxxx
    const u32 ge_f8 = f.ge_f0 & s4;   // 11111xxx: never valid
    const u32 f4xx = f.ge_f0 & ~s4 & s5;                         // 111101xx
    const u32 f5_7 = f4xx & (s6 | s7);                           // F5, F6, F7: > U+10FFFF
    const u32 low3_zero = ~s5 & ~s6 & ~s7;
    f.is_e0 = b.is_e0;                                           // 11100000
    f.is_ed = b.is_ed;                                           // 11101101
    f.is_f0 = f.ge_f0 & ~s4 & low3_zero;                         // 11110000
    f.is_f4 = f4xx & ~s6 & ~s7;                                  // 11110100
    const u32 must = prev_bytes(f.ge_c0, pf.ge_c0, 1) | prev_bytes(f.ge_e0, pf.ge_e0, 2) |
                     prev_bytes(f.ge_f0, pf.ge_f0, 3);
    // narrowed second byte: E0 -> A0..BF (bit 5), ED -> 80..9F, F0 -> 90..BF (bit 5 or 4), F4 -> 80..8F
    const u32 second = (prev_bytes(f.is_e0, pf.is_e0, 1) & ~s2) | (prev_bytes(f.is_ed, pf.is_ed, 1) & s2) |
                       (prev_bytes(f.is_f0, pf.is_f0, 1) & ~s2 & ~s3) | (prev_bytes(f.is_f4, pf.is_f4, 1) & (s2 | s3));
    pf = f;
    return ((must ^ b.cont) | ge_f8 | b.c0c1 | f5_7 | second) & 0x80808080u;
}

// the aligned dword at abase[q, q + 4) with bytes outside [lo, hi) zeroed
__device__ __forceinline__ u32 masked_dword(const Utf8Range& r, u64 q) {
    u32 w = *reinterpret_cast<const u32*>(r.abase + q);
    if (q < r.lo || q + 4 > r.hi) {
        u32 keep = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q + (u64)k >= r.lo && q + (u64)k < r.hi) keep |= 0xffu << (8 * k);
        w &= keep;
    }
    return w;
}

// n_chunks counts one chunk past the end of the buffer (it reads as zeros): that is where a lead in
// the very last bytes finds its missing continuation
__global__ __launch_bounds__(256) void utf8_scan_kernel(Utf8Range r, u64 n_chunks, u64 n_real_chunks, u64* result) {
    const u32 lane = threadIdx.x & 63u;
    // wave-uniform by construction; saying so lets the halo load below be a scalar load
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 n_blocks = (n_chunks + 255) / 256;  // 4 KiB = 256 chunks of 16 bytes per wave iteration
    u64 first = ~0ull;
    for (u64 blk = wave; blk < n_blocks; blk += n_waves) {
        u32x4 v[4];
        u32 hib[4];
        // buffer loads with the hardware range check: chunks past the end of the buffer read as zeros,
        // no branch around the loads (the descriptor is rebuilt per 4-KiB block: its size field is 32 bit)
        const u64 blk0 = blk * 4096;
        const u64 avail = n_real_chunks * 16 > blk0 ? n_real_chunks * 16 - blk0 : 0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(r.abase) + blk0, 0, (int)(avail < 4096 ? avail : 4096), 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((u32)j * 1024u + lane * 16u), 0, 2 /* nt */);
        // the four bytes before the block (wave-uniform address: a scalar load)
        const u32 halo = blk ? masked_dword(r, blk * 4096 - 4) : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // only the first and the last chunk of the buffer can hold bytes outside [lo, hi): zero them
            const u64 off = (blk * 256 + (u64)j * 64 + lane) * 16;
            if (off < r.lo || off + 16 > r.hi) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    u32 keep = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u64 i = off + (u64)(4 * d + k);
                        if (i >= r.lo && i < r.hi) keep |= 0xffu << (8 * k);
                    }
                    v[j][d] &= keep;
                }
            }
            hib[j] = (v[j][0] | v[j][1] | v[j][2] | v[j][3]) & 0x80808080u;
        }
        // pure ASCII and nothing pending from the bytes before the block
        if (__ballot(((hib[0] | hib[1] | hib[2] | hib[3]) | (halo & 0x80808080u)) != 0u) == 0ull) continue;

#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the dword before this lane's chunk: the previous lane's last dword
            u32 w0 = (u32)__shfl_up((int)v[j][3], 1);
            const u32 edge = j == 0 ? halo : (u32)__builtin_amdgcn_readlane((int)v[j > 0 ? j - 1 : 0][3], 63);
            if (lane == 0) w0 = edge;
            u32 e[4];
            Utf8Basic bf[4];
            const Utf8Basic b0 = utf8_basic(w0);
            u32 sp_f = b0.ge_f0, sp_e = b0.is_e0 | b0.is_ed;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf[d] = utf8_basic(v[j][d]);
                sp_f |= bf[d].ge_f0;
                sp_e |= bf[d].is_e0 | bf[d].is_ed;
            }
            if (__ballot((sp_f & 0x80808080u) != 0u) == 0ull) {
                // no byte >= F0 in this 1-KiB chunk (nor in the dword before any lane's 16 bytes): the basic rule ...
                e[0] = utf8_basic_errors(bf[0], b0);
#pragma unroll
                for (int d = 1; d < 4; ++d) e[d] = utf8_basic_errors(bf[d], bf[d - 1]);
                if (__ballot((sp_e & 0x80808080u) != 0u) != 0ull) {  // ... and, with E0 / ED leads about, their second byte
                    e[0] |= utf8_e0ed_errors(v[j][0], bf[0], b0);
#pragma unroll
                    for (int d = 1; d < 4; ++d) e[d] |= utf8_e0ed_errors(v[j][d], bf[d], bf[d - 1]);
                }
            } else {
                Utf8Flags pf = {0, 0, 0, 0, 0, 0, 0};
                (void)utf8_swar(w0, b0, pf);  // only its flags matter
#pragma unroll
                for (int d = 0; d < 4; ++d) e[d] = utf8_swar(v[j][d], bf[d], pf);
            }
            if ((e[0] | e[1] | e[2] | e[3]) != 0u) {
                const u32 d = e[0] ? 0u : e[1] ? 1u : e[2] ? 2u : 3u;
                const u32 ed = e[0] ? e[0] : e[1] ? e[1] : e[2] ? e[2] : e[3];
                const u64 i = (blk * 256 + (u64)j * 64 + lane) * 16 + 4 * d + ((u32)__builtin_ctz(ed) >> 3);
                first = i < first ? i : first;
            }
        }
    }
    // wave minimum, one atomic per wave that found anything
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(first >> 32), d) << 32) | (u32)__shfl_xor((int)(u32)first, d);
        first = o < first ? o : first;
    }
    if (lane == 0 && first != ~0ull) atomicMin((unsigned long long*)result, (unsigned long long)(first - r.lo));
}

// result[0] = a flagged offset within [start, start + 3] of the first ill-formed sequence (or
// UINT64_MAX): back up to a sequence start and decode forward to the exact offset.  One thread.
__global__ void utf8_refine_kernel(const uint8_t* buf, u64 len, u64* result) {
    const u64 p = result[0];
    if (p == ~0ull) return;
    u64 s = p > 3 ? p - 3 : 0;
    if (s > len) s = len;
    // everything before the error is valid: at most 3 continuation bytes of a sequence, plus the stray
    // continuation byte itself when that is the error
    for (int k = 0; k < 4 && s > 0 && s < len && (buf[s] & 0xC0) == 0x80; ++k) --s;
    u64 i = s;
    while (i < len) {
        const u32 b = buf[i];
        if (b < 0x80u) { ++i; continue; }
        u32 need, lo1 = 0x80u, hi1 = 0xBFu;
        if (b >= 0xC2u && b <= 0xDFu) need = 1;
        else if (b >= 0xE0u && b <= 0xEFu) { need = 2; if (b == 0xE0u) lo1 = 0xA0u; if (b == 0xEDu) hi1 = 0x9Fu; }
        else if (b >= 0xF0u && b <= 0xF4u) { need = 3; if (b == 0xF0u) lo1 = 0x90u; if (b == 0xF4u) hi1 = 0x8Fu; }
        else break;
        bool ok = true;
        for (u32 k = 1; k <= need && ok; ++k) {
            if (i + k >= len) { ok = false; break; }
            const u32 c = buf[i + k];
            ok = c >= lo1 && c <= hi1;
            lo1 = 0x80u;
            hi1 = 0xBFu;
        }
        if (!ok) break;
        i += need + 1;
        if (i > p + 4) break;  // cannot happen: pass 1 flagged something at or before p
    }
    result[0] = i < len ? i : (p < len ? p : len - 1);
}

__global__ void utf8_init_kernel(u64* result) {
    result[0] = ~0ull;
    result[1] = 0;
}

hipError_t launch_utf8_validate(const void* dbuf, u64 len, void* d_result, int n_cus, hipStream_t stream) {
    hipLaunchKernelGGL(utf8_init_kernel, dim3(1), dim3(1), 0, stream, (u64*)d_result);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || len == 0) return e;
    const uintptr_t addr = (uintptr_t)dbuf;
    Utf8Range r;
    r.abase = (const uint8_t*)(addr & ~(uintptr_t)127);  // wave loads on whole 128-byte lines
    r.lo = (u64)(addr & 127);
    r.hi = r.lo + len;
    const u64 n_real = (r.hi + 15) / 16;
    const u64 n_chunks = n_real + 1;
    const u64 n_blocks = (n_chunks + 255) / 256;          // wave iterations
    u64 grid = (n_blocks + 3) / 4;                        // 4 waves per workgroup
    const u64 cap = (u64)(n_cus > 0 ? n_cus : 256) * 8;   // 32 waves per CU: enough loads in flight to stream
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(utf8_scan_kernel, dim3((u32)grid), dim3(256), 0, stream, r, n_chunks, n_real, (u64*)d_result);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(utf8_refine_kernel, dim3(1), dim3(1), 0, stream, (const uint8_t*)dbuf, len, (u64*)d_result);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// span trimming: [begin, end) -> without leading / trailing 0x20 (the reference's class 4,
// src/stage1.rs:41-48: `todo: trim " xx "`), then optionally without one enclosing quote pair.
// ---------------------------------------------------------------------------------------------
__global__ void trim_spans_kernel(const uint8_t* __restrict__ bytes, u64* __restrict__ begin, u64* __restrict__ end,
                                  u64 n, u32 flags, u32 quote) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 b = begin[i], e = end[i];
        if (e < b) e = b;  // the empty last field of a CRLF row
        if (flags & CSVSIMD_TRIM_SPACE) {
            while (b < e && bytes[b] == 0x20) ++b;
            while (e > b && bytes[e - 1] == 0x20) --e;
        }
        if ((flags & CSVSIMD_TRIM_QUOTES) && e - b >= 2 && bytes[b] == quote && bytes[e - 1] == quote) {
            ++b;
            --e;
        }
        begin[i] = b;
        end[i] = e;
    }
}

hipError_t launch_trim_spans(const void* dbytes, void* d_begin, void* d_end, u64 n, u32 flags, u32 quote,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    u64 blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(trim_spans_kernel, dim3((u32)blocks), dim3(256), 0, stream, (const uint8_t*)dbytes,
                       (u64*)d_begin, (u64*)d_end, n, flags, quote);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ingest: the way back of a chunk's tape (csvsimd_stage1_index).  The chunk's entries (u64, absolute) sit in device
// memory; what crosses PCIe is HALF of that: 32-bit offsets relative to the chunk (a chunk is <= 32 MiB), written by
// this kernel straight into the pinned host slot (device-visible host memory: posted writes, no copy engine — measured,
// a D2H copy of the tape is served one after the other with the H2D copies of the following chunks on the pool's
// hosts, +10 ms on a 38 ms call).  The entry count is read from the chunk's result record ON THE DEVICE, so the kernel
// is enqueued right behind the stage-1 launch without the host knowing the count; the host expands the offsets into
// the caller's tape when it has read the record.
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2t __attribute__((ext_vector_type(2)));
// The kernel is also the chunk's PUBLISHER (round 5): its last workgroup to finish copies the chunk's 64-byte result record
// into pinned host memory and then stores the chunk's sequence number behind it, which the host polls — no copy-out, no
// event.  Order: every thread's stores to the slot, system-scope fence, workgroup barrier, one arrival per workgroup on a
// device counter (atomicInc wraps it back to 0 for the next chunk); the last arrival writes record, fence, sequence word.
// Posted writes of one device reach host memory in order, and the fences keep them in order on the way to the link.
__global__ __launch_bounds__(256) void narrow_tape_kernel(const u64* __restrict__ tape, const csvsimd_shard_result* __restrict__ res,
                                                          u64 cap, u64 base, u32* __restrict__ out, u64* __restrict__ h_rec,
                                                          u64 seq, u32* __restrict__ arrivals) {
    const u64 count = res->count;
    const u64 n = (out && count < cap) ? count : (out ? cap : 0);
    const u64 n4 = n / 4;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (u64)gridDim.x * blockDim.x) {
        const u64x2t a = __builtin_nontemporal_load(reinterpret_cast<const u64x2t*>(tape) + 2 * j);
        const u64x2t b = __builtin_nontemporal_load(reinterpret_cast<const u64x2t*>(tape) + 2 * j + 1);
        const u32x4t v = {(u32)(a.x - base), (u32)(a.y - base), (u32)(b.x - base), (u32)(b.y - base)};
        reinterpret_cast<u32x4t*>(out)[j] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[4 * n4 + threadIdx.x] = (u32)(tape[4 * n4 + threadIdx.x] - base);
    if (!h_rec) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 old = atomicInc(arrivals, gridDim.x - 1);  // ((old >= gridDim.x - 1) ? 0 : old + 1)
        if (old == gridDim.x - 1) {
            __threadfence_system();
            const u64* r = reinterpret_cast<const u64*>(res);
#pragma unroll
            for (int i = 0; i < 8; ++i) h_rec[i] = r[i];
            __threadfence_system();
            __hip_atomic_store(h_rec + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Which instantiation for data nobody has seen yet?  Sixteen 64-KiB windows spread over the buffer, one workgroup each:
// bytes equal to the delimiter, CR or LF (quotes ignored: an upper bound of the entries, close enough to put the data on
// one side of 0.125 entries per byte).  1 MiB read: a few microseconds, once per context (capi.cpp: the synchronous device
// entry point of a context that knows nothing about its data yet).
// ---------------------------------------------------------------------------------------------
constexpr u32 kSampleWindows = 16, kSampleBytes = 64u << 10;
__global__ __launch_bounds__(256) void density_sample_kernel(const uint8_t* __restrict__ buf, u64 len, u32 delim4, unsigned long long* __restrict__ out) {
    const u64 span = len > kSampleBytes ? len - kSampleBytes : 0;
    const u64 start = ((span / (kSampleWindows - 1)) * blockIdx.x) & ~(u64)3;   // (dword loads: any 4-byte aligned start)
    const uintptr_t mis = (uintptr_t)buf & 3u;
    const uint8_t* const base = buf + start + (mis ? 4 - mis : 0);
    const u64 avail = (buf + len) - base;
    const u32 words = (u32)((avail < kSampleBytes ? avail : kSampleBytes) / 4);
    u32 cnt = 0;
    for (u32 i = threadIdx.x; i < words; i += blockDim.x) {
        const u32 x = reinterpret_cast<const u32*>(base)[i];
        // zero-byte test of x ^ pattern for the three bytes (exact per byte: (y - 0x01..) & ~y & 0x80.. over-reports only
        // above a true zero byte, which a count used as a threshold can live with)
        const u32 a = x ^ delim4, b = x ^ 0x0a0a0a0au, c = x ^ 0x0d0d0d0du;
        const u32 z = (((a - 0x01010101u) & ~a) | ((b - 0x01010101u) & ~b) | ((c - 0x01010101u) & ~c)) & 0x80808080u;
        cnt += (u32)__builtin_popcount(z);
    }
    for (int d = 32; d > 0; d >>= 1) cnt += (u32)__shfl_xor((int)cnt, d);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
    if (threadIdx.x == 0) atomicAdd(out + 1, (unsigned long long)words * 4ull);
}

hipError_t launch_density_sample(const void* dbuf, u64 len, u32 delimiter, void* d_out2, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_out2, 0, 16, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(density_sample_kernel, dim3(kSampleWindows), dim3(256), 0, stream, (const uint8_t*)dbuf, len,
                       delimiter * 0x01010101u, (unsigned long long*)d_out2);
    return hipGetLastError();
}

hipError_t launch_narrow_tape(const void* d_tape, const void* d_result, u64 cap, u64 base, void* d_out, int workgroups,
                              hipStream_t stream, void* h_rec_dev, u64 seq, void* d_arrivals) {
    // 16-byte aligned tape and slot (hipMalloc / hipHostMalloc); the caller sizes the grid to the bytes it expects
    // (capi.cpp: the writes cross PCIe against the H2D copies' read requests, and are better trickled than dumped).
    // d_out == nullptr: nothing to pack (a count-only call): the launch only publishes the record.
    static_assert(sizeof(csvsimd_shard_result) == 64, "the publisher copies eight 64-bit words");
    hipLaunchKernelGGL(narrow_tape_kernel, dim3((u32)(workgroups > 0 ? workgroups : 1)), dim3(256), 0, stream, (const u64*)d_tape,
                       (const csvsimd_shard_result*)d_result, cap, base, (u32*)d_out, (u64*)h_rec_dev, seq, (u32*)d_arrivals);
    return hipGetLastError();
}

}  // namespace csvsimd
