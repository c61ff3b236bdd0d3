// stage1_kernels.hip — gfx950 (MI355X / CDNA4) kernels for CSV stage 1: bytes -> tape.
//
// What it computes (bit-exact with reference reader::read, src/reader.rs:150-306):
//   tape entry for byte i  <=>  byte in {',', CR, LF}  (class & 3,  src/avx/stage1.rs:394)
//                               and the inclusive prefix-xor of '"' bits at i is 0
//                               (src/avx/stage1.rs:342-407), emitted ascending as u64
//                               (src/stage1.rs:162-296).
//
// How (MI355X-first; nothing here mirrors the SSE code's structure) — DESIGN.md §4 / §7 have the numbers:
//   * one pass over the input: every byte is read once from HBM by LDS-DMA
//     (buffer_load_dwordx4 ... lds, non-temporal, 1 KiB per wave instruction, fully coalesced, no
//     VGPRs); the buffer descriptor's range check makes the ragged last tile branch-free.
//   * measured on gfx950: every 32-bit VALU instruction costs 4 cycles per wave64 per SIMD, so the
//     kernel is bound by VALU ISSUE, not HBM, unless the per-byte instruction count is tiny.
//     Hence: (a) classification = one v_perm 8-entry LUT keyed by 3 hashed bits + xor +
//     v_lerp_u8 (carry-free per-byte add) + two v_bitop3 + two v_dot4 bit gathers = 10 VALU per
//     4 bytes for BOTH masks; (b) the transpose that gives each lane one 64-byte stripe (= one
//     64-bit structural mask + one 64-bit quote mask) is done by the LDS image itself: swizzle
//     on the DMA source side, conflict-free ds_read_b128 on the reader side, zero VALU.
//   * in-string mask = 6 shift-xor steps on the lane's 64-bit word (CDNA has no carry-less
//     multiply) + ONE ballot/mbcnt per 4 KiB for the carry across lanes + a scalar carry across
//     rounds.  Masks of the whole 32-KiB wave span stay in registers (2 x u64 per round).
//   * the two loop-carried quantities of the reference (inside_str, array_idx:
//     src/reader.rs:217-218) become a composable tile descriptor
//     (quote parity, count if entered outside, count if entered inside) resolved across
//     workgroups by a single-pass decoupled look-back over one 64-bit word per tile
//     (relaxed agent-scope atomics: the data is the flag, no fences; status tag in both halves).
//     The resolve of a tile is lagged by one tile (its masks are held in registers meanwhile).
//   * ordered compaction: one DPP wave scan of per-lane popcounts per 4 KiB, set bits scattered as
//     u16 offsets into a wave-private LDS window (rounds batched), flushed as fully coalesced
//     non-temporal 16-byte stores.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "stage1_kernels.h"

// This file is compiled TWICE (Makefile): as itself — every kernel and launcher, the 8-round geometry (256-KiB tiles) —
// and through stage1_dense.hip with CSVSIMD_DENSE_TU: only stage1_kernel<..., DENSE> at 2 rounds per wave (64-KiB tiles)
// and its launcher, in namespace csvsimd_dense.  Write-heavy streams run faster the smaller the unit a workgroup draws
// (the delimiter-dense corpus writes 1.6 bytes per byte read), the sparse corpora slower: the geometry is
// a per-instantiation choice, made per launch by the data's density.
#ifdef CSVSIMD_DENSE_TU
#define CSVSIMD_KERNEL_NS csvsimd_dense
namespace csvsimd_dense {
using namespace csvsimd;
#else
#define CSVSIMD_KERNEL_NS csvsimd
namespace csvsimd {
#endif

// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32;
typedef uint64_t u64;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

static constexpr int kWaves = CSVSIMD_COMPUTE_WAVES;    // waves per workgroup (one 32-KiB span each)
static constexpr int kThreads = kWaves * 64;
static constexpr int kRows = 4;                         // 1-KiB rows (dwordx4 wave loads) per round
static constexpr int kRoundBytes = kRows * 1024;        // 4 KiB per wave per round = 64 stripes of 64 B
static constexpr int kRounds = CSVSIMD_ROUNDS;          // rounds per wave per tile
static constexpr int kSpanBytes = kRounds * kRoundBytes;  // 32 KiB contiguous per wave
static constexpr int kTileBytes = kWaves * kSpanBytes;    // 256 KiB per workgroup tile
#ifndef CSVSIMD_COMP_CAP
#define CSVSIMD_COMP_CAP 2048
#endif
static constexpr int kCompCap = CSVSIMD_COMP_CAP;
#ifndef CSVSIMD_STORE_ALIGN
#define CSVSIMD_STORE_ALIGN 128
#endif
static constexpr int kStoreAlignEntries = CSVSIMD_STORE_ALIGN / 8;  // tape entries per aligned store unit
// every input byte is read exactly once and every tape byte written exactly once: non-temporal
// on both sides (measured on the same traffic mix: +11 % over default-policy loads and stores)
static constexpr int kLoadAux = 2;                      // buffer-load cache policy bits: nt

static_assert(kTileBytes == CSVSIMD_TILE_BYTES, "tile geometry must match the host header");

// descriptor word = two 32-bit halves, EACH tagged with the 2-bit status and the launch epoch:
//   lo = status << 30 | epoch << 20 | x[19:0]     hi = status << 30 | epoch << 20 | x[39:20]
//   aggregate  (status 1): x = P | A << 1 | B << 20          (A, B <= kTileBytes < 2^19)
//   inclusive  (status 2): x = state | running_count << 1    (count < 2^39)
// A word is valid only if both halves carry the same status AND the epoch of the running launch.
// Measured on MI355X: a 64-bit sc1 load that races with the aggregate -> inclusive rewrite of the
// same word can return one half of each (observed as "status = aggregate, payload = an inclusive
// count", ~1 in 10^5 polls once the look-back polls words while they are being rewritten); the
// duplicated tag turns that into a harmless "not published yet".
// The epoch is what replaces the per-launch zeroing of the descriptor array (round 1: a separate
// zero_kernel, 3.9 us + a kernel boundary): words left behind by earlier launches carry another
// epoch and read as "not published".  The epoch lives in device memory (control block) and is
// advanced by the last workgroup of every launch, so a captured hipGraph replays correctly; when
// it wraps (every 1024 launches) that workgroup zeroes the words used since the last wrap.
static constexpr u32 kStatusAgg = 1u;
static constexpr u32 kStatusInc = 2u;
static constexpr int kEpochBits = 10;
static constexpr int kHalfBits = 30 - kEpochBits;  // payload bits per 32-bit half
static constexpr u32 kHalfMask = (1u << kHalfBits) - 1u;
static constexpr u32 kEpochMask = (1u << kEpochBits) - 1u;
static constexpr int kAggShiftB = kHalfBits;  // B sits in the high half
static_assert(kSpanBytes <= 65536, "the compaction window holds span-relative offsets as u16");
static_assert(kTileBytes < (1 << (kHalfBits - 1)), "tile aggregates A and B must fit their descriptor fields");
static_assert(kWaves % 4 == 0 && kRows == 4, "the dispatcher reserves ceil(waves / 4) slots on every SIMD; rows are dwordx4 wave loads");
static constexpr uint32_t kSpinLimit = 1u << 20;  // bounded spins: a protocol bug must end the kernel, not hang the GPU


// ---------------------------------------------------------------------------------------------
// wavefront primitives (wave64)
// ---------------------------------------------------------------------------------------------
// number of set bits of `mask` in lanes below this one
__device__ __forceinline__ u32 mbcnt64(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_or_zero(u32 v) {
    // lanes whose source is out of range / row-masked read 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// inclusive +scan over the 64 lanes; fields packed inside v must not carry into each other
__device__ __forceinline__ u32 wave_incl_scan_add(u32 v) {
    v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
    return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan_add(v), 63);
}

// ---------------------------------------------------------------------------------------------
// byte classification: 4 bytes -> 0x80 flags per byte for "structural" and for "quote".
// Equal to the reference's class table (src/stage1.rs:23-48): structural = class & 3
// ({0x2c, 0x0a, 0x0d}), quote = class & 16 ({0x22}); every other byte incl. >= 0x80 is 0.
//
//   key  = (b ^ (b >> 3)) & 7         0x0a->3  0x0d->4  0x22->6  0x2c->1   (all distinct)
//   e    = LUT[key]                   the only special byte that has this key (v_perm_b32)
//   nz   = (b ^ e) != 0               v_lerp_u8(y, 0xff, 0) = (y + 255) >> 1: bit 7 <=> y >= 1,
//                                     a per-byte add that cannot carry into the next byte
//   special bytes with bit 3 set are structural, the one with bit 3 clear (0x22) is the quote.
// Unused LUT slots hold 0x2c, whose own key (1) differs from theirs, so they can never match.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void classify_dword(u32 x, u32 w, u32& acc_s, u32& acc_q) {
    const u32 key = ((x >> 3) ^ x) & 0x07070707u;
    const u32 e = __builtin_amdgcn_perm(0x2c222c0du, 0x0a2c2c2cu, key);
    const u32 nz = __builtin_amdgcn_lerp(x ^ e, 0xffffffffu, 0u);
    const u32 xs = x << 4;  // bit 3 of every byte -> bit 7
    const u32 fs = ~nz & xs & 0x80808080u;
    const u32 fq = ~nz & ~xs & 0x80808080u;
    // gather the four bit-7 flags: sum(0x80 * weight) — weights are 1 << i, no carries
    acc_s = __builtin_amdgcn_udot4(fs, w, acc_s, false);
    acc_q = __builtin_amdgcn_udot4(fq, w, acc_q, false);
}

// 16 bytes -> 16 structural bits + 16 quote bits (bit i = byte i)
__device__ __forceinline__ void classify16(uint4 v, u32& st16, u32& q16) {
    u32 s_lo = 0, q_lo = 0, s_hi = 0, q_hi = 0;
    classify_dword(v.x, 0x08040201u, s_lo, q_lo);
    classify_dword(v.y, 0x80402010u, s_lo, q_lo);
    classify_dword(v.z, 0x08040201u, s_hi, q_hi);
    classify_dword(v.w, 0x80402010u, s_hi, q_hi);
    // acc = 128 * (8-bit mask)
    st16 = (s_lo >> 7) | (s_hi << 1);
    q16 = (q_lo >> 7) | (q_hi << 1);
}

// ---------------------------------------------------------------------------------------------
// Dialect extension (NOT reference behaviour: the reference hard-wires ','
// and '"', src/avx/stage1.rs:392-394, and lists escapes as a TODO, README.md:32).  Any delimiter /
// quote / escape byte: no hash of arbitrary bytes into the 8 LUT slots is guaranteed to be
// collision free, so these variants compare against each special byte directly (xor + v_lerp_u8
// zero test per special): 13 VALU per dword without, 16 with an escape byte, instead of 10.
//   DIALECT 0 = the reference dialect above (the only one the headline numbers are quoted on)
//   DIALECT 1 = runtime delimiter and quote byte (quote_mask 0 switches quoting off)
//   DIALECT 2 = DIALECT 1 + an escape byte: the byte after an unescaped escape byte is literal
// ---------------------------------------------------------------------------------------------
struct DialectRegs {
    u32 delim, quote, esc;  // the byte replicated into all four lanes of a dword
    u32 qmask;              // 0x80808080, or 0 when the dialect has no quote byte
    // DIALECT 3 (hashed): key = ((b >> sh1) ^ (b >> sh2)) & 7 is distinct for the dialect's special bytes (found by the
    // host); LUT slot `key` holds the only special byte with that key / its class (0x80 structural, 0x40 quote, 0x20 escape)
    u32 sh1, sh2, lut_lo, lut_hi, cls_lo, cls_hi;
};

// DIALECT 3: the default kernel's trick for ANY delimiter / quote / escape byte whose 3-bit two-shift hash is collision
// free (96 % of all triples; the host searches the two shifts, csvsimd::dialect_hash, and falls back to DIALECT 2's
// direct compares otherwise): one v_perm LUT for "the only special byte with this key", ONE zero test instead of five,
// a second v_perm for that byte's class.  15 VALU per dword for three masks (direct compares: 17).
__device__ __forceinline__ void classify_dword_h(u32 x, u32 w, const DialectRegs& dr, u32& acc_s, u32& acc_q, u32& acc_e) {
    const u32 key = ((x >> dr.sh1) ^ (x >> dr.sh2)) & 0x07070707u;
    const u32 e = __builtin_amdgcn_perm(dr.lut_hi, dr.lut_lo, key);
    const u32 nz = __builtin_amdgcn_lerp(x ^ e, 0xffffffffu, 0u);  // bit 7 of a byte: it is NOT its slot's special byte
    const u32 cls = __builtin_amdgcn_perm(dr.cls_hi, dr.cls_lo, key);
    const u32 fs = ~nz & cls & 0x80808080u;
    const u32 fq = ~nz & (cls << 1) & 0x80808080u;
    const u32 fe = ~nz & (cls << 2) & 0x80808080u;
    acc_s = __builtin_amdgcn_udot4(fs, w, acc_s, false);
    acc_q = __builtin_amdgcn_udot4(fq, w, acc_q, false);
    acc_e = __builtin_amdgcn_udot4(fe, w, acc_e, false);
}

template <int DIALECT>
__device__ __forceinline__ void classify_dword_d(u32 x, u32 w, const DialectRegs& dr, u32& acc_s, u32& acc_q,
                                                 u32& acc_e) {
    const u32 nzd = __builtin_amdgcn_lerp(x ^ dr.delim, 0xffffffffu, 0u);
    const u32 nzc = __builtin_amdgcn_lerp(x ^ 0x0d0d0d0du, 0xffffffffu, 0u);
    const u32 nzl = __builtin_amdgcn_lerp(x ^ 0x0a0a0a0au, 0xffffffffu, 0u);
    const u32 nzq = __builtin_amdgcn_lerp(x ^ dr.quote, 0xffffffffu, 0u);
    const u32 fs = ~(nzd & nzc & nzl) & 0x80808080u;
    const u32 fq = ~nzq & dr.qmask;
    acc_s = __builtin_amdgcn_udot4(fs, w, acc_s, false);
    acc_q = __builtin_amdgcn_udot4(fq, w, acc_q, false);
    if (DIALECT == 2) {
        const u32 nze = __builtin_amdgcn_lerp(x ^ dr.esc, 0xffffffffu, 0u);
        acc_e = __builtin_amdgcn_udot4(~nze & 0x80808080u, w, acc_e, false);
    }
}

template <int DIALECT>
__device__ __forceinline__ void classify16_d(uint4 v, const DialectRegs& dr, u32& st16, u32& q16, u32& e16) {
    u32 s_lo = 0, q_lo = 0, e_lo = 0, s_hi = 0, q_hi = 0, e_hi = 0;
    if (DIALECT == 3) {
        classify_dword_h(v.x, 0x08040201u, dr, s_lo, q_lo, e_lo);
        classify_dword_h(v.y, 0x80402010u, dr, s_lo, q_lo, e_lo);
        classify_dword_h(v.z, 0x08040201u, dr, s_hi, q_hi, e_hi);
        classify_dword_h(v.w, 0x80402010u, dr, s_hi, q_hi, e_hi);
    } else {
        classify_dword_d<DIALECT>(v.x, 0x08040201u, dr, s_lo, q_lo, e_lo);
        classify_dword_d<DIALECT>(v.y, 0x80402010u, dr, s_lo, q_lo, e_lo);
        classify_dword_d<DIALECT>(v.z, 0x08040201u, dr, s_hi, q_hi, e_hi);
        classify_dword_d<DIALECT>(v.w, 0x80402010u, dr, s_hi, q_hi, e_hi);
    }
    st16 = (s_lo >> 7) | (s_hi << 1);
    q16 = (q_lo >> 7) | (q_hi << 1);
    e16 = (e_lo >> 7) | (e_hi << 1);
}

// Escaped-byte mask of one 64-byte stripe from its escape-byte mask `bs` and `in` (= byte 0 is
// escaped).  Same carry-propagating-add idea simdjson uses for JSON backslashes: a run of escape
// bytes escapes the byte after it iff its length is odd; runs are separated by parity of their
// start position and resolved with one 64-bit add.
__device__ __forceinline__ u64 escaped_mask(u64 bs, u32 in) {
    constexpr u64 kEven = 0x5555555555555555ull;
    const u64 b1 = bs & ~(u64)in;
    const u64 follows = (b1 << 1) | (u64)in;
    const u64 odd_starts = b1 & ~kEven & ~follows;
    const u64 inv = (odd_starts + b1) << 1;
    return (kEven ^ inv) & follows;
}

// Parity of the run of escape bytes that ends right before abase[pos] (valid bytes are
// abase[lo, hi)); a run that reaches the shard start continues into esc_in.  Whole wave, uniform.
__device__ __forceinline__ u32 escape_run_parity(const uint8_t* abase, u64 lo, u64 hi, u64 pos, u32 esc_byte,
                                                 u32 esc_in, u32 lane) {
    u32 total = 0;
    for (;;) {
        // lane k looks at byte pos - 1 - k
        const bool inside = pos >= lo + 1 + lane && pos - 1 - lane < hi;
        bool is = false;
        if (inside) is = abase[pos - 1 - lane] == (uint8_t)esc_byte;
        const u64 nb = ~__ballot(is);
        const u32 run = nb ? (u32)__builtin_ctzll(nb) : 64u;
        total += run;
        if (run < 64u) {
            const bool reached_lo = pos < lo + 1 + run;  // the run was ended by the shard start
            return (total & 1u) ^ (reached_lo ? esc_in : 0u);
        }
        pos -= 64;
    }
}

// ---------------------------------------------------------------------------------------------
// descriptor algebra: f = (P, A, B): parity flip, count if entered outside, count if inside
// compose(e, l) = "e happens first, then l"
// ---------------------------------------------------------------------------------------------
struct Desc {
    u32 p, a, b;
};
__device__ __forceinline__ Desc compose(Desc e, Desc l) {
    Desc r;
    r.p = e.p ^ l.p;
    r.a = e.a + (e.p ? l.b : l.a);
    r.b = e.b + (e.p ? l.a : l.b);
    return r;
}

// lanes hold f_k for sequence position (-k) (lane 0 = latest); lanes >= m are ignored.
// Returns in lane 0 the composition earliest..latest over lanes [0, m).
__device__ __forceinline__ Desc wave_compose_ordered(Desc f, u32 lane, u32 m) {
    // opaque: the six `lane + d >= 64` predicates below are loop invariant, and hoisted out of the tile loop they hold
    // twelve SGPRs for the whole kernel — of a uniform state that already exceeds what a wave has
    asm volatile("" : "+v"(lane));
    if (lane >= m) { f.p = 0; f.a = 0; f.b = 0; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Desc g;
        g.p = (u32)__shfl_down((int)f.p, d);
        g.a = (u32)__shfl_down((int)f.a, d);
        g.b = (u32)__shfl_down((int)f.b, d);
        if (lane + d >= 64) { g.p = 0; g.a = 0; g.b = 0; }
        f = compose(g, f);  // g covers earlier positions
    }
    return f;
}

// ---------------------------------------------------------------------------------------------
// single-pass look-back (wave 0 of the workgroup).  Returns entering state and tape base of
// `tile`, publishes this tile's aggregate and inclusive words.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 encode_desc(u32 status, u32 epoch, u64 x) {
    const u32 tag = (status << 30) | (epoch << kHalfBits);
    const u32 lo = tag | ((u32)x & kHalfMask);
    const u32 hi = tag | ((u32)(x >> kHalfBits) & kHalfMask);
    return ((u64)hi << 32) | lo;
}
// status (0 = not published, torn, or left behind by an earlier launch) and 40-bit payload
__device__ __forceinline__ u32 decode_desc(u64 d, u32 epoch, u64& x) {
    const u32 lo = (u32)d, hi = (u32)(d >> 32);
    x = (u64)(lo & kHalfMask) | ((u64)(hi & kHalfMask) << kHalfBits);
    const u32 tag_lo = lo >> kHalfBits, tag_hi = hi >> kHalfBits;  // status : epoch
    return (tag_lo == tag_hi && (tag_lo & kEpochMask) == epoch) ? (lo >> 30) : 0u;
}
__device__ __forceinline__ void store_desc(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 load_desc(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void publish_aggregate(u64* desc, u32 tile, u32 epoch, Desc agg) {
    store_desc(desc + tile,
               encode_desc(kStatusAgg, epoch, (u64)agg.p | ((u64)agg.a << 1) | ((u64)agg.b << kAggShiftB)));
}

// Resolves the entering state and tape base of `tile` (whose aggregate is already published) and
// publishes its inclusive word.  Whole wave; every poll looks at 256 predecessors (4 per lane):
// on MI355X a cross-XCD poll costs 1-2 us while tiles complete every ~30 ns chip-wide, so the
// nearest inclusive word is routinely > 64 tiles back.
// PRE: the first window (positions 4 * lane .. + 3 behind `tile`) was requested earlier and arrives in `pre`; the loads'
// latency was spent on other work.  If that window does not resolve the tile, live polls follow.
//
// lookback_issue: the 256 words behind `tile` — desc[tile - 256, tile), 2 KiB — are copied to LDS by eight LDS-DMA
// instructions (buffer_load_dword ... lds, sc1: like the relaxed agent-scope loads of the live polls they bypass the
// CU's L1; a word's two halves may come from different versions, which the tag in both halves was made for): no VGPR is
// held while the requests are in flight.  Words before desc[0] are out of the descriptor's range and arrive as 0.
// lookback_fetch (after s_waitcnt vmcnt(0)): lane l picks up positions 4l .. 4l+3 = words [252 - 4l, 255 - 4l] of the copy.
__device__ __forceinline__ void lookback_issue(const u64* desc, u32 tile, u32 lane, uint4* lds) {
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(desc), 0, (int)(tile * 8u), 0x00020000);
    // byte offset of dword (64 q + lane) of the window; "negative" offsets wrap to out-of-range
    u32 voff = (tile - 256u) * 8u + lane * 4u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(reinterpret_cast<u32*>(lds) + 64 * q), 4, (int)voff, 0, 0,
                                                 16 /* sc1 */);
        voff += 256u;
    }
}
__device__ __forceinline__ void lookback_fetch(const uint4* lds, u32 tile, u32 lane, u64 (&pre)[4]) {
    asm volatile("" : "+v"(lane));               // the address is recomputed per tile, not held across the loop
    const uint4 lo = lds[126u - 2u * lane];      // words 252 - 4l, 253 - 4l
    const uint4 hi = lds[127u - 2u * lane];      // words 254 - 4l, 255 - 4l
    pre[3] = ((u64)lo.y << 32) | lo.x;
    pre[2] = ((u64)lo.w << 32) | lo.z;
    pre[1] = ((u64)hi.y << 32) | hi.x;
    pre[0] = ((u64)hi.w << 32) | hi.z;
    (void)tile;
}

template <bool PRE = false>
__device__ __forceinline__ void resolve(u64* desc, u32 tile, u32 epoch, Desc agg, u32 in_quote_in, u32 lane_in,
                                        u32& pin_out, u64& base_out, u32& err, const u64* pre = nullptr,
                                        u32* dbg_windows_spins = nullptr, u32 first_tile = 0) {
    // first_tile: batched launches — the first tile of the BUFFER `tile` belongs to (tiles of all buffers share one index
    // space and one descriptor array); the words before it are other buffers' and read as the virtual word
    const u32 lane = lane_in;
    u32 dbg_windows = 0;
    u32 pin = in_quote_in;
    u64 base = 0;
    // acc = composition of the tiles in (hi, tile): function of the state entering tile hi+1
    u32 acc_p = 0;
    u64 acc_a = 0, acc_b = 0;
    int64_t hi = (int64_t)tile - 1;  // position 0 of the window; tile 0 sees only the virtual word
    u32 spins = 0;
    // Polite start: wait for the nearest predecessor with ONE 8-byte poll at a time.  It started just
    // before this tile and normally publishes last of the ~100 tiles the window needs; 1024 control
    // waves each re-reading 256 words every few hundred cycles would cost more fabric bandwidth than
    // the CSV stream itself (measured: -20 % chip throughput).
    if (!PRE && tile != first_tile) {
        for (;;) {
            u64 x;
            const u32 st = decode_desc(load_desc(desc + (tile - 1)), epoch, x);
            if (__builtin_amdgcn_readfirstlane((int)st) != 0) break;
            __builtin_amdgcn_s_sleep(16);
            if (++spins > kSpinLimit) { err = 1; break; }
        }
    }
    bool use_pre = PRE;
    for (;;) {
        // lane k holds window positions 4k .. 4k+3 (position 0 = nearest predecessor)
        ++dbg_windows;
        u64 d[4], x[4];
        u32 linv = 4, linc = 4;  // first invalid / first inclusive among this lane's four
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const int64_t j = hi - (int64_t)(4 * lane + i);
            // virtual tile -1 = inclusive (in_quote_in, 0): the shard's entering state
            d[i] = encode_desc(kStatusInc, epoch, (u64)in_quote_in);
            if (j >= (int64_t)first_tile) d[i] = (PRE && use_pre) ? pre[i] : load_desc(desc + j);
        }
        use_pre = false;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const u32 status = decode_desc(d[i], epoch, x[i]);
            if (status == 0) linv = (u32)i;
            if (status == kStatusInc) linc = (u32)i;
        }
        const u64 minv = __ballot(linv < 4);
        const u64 minc = __ballot(linc < 4);
        u32 first_inv = 256, first_inc = 256;
        if (minv) {
            const u32 l = (u32)__builtin_ctzll(minv);
            first_inv = 4 * l + (u32)__builtin_amdgcn_readlane((int)linv, (int)l);
        }
        if (minc) {
            const u32 l = (u32)__builtin_ctzll(minc);
            first_inc = 4 * l + (u32)__builtin_amdgcn_readlane((int)linc, (int)l);
        }
        const bool done = first_inc < first_inv;
        const u32 limit = done ? first_inc : first_inv;
        // compose this lane's aggregates at positions < limit (earliest = largest position first)
        Desc f = {0, 0, 0};
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            Desc g;
            g.p = (u32)x[i] & 1u;
            g.a = (u32)(x[i] >> 1) & (kHalfMask >> 1);
            g.b = (u32)(x[i] >> kAggShiftB) & kHalfMask;
            if (4 * lane + (u32)i >= limit) { g.p = 0; g.a = 0; g.b = 0; }
            f = compose(f, g);
        }
        const Desc F = wave_compose_ordered(f, lane, 64);
        const u32 Fp = (u32)__builtin_amdgcn_readfirstlane((int)F.p);
        const u32 Fa = (u32)__builtin_amdgcn_readfirstlane((int)F.a);
        const u32 Fb = (u32)__builtin_amdgcn_readfirstlane((int)F.b);
        if (done) {
            const u32 sel = first_inc & 3u;  // wave-uniform
            const u64 xsel = sel == 0 ? x[0] : sel == 1 ? x[1] : sel == 2 ? x[2] : x[3];
            const u32 xlo = (u32)__builtin_amdgcn_readlane((int)(u32)xsel, (int)(first_inc >> 2));
            const u32 xhi = (u32)__builtin_amdgcn_readlane((int)(u32)(xsel >> 32), (int)(first_inc >> 2));
            const u64 xinc = ((u64)xhi << 32) | xlo;
            u32 st = (u32)xinc & 1u;
            u64 n = xinc >> 1;
            n += st ? Fb : Fa;
            st ^= Fp;
            n += st ? acc_b : acc_a;
            st ^= acc_p;
            pin = st;
            base = n;
            break;
        }
        // fold the resolved-aggregate prefix [0, first_inv) and slide the window past it
        const u64 na = (u64)Fa + (Fp ? acc_b : acc_a);
        const u64 nb = (u64)Fb + (Fp ? acc_a : acc_b);
        acc_a = na;
        acc_b = nb;
        acc_p ^= Fp;
        hi -= (int64_t)first_inv;
        if (first_inv < 256) {  // a predecessor has not published yet: back off, bounded
            __builtin_amdgcn_s_sleep(32);
            if (++spins > kSpinLimit) { err = 1; break; }
        }
    }
    if (dbg_windows_spins) *dbg_windows_spins = (dbg_windows << 16) | (spins & 0xffffu);
    const u32 state_out = pin ^ agg.p;
    const u64 count_out = base + (pin ? agg.b : agg.a);
    if (lane == 0) store_desc(desc + tile, encode_desc(kStatusInc, epoch, (u64)state_out | (count_out << 1)));
    pin_out = pin;
    base_out = base;
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel
// ---------------------------------------------------------------------------------------------
struct RoundMasks {
    u64 st;  // bit i = byte i of this lane's 64-byte stripe is ',', CR or LF
    u64 s;   // in-string mask relative to the wave span's start (span entered outside a string)
};

__device__ __forceinline__ void load_round(rsrc_t rsrc, u32 voff, uint4 (&v)[kRows]) {
    // The whole tile-relative offset lives in voff: the hardware range check covers
    // voffset + immediate only (soffset is excluded from bounds checking), and the check is what
    // makes reading "past the end" of the last tile safe.  j * 1024 folds into the 12-bit immediate.
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const auto x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + (u32)j * 1024u), 0, kLoadAux);
        v[j] = make_uint4(x[0], x[1], x[2], x[3]);
    }
}

// A shard whose start is not 16-byte aligned, or whose end is not a multiple of 64, has one
// partially valid stripe at each end.  (Addressing the shard from its 128-byte line instead would
// save the 1.7 % that line-straddling wave loads cost a misaligned shard, but the two-stripe front
// edge it needs cost the aligned case 2.3 % in this kernel: measured and not adopted.)
// Loads always fetch whole 16-byte chunks (a chunk never
// crosses a page; chunks entirely past the end read as zero through the buffer descriptor — zero
// bytes are class 0, exactly like the reference's zero padding of its last block,
// src/avx/stage1.rs:54-92) and the stray bytes of those two stripes are dropped at the bit level
// after classification.  Each lane knows, per tile, at most one "back" special stripe; the
// "front" one can only be stripe 0 of the shard (tile 0, wave 0, round 0, lane 0).
struct EdgeKeep {
    u32 back_round;  // round index of this lane's partially valid last stripe, or 0xff
    u64 back_keep;   // bits to keep in that round
    u64 front_keep;  // bits to keep in round 0 (all ones unless this lane holds stripe 0 and lo > 0)
    u64 front_esc;   // escape dialect only: a synthetic escape byte right before a misaligned shard start
};

__device__ __forceinline__ EdgeKeep edge_keep_of_tile(u32 lane, u32 w, u32 lo_rel, u32 hi_rel, u32 esc_in = 0) {
    EdgeKeep e;
    e.front_esc = 0;
    e.back_round = 0xffu;
    e.back_keep = ~0ull;
    e.front_keep = ~0ull;
    if ((hi_rel & 63u) != 0u) {
        const u32 sb = hi_rel >> 6;  // tile-relative index of the partial stripe
        constexpr u32 kStripesPerSpan = (u32)kSpanBytes / 64u;
        if (sb / kStripesPerSpan == w && (sb & 63u) == lane) {
            e.back_round = (sb % kStripesPerSpan) >> 6;
            e.back_keep = (1ull << (hi_rel & 63u)) - 1ull;
        }
    }
    if (lo_rel != 0u && w == 0 && lane == 0) {
        e.front_keep = ~((1ull << lo_rel) - 1ull);  // lo_rel < 16
        // "the first valid byte is escaped" = an unescaped escape byte in the dropped slot before it
        if (esc_in) e.front_esc = 1ull << (lo_rel - 1u);
    }
    return e;
}

// wave-private 4-KiB LDS image of one round, in 16-byte slots.  Chunk (stripe s, k) lives in slot
// 4 s + (k ^ ((s >> 2) & 3)): the stripe-owning readers (ds_read_b128 lane groups) are bank-conflict
// free.  The image is filled by LDS-DMA (buffer_load_dwordx4 ... lds): the destination is linear
// (slot = 64 j + lane for row j), so the swizzle sits on the SOURCE side — lane i of row j fetches
// the chunk that belongs in slot 64 j + i.  Each quad of lanes still covers one whole 64-byte
// segment, so the global access stays fully coalesced, and the data never touches a VGPR.
struct StageAddr {
    u32 src;    // tile-relative byte offset this lane fetches for row 0, round 0 of wave 0
    u32 rslot;  // slot of chunk 0 of this lane's stripe (chunk k: rslot ^ k)
};
__device__ __forceinline__ StageAddr stage_addr_of_lane(u32 lane) {
    StageAddr a;
    a.src = (lane >> 2) * 64u + (((lane & 3u) ^ ((lane >> 4) & 3u)) * 16u);
    a.rslot = lane * 4u + ((lane >> 2) & 3u);
    return a;
}

__device__ __forceinline__ void dma_round(rsrc_t rsrc, u32 voff, uint4* stage) {
    // The whole tile-relative offset lives in voff (+ j * 1024 added in a VGPR): the hardware
    // range check covers voffset + immediate only, and the check is what makes reading "past the
    // end" of the last tile safe — out-of-range lanes deposit zeros.
#pragma unroll
    for (int j = 0; j < kRows; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + 64 * j), 16, (int)(voff + (u32)j * 1024u), 0,
                                                 0, kLoadAux);
}

// Count phase of one wave span.  Round r+2 is requested as soon as round r's image has been read
// into registers.  The body must stay ONE
// basic block and must not be duplicated under a branch: with control flow around it LLVM
// hoists/sinks the classification across all eight rounds (128+ live VGPRs).  The sched_barriers
// keep the machine scheduler from doing the same and the opaque asm anchors each round's results.
template <int DIALECT>
__device__ __forceinline__ void count_phase(rsrc_t rsrc, u32 lane, u32 w, const EdgeKeep& ek, uint4* stage0,
                                            uint4* stage1, RoundMasks (&m)[kRounds],
                                            u32& carry, u32& cnt_a, u32& cnt_t, const DialectRegs& dr,
                                            u32 esc_carry) {
    // derived from the lane id once per tile, behind a fence: computed once per kernel these two addresses stay
    // live across the emit phase, where the register peak is (two tiles' masks: 64 VGPRs)
    u32 l_ = lane;
    asm volatile("" : "+v"(l_));
    const StageAddr sa = stage_addr_of_lane(l_);
    // two images per wave: rounds r+1 and r+2 stream in (8 KiB per wave in flight, no VGPRs) while
    // round r is classified
    u32 voff = w * (u32)kSpanBytes + sa.src;  // one running VGPR, advanced per round
    u32 rslot = sa.rslot;
    dma_round(rsrc, voff, stage0);
    voff += (u32)kRoundBytes;
    asm volatile("" : "+v"(voff));
    dma_round(rsrc, voff, stage1);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        uint4* const stage = (r & 1) ? stage1 : stage0;
        __builtin_amdgcn_sched_barrier(0);
        // this round's image has landed once all but the next round's 4 DMAs have retired (LDS-DMA
        // is ordered for a ds_read only by the issuing wave's vmcnt, which counts in issue order)
        if (r + 1 < kRounds)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        uint4 stripe[kRows];  // lane l: the 64 contiguous bytes of stripe l of this round
        // opaque: the eight read addresses (4 chunks x 2 images) are cheap to rebuild (one v_xad each) but,
        // hoisted out of the tile loop, they are what hipcc spills — and a scratch reload in here
        // waits for vmcnt(0), i.e. for the LDS-DMA prefetch of the next rounds (measured: -9 %)
        if (DIALECT >= 2) {
            // the escape variant is short of registers by its third mask: there even `rslot` gets spilled,
            // so it is rebuilt from the lane id (always live) behind the same kind of fence
            u32 l = lane;
            asm volatile("" : "+v"(l));
            rslot = l * 4u + ((l >> 2) & 3u);
        } else {
            asm volatile("" : "+v"(rslot));
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k) stripe[k] = stage[rslot ^ (u32)k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // the image is free again: request round r+2 into it
        if (r + 2 < kRounds) {
            voff += (u32)kRoundBytes;
            asm volatile("" : "+v"(voff));  // opaque: keeps hipcc from materialising 32 offsets up front
            dma_round(rsrc, voff, stage);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32 st16[kRows], q16[kRows], e16[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            if (DIALECT == 0)
                classify16(stripe[k], st16[k], q16[k]);
            else
                classify16_d<DIALECT>(stripe[k], dr, st16[k], q16[k], e16[k]);
        }
        u64 keep = ek.back_round == (u32)r ? ek.back_keep : ~0ull;
        if (r == 0) keep &= ek.front_keep;
        u64 st = ((u64)(st16[0] | (st16[1] << 16)) | ((u64)(st16[2] | (st16[3] << 16)) << 32)) & keep;
        u64 x = ((u64)(q16[0] | (q16[1] << 16)) | ((u64)(q16[2] | (q16[3] << 16)) << 32)) & keep;
        if (DIALECT >= 2) {
            u64 bs = ((u64)(e16[0] | (e16[1] << 16)) | ((u64)(e16[2] | (e16[3] << 16)) << 32)) & keep;
            if (r == 0) bs |= ek.front_esc;
            // escape bytes are rare: a round whose 4 KiB hold none, entered with no pending escape, skips the rest
            // (wave-uniform; ~40 of the round's ~360 VALU)
            if (__ballot(bs != 0) != 0 || esc_carry != 0) {
                // does this stripe END inside an odd run of escape bytes (O), or is it one whole run (A)?
                // The carry into every lane is then the carry chain of the scalar add (O|A) + O + carry-in:
                // O generates, A propagates.
                const u64 nb = ~bs;
                const u32 lead = nb ? (u32)__builtin_clzll(nb) : 64u;
                const u64 gen = __ballot(lead < 64u && (lead & 1u));
                const u64 prop = __ballot(lead == 64u);
                const u64 a = gen | prop;
                const u64 s1 = a + gen;
                const u64 s2 = s1 + esc_carry;
                const u64 into = s2 ^ prop;  // bit l = the first byte of lane l's stripe is escaped
                esc_carry = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(s1 < a) | (u32)(s2 < s1)));
                const u64 escaped = escaped_mask(bs, (u32)(into >> lane) & 1u);
                st &= ~escaped;
                x &= ~escaped;
            }
        }
        // inclusive prefix-xor over the stripe's 64 bits (src/avx/stage1.rs:342-361 does this
        // with one PCLMULQDQ; CDNA has no carry-less multiply)
        x ^= x << 1;
        x ^= x << 2;
        x ^= x << 4;
        x ^= x << 8;
        x ^= x << 16;
        x ^= x << 32;
        // carry across lanes (one ballot + mbcnt) and across rounds (scalar)
        const u64 par = __ballot((x >> 63) != 0);
        const u32 enter = (mbcnt64(par) ^ carry) & 1u;
        carry ^= (u32)__builtin_popcountll(par) & 1u;
        m[r].st = st;
        m[r].s = x ^ (enter ? ~0ull : 0ull);
        cnt_a += (u32)__builtin_popcountll(m[r].st & ~m[r].s);
        cnt_t += (u32)__builtin_popcountll(m[r].st);
        // anchor this round's results here: an opaque asm cannot be sunk or re-ordered
        if (DIALECT >= 2)  // (an "s" operand downstream of the round's uniform branch does not compile: hipcc 7.2)
            asm volatile("" : "+v"(m[r].st), "+v"(m[r].s), "+v"(cnt_a), "+v"(cnt_t));
        else
            asm volatile("" : "+v"(m[r].st), "+v"(m[r].s), "+v"(cnt_a), "+v"(cnt_t), "+s"(carry));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Control block at the head of the context scratch (stage1_kernels.h).  Zeroed once when the scratch
// is allocated; from then on every launch leaves it ready for the next one (the last workgroup to
// finish resets ticket / done_tot / err and advances the epoch), so a launch is ONE kernel: no
// memset before it, no reduction kernel after it, and a captured graph replays correctly.
struct Control {
    u32 ticket;    // next tile id (atomic)
    u32 epoch;     // tag of this launch's descriptor words
    u64 done_tot;  // (workgroups finished) << 48 | comma/CR/LF bytes they saw  (one atomic per workgroup)
    u32 err;       // look-back spin bound hit (atomic or)
    u32 hwm;       // descriptor words possibly dirty since the last wrap of the epoch
    u32 probe_ticket, probe_done;  // csvsimd_hbm_probe_device's own pair
    u32 guess;     // CSVSIMD_ENTER_GUESS launches: 0 = not decided yet, 2 | s = state s was chosen (by the workgroup whose
                   // aggregate was the last of the shard's first kGuessTiles to arrive)
    u32 guess_cnt; // ... how many of those aggregates have arrived
};
static_assert(sizeof(Control) <= CSVSIMD_SCRATCH_CTL_BYTES, "control block must fit its slot");

struct BatchItem {          // device, 64 bytes; written by the host before the launch (csvsimd_stage1_index_batch_device_async)
    const uint8_t* abase;   // 16-byte aligned
    u64 lo, hi;             // valid bytes are abase[lo, hi)
    u64 base_off;           // tape value of byte abase[lo]
    u64* tape;
    u64 tape_cap;
    u32 first_tile;         // global index of this buffer's first tile
    u32 in_quote_in;        // 0 / 1
    u64 reserved;
};
// The table never changes while a launch runs, and every wave reads it with wave-uniform addresses: through the
// CONSTANT address space those reads are scalar loads (s_load via the scalar cache) — measured, the same reads as
// vector loads queue behind the CU's streaming input for 3-5 us each and cost a batched launch 50 % of its rate.
typedef const __attribute__((address_space(4))) BatchItem* batch_cptr;
typedef const __attribute__((address_space(4))) u32* u32_cptr;
static_assert(sizeof(BatchItem) == 64, "one line per buffer");

struct KernelArgs {
    const uint8_t* abase;  // 16-byte aligned
    u64 lo, hi;            // valid bytes are abase[lo, hi)
    u64 base_off;          // tape value of byte abase[lo]
    u32 in_quote_in;
    u32 num_tiles;
    u64* tape;
    u64 tape_cap;
    u64* desc;      // num_tiles words of this context's scratch (epoch-tagged, never zeroed per launch)
    Control* ctl;
    csvsimd_shard_result* result;
    // optional (sharded re-emit): the device csvsimd_stitch of this shard.  When set, the launch does nothing at all
    // unless its `reemit` word (index 9) is 1, and then runs with the stitch's in_quote_in (word 0) — 0 or 1: a wrong
    // CSVSIMD_ENTER_GUESS can err either way.
    const u32* state_ptr;
    // optional (chunked ingest): the device result record of the chunk right before this one.  When set, the launch takes
    // its entering state (and, escape dialects, its escape_in) from that record when it starts — chunk i + 1 is
    // enqueued behind chunk i without the host ever reading chunk i's record: the two values the reference carries
    // between 64-byte blocks (inside_str, src/reader.rs:218) carried between launches on the device.
    const csvsimd_shard_result* chain;
    // optional (batched launch, BATCH instantiation): n_items independent buffers in ONE persistent launch.  Their tiles
    // share one index space (buffer b owns tiles [first_tile[b], first_tile[b + 1])), one ticket, one descriptor array;
    // a tile's look-back stops at its buffer's first tile.  abase / lo / hi / base_off / in_quote_in / tape / tape_cap above
    // are unused then, `result` is an array of n_items records.
    const struct BatchItem* batch;
    const u32* batch_first;  // first_tile of every buffer again, compact: what the tile -> buffer search reads
    u64* batch_tot;          // per buffer: comma/CR/LF bytes, added to by the launch, read and reset by its last workgroup
    u32 n_items;
    // dialect variants only (DIALECT != 0)
    u32 delim, quote, escape;  // bytes; quote / escape 0 = feature off
    u32 escape_in;             // the first byte of the shard is escaped
    u32 hash_sh1, hash_sh2, hash_lut_lo, hash_lut_hi, hash_cls_lo, hash_cls_hi;  // DIALECT 3 (see DialectRegs)
    // pacing knobs, chosen by the host from the launch size
    u32 emit_delay;  // x 640 cycles of s_sleep between barrier B and the emit phase
    u32 count_prio;  // 1: count phases run at s_setprio 3
    u32* cu_token;   // non-null: the workgroups that share a CU take turns in the count phase (one u32 per CU)
    u32 token_mode;  // 1 = token, then ticket; 2 = both atomics in flight together
#ifdef CSVSIMD_DEV_PROBES
    u64* prof;  // timing build: per-phase stamp sums
#endif
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// "the first byte of the shard is escaped": the launch's own argument, or the previous chunk's escape_out (chained
// launches; re-read where it is needed rather than carried through the tile loop in a register)
template <int DIALECT>
__device__ __forceinline__ u32 escape_in_of(const KernelArgs& args) {
    if (DIALECT < 2) return 0u;
    const csvsimd_shard_result* c = args.chain;
    asm volatile("" : "+s"(c));  // the test is redone here, from the SGPR pair: hoisted out of the tile loop it becomes a spilled VGPR
    if (c)
        return (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&c->escape_out, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT)) & 1u;
    return args.escape_in;
}

// Window slot of entry k (opt-in build knob, default off).  A lane scatters its stripe's entries to consecutive
// ranks, so in one ds_write_b16 the 64 lanes hit ranks that are `entries per stripe` apart: on a regular dense file
// (1024 x 4: 12.8 per stripe) that is a stride of 6.4 dwords, i.e. only five of the 32 banks (SQ_LDS_BANK_CONFLICT =
// 43.6 M per GiB, profiles/r02_pmc_1024x4_dense_1GiB.json).  XOR-ing the bank bits (1..5 of the u16 index) with
// the next five index bits removes the conflicts (profiles/r02_pmc_dense_swizzled_window.json) — and changes the
// kernel time by less than 1 %: the scatter is not what bounds the dense corpus (its HBM write share is, DESIGN.md
// §4), and the three extra VALU per entry cost the sparse corpora 1-3 %.  Measured, kept as a knob, not the default.
#ifndef CSVSIMD_WINDOW_SWIZZLE
#define CSVSIMD_WINDOW_SWIZZLE 0
#endif
__device__ __forceinline__ u32 comp_slot(u32 k) {
#if CSVSIMD_WINDOW_SWIZZLE
    return k ^ (((k >> 6) & 31u) << 1);
#else
    return k;
#endif
}

// Writes window entries comp[0, n) (u16 offsets relative to the span) to tape[run, run + n) as
// fully coalesced non-temporal stores, 16 bytes (two entries) per lane wherever the address allows.
template <bool NOSTORE = false>
__device__ __forceinline__ void flush_window(u64* const tape, const u64 tape_cap_in, const unsigned short* comp, u32 n, u64 run,
                                             u64 span_off, u32 lane) {
    if (n == 0) return;
    // NOSTORE (development probe): an impossible capacity keeps the loop but drops the stores
    const u64 tape_cap = NOSTORE ? (tape_cap_in & 1ull) : tape_cap_in;
    // Entry k sits at byte address tape + 8 (run + k).  Head entries are peeled so that the main
    // loop's wave stores start on a 128-byte boundary (the L2 line): each 1-KiB store then covers
    // whole lines only.  Measured (scripts/ubench_mem.hip, scripts/exp_width.py): wave stores that
    // straddle lines cost 4 % of the whole stream at a 20 % write share and 13 % on the dense
    // corpus, whose tape offsets are not a multiple of the line.
    u32 head = (0u - (u32)(((uintptr_t)tape >> 3) + run)) & (u32)(kStoreAlignEntries - 1);
    head = head < n ? head : n;
    if (lane < head && run + lane < tape_cap)
        __builtin_nontemporal_store(span_off + comp[comp_slot(lane)], tape + run + lane);
    const u32 npairs = (n - head) >> 1;
    for (u32 i = lane; i < npairs; i += 64) {
        const u32 k = head + 2 * i;
        const u64 idx = run + k;
        const u64 e0 = span_off + comp[comp_slot(k)], e1 = span_off + comp[comp_slot(k + 1)];
        if (idx + 1 < tape_cap) {
            const u32x4 x = {(u32)e0, (u32)(e0 >> 32), (u32)e1, (u32)(e1 >> 32)};
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(tape + idx));
        } else if (idx < tape_cap) {
            __builtin_nontemporal_store(e0, tape + idx);
        }
    }
    if (((n - head) & 1u) && lane == 0) {
        const u64 idx = run + n - 1;
        if (idx < tape_cap) __builtin_nontemporal_store(span_off + comp[comp_slot(n - 1)], tape + idx);
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// scatters the set bits of R (this lane's stripe of round r) as u16 span-relative offsets to
// comp[p], comp[p+1], ... keeping only positions < kCompCap (p wraps for entries before a window)
__device__ __forceinline__ void scatter_bits(unsigned short* comp, u64 R, u32 p, u32 stripe_rel) {
    u32 lo = (u32)R, hi = (u32)(R >> 32);
    while (lo) {
        const u32 b = (u32)__builtin_ctz(lo);
        lo &= lo - 1;
        if (p < (u32)kCompCap) comp[comp_slot(p)] = (unsigned short)(stripe_rel + b);
        ++p;
    }
    while (hi) {
        const u32 b = (u32)__builtin_ctz(hi) + 32u;
        hi &= hi - 1;
        if (p < (u32)kCompCap) comp[comp_slot(p)] = (unsigned short)(stripe_rel + b);
        ++p;
    }
}

// Emits the tape entries of one wave span from its held masks: ordered compaction through the
// wave-private LDS window.  Rounds are batched into the window until it is full, so a sparse span
// (CSV with long fields) leaves as one long run of 16-byte stores.
//   wstate: absolute in-string state entering the span; run: tape index of the span's first entry
template <bool NOSTORE = false>
__device__ __forceinline__ void emit_span(u64* const tape, const u64 tape_cap, const RoundMasks (&m)[kRounds], u32 lane,
                                          const u64 span_off /* tape value of the span's byte 0 */, u32 wstate, u64 run,
                                          unsigned short* comp) {
    const u64 flipall = wstate ? ~0ull : 0ull;
    u32 fill = 0;                                           // entries waiting in the window

    // wave-uniform by construction; tell the compiler (it arrives through LDS, i.e. in a VGPR)
    run = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(run >> 32)) << 32) |
          (u32)__builtin_amdgcn_readfirstlane((int)(u32)run);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const u64 R = m[r].st & ~(m[r].s ^ flipall);
        const u32 c = (u32)__builtin_popcountll(R);
        const u32 incl = wave_incl_scan_add(c);
        const u32 n_r = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 excl = incl - c;
        // the round offset goes through an opaque SGPR: folded, hipcc keeps seven hoisted copies of
        // lane * 64 + r * 4096 alive across the whole tile loop — exactly the registers whose absence makes
        // it spill (round 1: 11 VGPRs to scratch; now none)
        u32 roff = (u32)r * kRoundBytes;
        asm volatile("" : "+s"(roff));
        const u32 stripe_rel = roff + lane * 64u;
        if (fill + n_r > (u32)kCompCap) {
            wave_lds_fence();
            flush_window<NOSTORE>(tape, tape_cap, comp, fill, run, span_off, lane);
            wave_lds_fence();
            run += fill;
            fill = 0;
        }
        if (n_r <= (u32)kCompCap) {
            scatter_bits(comp, R, fill + excl, stripe_rel);
            fill += n_r;
        } else {
            // more than half of this round's bytes are structural: several window passes
            for (u32 win = 0; win < n_r; win += kCompCap) {
                scatter_bits(comp, R, excl - win, stripe_rel);
                wave_lds_fence();
                const u32 n_win = (n_r - win) < (u32)kCompCap ? (n_r - win) : (u32)kCompCap;
                flush_window<NOSTORE>(tape, tape_cap, comp, n_win, run, span_off, lane);
                wave_lds_fence();
                run += n_win;
            }
        }
        // keep the rounds sequential: interleaving all eight scans costs ~40 VGPRs
        asm volatile("" : "+s"(fill), "+s"(run));
        __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
    flush_window<NOSTORE>(tape, tape_cap, comp, fill, run, span_off, lane);
    wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// DENSE instantiation (round 4): the emit phase of a delimiter-dense file.  On the 1024 x 4 corpus a wave span holds
// ~6 500 entries, the emit phase is ~21 us of INSTRUCTIONS per tile (profiles/r03_tile_timeline_dense_nostore.txt) and
// the kernel is bound by their issue, not by HBM.  emit_span above spends ~16 instructions per entry and loop trip in
// its scatter (a capacity test around every LDS store: its window may be smaller than a round) and ~25 per pair in its
// flush (a tape-capacity test around every global store).  Here:
//   * the window is the wave's two stage images side by side, 4 096 entries: a round (<= 4 096 set bits) ALWAYS fits, so
//     the scatter needs no test and never takes several passes, and a dense span flushes twice instead of four times;
//   * the tape's capacity is tested once per window: the flush loop is loads, adds, store;
//   * both halves of a pair come from one 32-bit LDS read when the window index is even, and the 64-bit add of the
//     span's offset shrinks to a 32-bit one when it cannot carry (tested once per flush).
// A window that does not fit the tape's capacity leaves entry by entry (the capacity protocol is the rare case).
// Chosen per launch by the entries-per-byte the context has last seen (capi.cpp); results are identical by
// construction and by test (tests/test_gpu_dense_variant.py runs every configuration through both).
// ---------------------------------------------------------------------------------------------
static constexpr int kDenseCap = 2 * kRoundBytes / 2;  // u16 entries in 8 KiB
static_assert(kDenseCap >= kRoundBytes, "a round's set bits always fit the dense window");

__device__ __forceinline__ void scatter_bits_nocheck(unsigned short* comp, u64 R, u32 p, u32 stripe_rel) {
    u32 lo = (u32)R, hi = (u32)(R >> 32);
    unsigned short* q = comp + p;
    while (lo) {
        *q++ = (unsigned short)(stripe_rel + (u32)__builtin_ctz(lo));
        lo &= lo - 1;
    }
    stripe_rel += 32u;
    while (hi) {
        *q++ = (unsigned short)(stripe_rel + (u32)__builtin_ctz(hi));
        hi &= hi - 1;
    }
}

// comp[0, n) -> tape[run, run + n).  ONE capacity test per window: a window that does not fit the tape whole (the caller's
// tape is too small: the capacity protocol, never the timed case) leaves entry by entry, each store guarded — a loop of
// a dozen instructions (emit_span inlined as the fallback cost this instantiation two spilled VGPRs).
__device__ __forceinline__ void flush_window_dense(u64* const tape, const u64 tape_cap, const unsigned short* comp, u32 n, u64 run,
                                                   u64 span_off, u32 lane) {
    if (n == 0) return;
    if (run + n > tape_cap) {  // wave-uniform
        for (u32 k = lane; k < n; k += 64)
            if (run + k < tape_cap) __builtin_nontemporal_store(span_off + comp[k], tape + run + k);
        return;
    }
    u32 head = (0u - (u32)(((uintptr_t)tape >> 3) + run)) & (u32)(kStoreAlignEntries - 1);  // see flush_window
    head = head < n ? head : n;
    if (lane < head) __builtin_nontemporal_store(span_off + comp[lane], tape + run + lane);
    const u32 npairs = (n - head) >> 1;
    const u32 base_lo = (u32)span_off, base_hi = (u32)(span_off >> 32);
    u64* const out = tape + run + head;
    if (base_lo <= 0xffff0000u) {  // offset + u16 cannot carry: 32-bit adds (wave-uniform)
        if ((head & 1u) == 0u) {   // pairs sit in aligned dwords of the window (wave-uniform)
            const u32* const pairs = reinterpret_cast<const u32*>(comp + head);
            for (u32 i = lane; i < npairs; i += 64) {
                const u32 pr = pairs[i];
                const u32x4 x = {base_lo + (pr & 0xffffu), base_hi, base_lo + (pr >> 16), base_hi};
                __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out + 2 * i));
            }
        } else {
            for (u32 i = lane; i < npairs; i += 64) {
                const u32 c0 = comp[head + 2 * i], c1 = comp[head + 2 * i + 1];
                const u32x4 x = {base_lo + c0, base_hi, base_lo + c1, base_hi};
                __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out + 2 * i));
            }
        }
    } else {
        for (u32 i = lane; i < npairs; i += 64) {
            const u64 e0 = span_off + comp[head + 2 * i], e1 = span_off + comp[head + 2 * i + 1];
            const u32x4 x = {(u32)e0, (u32)(e0 >> 32), (u32)e1, (u32)(e1 >> 32)};
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out + 2 * i));
        }
    }
    if (((n - head) & 1u) && lane == 0) __builtin_nontemporal_store(span_off + comp[n - 1], tape + run + n - 1);
}

// The DENSE instantiation's speculative path, workgroup wide: its tile is 64 KiB, so an offset relative to the TILE fits the
// window's u16, and the eight waves' 8-KiB windows are one contiguous 64-KiB block: every wave scatters its span's entries
// at their place in the TILE's order (its entries start behind those of the waves before it: known from the aggregates,
// before the look-back), and after barrier B all 512 threads write the tile's run of the tape front to back — one stream
// of stores per workgroup instead of eight (a bare stream of this write-heavy mix gains 4-7 % from that order alone,
// profiles/r03_ubench_dense_write_patterns.txt).
static constexpr u32 kWgCap = (u32)kWaves * (u32)kDenseCap;  // entries the workgroup's window holds
__device__ __forceinline__ void scatter_span_tile(const RoundMasks (&m)[kRounds], u32 lane, u32 wstate, unsigned short* win,
                                                  u32 first, u32 span_rel) {
    const u64 flipall = wstate ? ~0ull : 0ull;
    u32 fill = first;  // the tile-order index of this wave's first entry
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const u64 R = m[r].st & ~(m[r].s ^ flipall);
        const u32 c = (u32)__builtin_popcountll(R);
        const u32 incl = wave_incl_scan_add(c);
        const u32 n_r = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        u32 roff = span_rel + (u32)r * kRoundBytes;
        asm volatile("" : "+s"(roff));
        scatter_bits_nocheck(win, R, fill + incl - c, roff + lane * 64u);
        fill += n_r;
        asm volatile("" : "+s"(fill));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// win[0, n) (tile-relative u16 offsets, tile order) -> tape[run, run + n) by ALL threads of the workgroup (t = thread id)
__device__ __forceinline__ void flush_tile_dense(u64* const tape, const u64 tape_cap, const unsigned short* win, u32 n, u64 run,
                                                 u64 tile_off, u32 t) {
    if (n == 0) return;
    if (run + n > tape_cap) {  // workgroup-uniform: the caller's tape is too small (capacity protocol)
        for (u32 k = t; k < n; k += (u32)kThreads)
            if (run + k < tape_cap) __builtin_nontemporal_store(tile_off + win[k], tape + run + k);
        return;
    }
    u32 head = (0u - (u32)(((uintptr_t)tape >> 3) + run)) & (u32)(kStoreAlignEntries - 1);  // see flush_window
    head = head < n ? head : n;
    if (t < head) __builtin_nontemporal_store(tile_off + win[t], tape + run + t);
    const u32 npairs = (n - head) >> 1;
    const u32 base_lo = (u32)tile_off, base_hi = (u32)(tile_off >> 32);
    u64* const out = tape + run + head;
    if (base_lo <= 0xffff0000u && (head & 1u) == 0u) {  // 32-bit adds, pairs in aligned dwords (workgroup-uniform)
        const u32* const pairs = reinterpret_cast<const u32*>(win + head);
        for (u32 i = t; i < npairs; i += (u32)kThreads) {
            const u32 pr = pairs[i];
            const u32x4 x = {base_lo + (pr & 0xffffu), base_hi, base_lo + (pr >> 16), base_hi};
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out + 2 * i));
        }
    } else {
        for (u32 i = t; i < npairs; i += (u32)kThreads) {
            const u64 e0 = tile_off + win[head + 2 * i], e1 = tile_off + win[head + 2 * i + 1];
            const u32x4 x = {(u32)e0, (u32)(e0 >> 32), (u32)e1, (u32)(e1 >> 32)};
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out + 2 * i));
        }
    }
    if (((n - head) & 1u) && t == 0) __builtin_nontemporal_store(tile_off + win[n - 1], tape + run + n - 1);
}

// emit_span for the DENSE instantiation: comp = the wave's 8-KiB window
__device__ __forceinline__ void emit_span_dense(u64* const tape, const u64 tape_cap, const RoundMasks (&m)[kRounds], u32 lane,
                                                const u64 span_off, u32 wstate, u64 run, unsigned short* comp) {
    const u64 flipall = wstate ? ~0ull : 0ull;
    u32 fill = 0;
    run = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(run >> 32)) << 32) |
          (u32)__builtin_amdgcn_readfirstlane((int)(u32)run);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const u64 R = m[r].st & ~(m[r].s ^ flipall);
        const u32 c = (u32)__builtin_popcountll(R);
        const u32 incl = wave_incl_scan_add(c);
        const u32 n_r = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        u32 roff = (u32)r * kRoundBytes;
        asm volatile("" : "+s"(roff));  // (see emit_span)
        if (fill + n_r > (u32)kDenseCap) {
            wave_lds_fence();
            flush_window_dense(tape, tape_cap, comp, fill, run, span_off, lane);
            wave_lds_fence();
            run += fill;
            fill = 0;
        }
        scatter_bits_nocheck(comp, R, fill + incl - c, roff + lane * 64u);
        fill += n_r;
        asm volatile("" : "+s"(fill), "+s"(run));
        __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
    flush_window_dense(tape, tape_cap, comp, fill, run, span_off, lane);
    wave_lds_fence();
}

// The end of a launch (wave 0 of every workgroup): count this workgroup done; the workgroup whose add completes the
// count writes the result record from the last tile's inclusive word and leaves the control block ready for the next
// launch.
struct Control;
__device__ __forceinline__ u32 wait_for_guess(Control* ctl, u32& err);
constexpr u32 kEnterGuessFwd = CSVSIMD_ENTER_GUESS;
template <int DIALECT, bool NO_LOOKBACK, bool BATCH = false>
__device__ __forceinline__ void finish_launch(const KernelArgs& args, u32 epoch, u32 inq_in, u64 wg_tot, u32 err, u32 lane) {
    Control* const ctl = args.ctl;
    if (err && lane == 0) atomicOr(&ctl->err, 1u);
    // every descriptor word this workgroup published (and its error flag) must have left before it
    // counts itself done: the workgroup that finishes last reads the last tile's inclusive word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u64 old = 0;
    if (lane == 0)
        old = atomicAdd((unsigned long long*)&ctl->done_tot, (unsigned long long)((1ull << 48) | wg_tot));
    const u32 old_lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)old);
    const u32 old_hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(old >> 32));
    if ((old_hi >> 16) != gridDim.x - 1u) return;

    const u64 total = ((((u64)old_hi << 32) | old_lo) & ((1ull << 48) - 1ull)) + wg_tot;
    // the launch ends when this workgroup does: the control words it needs are requested together with the first poll
    // of the last tile's word (three round trips in flight at once instead of one after the other)
    const u32 err_seen = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 hwm_seen = __hip_atomic_load(&ctl->hwm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 e = 0;
    // CSVSIMD_ENTER_GUESS and this workgroup never resolved a tile: the choice of tile 0's workgroup is there by now
    // (every workgroup has finished); an empty shard has no tile 0 and is "entered outside"
    if (DIALECT >= 2) asm volatile("" : "+s"(inq_in));  // see uniform_again
    if (inq_in == kEnterGuessFwd) inq_in = args.num_tiles ? wait_for_guess(ctl, e) : 0u;
    u32 state_out = inq_in;
    u64 count = 0;
    if (args.num_tiles > 0 && !NO_LOOKBACK && !BATCH) {
        // published by whichever workgroup resolved the last tile, before it counted itself done
        u64 x = 0;
        for (u32 spins = 0;; ++spins) {
            if (decode_desc(load_desc(args.desc + (args.num_tiles - 1)), epoch, x) == kStatusInc) break;
            if (spins > kSpinLimit) { e = 1; x = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        state_out = (u32)x & 1u;
        count = x >> 1;
    }
    e |= err_seen;  // every other workgroup raised its flag before it counted itself done
    // escape dialect: is the byte after the shard escaped? (chains into the next shard's escape_in)
    u32 esc_out = 0;
    if (DIALECT >= 2)
        esc_out = escape_run_parity(args.abase, args.lo, args.hi, args.hi, args.escape, escape_in_of<DIALECT>(args), lane);
    if (BATCH) {
        // one record per buffer, 64 buffers at a time: the inclusive word of the buffer's LAST tile holds its count and
        // leaving state (published by whichever workgroup resolved that tile, before it counted itself done); an empty
        // buffer has no tile: nothing counted, the state passes through
        for (u32 b = lane; b < args.n_items; b += 64u) {
            const BatchItem* const it = args.batch + b;
            const u32 first = it->first_tile;
            const u32 next = b + 1u < args.n_items ? args.batch[b + 1u].first_tile : args.num_tiles;
            const u32 inq = it->in_quote_in & 1u;
            u32 st = inq, eb = e;
            u64 cnt = 0;
            if (next > first) {
                u64 x = 0;
                for (u32 spins = 0;; ++spins) {
                    if (decode_desc(load_desc(args.desc + (next - 1u)), epoch, x) == kStatusInc) break;
                    if (spins > kSpinLimit) { eb = 1; x = inq; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                st = (u32)x & 1u;
                cnt = x >> 1;
            }
            const u64 tot = __hip_atomic_load(&args.batch_tot[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            csvsimd_shard_result* const r = args.result + b;
            r->count = cnt;
            r->count_enter_outside = inq ? tot - cnt : cnt;
            r->count_enter_inside = inq ? cnt : tot - cnt;
            r->quote_parity = st ^ inq;
            r->in_quote_out = st;
            r->error = eb;
            r->escape_out = 0;
            const u64 cap = it->tape ? it->tape_cap : 0;
            r->written = cnt < cap ? cnt : cap;
            r->in_quote_in_used = inq;
            r->reserved0 = 0;
            r->reserved1 = 0;
            args.batch_tot[b] = 0;  // ready for the next launch over the same buffers (a replayed graph)
        }
    }
    // ONLY NOW — every inclusive word this function reads (the shard's last tile above, every buffer's last tile in the
    // batched loop) has been read — may the words be cleared:
    const u32 next_epoch = (epoch + 1u) & kEpochMask;
    u32 hwm = hwm_seen > args.num_tiles ? hwm_seen : args.num_tiles;
    if (next_epoch == 0u) {
        // the epoch wraps: words tagged in earlier rounds of the counter must not be mistaken for
        // the next round's, so everything used since the last wrap is cleared (once per 1024 launches)
        // (round 3's first batched kernel cleared BEFORE its per-buffer loop: every 1024th batch reported the spin bound —
        // found by scripts/soak.py's batched mode, pinned by test_batch_across_an_epoch_wrap)
        for (u32 i = lane; i < hwm; i += 64u) args.desc[i] = 0;
        hwm = 0;
    }
    if (lane == 0) {
        csvsimd_shard_result* const r = BATCH ? nullptr : args.result;
        if (!BATCH) {
        r->count = count;
        // total = count_enter_outside + count_enter_inside, whichever hypothesis was run
        r->count_enter_outside = inq_in ? total - count : count;
        r->count_enter_inside = inq_in ? count : total - count;
        r->quote_parity = state_out ^ inq_in;
        r->in_quote_out = state_out;
        r->error = e;
        r->escape_out = esc_out;
        r->written = count < args.tape_cap ? count : args.tape_cap;
        r->in_quote_in_used = inq_in;
        r->reserved0 = 0;
        r->reserved1 = 0;
        }
        // ready for the next launch (made visible by the end-of-kernel release)
        ctl->ticket = 0;
        ctl->done_tot = 0;
        ctl->err = 0;
        ctl->guess = 0;
        ctl->guess_cnt = 0;
        ctl->hwm = hwm;
        ctl->epoch = next_epoch;
    }
}

// Speculative scatter of a whole wave span into the window, done BEFORE the tile is resolved (while wave 0's look-back
// polls are in flight): the tile's entering state is GUESSED — as the one of the two hypotheses under which the tile
// has more entries (both counts are in its aggregate: read with the wrong quote parity, text outside strings looks
// quoted and nearly every separator disappears; a quote-free tile has no entries at all "entered inside") — and only
// the tape index base is still missing, which the window does not need.  A wrong guess costs the tile the ordinary
// emit after the look-back, never correctness.  The caller makes sure the span's entries fit one window.
__device__ __forceinline__ void scatter_bits_nocheck(unsigned short* comp, u64 R, u32 p, u32 stripe_rel);
template <bool NOCHECK = false>
__device__ __forceinline__ void scatter_span_spec(const RoundMasks (&m)[kRounds], u32 lane, u32 wstate, unsigned short* comp) {
    const u64 flipall = wstate ? ~0ull : 0ull;
    u32 fill = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const u64 R = m[r].st & ~(m[r].s ^ flipall);
        const u32 c = (u32)__builtin_popcountll(R);
        const u32 incl = wave_incl_scan_add(c);
        const u32 n_r = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        u32 roff = (u32)r * kRoundBytes;
        asm volatile("" : "+s"(roff));
        if (NOCHECK) scatter_bits_nocheck(comp, R, fill + incl - c, roff + lane * 64u);  // (the caller made sure the span fits)
        else scatter_bits(comp, R, fill + incl - c, roff + lane * 64u);
        fill += n_r;
        asm volatile("" : "+s"(fill));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Workgroup barrier that also drains this wave's LDS traffic first.  hipcc (ROCm 7.2) was observed to
// emit a bare s_barrier for __syncthreads() when the preceding ds_write sits in a predecessor block
// across a loop back-edge; the released waves' ds_reads then overtook the write (1 tile in ~10^5
// read a stale tile id).  The wait is cheap and makes the hand-off independent of that analysis.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
// Barrier B.  __syncthreads() is a workgroup-scope release fence: behind a global store hipcc makes it
// `s_waitcnt vmcnt(0)`, i.e. all eight waves would wait for the acknowledgement of wave 0's descriptor word (and of their
// own previous tape stores) before the first tape store of this tile may leave.  The waves of a workgroup talk to each
// other through LDS only (tile id, wave descriptors, entering state and base, the windows); global memory carries the
// descriptors (agent-scope atomics of wave 0, ordered by their own tags) and the tape, which nobody reads before the
// kernel ends.  So the hand-off waits for LDS alone: +0.4-0.5 % on the 8-GiB shard (profiles/r03_ab_barrier.txt).
__device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DBG (development probes: instantiated only in -DCSVSIMD_DEV_PROBES builds, which the product library is
// not): 0 = normal, bit 0 = skip classification (loads only), bit 1 = static tiles (no ticket; only without
// look-back), bit 2 = no look-back, bit 3 = accumulate per-phase s_memrealtime stamps of waves 0 and 1
// into args.prof (timing build), bit 4 = emit phase without its global stores
//
// One iteration of the workgroup loop (three barriers):
//   ticket -> [T] -> count phase of tile_i (masks -> registers) -> [A] -> wave 0: publish the
//   aggregate of tile_i, then resolve tile_{i-1} (look-back) -> [B] -> all waves emit tile_{i-1}.
//
// The tile counted in iteration i-1 is HELD in registers (masks of the whole span + descriptors)
// and is resolved and emitted one iteration later.  Measured reasons (MI355X): a cross-XCD poll
// queues behind the CU's own streaming loads (3-5 us) while tiles complete every ~30 ns chip-wide,
// so a tile can only resolve once every predecessor back to the nearest inclusive word (~100-300
// tiles) has published.  Resolving right after the own count phase makes every workgroup wait for
// the slowest of those concurrently running predecessors (-25 % throughput); one tile-time later
// they have all long published and the look-back is one or two polls.  Variants measured and
// rejected: drawing the ticket early (scrambles the start order), a dedicated control wave per
// workgroup (5-wave groups halve residency: the dispatcher reserves ceil(waves/4) slots on every
// SIMD; 7+1-wave groups resolve early and their compute waves wait for it: 3.6 vs 4.4 TB/s).
// Count phases never wait on anything but their own loads, so every aggregate is eventually
// published by a running workgroup: the look-back always terminates, with no residency assumption
// (tile ids come from an atomic ticket drawn when the workgroup is ready to start the tile).
//
// The launch is self-contained (round 2): no memset before it (epoch-tagged descriptor words), no
// reduction kernel after it — the last workgroup to finish (one returning atomic per workgroup on
// Control::done_tot, which also carries the workgroups' comma/CR/LF byte totals) writes the whole
// result record from the last tile's inclusive word and leaves the control block ready for the
// next launch.
#if CSVSIMD_WAVES_PER_EU > 0
#define CSVSIMD_LAUNCH_BOUNDS __launch_bounds__(kThreads, CSVSIMD_WAVES_PER_EU)
#else
#define CSVSIMD_LAUNCH_BOUNDS __launch_bounds__(kThreads)
#endif
constexpr u32 kEnterGuess = CSVSIMD_ENTER_GUESS;
constexpr u32 kStitchReemitWord = 9;  // csvsimd_stitch::reemit as a u32 index
static_assert(offsetof(csvsimd_stitch, reemit) == 4 * kStitchReemitWord && offsetof(csvsimd_stitch, in_quote_in) == 0,
              "the re-emit launch reads these two words");

// CSVSIMD_ENTER_GUESS: a shard cut out of the middle of a file, entering state unknown.  The shard's first kGuessTiles
// tiles (2 MiB) vote: each of their workgroups counts itself in after publishing its aggregate (P, A, B); the one whose
// count completes the set reads the aggregates back, composes them IN ORDER — the counts of tiles 1.. depend on the
// parity of the tiles before them — and chooses the entering state under which those tiles together hold more entries:
// read with the wrong quote parity, text outside strings looks quoted and nearly every separator disappears.  (Round 2
// let tile 0 decide alone: one long quoted field at the start of a shard was enough to fool it; now it takes 2 MiB of
// them.)  The voters' count phases run concurrently at the start of the launch and wait for nothing, so the choice
// is there before any workgroup needs it (wave 0 of a workgroup that resolves a tile: wait_for_guess); no tile
// becomes inclusive before the choice exists, so the words read back here are still aggregates.
constexpr u32 kGuessTiles = (2u << 20) / (u32)kTileBytes;  // 8 tiles of 256 KiB (32 of the dense geometry's 64 KiB)
constexpr u32 kPinDeferred = 0xffffffffu;                  // s_pin: nothing was resolved in this iteration
__device__ __forceinline__ void guess_vote(u64* desc, Control* ctl, u32 num_tiles, u32 epoch, u32 lane, u32& err) {
    const u32 voters = num_tiles < kGuessTiles ? num_tiles : kGuessTiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's aggregate has left before it is counted in
    u32 old = 0;
    if (lane == 0) old = atomicAdd(&ctl->guess_cnt, 1u);
    old = (u32)__builtin_amdgcn_readfirstlane((int)old);
    if (old != voters - 1u) return;
    // one word after the other, in a loop that stays a loop: this runs once per launch, in ONE workgroup, at a point
    // where wave 0 holds two tiles' masks — a wave-parallel composition here cost the default kernel 4 spilled VGPRs
    Desc F = {0, 0, 0};
#pragma unroll 1
    for (u32 k = 0; k < voters; ++k) {
        u64 x = 0;
        for (u32 spins = 0; decode_desc(load_desc(desc + k), epoch, x) != kStatusAgg; ++spins) {
            // (a word still in flight between two XCDs: look again)
            if (spins > kSpinLimit) { err = 1; x = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        Desc d;
        d.p = (u32)x & 1u;
        d.a = (u32)(x >> 1) & (kHalfMask >> 1);
        d.b = (u32)(x >> kAggShiftB) & kHalfMask;
        F = compose(F, d);
    }
    if (lane == 0)
        __hip_atomic_store(&ctl->guess, 2u | (F.b > F.a ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CSVSIMD_ENTER_GUESS: the choice of the shard's first tiles (Control::guess), once it is there
__device__ __forceinline__ u32 wait_for_guess(Control* ctl, u32& err) {
    for (u32 spins = 0;; ++spins) {
        const u32 g = (u32)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_load(&ctl->guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (g) return g & 1u;
        if (spins > kSpinLimit) { err = 1; return 0u; }
        __builtin_amdgcn_s_sleep(16);
    }
}

// Escape dialects: a loop-invariant test of a kernel argument (count_prio != 0, in_quote_in == GUESS) is hoisted out of the
// tile loop by hipcc as a 0/1 VALUE — and, the uniform state of the loop exceeding a wave's SGPRs, that value ends up in a
// VGPR, which is then spilled to scratch.  Behind this fence the test is redone where it is used, from the SGPR.
template <bool FENCE>
__device__ __forceinline__ u32 uniform_again(u32 v) {
    if (FENCE) asm volatile("" : "+s"(v));
    return v;
}

// Batched launches (BATCH): which buffer does `tile` belong to?  first_tile[] ascends; the buffer is the LAST one whose
// first tile is <= tile (empty buffers share their successor's first tile and are never chosen).  Scalar code on scalar
// loads: a binary search down to a window of 16, then the window's loads go out together.
__device__ __forceinline__ u32 batch_item_of(const u32* first_tiles, u32 n_items, u32 tile) {
    const u32_cptr ft = (u32_cptr)first_tiles;
    u32 lo = 0, hi = n_items;
    while (hi - lo > 16u) {
        const u32 mid = (lo + hi) >> 1;
        if (ft[mid] <= tile) lo = mid; else hi = mid;
    }
    u32 idx = lo;
    for (u32 j = lo + 1u; j < hi; ++j)
        if (ft[j] <= tile) idx = j;
    return idx;
}

#ifdef CSVSIMD_WG_END_TRACE
// dev builds only: per workgroup {first instruction, last ticket drawn, exit} in s_memrealtime ticks + tiles it counted
__device__ u64 g_wg_trace[2048][4];
#endif
template <bool EMIT, int DBG = 0, int DIALECT = 0, bool BATCH = false, bool DENSE = false>
__global__ CSVSIMD_LAUNCH_BOUNDS void stage1_kernel(const KernelArgs args) {
#ifdef CSVSIMD_WG_END_TRACE
    const u64 wg_t0 = __builtin_amdgcn_s_memrealtime();
    u64 wg_tlast = 0;
    u32 wg_tiles = 0;
#endif
    __shared__ u32 s_tile;
    __shared__ u32 s_wdesc[kWaves][3];
    __shared__ u32 s_pin;
    __shared__ u64 s_base;
    // CSVSIMD_ENTER_GUESS launches on a grid smaller than the vote (round 5): tiles this workgroup counted, published and
    // voted for but could not keep — their masks were dropped when the choice was not there yet and the workgroup went on
    // drawing tickets; they are counted again (not published, not voted for again) once the choice exists
    __shared__ u32 s_owed[kGuessTiles];
    __shared__ u32 s_owed_n, s_owed_head, s_redo;  // s_redo: the tile of this iteration is an owed one (wave 0 reads it back)
    // wave-private images: input transpose in the count phase (two, double-buffered LDS-DMA), u16
    // compaction window in the emit phase (the uses never overlap in time within a wave).
    // DENSE: a wave's two images are ONE 8-KiB block, so that together they are its 4 096-entry emit window
    // (for the default instantiation the two stay separate arrays: side by side they cost it 6 %, NOTEBOOK.md round 2)
    constexpr int kImg = kRoundBytes / 16;
    __shared__ uint4 s_stage_a[kWaves][DENSE ? 2 * kImg : kImg];
    __shared__ uint4 s_stage_bb[DENSE ? 1 : kWaves][DENSE ? 1 : kImg];
    static_assert(kCompCap * 2 <= kRoundBytes, "compaction window must fit the stage image");
    static_assert(!DENSE || (EMIT && DBG == 0 && DIALECT <= 1 && !(DIALECT && BATCH)),
                  "the dense emit path exists for emitting launches of the reference dialect (one buffer or a batch) and of another delimiter / quote byte");
    // DENSE: wave 0's look-back window has a place of its own (2 KiB): the waves' images are the WORKGROUP's emit window there
    __shared__ uint4 s_lb[DENSE ? 128 : 1];
#define s_stage_of(wave) (s_stage_a[wave])
#define s_stage_b_of(wave) (DENSE ? s_stage_a[wave] + kImg : s_stage_bb[DENSE ? 0 : (wave)])
#define s_lookback_win() (DENSE ? s_lb : s_stage_b_of(0))
    // escape dialects only (the array does not exist in the other instantiations): the masks of the held tile's LAST
    // round are parked here across the count phase of the next tile — those variants are four VGPRs short there (a third
    // mask and the run-parity chain are in flight), and what hipcc spills otherwise is exactly this pair, to scratch
    // (a batched launch carries the held tile's buffer on top: same squeeze; the dense geometry holds two rounds per wave, not
    // eight, and has the registers)
    constexpr bool kPark = (DIALECT >= 2 || BATCH) && !DENSE;
    __shared__ uint4 s_park[kPark ? kWaves : 1][kPark ? 64 : 1];

    const u32 t = threadIdx.x;
    const u32 lane = t & 63u;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    u32 err = 0;  // wave 0 only

    // sharded re-emit: the true entering state sits in device memory (written by the stitch kernel
    // earlier on this stream); a shard that really is entered outside a string has nothing to redo
    u32 inq_in = args.in_quote_in;  // 0 / 1, or kEnterGuess until the choice of tile 0's workgroup is known
    if (args.state_ptr) {
        // {in_quote_in, ..., reemit}: csvsimd_stitch as the stitch kernel left it earlier on this stream
        const u32 redo = (u32)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_load(args.state_ptr + kStitchReemitWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (redo == 0u) return;
        inq_in = (u32)__builtin_amdgcn_readfirstlane(
                     (int)__hip_atomic_load(args.state_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u;
    }
    if (args.chain)  // chained launch: the state the previous chunk left (written by a launch earlier on this stream)
        inq_in = (u32)__builtin_amdgcn_readfirstlane(
                     (int)__hip_atomic_load(&args.chain->in_quote_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u;
    // A SCALAR load: the word was written by the previous launch's last workgroup (a kernel boundary in between) and is
    // advanced again only when every workgroup of this launch is done.  As a vector load, its first use cost every wave
    // an `s_waitcnt vmcnt(0)` behind barrier A in EVERY iteration (loads and stores share one in-order counter) — on
    // wave 0 that is the acknowledgement of the token-release store, in front of the aggregate's publication: the tiles
    // behind it resolved that much later.  +1.1 % on the 8-GiB shard (profiles/r03_ab_epoch.txt); a launch of one tile
    // per workgroup pays the dependent scalar load once (+0.8 us).
    typedef const __attribute__((address_space(4))) Control* ctl_cptr;
    const u32 epoch_v = ((ctl_cptr)args.ctl)->epoch;

    RoundMasks held[kRounds];
    Desc held_agg = {0, 0, 0}, held_before = {0, 0, 0};
    u32 held_tile = 0, held_wa = 0, held_wb = 0;  // wa / wb: this wave's own entry counts (entered outside / inside)
    u32 held_item = 0;                            // BATCH: the buffer the held tile belongs to
    bool have_held = false;
    u64 wg_tot = 0;  // wave 0: comma/CR/LF bytes in this workgroup's tiles

#ifdef CSVSIMD_DEV_PROBES
    u64 prof[6] = {0, 0, 0, 0, 0, 0};
    u64 stamp = 0;
#define CSVSIMD_STAMP(k)                                              \
    if (DBG & 8) {                                                    \
        const u64 now_ = __builtin_amdgcn_s_memrealtime();            \
        prof[k] += now_ - stamp;                                      \
        stamp = now_;                                                 \
    }
    // DBG bit 5 (with bit 3): wave 0 also leaves absolute stamps per tile in args.prof + 32:
    // [tile * 8 + k], k = 0 ticket drawn, 1 counted, 2 past barrier A, 3 resolved, 4 past barrier B, 5 emitted,
    // 6 = blockIdx | XCC id << 32
#define CSVSIMD_TRACE(k, tid)                                                                          \
    if ((DBG & 32) && w == 0 && lane == 0 && (tid) < args.num_tiles)                                   \
        args.prof[32 + (u64)(tid) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
    // second record per tile, behind the first ones (args.prof + 32 + 8 * num_tiles): stamps of the iteration that RESOLVES
    // the tile — k = 0 loop top, 1 past barrier A, 2 wave 0's speculative scatter done, 3 look-back window landed, 4 the
    // last wave's scatter done (arrives at barrier B), 5 the last wave past barrier B, 6 the last wave's stores issued,
    // 7 wave 0 about to flush
#define CSVSIMD_TRACEX(k, tid, wave, value)                                                            \
    if ((DBG & 32) && w == (u32)(wave) && lane == 0 && (tid) < args.num_tiles)                         \
        args.prof[32 + ((u64)args.num_tiles + (u64)(tid)) * 8 + (k)] = (value);
    if (DBG & 8) stamp = __builtin_amdgcn_s_memrealtime();
#else
#define CSVSIMD_STAMP(k)
#define CSVSIMD_TRACE(k, tid)
#define CSVSIMD_TRACEX(k, tid, wave, value)
#endif

    // the physical CU this workgroup runs on (fixed for its lifetime): XCC id x the cu/sh/se bits of HW_ID
    u32* const my_token = args.cu_token
        ? args.cu_token + ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u) << 8) +
              ((__builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11))) & 0xffu)
        : nullptr;
    bool hold_token = false;
    if (w == 0 && lane == 0) {
        s_owed_n = 0;
        s_owed_head = 0;
        s_redo = 0;
    }

    for (u32 iter = 0;; ++iter) {
#ifdef CSVSIMD_DEV_PROBES
        u64 trace_top = 0, trace_t = 0, trace_a = 0, trace_landed = 0;
        u32 trace_ws = 0;
        if (DBG & 32) trace_top = __builtin_amdgcn_s_memrealtime();
#endif
        if (w == 0 && lane == 0) {  // (not `t == 0`: threadIdx.x itself would have to stay live through the loop)
            if (my_token && args.token_mode == 2) {
                // both atomics in flight together (one round trip instead of two); a workgroup that then has to wait
                // for the token does so holding its ticket — its aggregate is late by at most one count phase
                const u32 tk = atomicAdd(&args.ctl->ticket, 1u);
                u32 got = atomicCAS(my_token, 0u, 1u);
                u32 spins = 0;
                while (got != 0u && ++spins < (1u << 16)) {
                    __builtin_amdgcn_s_sleep(8);
                    got = atomicCAS(my_token, 0u, 1u);
                }
                s_tile = tk;
            } else {
                if (my_token) {
                    // one workgroup per CU in the count phase at a time: taken BEFORE the ticket, so a waiting
                    // workgroup holds no tile anybody could depend on; released after barrier A.  Bounded spin.
                    u32 spins = 0;
                    while (atomicCAS(my_token, 0u, 1u) != 0u && ++spins < (1u << 16)) __builtin_amdgcn_s_sleep(8);
                }
                // an owed tile first, as soon as the shard's entering state has been chosen (see kOwed below); else a ticket
                // (GUESS launches) an owed tile first, as soon as the shard's entering state has been chosen; else a ticket — and
                // if the tickets have run out while tiles are owed, the choice is waited for here: every voter's ticket was
                // drawn long ago, by a workgroup that is running, and this workgroup's own voters have all voted
                bool took = false;
                if (!BATCH && uniform_again<true>(inq_in) == kEnterGuess) {
                    const u32 head = s_owed_head;
                    if (head != s_owed_n) {
                        u32 g = __hip_atomic_load(&args.ctl->guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        u32 tk = 0;
                        if (g == 0u) {
                            tk = atomicAdd(&args.ctl->ticket, 1u);
                            if (tk >= args.num_tiles) {
                                for (u32 spins = 0; g == 0u && spins < kSpinLimit; ++spins) {
                                    __builtin_amdgcn_s_sleep(16);
                                    g = __hip_atomic_load(&args.ctl->guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                if (g == 0u) atomicOr(&args.ctl->err, 1u);  // (the spin bound: the launch ends with its error flag set)
                                g = 2u;
                            }
                        }
                        if (g != 0u) {
                            tk = s_owed[head];
                            s_owed_head = head + 1u;
                        }
                        s_redo = g != 0u ? 1u : 0u;
                        s_tile = tk;
                        took = true;
                    }
                }
                if (!took)
                s_tile = (DBG & 2) ? blockIdx.x + iter * gridDim.x : atomicAdd(&args.ctl->ticket, 1u);
            }
        }
        hold_token = my_token != nullptr;
        wg_barrier();  // barrier T
#ifdef CSVSIMD_DEV_PROBES
        if (DBG & 32) trace_t = __builtin_amdgcn_s_memrealtime();
#endif
        CSVSIMD_STAMP(0)
        const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)s_tile);
        CSVSIMD_TRACE(0, tile)
        const bool have_cur = tile < args.num_tiles;
#ifdef CSVSIMD_WG_END_TRACE
        if (have_cur) { ++wg_tiles; wg_tlast = __builtin_amdgcn_s_memrealtime(); }
#endif
        if (hold_token && !have_cur) {  // nothing to count: the partner workgroup need not wait for this one
            if (w == 0 && lane == 0) __hip_atomic_store(my_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hold_token = false;
        }
        if (!have_cur && !have_held) break;

        RoundMasks m[kRounds];
        Desc agg = {0, 0, 0}, before = {0, 0, 0};
        u32 cur_wa = 0, cur_wb = 0;
        u32 cur_item = 0;
        if (have_cur) {
            // the buffer this tile reads: the launch's one buffer, or (BATCH) the one the tile index falls into
            const uint8_t* t_abase = args.abase;
            u64 t_lo = args.lo, t_hi = args.hi;
            u32 t_first = 0;
            if (BATCH) {
                cur_item = batch_item_of(args.batch_first, args.n_items, tile);
                const batch_cptr it = (batch_cptr)args.batch + cur_item;
                t_abase = it->abase;
                t_lo = it->lo;
                t_hi = it->hi;
                t_first = it->first_tile;
            }
            const u64 tile0 = (u64)(tile - t_first) * kTileBytes;  // relative to abase
            // descriptor over this tile's valid bytes, rounded up to whole 16-byte chunks (a chunk
            // never straddles a page, so the <= 15 extra bytes are always mapped)
            const u64 hi16 = (t_hi + 15) & ~15ull;
            const u64 avail = hi16 - tile0;
            const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(t_abase) + tile0, 0,
                (int)(avail < (u64)kTileBytes ? avail : (u64)kTileBytes), 0x00020000);
            // valid bytes of this tile are [lo_rel, hi_rel) relative to the tile start
            const u32 lo_rel = t_lo > tile0 ? (u32)(t_lo - tile0) : 0u;  // lo < 16
            const u32 hi_rel = t_hi - tile0 < (u64)kTileBytes ? (u32)(t_hi - tile0) : (u32)kTileBytes;

            // ---- count phase: masks for the whole span stay in registers ---------------------
            u32 carry = 0, cnt_a = 0, cnt_t = 0;
            const EdgeKeep ek = edge_keep_of_tile(lane, w, lo_rel, hi_rel, escape_in_of<DIALECT>(args));
            DialectRegs dr = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            u32 esc_carry = 0;
            if (DIALECT == 1 || DIALECT == 2) {
                dr.delim = args.delim * 0x01010101u;
                dr.quote = args.quote * 0x01010101u;
                dr.esc = args.escape * 0x01010101u;
                dr.qmask = args.quote ? 0x80808080u : 0u;
            }
            if (DIALECT == 3) {
                dr.sh1 = args.hash_sh1;
                dr.sh2 = args.hash_sh2;
                dr.lut_lo = args.hash_lut_lo;
                dr.lut_hi = args.hash_lut_hi;
                dr.cls_lo = args.hash_cls_lo;
                dr.cls_hi = args.hash_cls_hi;
            }
            if (DIALECT >= 2) {
                // escape state entering this wave span: parity of the escape run that ends right
                // before it (one 64-byte peek; a misaligned shard start is handled by ek.front_esc)
                const u64 span0 = tile0 + (u64)w * kSpanBytes;
                if (span0 > args.lo || lo_rel == 0u)
                    esc_carry = (u32)__builtin_amdgcn_readfirstlane(
                        (int)escape_run_parity(args.abase, args.lo, args.hi, span0, args.escape, escape_in_of<DIALECT>(args), lane));
            }
            if (DBG & 1) {
                uint4 v[kRows];
                u32 acc = 0;
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    load_round(rsrc, w * (u32)kSpanBytes + lane * 16u + (u32)r * kRoundBytes, v);
#pragma unroll
                    for (int j = 0; j < kRows; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
                    m[r].st = 0;
                    m[r].s = 0;
                }
                cnt_a = acc & 1u;
            } else {
                // pacing (NOTEBOOK.md "Pacing"): the phase that keeps HBM loads in flight gets the SIMD's issue
                // priority over the partner workgroup's resolve / emit phase
                if (uniform_again<(DIALECT >= 2 || BATCH)>(args.count_prio)) __builtin_amdgcn_s_setprio(3);
                count_phase<DIALECT>(rsrc, lane, w, ek, s_stage_of(w), s_stage_b_of(w), m, carry, cnt_a, cnt_t, dr,
                                     esc_carry);
                if (uniform_again<(DIALECT >= 2 || BATCH)>(args.count_prio)) __builtin_amdgcn_s_setprio(0);
            }
            const u32 wave_a = wave_sum(cnt_a);
            const u32 wave_t = wave_sum(cnt_t);
            cur_wa = wave_a;
            cur_wb = wave_t - wave_a;
            if (lane == 0) {
                s_wdesc[w][0] = carry;
                s_wdesc[w][1] = wave_a;
                s_wdesc[w][2] = wave_t - wave_a;
            }
        }
        CSVSIMD_STAMP(1)  // count phase
        CSVSIMD_TRACE(1, tile)
        wg_barrier();     // barrier A
        if (hold_token && w == 0 && lane == 0)
            __hip_atomic_store(my_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CSVSIMD_DEV_PROBES
        if (DBG & 32) trace_a = __builtin_amdgcn_s_memrealtime();
#endif
        CSVSIMD_STAMP(2)
        CSVSIMD_TRACE(2, tile)

        const u32 epoch = epoch_v & kEpochMask;
        if (have_cur) {
            // ---- tile aggregate; this wave's entering state/offset relative to the tile ------
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                Desc d = {s_wdesc[k][0], s_wdesc[k][1], s_wdesc[k][2]};
                if ((u32)k == w) before = agg;
                agg = compose(agg, d);
            }
        }

        // Wave 0 publishes this tile's aggregate and REQUESTS the look-back window of the held tile; then every wave
        // scatters its span of the held tile speculatively (see scatter_span_spec) — the 3-5 us the polls take behind the
        // CU's streaming loads used to be seven idle waves at barrier B — and only then does wave 0 consume the window.
        if (w == 0) {
            // (GUESS launches) an owed tile was counted, published and voted for before: only its masks were needed again
            const bool redo = !BATCH && uniform_again<true>(inq_in) == kEnterGuess &&
                              (u32)__builtin_amdgcn_readfirstlane((int)s_redo) != 0u;
            bool defer = false;
            if (have_cur && !redo) {
                if (!(DBG & 4) && lane == 0) publish_aggregate(args.desc, tile, epoch, agg);
                wg_tot += (u64)(u32)__builtin_amdgcn_readfirstlane((int)(agg.a + agg.b));
                if (BATCH && lane == 0)  // the buffer's own comma/CR/LF total, for its result record
                    atomicAdd((unsigned long long*)&args.batch_tot[cur_item], (unsigned long long)(agg.a + agg.b));
            }
            // a shard whose entering state nobody knows: the first kGuessTiles tiles vote (see guess_vote)
            if (uniform_again<(DIALECT >= 2 || BATCH)>(inq_in) == kEnterGuess && have_cur && !redo && tile < kGuessTiles && !(DBG & 4)) {
                guess_vote(args.desc, args.ctl, args.num_tiles, epoch, lane, err);
                // This workgroup now holds two counted tiles and the held one cannot be resolved before the choice exists.
                // Waiting for it here assumes that the voters still missing are being counted by OTHER resident workgroups
                // (rounds 2-4: a GUESS launch needed 4 — dense geometry: 16 — of its workgroups resident at once, and a launch
                // squeezed in beside other contexts' grids ended in the spin bound).  Instead: the HELD tile is OWED — its
                // aggregate is out, its vote is in, its masks are dropped (the tile just counted takes its place as usual) —
                // and the workgroup draws the next ticket, i.e. the next voter: a grid of ONE workgroup completes the vote by
                // itself.  Tiles behind the voters wait for the choice as before: every voter ticket was drawn before
                // theirs, by a running workgroup.
                if (have_held &&
                    (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&args.ctl->guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                    defer = true;
                    if (lane == 0) {
                        const u32 k = s_owed_n;
                        s_owed[k] = held_tile;
                        s_owed_n = k + 1u;
                    }
                }
            }
            // (GUESS launches) what this iteration does with the held tile, for every wave behind barrier B — and for this
            // wave's own resolve below
            if (!BATCH && uniform_again<true>(inq_in) == kEnterGuess && lane == 0) s_pin = defer ? kPinDeferred : 0u;
            // into wave 0's second stage image: idle until the next count phase
            if (have_held && !defer && !(DBG & 4)) lookback_issue(args.desc, held_tile, lane, s_lookback_win());
            if (redo && lane == 0) s_redo = 0;
        }
        bool spec_done = false;
        const u32 spec_pin = held_agg.b > held_agg.a ? 1u : 0u;      // the guess: the hypothesis with more entries
        const u32 spec_state = spec_pin ^ held_before.p;             // this wave's entering state under it
        const u32 spec_n = spec_state ? held_wb : held_wa;           // ... and its entry count
        const u32 spec_tile_n = spec_pin ? held_agg.b : held_agg.a;  // the whole tile's entries under the guess
        if (DENSE) {
            if (EMIT && have_held && spec_tile_n <= kWgCap) {
                scatter_span_tile(held, lane, spec_state, reinterpret_cast<unsigned short*>(s_stage_a[0]),
                                  spec_pin ? held_before.b : held_before.a, w * (u32)kSpanBytes);
                spec_done = true;
            }
        } else if (EMIT && have_held && !(DBG & 16) && spec_n <= (u32)kCompCap) {
            if (kPark) {
                const uint4 pk = s_park[w][lane];
                held[kRounds - 1].st = ((u64)pk.y << 32) | pk.x;
                held[kRounds - 1].s = ((u64)pk.w << 32) | pk.z;
            }
            scatter_span_spec(held, lane, spec_state, reinterpret_cast<unsigned short*>(s_stage_of(w)));
            spec_done = true;
        }
        if (have_held) {
            CSVSIMD_TRACEX(2, held_tile, 0, __builtin_amdgcn_s_memrealtime())
            CSVSIMD_TRACEX(4, held_tile, kWaves - 1, __builtin_amdgcn_s_memrealtime())
        }
        if (w == 0 && have_held &&
            !(!BATCH && uniform_again<true>(inq_in) == kEnterGuess &&
              (u32)__builtin_amdgcn_readfirstlane((int)s_pin) == kPinDeferred)) {
            u32 pin = 0;
            u64 base = 0;
            if (!(DBG & 4)) {
                u64 pre[4];
                // CSVSIMD_ENTER_GUESS: the choice was published before tile 0's aggregate, a tile-time ago at least
                const u32 inq_now = uniform_again<(DIALECT >= 2 || BATCH)>(inq_in);
                u32 inq_eff = inq_now == kEnterGuess ? wait_for_guess(args.ctl, err) : inq_now;
                u32 held_first = 0;
                if (BATCH) {
                    const batch_cptr it = (batch_cptr)args.batch + held_item;
                    inq_eff = it->in_quote_in & 1u;
                    held_first = it->first_tile;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CSVSIMD_DEV_PROBES
                if (DBG & 32) trace_landed = __builtin_amdgcn_s_memrealtime();
                lookback_fetch(s_lookback_win(), held_tile, lane, pre);
                resolve<true>(args.desc, held_tile, epoch, held_agg, inq_eff, lane, pin, base, err, pre,
                              (DBG & 32) ? &trace_ws : nullptr, held_first);
#else
                lookback_fetch(s_lookback_win(), held_tile, lane, pre);
                resolve<true>(args.desc, held_tile, epoch, held_agg, inq_eff, lane, pin, base, err, pre, nullptr, held_first);
#endif
            }
            if (lane == 0) {
                s_pin = pin;
                s_base = base;
            }
        }
        CSVSIMD_STAMP(3)  // publish + resolve (wave 0)
        if (have_held) { CSVSIMD_TRACE(3, held_tile) }
#ifdef CSVSIMD_DEV_PROBES
        if ((DBG & 32) && w == 0 && lane == 0 && have_held && held_tile < args.num_tiles) {
            // the iteration that resolved the held tile, in 10-ns ticks: upper half of slot 6 = loop top -> barrier T
            // (token + ticket) | barrier T -> barrier A (the count phase of the OTHER tile, if any); upper half of slot 7 =
            // barrier A -> look-back window landed (12 bits) | windows walked (4 bits) | spins (16 bits)
            u32* const s6 = reinterpret_cast<u32*>(&args.prof[32 + (u64)held_tile * 8 + 6]);
            u32* const s7 = reinterpret_cast<u32*>(&args.prof[32 + (u64)held_tile * 8 + 7]);
            const u32 d_top = (u32)(trace_t - trace_top), d_cnt = (u32)(trace_a - trace_t), d_land = (u32)(trace_landed - trace_a);
            CSVSIMD_TRACEX(0, held_tile, 0, trace_top)
            CSVSIMD_TRACEX(1, held_tile, 0, trace_a)
            CSVSIMD_TRACEX(3, held_tile, 0, trace_landed)
            s6[1] = (d_top > 0xffffu ? 0xffffu : d_top) | ((d_cnt > 0xffffu ? 0xffffu : d_cnt) << 16);
            s7[1] = ((d_land > 0xfffu ? 0xfffu : d_land) << 20) | ((trace_ws >> 16 > 15u ? 15u : trace_ws >> 16) << 16) | (trace_ws & 0xffffu);
        }
#endif
        wg_barrier_lds();     // barrier B
        CSVSIMD_STAMP(4)
        if (have_held) { CSVSIMD_TRACE(4, held_tile) }
        if (have_held) { CSVSIMD_TRACEX(5, held_tile, kWaves - 1, __builtin_amdgcn_s_memrealtime()) }
        // (s_pin == kPinDeferred, GUESS launches only: the held tile is owed — nothing was resolved, nothing is emitted, the tile
        // counted in this iteration becomes the held one as always)
        if (EMIT && have_held && (u32)__builtin_amdgcn_readfirstlane((int)s_pin) != kPinDeferred) {
            // pacing knob, 0 by default since the per-CU token: a pause between barrier B and the flush
            for (u32 z = 0; z < args.emit_delay; ++z) __builtin_amdgcn_s_sleep(10);
            const u32 pin = s_pin;
            // state entering this wave's span and tape index of its first entry
            const u32 wstate = pin ^ held_before.p;
            const u64 run = s_base + (pin ? held_before.b : held_before.a);
            // where the held tile's entries go, and the tape value of this span's byte 0
            u64* e_tape = args.tape;
            u64 e_cap = args.tape_cap, e_off = args.base_off - args.lo;
            u32 e_first = 0;
            if (BATCH) {
                const batch_cptr it = (batch_cptr)args.batch + held_item;
                e_tape = it->tape;
                e_cap = it->tape ? it->tape_cap : 0;
                e_off = it->base_off - it->lo;
                e_first = it->first_tile;
            }
            const u64 span0 = (u64)(held_tile - e_first) * kTileBytes + (u64)w * kSpanBytes;
            CSVSIMD_TRACEX(7, held_tile, 0, __builtin_amdgcn_s_memrealtime())
            if (DENSE && spec_done && pin == (held_agg.b > held_agg.a ? 1u : 0u)) {
                // the guess was right: the workgroup's window holds the TILE's entries in order (barrier B made every wave's
                // scatter visible); all threads write them out front to back
                const u64 sb = s_base;
                const u64 base_u = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(sb >> 32)) << 32) |
                                   (u32)__builtin_amdgcn_readfirstlane((int)(u32)sb);
                flush_tile_dense(e_tape, e_cap, reinterpret_cast<const unsigned short*>(s_stage_a[0]), spec_tile_n, base_u,
                                 e_off + (u64)(held_tile - e_first) * kTileBytes, w * 64u + lane);
            } else if (spec_done && pin == (held_agg.b > held_agg.a ? 1u : 0u)) {
                // the guess was right: the window already holds the span's entries, only the stores are left
                wave_lds_fence();
                const u64 run_u = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(run >> 32)) << 32) |
                                  (u32)__builtin_amdgcn_readfirstlane((int)(u32)run);
                flush_window(e_tape, e_cap, reinterpret_cast<unsigned short*>(s_stage_of(w)), spec_n, run_u, e_off + span0, lane);
                wave_lds_fence();
            } else if (DENSE) {
                // (the guess was wrong, or the tile holds more entries than the workgroup's window: wave by wave)
                emit_span_dense(e_tape, e_cap, held, lane, e_off + span0, wstate, run, reinterpret_cast<unsigned short*>(s_stage_of(w)));
            } else {
                if (kPark) {  // (the speculative scatter may not have run: the pair is fetched again)
                    const uint4 pk = s_park[w][lane];
                    held[kRounds - 1].st = ((u64)pk.y << 32) | pk.x;
                    held[kRounds - 1].s = ((u64)pk.w << 32) | pk.z;
                }
                emit_span<(DBG & 16) != 0>(e_tape, e_cap, held, lane, e_off + span0, wstate, run,
                                           reinterpret_cast<unsigned short*>(s_stage_of(w)));
            }
        }
        if (have_held) { CSVSIMD_TRACE(5, held_tile) }
        if (have_held) { CSVSIMD_TRACEX(6, held_tile, kWaves - 1, __builtin_amdgcn_s_memrealtime()) }
#ifdef CSVSIMD_DEV_PROBES
        if ((DBG & 32) && w == 0 && lane == 0 && have_cur) {
            // lower halves only: the upper halves belong to the iteration that resolves this tile (above)
            reinterpret_cast<u32*>(&args.prof[32 + (u64)tile * 8 + 6])[0] =
                blockIdx.x | ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu) << 16);
            reinterpret_cast<u32*>(&args.prof[32 + (u64)tile * 8 + 7])[0] =
                (u32)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID
        }
#endif
        // the tile counted in this iteration becomes the held one
        have_held = have_cur;
        held_tile = tile;
        held_item = cur_item;
        held_agg = agg;
        held_before = before;
        held_wa = cur_wa;
        held_wb = cur_wb;
#pragma unroll
        for (int r = 0; r < kRounds - (kPark ? 1 : 0); ++r) held[r] = m[r];
        if (kPark)
            s_park[w][lane] = make_uint4((u32)m[kRounds - 1].st, (u32)(m[kRounds - 1].st >> 32), (u32)m[kRounds - 1].s,
                                         (u32)(m[kRounds - 1].s >> 32));
        CSVSIMD_STAMP(5)  // emit
        // the tickets are exhausted for good and nothing is held any more: leave without another trip through the token
        // and the ticket counter.  Measured and rejected around this hand-off (ab_variants, 8 GiB / 1 GiB):
        //  * reading the ticket counter next to the token CAS, so that a workgroup with nothing left to count never
        //    waits for its partner's count phase: the extra load sits on the count phases' critical path, -35 %;
        //  * drawing the NEXT ticket right after barrier A, so that only the token is left to wait for at the loop
        //    top: -0.5 ... -2 % (a shorter hand-off does not help: the steady state is bound by HBM, not by this chain);
        //  * resolving and emitting a tile that is among the last gridDim.x of the shard in the iteration that counted
        //    it (un-lagged, right behind the held one) so that the launch drains in two back-to-back resolves: 0 ... -0.5 %
        //    (the un-lagged look-back waits for the predecessors that are still counting), and 2 VGPR spills;
        //  * no token for a workgroup's FIRST tile (the two workgroups of a CU count their first tiles at once instead
        //    of the second one sitting out the first one's count phase): 0 ... -2 %, worst on the shortest launches.
        if (!have_cur) break;
    }
#ifdef CSVSIMD_DEV_PROBES
    if ((DBG & 8) && lane == 0 && w < 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            atomicAdd((unsigned long long*)(args.prof + w * 8 + k), (unsigned long long)prof[k]);
        if (w == 0) atomicAdd((unsigned long long*)(args.prof + 16), 1ull);
    }
#endif
#undef CSVSIMD_STAMP
#undef CSVSIMD_TRACE
#undef CSVSIMD_TRACEX

#undef s_stage_of
#undef s_stage_b_of
#undef s_lookback_win
    // ---- this workgroup is done; the last one to get here completes the launch --------------------
    if (w != 0) return;
#ifdef CSVSIMD_WG_END_TRACE
    if (blockIdx.x < 2048 && (threadIdx.x & 63u) == 0) {
        g_wg_trace[blockIdx.x][0] = wg_t0;
        g_wg_trace[blockIdx.x][1] = wg_tlast;
        g_wg_trace[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
        g_wg_trace[blockIdx.x][3] = wg_tiles;
    }
#endif
    finish_launch<DIALECT, (DBG & 4) != 0, BATCH>(args, epoch_v & kEpochMask, inq_in,
                                           wg_tot, err,
                                           // the lane id again, from the execution mask: `lane` as derived from threadIdx.x
                                           // would otherwise have to survive the tile loop in a register the loop needs
                                           __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

#if defined(CSVSIMD_WG_END_TRACE) && !defined(CSVSIMD_DENSE_TU)
}  // namespace csvsimd
extern "C" int csvsimd_dev_wg_trace(uint64_t* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(csvsimd::g_wg_trace), sizeof(uint64_t) * 2048 * 4);
}
namespace csvsimd {
#endif
// KernelArgs of one launch over L's buffer (everything but the dialect's hashed tables)
static void fill_kernel_args(const Stage1Launch& L, KernelArgs& a) {
    const uintptr_t addr = (uintptr_t)L.dbuf;
    a.abase = (const uint8_t*)(addr & ~(uintptr_t)15);
    a.lo = (u64)(addr & 15);
    a.hi = a.lo + L.len;
    a.base_off = L.base_off;
    a.in_quote_in = L.in_quote_in <= CSVSIMD_ENTER_GUESS ? L.in_quote_in : 1u;  // 0, 1 or CSVSIMD_ENTER_GUESS
    a.num_tiles = (u32)((a.hi + kTileBytes - 1) / kTileBytes);
    if (L.len == 0) a.num_tiles = 0;
    a.tape = (u64*)L.dtape;
    a.tape_cap = L.dtape ? L.tape_cap : 0;
    a.desc = L.scratch_desc;
    a.ctl = reinterpret_cast<Control*>(L.scratch_base);
    a.result = L.d_result;
    a.state_ptr = L.d_state;
    a.chain = L.d_chain;
    a.delim = L.delimiter;
    a.quote = L.quote;
    a.escape = L.escape;
    a.escape_in = (L.escape && L.escape_in) ? 1u : 0u;
    // Pacing: the two workgroups of a CU take turns in the count phase (a token per physical CU in the scratch block,
    // taken by thread 0 before it draws the ticket, released after barrier A), and count phases run at s_setprio 3.
    // Measured on MI355X (scripts/sweep_token.sh, probe build; kernel ms: 8 GiB / 2 GiB 64x31, 1 GiB 16x32, 1 GiB dense):
    //   no token, no pause            1.848  -      0.2506  0.528   both workgroups often count at once: they share the
    //   no token, ~2 us pause before  1.739  0.463  0.2504  0.530   SIMDs' issue slots and double the loads in flight — the
    //     emit (interim default)                                    bare stream, too, is fastest at 8 waves per CU
    //   token + priority              1.681  0.439  0.2315  0.522   <- default: 63.9 / 61.1 / 58.0 / 25.7 % of 8 TB/s
    //   token + priority + pause      1.691  0.442  0.2332  0.518
    //   token, both atomics at once   1.789  0.463  0.2413  0.526   (a workgroup may then wait for the token holding a ticket)
    // (taken before the speculative scatter went in; with it the default row reads 1.645 / 0.427 / 0.2235 / 0.523 =
    // 65.3 / 62.9 / 60.0 / 25.6 %, and the ordering of the rows is unchanged.)
    // The pause knob stays (0 by default); the probe build's environment hooks override all three per launch.
    a.emit_delay = L.pace_emit_delay >= 0 ? (u32)L.pace_emit_delay : 0u;
    a.count_prio = L.pace_count_prio >= 0 ? (u32)L.pace_count_prio : 1u;
    const int token_mode = L.pace_cu_token >= 0 ? L.pace_cu_token : 1;
    a.cu_token = token_mode > 0
        ? reinterpret_cast<u32*>(reinterpret_cast<char*>(L.scratch_base) + CSVSIMD_SCRATCH_TOKEN_OFFSET) : nullptr;
    a.token_mode = (u32)token_mode;
#ifdef CSVSIMD_DEV_PROBES
    a.prof = L.scratch_prof;
#endif
    a.hash_sh1 = a.hash_sh2 = a.hash_lut_lo = a.hash_lut_hi = a.hash_cls_lo = a.hash_cls_hi = 0;
}

#ifdef CSVSIMD_DENSE_TU
// The dense geometry's launchers: an emitting launch of the reference dialect or of another delimiter / quote byte (no
// escape byte), one buffer or — reference dialect — a batch (launch_stage1 / launch_stage1_batch of the other compilation
// hand over).  ONE instantiation per translation unit (CSVSIMD_DENSE_WHICH: stage1_dense.hip, stage1_dense_d1.hip,
// stage1_dense_batch.hip): compiled next to its siblings the default dense kernel came out with other register allocation
// (55 -> 91 SGPRs spilled to VGPR lanes), alone it is the kernel round 4 measured.
#if CSVSIMD_DENSE_WHICH == 0
hipError_t launch_stage1_dense(const Stage1Launch& L, hipStream_t stream) {
    if (L.delimiter != ',' || L.quote != '"') return launch_stage1_dense_d1(L, stream);
    KernelArgs a;
    fill_kernel_args(L, a);
    const u32 want = a.num_tiles ? a.num_tiles : 1u;
    const u32 grid = want < L.max_blocks ? want : L.max_blocks;
    hipError_t e;
    if (L.ev_begin && (e = hipEventRecord(L.ev_begin, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL((stage1_kernel<true, 0, 0, false, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (L.ev_end && (e = hipEventRecord(L.ev_end, stream)) != hipSuccess) return e;
    return hipSuccess;
}
#elif CSVSIMD_DENSE_WHICH == 1
hipError_t launch_stage1_dense_d1(const Stage1Launch& L, hipStream_t stream) {
    KernelArgs a;
    fill_kernel_args(L, a);
    const u32 want = a.num_tiles ? a.num_tiles : 1u;
    const u32 grid = want < L.max_blocks ? want : L.max_blocks;
    hipError_t e;
    if (L.ev_begin && (e = hipEventRecord(L.ev_begin, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL((stage1_kernel<true, 0, 1, false, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (L.ev_end && (e = hipEventRecord(L.ev_end, stream)) != hipSuccess) return e;
    return hipSuccess;
}
#else
// (first_tile of every buffer counts tiles of THIS geometry: CSVSIMD_MIN_TILE_BYTES)
hipError_t launch_stage1_batch_dense(void* d_items, void* d_first_tiles, void* d_tots, u32 n_items, u32 total_tiles,
                                     csvsimd_shard_result* d_results, void* scratch_base, u64* scratch_desc, u32 max_blocks,
                                     hipStream_t stream) {
    static_assert(kTileBytes == CSVSIMD_MIN_TILE_BYTES, "the host lays a dense batch out in tiles of this size");
    KernelArgs a = {};
    a.num_tiles = total_tiles;
    a.desc = scratch_desc;
    a.ctl = reinterpret_cast<Control*>(scratch_base);
    a.result = d_results;
    a.batch = reinterpret_cast<const BatchItem*>(d_items);
    a.batch_first = reinterpret_cast<const u32*>(d_first_tiles);
    a.batch_tot = reinterpret_cast<u64*>(d_tots);
    a.n_items = n_items;
    a.count_prio = 1u;
    a.cu_token = reinterpret_cast<u32*>(reinterpret_cast<char*>(scratch_base) + CSVSIMD_SCRATCH_TOKEN_OFFSET);
    a.token_mode = 1u;
    const u32 want = total_tiles ? total_tiles : 1u;
    const u32 grid = want < max_blocks ? want : max_blocks;
    hipLaunchKernelGGL((stage1_kernel<true, 0, 0, true, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}
#endif
}  // namespace csvsimd_dense
#else

// ---------------------------------------------------------------------------------------------
// utilities: synthetic corpus, checksum, self-test
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ u32 synth_byte(u64 pos, u32 cols, u32 width, u64 seed, u32 quote_pct) {
    const u64 row_bytes = (u64)cols * (width + 1);
    const u64 r = pos / row_bytes;
    const u32 within = (u32)(pos - r * row_bytes);
    const u32 c = within / (width + 1);
    const u32 k = within - c * (width + 1);
    if (k == width) return c == cols - 1 ? '\n' : ',';
    const u64 key = seed ^ (r << 20) ^ ((u64)c << 8);
    if (quote_pct && r > 0 && width >= 22 && splitmix64(key ^ 0xFF) % 100 < quote_pct) {
        if (k == 0 || k == width - 1) return '"';
        if (k == 8) return ',';
        if (k == 20) return '\n';
    }
    const u32 a = (u32)(splitmix64(key ^ k) % 36);
    return a < 26 ? 'a' + a : '0' + (a - 26);
}

__global__ void synth_kernel(uint8_t* dst, u64 file_off, u64 len, u32 cols, u32 width, u64 seed,
                             u32 quote_pct) {
    // 4 bytes per thread; dst is at least 4-byte aligned (checked on the host)
    const u64 n4 = (len + 3) / 4;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = i * 4;
        if (p + 4 <= len) {
            u32 wv = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) wv |= synth_byte(file_off + p + b, cols, width, seed, quote_pct) << (8 * b);
            *reinterpret_cast<u32*>(dst + p) = wv;
        } else {
            for (u64 q = p; q < len; ++q) dst[q] = (uint8_t)synth_byte(file_off + q, cols, width, seed, quote_pct);
        }
    }
}

__global__ void checksum_kernel(const u64* tape, u64 n, u64 first_index, u64* out) {
    u64 a = 0, b = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 gi = first_index + i;
        const u64 e = tape[i];
        a += splitmix64(e ^ (gi * 0x9E3779B97F4A7C15ull));
        b += e * (2 * gi + 1);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += ((u64)(u32)__shfl_xor((int)(u32)(a >> 32), d) << 32) | (u32)__shfl_xor((int)(u32)a, d);
        b += ((u64)(u32)__shfl_xor((int)(u32)(b >> 32), d) << 32) | (u32)__shfl_xor((int)(u32)b, d);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd((unsigned long long*)&out[0], (unsigned long long)a);
        atomicAdd((unsigned long long*)&out[1], (unsigned long long)b);
    }
}

// HBM streaming probe with the stage-1 traffic shape and none of its work: the achievable ceiling
// the roofline fraction can be read against (bench.py reports it next to the 8 TB/s spec peak).
// 128-KiB tiles from an atomic ticket, 4 waves x 8 rounds x 4 KiB, non-temporal both ways, line-aligned 1-KiB
// wave stores.  WR16 = bytes written per 16 bytes read: 0 = read only, 4 = the 64x31 corpus (8 B of tape per
// 32 B), 25 = the dense corpus (1024 x 4: 8 B of tape per 5 B read = 25.6 / 16; 25 keeps the output inside a
// tape-sized buffer).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"  // the WR16 == 0 instantiation has nothing to unroll in its store loop
template <int WR16>
__global__ __launch_bounds__(256) void hbm_probe_kernel(const uint8_t* __restrict__ in, uint4* __restrict__ out,
                                                        Control* ctl, u32 num_tiles) {
    __shared__ u32 s_tile;
    const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (;;) {
        if (t == 0) s_tile = atomicAdd(&ctl->probe_ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        __syncthreads();
        if (tile >= num_tiles) break;
        const u64 tile0 = (u64)tile * 131072;
        const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in) + tile0, 0, 131072, 0x00020000);
        constexpr u32 kOutKiBPerSpan = WR16 * 2;  // this wave's output per 32-KiB span, in 1-KiB wave stores
        uint4* const obase = out + ((u64)tile * 4 + w) * (kOutKiBPerSpan * 64);
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += 2) {
            uint4 v[2][4];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto x = __builtin_amdgcn_raw_buffer_load_b128(
                        rsrc, (int)(w * 32768u + (u32)(r0 + d) * 4096u + (u32)j * 1024u + lane * 16u), 0, kLoadAux);
                    v[d][j] = make_uint4(x[0], x[1], x[2], x[3]);
                }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                uint4 o;
                o.x = v[d][0].x ^ v[d][1].x ^ v[d][2].x ^ v[d][3].x;
                o.y = v[d][0].y ^ v[d][1].y ^ v[d][2].y ^ v[d][3].y;
                o.z = v[d][0].z ^ v[d][1].z ^ v[d][2].z ^ v[d][3].z;
                o.w = v[d][0].w ^ v[d][1].w ^ v[d][2].w ^ v[d][3].w;
                if (WR16 == 0) {
                    acc.x ^= o.x; acc.y ^= o.y; acc.z ^= o.z; acc.w ^= o.w;
                } else {
                    // 1-KiB stores due after this round: [from, upto)
                    const u32 from = (u32)(r0 + d) * kOutKiBPerSpan / 8, upto = (u32)(r0 + d + 1) * kOutKiBPerSpan / 8;
                    for (u32 q = from; q < upto; ++q) {
                        const u32x4 x = {o.x + q, o.y, o.z, o.w};
                        __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(obase + q * 64 + lane));
                    }
                }
            }
        }
    }
    if (WR16 == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = acc;  // keeps the loads alive
    // the last workgroup out leaves the ticket pair ready for the next launch (as stage1_kernel does)
    if (t == 0 && atomicAdd(&ctl->probe_done, 1u) == gridDim.x - 1u) {
        ctl->probe_ticket = 0;
        ctl->probe_done = 0;
    }
}

#pragma clang diagnostic pop

// Independent yardstick for the probe above: the textbook copy — one 16-byte element per thread, a
// grid as large as the buffer, no tickets, no tiles — the shape behind the guide's "6.29 TB/s measured (float4 copy)"
// (MI355X_MICROARCH.md:36).  NT: non-temporal both ways, as the stage-1 kernel's loads and stores are.
template <bool NT>
__global__ __launch_bounds__(256) void copy_probe_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, u64 n16) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= n16) return;
    if (NT)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
    else
        out[i] = in[i];
}

// self-test of the wavefront primitives against plain loops (one wave); out[0] = failure bits
__global__ void selftest_kernel(u32* out) {
    __shared__ u32 s_v[64];
    __shared__ u32 s_d[64][3];
    const u32 lane = threadIdx.x;
    u32 fail = 0;
    if (mbcnt64(~0ull) != lane) fail |= 1;
    // scan
    const u32 v = (lane * 2654435761u >> 20) & 0x03ff03ffu;
    s_v[lane] = v;
    __syncthreads();
    u32 ref = 0;
    for (u32 k = 0; k <= lane; ++k) ref += s_v[k];
    if (wave_incl_scan_add(v) != ref) fail |= 2;
    u32 tot = 0;
    for (u32 k = 0; k < 64; ++k) tot += s_v[k];
    if (wave_sum(v) != tot) fail |= 4;
    // mbcnt
    const u64 mask = __ballot((v >> 3) & 1u);
    u32 below = 0;
    for (u32 k = 0; k < lane; ++k) below += (s_v[k] >> 3) & 1u;
    if (mbcnt64(mask) != below) fail |= 8;
    // ordered composition over lanes [0, m)
    Desc f = {(v >> 5) & 1u, v & 0xffu, (v >> 16) & 0xffu};
    s_d[lane][0] = f.p; s_d[lane][1] = f.a; s_d[lane][2] = f.b;
    __syncthreads();
    for (u32 m = 0; m <= 64; m += 7) {
        const Desc F = wave_compose_ordered(f, lane, m);
        Desc R = {0, 0, 0};
        for (int k = (int)m - 1; k >= 0; --k) {  // earliest (largest lane) first
            Desc d = {s_d[k][0], s_d[k][1], s_d[k][2]};
            R = compose(R, d);
        }
        if (lane == 0 && (F.p != R.p || F.a != R.a || F.b != R.b)) fail |= 16;
    }
    // classification of all 256 byte values, 16 at a time
    for (u32 base = 0; base < 256; base += 16) {
        uint4 x;
        x.x = (base + 0) | ((base + 1) << 8) | ((base + 2) << 16) | ((base + 3) << 24);
        x.y = (base + 4) | ((base + 5) << 8) | ((base + 6) << 16) | ((base + 7) << 24);
        x.z = (base + 8) | ((base + 9) << 8) | ((base + 10) << 16) | ((base + 11) << 24);
        x.w = (base + 12) | ((base + 13) << 8) | ((base + 14) << 16) | ((base + 15) << 24);
        u32 st, q;
        classify16(x, st, q);
        u32 est = 0, eq = 0;
        for (u32 b = 0; b < 16; ++b) {
            const u32 c = base + b;
            if (c == 0x2c || c == 0x0a || c == 0x0d) est |= 1u << b;
            if (c == 0x22) eq |= 1u << b;
        }
        if (st != est || q != eq) fail |= 32;
    }
    const u64 any = __ballot(fail != 0);
    if (lane == 0) out[0] = any ? (fail | 0x80000000u) : 0u;
    if (fail) atomicOr(&out[1], fail);
}

// ---------------------------------------------------------------------------------------------
// multi-GPU stitch on the device: csvsimd_stitch_shards (capi.cpp) for shard `rank`, run by one lane
// right after the all-gather on the same stream.  The two values the reference carries between
// 64-byte blocks (inside_str, array_idx: src/reader.rs:217-218) carried between GPUs; the re-emit
// launch reads out->in_quote_in from device memory, so the step never visits the host.
// ---------------------------------------------------------------------------------------------
__global__ void stitch_kernel(const csvsimd_shard_result* __restrict__ results, u32 n_shards, u32 rank,
                              u32 file_in_quote_in, csvsimd_stitch* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 state = file_in_quote_in ? 1u : 0u, err = 0;
    u64 idx = 1;  // the sentinel occupies global index 0
    csvsimd_stitch o = {};
    for (u32 i = 0; i < n_shards; ++i) {
        const u64 cnt = state ? results[i].count_enter_inside : results[i].count_enter_outside;
        if (i == rank) {
            o.in_quote_in = state;
            o.count = cnt;
            o.tape_index_base = idx;
            o.reemit = (results[i].in_quote_in_used & 1u) != state ? 1u : 0u;
        }
        idx += cnt;
        state ^= results[i].quote_parity & 1u;
        err |= results[i].error;
    }
    o.in_quote_final = state;
    o.total_entries = idx;
    o.error = err ? 1u : 0u;
    *out = o;
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (no allocation, no synchronisation: graph-capturable)
// ---------------------------------------------------------------------------------------------
#ifdef CSVSIMD_DEV_PROBES
#define CSVSIMD_PROBE_LAUNCH(MODE, EMITV)                                                                         \
    if (L.debug_mode == (MODE) && (EMITV) == (a.tape != nullptr)) {                                               \
        hipLaunchKernelGGL((stage1_kernel<EMITV, MODE>), dim3(grid), dim3(kThreads), 0, stream, a);               \
        launched = true;                                                                                          \
    }
#endif

// DIALECT 3's tables for the special bytes {delimiter, CR, LF, quote, escape} (quote 0 = none): two shifts in 0..5 (the
// three key bits of a byte must not reach into the next byte of the dword) whose xor gives every special byte its own
// 3-bit key.  An unused LUT slot holds a special byte whose OWN key is another one, so no byte with that key can equal it.
bool dialect_hash(u32 delim, u32 quote, u32 esc, DialectHash& h) {
    u32 sp[5], cls[5], n = 0;
    sp[n] = delim; cls[n++] = 0x80;
    sp[n] = 0x0d; cls[n++] = 0x80;
    sp[n] = 0x0a; cls[n++] = 0x80;
    if (quote) { sp[n] = quote; cls[n++] = 0x40; }
    if (esc) { sp[n] = esc; cls[n++] = 0x20; }
    for (u32 s1 = 0; s1 <= 5; ++s1)
        for (u32 s2 = s1 + 1; s2 <= 5; ++s2) {
            u32 key[5], seen = 0;
            bool ok = true;
            for (u32 i = 0; i < n && ok; ++i) {
                key[i] = ((sp[i] >> s1) ^ (sp[i] >> s2)) & 7u;
                ok = !((seen >> key[i]) & 1u);
                seen |= 1u << key[i];
            }
            if (!ok) continue;
            uint8_t lut[8], cl[8];
            for (u32 k = 0; k < 8; ++k) {
                u32 pick = 0;
                while (key[pick] == k) ++pick;  // n >= 3 distinct keys: one of the first two differs from k
                lut[k] = (uint8_t)sp[pick];
                cl[k] = 0;
            }
            for (u32 i = 0; i < n; ++i) { lut[key[i]] = (uint8_t)sp[i]; cl[key[i]] = (uint8_t)cls[i]; }
            h.sh1 = s1;
            h.sh2 = s2;
            h.lut_lo = lut[0] | lut[1] << 8 | lut[2] << 16 | (u32)lut[3] << 24;
            h.lut_hi = lut[4] | lut[5] << 8 | lut[6] << 16 | (u32)lut[7] << 24;
            h.cls_lo = cl[0] | cl[1] << 8 | cl[2] << 16 | (u32)cl[3] << 24;
            h.cls_hi = cl[4] | cl[5] << 8 | cl[6] << 16 | (u32)cl[7] << 24;
            return true;
        }
    return false;
}

hipError_t launch_stage1(const Stage1Launch& L, hipStream_t stream) {
    // delimiter-dense data, emitting launch, no escape byte: the other geometry (stage1_dense.hip)
    if (L.dense && L.dtape && !L.escape && L.debug_mode == 0)
        return csvsimd_dense::launch_stage1_dense(L, stream);
    KernelArgs a;
    fill_kernel_args(L, a);
    // 0 = the reference dialect (the tuned LUT classification), 1 = other delimiter / quote, 2 = + escape (direct
    // compares), 3 = + escape with the hashed LUT classification (when the special bytes have a collision-free hash)
    int dialect = L.escape ? 2 : (L.delimiter != ',' || L.quote != '"') ? 1 : 0;
    DialectHash dh;
    if (dialect == 2 && L.allow_hashed_dialect && dialect_hash(L.delimiter, L.quote, L.escape, dh)) {
        dialect = 3;
        a.hash_sh1 = dh.sh1; a.hash_sh2 = dh.sh2;
        a.hash_lut_lo = dh.lut_lo; a.hash_lut_hi = dh.lut_hi;
        a.hash_cls_lo = dh.cls_lo; a.hash_cls_hi = dh.cls_hi;
    }

    // ONE kernel per launch: an empty shard still runs one workgroup, which writes the result record
    const u32 want = a.num_tiles ? a.num_tiles : 1u;
    const u32 grid = want < L.max_blocks ? want : L.max_blocks;
    hipError_t e;
    if (L.ev_begin && (e = hipEventRecord(L.ev_begin, stream)) != hipSuccess) return e;
    bool launched = false;
#ifdef CSVSIMD_DEV_PROBES
    CSVSIMD_PROBE_LAUNCH(1, false)
    CSVSIMD_PROBE_LAUNCH(4, false)
    CSVSIMD_PROBE_LAUNCH(6, false)
    CSVSIMD_PROBE_LAUNCH(7, false)
    CSVSIMD_PROBE_LAUNCH(16, true)
    CSVSIMD_PROBE_LAUNCH(8, true)
    CSVSIMD_PROBE_LAUNCH(40, true)
    CSVSIMD_PROBE_LAUNCH(56, true)   // the timeline of a launch whose emit phases drop their tape stores
    CSVSIMD_PROBE_LAUNCH(8, false)
#endif
    if (launched) {
    } else if (dialect == 3 && a.tape)
        hipLaunchKernelGGL((stage1_kernel<true, 0, 3>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (dialect == 3)
        hipLaunchKernelGGL((stage1_kernel<false, 0, 3>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (dialect == 2 && a.tape)
        hipLaunchKernelGGL((stage1_kernel<true, 0, 2>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (dialect == 2)
        hipLaunchKernelGGL((stage1_kernel<false, 0, 2>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (dialect == 1 && a.tape)
        hipLaunchKernelGGL((stage1_kernel<true, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (dialect == 1)
        hipLaunchKernelGGL((stage1_kernel<false, 0, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (a.tape)
        hipLaunchKernelGGL((stage1_kernel<true, 0>), dim3(grid), dim3(kThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((stage1_kernel<false, 0>), dim3(grid), dim3(kThreads), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (L.ev_end && (e = hipEventRecord(L.ev_end, stream)) != hipSuccess) return e;
    return hipSuccess;
}

// Batched launch: `d_items` = n_items BatchItem lines in device memory (first_tile ascending, filled by the host),
// `d_first_tiles` = their first_tile fields again as a compact u32 array, `d_tots` = n_items zeroed u64 counters,
// `d_results` = n_items result records.  Reference dialect, entering states 0 / 1 only.
hipError_t launch_stage1_batch(void* d_items, void* d_first_tiles, void* d_tots, u32 n_items, u32 total_tiles,
                               csvsimd_shard_result* d_results, void* scratch_base, u64* scratch_desc, u32 max_blocks,
                               hipStream_t stream) {
    KernelArgs a = {};
    a.num_tiles = total_tiles;
    a.desc = scratch_desc;
    a.ctl = reinterpret_cast<Control*>(scratch_base);
    a.result = d_results;
    a.batch = reinterpret_cast<const BatchItem*>(d_items);
    a.batch_first = reinterpret_cast<const u32*>(d_first_tiles);
    a.batch_tot = reinterpret_cast<u64*>(d_tots);
    a.n_items = n_items;
    a.count_prio = 1u;
    a.cu_token = reinterpret_cast<u32*>(reinterpret_cast<char*>(scratch_base) + CSVSIMD_SCRATCH_TOKEN_OFFSET);
    a.token_mode = 1u;
    const u32 want = total_tiles ? total_tiles : 1u;
    const u32 grid = want < max_blocks ? want : max_blocks;
    hipLaunchKernelGGL((stage1_kernel<true, 0, 0, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

// name of the kernel launch_stage1 runs for this configuration (bench.py reports it next to the time)
const char* stage1_kernel_name(bool emit, int dialect, bool dense) {
    static const char* names[2][4] = {
        {"void csvsimd::stage1_kernel<false, 0, 0, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<false, 0, 1, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<false, 0, 2, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<false, 0, 3, false, false>(csvsimd::KernelArgs)"},
        {"void csvsimd::stage1_kernel<true, 0, 0, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<true, 0, 1, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<true, 0, 2, false, false>(csvsimd::KernelArgs)",
         "void csvsimd::stage1_kernel<true, 0, 3, false, false>(csvsimd::KernelArgs)"}};
    if (emit && dense && dialect == 0) return "void csvsimd_dense::stage1_kernel<true, 0, 0, false, true>(csvsimd_dense::KernelArgs)";
    if (emit && dense && dialect == 1) return "void csvsimd_dense::stage1_kernel<true, 0, 1, false, true>(csvsimd_dense::KernelArgs)";
    return names[emit ? 1 : 0][dialect < 0 || dialect > 3 ? 0 : dialect];
}

hipError_t launch_synth(void* dbuf, u64 file_off, u64 len, u32 cols, u32 width, u64 seed, u32 quote_pct,
                        hipStream_t stream) {
    if (len == 0) return hipSuccess;
    const u64 n4 = (len + 3) / 4;
    u64 blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(synth_kernel, dim3((u32)blocks), dim3(256), 0, stream, (uint8_t*)dbuf, file_off, len,
                       cols, width, seed, quote_pct);
    return hipGetLastError();
}

hipError_t launch_checksum(const void* dtape, u64 n, u64 first_index, void* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    u64 blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(checksum_kernel, dim3((u32)blocks), dim3(256), 0, stream, (const u64*)dtape, n,
                       first_index, (u64*)d_out);
    return hipGetLastError();
}

hipError_t launch_hbm_probe(const void* din, u64 len, void* dout, int write_div, void* scratch_base, u32 blocks,
                            hipStream_t stream) {
    const u32 tiles = (u32)(len / 131072);
    if (tiles == 0) return hipSuccess;
    Control* const ctl = reinterpret_cast<Control*>(scratch_base);
    if (write_div == 4)
        hipLaunchKernelGGL(hbm_probe_kernel<4>, dim3(blocks), dim3(256), 0, stream, (const uint8_t*)din, (uint4*)dout,
                           ctl, tiles);
    else if (write_div == 25)
        hipLaunchKernelGGL(hbm_probe_kernel<25>, dim3(blocks), dim3(256), 0, stream, (const uint8_t*)din, (uint4*)dout,
                           ctl, tiles);
    else
        hipLaunchKernelGGL(hbm_probe_kernel<0>, dim3(blocks), dim3(256), 0, stream, (const uint8_t*)din, (uint4*)dout,
                           ctl, tiles);
    return hipGetLastError();
}

hipError_t launch_copy_probe(const void* din, void* dout, u64 len, int mode, hipStream_t stream) {
    const u64 n16 = len / 16;
    if (n16 == 0) return hipSuccess;
    if (mode == 0) return hipMemcpyDtoDAsync((hipDeviceptr_t)dout, (hipDeviceptr_t)const_cast<void*>(din), n16 * 16, stream);
    const u64 blocks = (n16 + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (mode == 2)
        hipLaunchKernelGGL(copy_probe_kernel<true>, dim3((u32)blocks), dim3(256), 0, stream, (const u32x4*)din, (u32x4*)dout, n16);
    else
        hipLaunchKernelGGL(copy_probe_kernel<false>, dim3((u32)blocks), dim3(256), 0, stream, (const u32x4*)din, (u32x4*)dout, n16);
    return hipGetLastError();
}

hipError_t launch_stitch(const void* d_results, u32 n_shards, u32 rank, u32 file_in_quote_in, void* d_stitch,
                         hipStream_t stream) {
    hipLaunchKernelGGL(stitch_kernel, dim3(1), dim3(64), 0, stream, (const csvsimd_shard_result*)d_results, n_shards,
                       rank, file_in_quote_in, (csvsimd_stitch*)d_stitch);
    return hipGetLastError();
}

hipError_t launch_selftest(u32* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, d_out);
    return hipGetLastError();
}

int stage1_max_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, stage1_kernel<true, 0>, kThreads, 0) != hipSuccess || n < 1)
        n = 2;
    return n;
}

}  // namespace csvsimd
#endif  // CSVSIMD_DENSE_TU
