// stage1_kernels.hip — gfx950 (MI355X / CDNA4) kernels for CSV stage 1: bytes -> tape.
//
// What it computes (bit-exact with reference reader::read, src/reader.rs:150-306):
//   tape entry for byte i  <=>  byte in {',', CR, LF}  (class & 3,  src/avx/stage1.rs:394)
//                               and the inclusive prefix-xor of '"' bits at i is 0
//                               (src/avx/stage1.rs:342-407), emitted ascending as u64
//                               (src/stage1.rs:162-296).
//
// How (MI355X-first; nothing here mirrors the SSE code's structure):
//   * one pass over the input, HBM-bound: every byte is read once with 16-B/lane coalesced
//     loads (a wave covers 1 KiB per instruction); no table lookups, the byte classes come
//     from SWAR compares + v_dot4 bit gathers.
//   * a wave owns a contiguous span; per round it holds 4 KiB as 64 lanes x 4 chunks of 16 B and
//     keeps only two 64-bit masks per lane (4 x 16-bit fields: structural bits, in-string bits).
//   * in-string mask = per-field prefix-xor (4 shift-xor steps) + ballot/mbcnt carry across
//     lanes + scalar carry across rows/rounds (CDNA has no carry-less multiply).
//   * the two loop-carried quantities of the reference (inside_str, array_idx:
//     src/reader.rs:217-218) become a composable tile descriptor
//     (quote parity, count if entered outside, count if entered inside) resolved across
//     workgroups by a single-pass decoupled look-back over one 64-bit word per tile
//     (relaxed agent-scope atomics: the data is the flag, no fences).
//   * ordered compaction: per-lane counts -> packed DPP wave scan -> u16 offsets scattered into a
//     wave-private LDS window -> fully coalesced 8-B tape stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_kernels.h"

namespace csvsimd {

// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
static constexpr int kWaves = 4;                        // waves per workgroup
static constexpr int kThreads = kWaves * 64;            // 256
static constexpr int kRows = 4;                         // 1-KiB rows (dwordx4 wave loads) per round
static constexpr int kRoundBytes = kRows * 1024;        // 4 KiB per wave per round
static constexpr int kRounds = 8;                       // rounds per wave per tile
static constexpr int kSpanBytes = kRounds * kRoundBytes;  // 32 KiB contiguous per wave
static constexpr int kTileBytes = kWaves * kSpanBytes;    // 128 KiB per workgroup tile
static constexpr int kCompCap = 2048;                   // u16 entries per wave compaction window

static_assert(kTileBytes == CSVSIMD_TILE_BYTES, "tile geometry must match the host header");

// descriptor word: [63:62] status, [61] parity/state, aggregate: [47:24] B, [23:0] A
//                                                     inclusive: [60:0] running count
static constexpr uint64_t kStatusAgg = 1ull << 62;
static constexpr uint64_t kStatusInc = 2ull << 62;
static constexpr uint32_t kSpinLimit = 1u << 24;

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// wavefront primitives (wave64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

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
// byte classification: 16 bytes -> 16 structural bits + 16 quote bits (bit i = byte i)
// Equal to the reference's class table (src/stage1.rs:23-48): structural = class & 3
// ({0x2c, 0x0a, 0x0d}), quote = class & 16 ({0x22}); every other byte incl. >= 0x80 is 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void classify_dword(u32 x, u32 w, u32& acc_ns, u32& acc_nq) {
    // exact SWAR zero-byte test on the low 7 bits; bit 7 of x set => never a match.
    const u32 vm = x & 0x7f7f7f7fu;
    const u32 tc = (vm ^ 0x2c2c2c2cu) + 0x7f7f7f7fu;  // bit7(byte) = 1 iff low7 != ','
    const u32 tl = (vm ^ 0x0a0a0a0au) + 0x7f7f7f7fu;
    const u32 tr = (vm ^ 0x0d0d0d0du) + 0x7f7f7f7fu;
    const u32 tq = (vm ^ 0x22222222u) + 0x7f7f7f7fu;
    const u32 ns = (((tc & tl) & tr) | x) & 0x80808080u;  // 0x80 per byte that is NOT structural
    const u32 nq = (tq | x) & 0x80808080u;                // 0x80 per byte that is NOT a quote
    // gather the four bit-7 flags: sum(0x80 * weight) — weights are 1<<i, no carries
    acc_ns = __builtin_amdgcn_udot4(ns, w, acc_ns, false);
    acc_nq = __builtin_amdgcn_udot4(nq, w, acc_nq, false);
}

__device__ __forceinline__ void classify16(uint4 v, u32& st16, u32& q16) {
    u32 ns_lo = 0, nq_lo = 0, ns_hi = 0, nq_hi = 0;
    classify_dword(v.x, 0x08040201u, ns_lo, nq_lo);
    classify_dword(v.y, 0x80402010u, ns_lo, nq_lo);
    classify_dword(v.z, 0x08040201u, ns_hi, nq_hi);
    classify_dword(v.w, 0x80402010u, ns_hi, nq_hi);
    // acc = 128 * (8-bit "not" mask); assemble 16 bits and invert
    st16 = (((ns_lo >> 7) | (ns_hi << 1)) ^ 0xffffu) & 0xffffu;
    q16 = (((nq_lo >> 7) | (nq_hi << 1)) ^ 0xffffu) & 0xffffu;
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
__device__ __forceinline__ void store_desc(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 load_desc(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void lookback(u64* desc, u32 tile, Desc agg, u32 in_quote_in, u32 lane,
                                         u32& pin_out, u64& base_out, u32& err) {
    u32 pin = in_quote_in;
    u64 base = 0;
    if (tile != 0) {
        if (lane == 0)
            store_desc(desc + tile, kStatusAgg | ((u64)agg.p << 61) | ((u64)agg.b << 24) | (u64)agg.a);
        // acc = composition of the tiles in (hi, tile): function of the state entering tile hi+1
        u32 acc_p = 0;
        u64 acc_a = 0, acc_b = 0;
        int64_t hi = (int64_t)tile - 1;
        u32 spins = 0;
        for (;;) {
            const int64_t j = hi - (int64_t)lane;
            // virtual tile -1 = inclusive (in_quote_in, 0): the shard's entering state
            u64 d = kStatusInc | ((u64)in_quote_in << 61);
            if (j >= 0) d = load_desc(desc + j);
            const u32 status = (u32)(d >> 62);
            const u64 inv = __ballot(status == 0);
            const u64 inc = __ballot(status == 2);
            const u32 first_inv = inv ? (u32)__builtin_ctzll(inv) : 64u;
            const u32 first_inc = inc ? (u32)__builtin_ctzll(inc) : 64u;
            Desc f;
            f.p = (u32)(d >> 61) & 1u;
            f.a = (u32)d & 0xffffffu;
            f.b = (u32)(d >> 24) & 0xffffffu;
            if (first_inc < first_inv) {
                const Desc F = wave_compose_ordered(f, lane, first_inc);
                const u32 Fp = (u32)__builtin_amdgcn_readfirstlane((int)F.p);
                const u32 Fa = (u32)__builtin_amdgcn_readfirstlane((int)F.a);
                const u32 Fb = (u32)__builtin_amdgcn_readfirstlane((int)F.b);
                const u32 dlo = (u32)__builtin_amdgcn_readlane((int)(u32)d, (int)first_inc);
                const u32 dhi = (u32)__builtin_amdgcn_readlane((int)(u32)(d >> 32), (int)first_inc);
                const u64 dinc = ((u64)dhi << 32) | dlo;
                u32 s = (u32)(dinc >> 61) & 1u;
                u64 n = dinc & ((1ull << 61) - 1);
                n += s ? Fb : Fa;
                s ^= Fp;
                n += s ? acc_b : acc_a;
                s ^= acc_p;
                pin = s;
                base = n;
                break;
            }
            // fold the resolved-aggregate prefix [0, first_inv) and slide the window past it
            if (first_inv > 0) {
                const Desc F = wave_compose_ordered(f, lane, first_inv);
                const u32 Fp = (u32)__builtin_amdgcn_readfirstlane((int)F.p);
                const u32 Fa = (u32)__builtin_amdgcn_readfirstlane((int)F.a);
                const u32 Fb = (u32)__builtin_amdgcn_readfirstlane((int)F.b);
                const u64 na = (u64)Fa + (Fp ? acc_b : acc_a);
                const u64 nb = (u64)Fb + (Fp ? acc_a : acc_b);
                acc_a = na;
                acc_b = nb;
                acc_p ^= Fp;
                hi -= (int64_t)first_inv;
            }
            if (first_inv < 64) {  // predecessor not published yet: back off, bounded
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinLimit) { err = 1; break; }
            }
        }
    }
    const u32 state_out = pin ^ agg.p;
    const u64 count_out = base + (pin ? agg.b : agg.a);
    if (lane == 0) store_desc(desc + tile, kStatusInc | ((u64)state_out << 61) | count_out);
    pin_out = pin;
    base_out = base;
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel
// ---------------------------------------------------------------------------------------------
struct RoundMasks {
    u64 st;  // 4 x 16-bit fields (row j in bits [16j, 16j+16)): comma/CR/LF bits
    u64 s;   // same layout: in-string mask relative to the wave span's start (entered outside)
};

// One round = 4 rows x 1 KiB of this wave's span, 16 B per lane per row, through a buffer
// descriptor that covers exactly the tile's valid bytes: chunks past the end read as zero in
// hardware (zero bytes are class 0, exactly like the reference's zero padding of the last
// block, src/avx/stage1.rs:54-92), so interior and last tiles share one branch-free path.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void load_round(rsrc_t rsrc, u32 voff, uint4 (&v)[kRows]) {
    // The whole tile-relative offset lives in voff: the hardware range check covers
    // voffset + immediate only (soffset is excluded from bounds checking), and the check is what
    // makes reading "past the end" of the last tile safe.  j * 1024 folds into the 12-bit immediate.
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const auto x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + (u32)j * 1024u), 0, 0);
        v[j] = make_uint4(x[0], x[1], x[2], x[3]);
    }
}

__device__ __forceinline__ RoundMasks masks_of_round(const uint4 (&v)[kRows], u32& carry, u64 keep) {
    u32 st16[kRows], q16[kRows];
#pragma unroll
    for (int j = 0; j < kRows; ++j) classify16(v[j], st16[j], q16[j]);
    const u64 st = ((u64)(st16[0] | (st16[1] << 16)) | ((u64)(st16[2] | (st16[3] << 16)) << 32)) & keep;
    u64 x = ((u64)(q16[0] | (q16[1] << 16)) | ((u64)(q16[2] | (q16[3] << 16)) << 32)) & keep;
    // inclusive prefix-xor inside each 16-bit field
    x ^= (x << 1) & 0xfffefffefffefffeull;
    x ^= (x << 2) & 0xfffcfffcfffcfffcull;
    x ^= (x << 4) & 0xfff0fff0fff0fff0ull;
    x ^= (x << 8) & 0xff00ff00ff00ff00ull;
    // carry across lanes (ballot + mbcnt) and across rows (scalar), sequence order = row, lane
    u64 flip = 0;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const u64 par = __ballot((x >> (16 * j + 15)) & 1ull);
        const u32 enter = (mbcnt64(par) ^ carry) & 1u;
        flip |= enter ? (0xffffull << (16 * j)) : 0ull;
        carry ^= (u32)__builtin_popcountll(par) & 1u;
    }
    RoundMasks m;
    m.st = st;
    m.s = x ^ flip;
    return m;
}

// A shard whose start is not 16-byte aligned, or whose end is not a multiple of 16, has one
// partially valid 16-byte chunk at each end.  Loads always fetch whole chunks (a chunk never
// crosses a page; chunks entirely past the end read as zero through the buffer descriptor) and
// the stray bytes of those two chunks are dropped at the bit level after classification.  Each
// lane knows, per tile, at most one "back" special chunk (round, 64-bit keep mask); the "front"
// special chunk can only be chunk 0 of the shard (tile 0, wave 0, round 0, row 0, lane 0).
struct EdgeKeep {
    u32 back_round;   // round index of this lane's partial last chunk, or 0xff
    u64 back_keep;    // bits to keep in that round
    u64 front_keep;   // bits to keep in round 0 (all ones unless this lane holds chunk 0 and lo > 0)
};

__device__ __forceinline__ EdgeKeep edge_keep_of_tile(u32 lane, u32 w, u32 lo_rel, u32 hi_rel) {
    EdgeKeep e;
    e.back_round = 0xffu;
    e.back_keep = ~0ull;
    e.front_keep = ~0ull;
    const u32 span_lo = w * (u32)kSpanBytes;
    if ((hi_rel & 15u) != 0u) {
        const u32 bp = hi_rel & ~15u;  // tile-relative position of the partial chunk
        const u32 rel = bp - span_lo;  // wraps when the chunk is not in this wave's span
        if (rel < (u32)kSpanBytes && ((bp >> 4) & 63u) == lane) {
            e.back_round = rel >> 12;
            const u32 j = (rel >> 10) & 3u;
            const u64 drop = (u64)(0xffffu & ~((1u << (hi_rel & 15u)) - 1u)) << (16 * j);
            e.back_keep = ~drop;
        }
    }
    if (lo_rel != 0u && w == 0 && lane == 0) e.front_keep = ~(u64)((1u << lo_rel) - 1u);  // lo_rel < 16
    return e;
}

// Count phase of one wave span.  Two rounds (8 KiB per wave) are in flight: round r+1 is
// requested before round r is classified.  The body must stay ONE basic block and must not be
// duplicated under a branch: with control flow around it LLVM hoists/sinks the classification
// across all eight rounds (128+ live VGPRs, one wave per SIMD).  The sched_barriers keep the
// machine scheduler from doing the same and the opaque asm anchors each round's results.
__device__ __forceinline__ void count_phase(rsrc_t rsrc, u32 lane, u32 w, const EdgeKeep& ek,
                                            RoundMasks (&m)[kRounds], u32& carry, u32& cnt_a, u32& cnt_t) {
    uint4 v[2][kRows];
    u32 voff = w * (u32)kSpanBytes + lane * 16u;  // one running VGPR, advanced per round
    load_round(rsrc, voff, v[0]);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        if (r + 1 < kRounds) {
            voff += (u32)kRoundBytes;
            asm volatile("" : "+v"(voff));  // opaque: keeps hipcc from materialising 32 offsets up front
            load_round(rsrc, voff, v[(r + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        u64 keep = ek.back_round == (u32)r ? ek.back_keep : ~0ull;
        if (r == 0) keep &= ek.front_keep;
        m[r] = masks_of_round(v[r & 1], carry, keep);
        cnt_a += (u32)__builtin_popcountll(m[r].st & ~m[r].s);
        cnt_t += (u32)__builtin_popcountll(m[r].st);
        // anchor this round's results here: an opaque asm cannot be sunk or re-ordered
        asm volatile("" : "+v"(m[r].st), "+v"(m[r].s), "+v"(cnt_a), "+v"(cnt_t), "+s"(carry));
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct KernelArgs {
    const uint8_t* abase;  // 16-byte aligned
    u64 lo, hi;            // valid bytes are abase[lo, hi)
    u64 base_off;          // tape value of byte abase[lo]
    u32 in_quote_in;
    u32 num_tiles;
    u64* tape;
    u64 tape_cap;
    u64* desc;      // num_tiles words, zeroed
    u32* ticket;    // zeroed
    u64* tot_struct;  // 64 sharded counters, zeroed: total comma/CR/LF bytes
    csvsimd_shard_result* result;
};

template <bool EMIT>
__global__ __launch_bounds__(kThreads) void stage1_kernel(const KernelArgs args) {
    __shared__ u32 s_tile;
    __shared__ u32 s_wdesc[kWaves][3];
    __shared__ u32 s_pin;
    __shared__ u64 s_base;
    __shared__ u32 s_err;
    __shared__ unsigned short s_comp[EMIT ? kWaves : 1][EMIT ? kCompCap : 1];

    const u32 t = threadIdx.x;
    const u32 lane = t & 63u;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    if (t == 0) s_err = 0;

    for (;;) {
        if (t == 0) s_tile = atomicAdd(args.ticket, 1u);
        __syncthreads();
        const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)s_tile);
        if (tile >= args.num_tiles) break;

        const u64 tile0 = (u64)tile * kTileBytes;                // relative to abase
        const u64 span0 = tile0 + (u64)w * kSpanBytes;
        // descriptor over this tile's valid bytes, rounded up to whole 16-byte chunks (a chunk
        // never straddles a page, so the <= 15 extra bytes are always mapped)
        const u64 hi16 = (args.hi + 15) & ~15ull;
        const u64 avail = hi16 - tile0;
        const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(args.abase) + tile0, 0, (int)(avail < (u64)kTileBytes ? avail : (u64)kTileBytes),
            0x00020000);
        // valid bytes of this tile are [lo_rel, hi_rel) relative to the tile start
        const u32 lo_rel = args.lo > tile0 ? (u32)(args.lo - tile0) : 0u;  // lo < 16
        const u32 hi_rel = args.hi - tile0 < (u64)kTileBytes ? (u32)(args.hi - tile0) : (u32)kTileBytes;

        // ---- count phase: masks for the whole span stay in registers -------------------------
        RoundMasks m[kRounds];
        u32 carry = 0, cnt_a = 0, cnt_t = 0;
        const EdgeKeep ek = edge_keep_of_tile(lane, w, lo_rel, hi_rel);
        count_phase(rsrc, lane, w, ek, m, carry, cnt_a, cnt_t);
        const u32 wave_a = wave_sum(cnt_a);
        const u32 wave_t = wave_sum(cnt_t);
        if (lane == 0) {
            s_wdesc[w][0] = carry;
            s_wdesc[w][1] = wave_a;
            s_wdesc[w][2] = wave_t - wave_a;
        }
        __syncthreads();

        // ---- tile aggregate; this wave's entering state/offset relative to the tile ----------
        Desc agg = {0, 0, 0}, before = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            Desc d = {s_wdesc[k][0], s_wdesc[k][1], s_wdesc[k][2]};
            if ((u32)k == w) before = agg;
            agg = compose(agg, d);
        }

        if (w == 0) {
            u32 pin, err = 0;
            u64 base;
            lookback(args.desc, tile, agg, args.in_quote_in, lane, pin, base, err);
            if (lane == 0) {
                s_pin = pin;
                s_base = base;
                if (err) s_err = 1;
                atomicAdd((unsigned long long*)(args.tot_struct + (tile & 63u)),
                          (unsigned long long)(agg.a + agg.b));
                if (tile == args.num_tiles - 1) {
                    const u64 count = base + (pin ? agg.b : agg.a);
                    args.result->count = count;
                    args.result->in_quote_out = pin ^ agg.p;
                    args.result->quote_parity = pin ^ agg.p ^ args.in_quote_in;
                    args.result->written = count < args.tape_cap ? count : args.tape_cap;
                }
            }
        }
        __syncthreads();
        if (EMIT) {
            const u32 pin = s_pin;
            // state entering this wave's span and tape index of its first entry
            const u32 wstate = pin ^ before.p;
            u64 run = s_base + (pin ? before.b : before.a);
            const u64 flipall = wstate ? ~0ull : 0ull;
            unsigned short* comp = s_comp[w];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const u64 R = m[r].st & ~(m[r].s ^ flipall);
                // per-row counts, packed 2 x 16 bit per register, scanned across lanes
                const u32 c01 = (u32)__builtin_popcount((u32)R & 0xffffu) |
                                ((u32)__builtin_popcount((u32)R >> 16) << 16);
                const u32 c23 = (u32)__builtin_popcount((u32)(R >> 32) & 0xffffu) |
                                ((u32)__builtin_popcount((u32)(R >> 48)) << 16);
                const u32 i01 = wave_incl_scan_add(c01);
                const u32 i23 = wave_incl_scan_add(c23);
                const u32 t01 = (u32)__builtin_amdgcn_readlane((int)i01, 63);
                const u32 t23 = (u32)__builtin_amdgcn_readlane((int)i23, 63);
                const u32 e01 = i01 - c01, e23 = i23 - c23;
                const u32 tot0 = t01 & 0xffffu, tot1 = t01 >> 16, tot2 = t23 & 0xffffu, tot3 = t23 >> 16;
                const u32 n_r = tot0 + tot1 + tot2 + tot3;
                u32 pos[kRows];
                pos[0] = (e01 & 0xffffu);
                pos[1] = tot0 + (e01 >> 16);
                pos[2] = tot0 + tot1 + (e23 & 0xffffu);
                pos[3] = tot0 + tot1 + tot2 + (e23 >> 16);
                // byte offset of this round relative to the shard's first valid byte
                const u64 round_off = args.base_off + (span0 + (u64)r * kRoundBytes) - args.lo;
                for (u32 win = 0; win < n_r; win += kCompCap) {
#pragma unroll
                    for (int j = 0; j < kRows; ++j) {
                        u32 bits = (u32)(R >> (16 * j)) & 0xffffu;
                        u32 p = pos[j] - win;  // wraps for entries before the window
                        const u32 off = (u32)j * 1024u + lane * 16u;
                        while (bits) {
                            const u32 b = (u32)__builtin_ctz(bits);
                            bits &= bits - 1;
                            if (p < (u32)kCompCap) comp[p] = (unsigned short)(off + b);
                            ++p;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const u32 n_win = (n_r - win) < (u32)kCompCap ? (n_r - win) : (u32)kCompCap;
                    for (u32 i = lane; i < n_win; i += 64) {
                        const u64 idx = run + win + i;
                        if (idx < args.tape_cap) args.tape[idx] = round_off + comp[i];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                run += n_r;
            }
        }
    }
    // the barrier at the loop head ordered every wave's s_err store of earlier tiles
    if (t == 0 && s_err) args.result->error = 1;
}

// sums the sharded structural-byte counters into the result (tiny, 1 wave)
__global__ void finalize_kernel(const u64* tot_struct, csvsimd_shard_result* result, u32 in_quote_in,
                                u32 num_tiles) {
    const u32 lane = threadIdx.x;
    u64 v = tot_struct[lane];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 lo = (u32)__shfl_xor((int)(u32)v, d);
        const u32 hi = (u32)__shfl_xor((int)(u32)(v >> 32), d);
        v += ((u64)hi << 32) | lo;
    }
    if (lane == 0) {
        if (num_tiles == 0) {
            result->count = 0;
            result->in_quote_out = in_quote_in;
            result->quote_parity = 0;
            result->written = 0;
        }
        const u64 total = v;
        const u64 count = result->count;
        // total = count_enter_outside + count_enter_inside, whichever hypothesis was run
        result->count_enter_outside = in_quote_in ? total - count : count;
        result->count_enter_inside = in_quote_in ? count : total - count;
    }
}

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

// self-test of the wavefront primitives against plain loops (one wave); out[0] = failure bits
__global__ void selftest_kernel(u32* out) {
    __shared__ u32 s_v[64];
    __shared__ u32 s_d[64][3];
    const u32 lane = threadIdx.x;
    u32 fail = 0;
    if (lane_id() != lane) fail |= 1;
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
// host-side launchers (no allocation, no synchronisation: graph-capturable)
// ---------------------------------------------------------------------------------------------
hipError_t launch_stage1(const Stage1Launch& L, hipStream_t stream) {
    const uintptr_t addr = (uintptr_t)L.dbuf;
    KernelArgs a;
    a.abase = (const uint8_t*)(addr & ~(uintptr_t)15);
    a.lo = (u64)(addr & 15);
    a.hi = a.lo + L.len;
    a.base_off = L.base_off;
    a.in_quote_in = L.in_quote_in ? 1u : 0u;
    a.num_tiles = (u32)((a.hi + kTileBytes - 1) / kTileBytes);
    if (L.len == 0) a.num_tiles = 0;
    a.tape = (u64*)L.dtape;
    a.tape_cap = L.dtape ? L.tape_cap : 0;
    a.desc = L.scratch_desc;
    a.ticket = L.scratch_ticket;
    a.tot_struct = L.scratch_tot;
    a.result = L.d_result;

    hipError_t e;
    // one memset covers ticket + sharded totals + descriptors (contiguous in the scratch block)
    e = hipMemsetAsync(L.scratch_base, 0, L.scratch_zero_bytes(a.num_tiles), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(L.d_result, 0, sizeof(csvsimd_shard_result), stream);
    if (e != hipSuccess) return e;
    if (a.num_tiles > 0) {
        const u32 grid = a.num_tiles < L.max_blocks ? a.num_tiles : L.max_blocks;
        if (L.ev_begin && (e = hipEventRecord(L.ev_begin, stream)) != hipSuccess) return e;
        if (a.tape)
            hipLaunchKernelGGL(stage1_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, a);
        else
            hipLaunchKernelGGL(stage1_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (L.ev_end && (e = hipEventRecord(L.ev_end, stream)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, stream, a.tot_struct, a.result,
                       a.in_quote_in, a.num_tiles);
    return hipGetLastError();
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

hipError_t launch_selftest(u32* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, d_out);
    return hipGetLastError();
}

int stage1_max_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, stage1_kernel<true>, kThreads, 0) != hipSuccess || n < 1)
        n = 2;
    return n;
}

}  // namespace csvsimd
