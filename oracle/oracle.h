/*
 * oracle.h — CPU restatement of the reference's stage 1.  TEST INFRASTRUCTURE ONLY: see oracle.c.
 * Nothing under csv-simd_amd/ may include this header.
 */
#ifndef CSVSIMD_ORACLE_H
#define CSVSIMD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_ERR_CAPACITY (-11)

/* class byte of the reference's nibble tables (src/stage1.rs:23-48) */
uint8_t oracle_byte_class(uint8_t b);
/* in-string mask of one 64-byte block entered outside a string (src/avx/stage1.rs:342-361): the clmul form
 * the reference executes and the bit loop design_notes_1.md:92-93 describes */
uint64_t oracle_string_mask_clmul(uint64_t quote_bits);
uint64_t oracle_string_mask_loop(uint64_t quote_bits);

/* scalar definition; offsets = base_off + i; no sentinel */
int oracle_scalar_index(const uint8_t* buf, uint64_t len, uint64_t base_off, uint32_t in_quote_in,
                        uint64_t* tape, uint64_t cap, uint64_t* n_out, uint32_t* in_quote_out);
/* scalar definition with the reference's layout: tape[0] = 0 sentinel (src/reader.rs:216) */
int oracle_scalar_read(const uint8_t* buf, uint64_t len, uint64_t* tape, uint64_t cap,
                       uint64_t* n_out);
/* (parity, count if the shard is entered outside a string, count if entered inside) */
void oracle_shard_descriptor(const uint8_t* buf, uint64_t len, uint32_t* parity,
                             uint64_t* cnt_enter_outside, uint64_t* cnt_enter_inside);

/* PARITY UNPINNED against the reference for everything marked "extension" below: the reference never
 * executes such behaviour, so these definitions are pinned on independent implementations instead
 * (Python's csv module, CPython's UTF-8 decoder — tests/test_oracle.py).
 *
 * Dialect extension (NOT a reference behaviour — the reference classifies
 * space/backslash, src/stage1.rs:41-48, but never uses them and hard-wires ',' and '"',
 * src/avx/stage1.rs:392-394).  Scalar definition the GPU dialect kernels are checked against:
 *   an escaped byte (the one after an unescaped `escape` byte) is literal; `quote` toggles the
 *   in-string state; `delimiter`, CR and LF outside a string are structural.  quote == 0 / escape == 0
 *   switch that feature off.  escape_in: the byte at offset 0 is escaped.
 * With (',', '"', 0) this is oracle_scalar_index. */
int oracle_dialect_index(const uint8_t* buf, uint64_t len, uint64_t base_off, uint8_t delimiter,
                         uint8_t quote, uint8_t escape, uint32_t in_quote_in, uint32_t escape_in,
                         uint64_t* tape, uint64_t cap, uint64_t* n_out, uint32_t* in_quote_out,
                         uint32_t* escape_out);

/* Extensions next to stage 1 (no executed reference counterpart: src/avx/utf8check.rs is dead code,
 * trimming is a todo in src/stage1.rs:41-48).  Sequential RFC 3629 decoder: offset of the first
 * byte that does not start / continue a well-formed sequence, UINT64_MAX if valid — pinned in
 * tests/test_oracle.py against CPython's bytes.decode (UnicodeDecodeError.start). */
uint64_t oracle_utf8_first_invalid(const uint8_t* buf, uint64_t len);
/* [*begin, *end) without leading/trailing 0x20 (flag 1) and one enclosing quote pair (flag 2) */
void oracle_trim_span(const uint8_t* bytes, uint64_t* begin, uint64_t* end, uint32_t flags, uint8_t quote);

/* faithful SSE restatement of reader::read (src/reader.rs:150-306); cap >= n + 64 */
int oracle_sse_read(const uint8_t* buf, uint64_t len, uint64_t* tape, uint64_t cap,
                    uint64_t* n_out);
/* same, output in a Vec-like growing buffer (the timed "ref_sse_1t" baseline); oracle_free it */
int oracle_sse_read_growing(const uint8_t* buf, uint64_t len, uint64_t** tape_out, uint64_t* n_out);
void oracle_free(void* p);
/* the same block loop on `threads` contiguous chunks with a quote-parity / count stitch (not a
 * reference behaviour: the reference is single-threaded); tape[0] = 0 sentinel */
int oracle_sse_read_mt(const uint8_t* buf, uint64_t len, int threads, uint64_t* tape, uint64_t cap,
                       uint64_t* n_out);

/* order-sensitive checksum of tape[0..n) whose first element has global index first_index */
void oracle_tape_checksum(const uint64_t* tape, uint64_t n, uint64_t first_index, uint64_t* s1,
                          uint64_t* s2);

/* synthetic corpus bytes [file_off, file_off+len) of the (cols x width) shape, SURVEY.md §8d */
void oracle_synth_fill(uint8_t* dst, uint64_t file_off, uint64_t len, uint32_t cols,
                       uint32_t width, uint64_t seed, uint32_t quote_pct);

#ifdef __cplusplus
}
#endif
#endif
