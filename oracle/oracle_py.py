"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / CPU baseline.  Nothing under csv-simd_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_u8p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        h.oracle_byte_class.restype, h.oracle_byte_class.argtypes = C.c_uint8, [C.c_uint8]
        for f in (h.oracle_string_mask_clmul, h.oracle_string_mask_loop):
            f.restype, f.argtypes = C.c_uint64, [C.c_uint64]
        h.oracle_scalar_index.restype = C.c_int
        h.oracle_scalar_index.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                          _u64p, C.POINTER(C.c_uint32)]
        h.oracle_dialect_index.restype = C.c_int
        h.oracle_dialect_index.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint8, C.c_uint8, C.c_uint8,
                                           C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, _u64p,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        h.oracle_utf8_first_invalid.restype = C.c_uint64
        h.oracle_utf8_first_invalid.argtypes = [C.c_void_p, C.c_uint64]
        h.oracle_trim_span.restype = None
        h.oracle_trim_span.argtypes = [C.c_void_p, _u64p, _u64p, C.c_uint32, C.c_uint8]
        h.oracle_scalar_read.restype = C.c_int
        h.oracle_scalar_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p]
        h.oracle_shard_descriptor.restype = None
        h.oracle_shard_descriptor.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), _u64p, _u64p]
        h.oracle_sse_read.restype = C.c_int
        h.oracle_sse_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p]
        h.oracle_sse_read_growing.restype = C.c_int
        h.oracle_sse_read_growing.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_u64p), _u64p]
        h.oracle_free.restype, h.oracle_free.argtypes = None, [C.c_void_p]
        h.oracle_sse_read_mt.restype = C.c_int
        h.oracle_sse_read_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, _u64p]
        h.oracle_tape_checksum.restype = None
        h.oracle_tape_checksum.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, _u64p, _u64p]
        h.oracle_synth_fill.restype = None
        h.oracle_synth_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_uint32]
        _lib = h
    return _lib


def aligned_copy(data, align: int = 64, head: int = 0) -> np.ndarray:
    """uint8 view whose address is `head` bytes past an `align` boundary (mmap-like when head=0)."""
    b = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    raw = np.zeros(b.size + 2 * align + head, dtype=np.uint8)
    off = (-raw.ctypes.data) % align + head
    a = raw[off: off + b.size]
    a[:] = b
    return a


def scalar_read(data) -> np.ndarray:
    a = aligned_copy(data)
    out = np.zeros(a.size + 2, dtype=np.uint64)
    n = C.c_uint64()
    rc = lib().oracle_scalar_read(a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def scalar_index(data, base_off: int = 0, in_quote_in: int = 0):
    """-> (entries uint64[], in_quote_out) without sentinel."""
    a = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    out = np.zeros(a.size + 1, dtype=np.uint64)
    n, q = C.c_uint64(), C.c_uint32()
    rc = lib().oracle_scalar_index(a.ctypes.data, a.size, base_off, in_quote_in, out.ctypes.data, out.size,
                                   C.byref(n), C.byref(q))
    assert rc == 0
    return out[: n.value].copy(), q.value


def dialect_index(data, delimiter=0x2c, quote=0x22, escape=0, base_off: int = 0, in_quote_in: int = 0,
                  escape_in: int = 0):
    """Dialect extension (no reference counterpart) -> (entries uint64[], in_quote_out, escape_out)."""
    a = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    out = np.zeros(a.size + 1, dtype=np.uint64)
    n, q, e = C.c_uint64(), C.c_uint32(), C.c_uint32()
    rc = lib().oracle_dialect_index(a.ctypes.data, a.size, base_off, delimiter, quote, escape, in_quote_in,
                                    escape_in, out.ctypes.data, out.size, C.byref(n), C.byref(q), C.byref(e))
    assert rc == 0
    return out[: n.value].copy(), q.value, e.value


def utf8_first_invalid(data):
    """None if valid UTF-8, else the offset of the first offending byte."""
    a = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    r = lib().oracle_utf8_first_invalid(a.ctypes.data if a.size else None, a.size)
    return None if r == 2**64 - 1 else int(r)


def trim_spans(data, begin, end, flags: int, quote: int = 0x22):
    a = np.ascontiguousarray(data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8))
    b, e = np.array(begin, dtype=np.uint64), np.array(end, dtype=np.uint64)
    for i in range(b.size):
        bb, ee = C.c_uint64(int(b[i])), C.c_uint64(int(e[i]))
        lib().oracle_trim_span(a.ctypes.data, C.byref(bb), C.byref(ee), flags, quote)
        b[i], e[i] = bb.value, ee.value
    return b, e


def sse_read(data, head: int = 0) -> np.ndarray:
    a = aligned_copy(data, head=head)
    out = np.zeros(a.size + 130, dtype=np.uint64)
    n = C.c_uint64()
    rc = lib().oracle_sse_read(a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def sse_read_growing_timed(a: np.ndarray):
    """Runs the Vec-growing restatement once; returns (n_entries, seconds)."""
    import time
    p, n = _u64p(), C.c_uint64()
    t0 = time.perf_counter()
    rc = lib().oracle_sse_read_growing(a.ctypes.data, a.size, C.byref(p), C.byref(n))
    dt = time.perf_counter() - t0
    assert rc == 0
    lib().oracle_free(p)
    return n.value, dt


def sse_read_growing_many_timed(arrays):
    """ref_sse_1t over many files on ONE thread (oracle_sse_read_growing_many): -> (entry counts, seconds).  The ctypes
    arrays are prepared before the clock starts and the tapes freed after it stops."""
    import time
    n = len(arrays)
    bufs = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
    lens = (C.c_uint64 * n)(*[a.size for a in arrays])
    tapes = (C.c_void_p * n)()
    counts = (C.c_uint64 * n)()
    fn = lib().oracle_sse_read_growing_many
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    t0 = time.perf_counter()
    done = fn(bufs, lens, n, tapes, counts)
    dt = time.perf_counter() - t0
    assert done == sum(1 for a in arrays if a.size >= 64)
    for p in tapes:
        if p:
            lib().oracle_free(C.c_void_p(p))
    return list(counts), dt


def sse_read_mt(data, threads: int) -> np.ndarray:
    a = aligned_copy(data)
    out = np.zeros(a.size + 2, dtype=np.uint64)
    n = C.c_uint64()
    rc = lib().oracle_sse_read_mt(a.ctypes.data, a.size, threads, out.ctypes.data, out.size, C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def shard_descriptor(data):
    a = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    p, c0, c1 = C.c_uint32(), C.c_uint64(), C.c_uint64()
    lib().oracle_shard_descriptor(a.ctypes.data, a.size, C.byref(p), C.byref(c0), C.byref(c1))
    return p.value, c0.value, c1.value


def tape_checksum(tape: np.ndarray, first_index: int = 0):
    t = np.ascontiguousarray(tape, dtype=np.uint64)
    s1, s2 = C.c_uint64(), C.c_uint64()
    lib().oracle_tape_checksum(t.ctypes.data, t.size, first_index, C.byref(s1), C.byref(s2))
    return s1.value, s2.value


def synth(file_off: int, length: int, cols: int, width: int, seed: int, quote_pct: int = 0) -> np.ndarray:
    out = aligned_copy(np.zeros(length, dtype=np.uint8))
    lib().oracle_synth_fill(out.ctypes.data, file_off, length, cols, width, seed, quote_pct)
    return out


# ---- consumers of the tape: scalar definitions (test infrastructure; pinned on Python's own string semantics) ----
def seek_field(data: bytes, index, field_cnt: int, crlf: bool, record_idx: int, field_idx: int):
    """RecordSource::seek_field (src/record_source.rs:106-140), line by line."""
    row_size = field_cnt + 1 if crlf else field_cnt
    record_cnt = (len(index) - 1) // row_size
    if record_idx + 1 >= record_cnt or field_idx >= field_cnt:
        return None
    idx_start = (record_idx + 1) * row_size + field_idx
    return bytes(data[int(index[idx_start]) + 1: int(index[idx_start + 1])])


def chunk_record_ids(chunk, field_cnt: int, crlf: bool):
    """Records (seek_field numbering) of a Chunk {id, start, end, record_cnt} of Tape::chunks (src/tape.rs:95-140)."""
    row_size = field_cnt + 1 if crlf else field_cnt
    _, start, end, _ = chunk
    return range(start // row_size - 1, end // row_size - 1)


def column_frequency(data: bytes, index, field_cnt: int, crlf: bool, chunks, field_idx: int):
    """design_notes_1.md:1-4 "frequency counts": collections.Counter over the column's field texts."""
    from collections import Counter
    c = Counter()
    for ch in chunks:
        for r in chunk_record_ids(ch, field_cnt, crlf):
            c[seek_field(data, index, field_cnt, crlf, r, field_idx)] += 1
    return c


def column_search(data: bytes, index, field_cnt: int, crlf: bool, chunk, field_idx: int, needle: bytes, mode: int):
    """design_notes_1.md:1-4 "function search": record ids whose field == / startswith / contains the needle."""
    test = (lambda v: v == needle, lambda v: v.startswith(needle), lambda v: needle in v)[mode]
    return [r for r in chunk_record_ids(chunk, field_cnt, crlf)
            if test(seek_field(data, index, field_cnt, crlf, r, field_idx))]
