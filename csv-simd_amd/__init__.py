"""csv-simd_amd — Python host side of the MI355X stage-1 CSV structural indexer.

Thin ctypes binding over the C ABI in ``include/csvsimd.h`` (``csrc/libcsvsimd_hip.so``, built
in-tree by ``__graft_entry__.build()`` / ``make -C csv-simd_amd/csrc``).  Names follow the
reference crate: ``read`` (src/reader.rs:150), ``create`` (src/lib.rs:61), ``Tape`` /
``boundaries`` / ``chunks`` (src/tape.rs), ``seek_record`` / ``seek_field``
(src/record_source.rs:70-140), ``StructureError`` (src/error.rs:7-21).

There is no CPU fallback: every compute call needs a HIP device and raises ``StructureError``
(or ``OSError`` if the shared library has not been built) otherwise.  The directory name holds a
hyphen, so import it with ``__graft_entry__.load_package()`` (module name ``csv_simd_amd``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSVSIMD_LIB") or os.path.join(_HERE, "csrc", "libcsvsimd_hip.so")  # override: tuning variants only

OK = 0
ERR_IO, ERR_MISSING_VALUE, ERR_INVALID_STATE, ERR_INVALID_CSV_FORMAT = -1, -2, -3, -4
ERR_INVALID_ARG, ERR_TAPE_CAPACITY, ERR_HIP, ERR_NO_DEVICE, ERR_INTERNAL, ERR_RCCL = -9, -11, -12, -13, -14, -15
NEWLINE_LF, NEWLINE_CRLF = 0, 1


class StructureError(RuntimeError):
    """Mirror of reference ``StructureError`` (src/error.rs:7-21) plus the ABI's own codes."""

    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = lib().csvsimd_strerror(code).decode()
        if code == ERR_HIP or detail:
            msg += ": " + (detail or lib().csvsimd_last_error().decode())
        super().__init__(f"[{code}] {msg}")


class ShardResult(C.Structure):
    _fields_ = [
        ("count", C.c_uint64),
        ("count_enter_outside", C.c_uint64),
        ("count_enter_inside", C.c_uint64),
        ("quote_parity", C.c_uint32),
        ("in_quote_out", C.c_uint32),
        ("error", C.c_uint32),
        ("escape_out", C.c_uint32),
        ("written", C.c_uint64),
        ("in_quote_in_used", C.c_uint32),   # the entering state the pass ran with (the kernel's own choice with ENTER_GUESS)
        ("reserved0", C.c_uint32),
        ("reserved1", C.c_uint64),
    ]


# in_quote_in of the device entry points (include/csvsimd.h)
ENTER_OUTSIDE, ENTER_INSIDE, ENTER_GUESS = 0, 1, 2


class Dialect(C.Structure):
    """csvsimd_dialect — extension beyond the reference (which hard-wires ',' and '"',
    src/avx/stage1.rs:392-394).  quote / escape 0 = off."""
    _fields_ = [("delimiter", C.c_uint8), ("quote", C.c_uint8), ("escape", C.c_uint8), ("escape_in", C.c_uint8),
                ("reserved", C.c_uint32)]

    def __init__(self, delimiter=",", quote='"', escape=None, escape_in: int = 0):
        def byte(v):
            if v is None:
                return 0
            return v if isinstance(v, int) else ord(v)
        super().__init__(byte(delimiter), byte(quote), byte(escape), 1 if escape_in else 0, 0)


class Utf8Result(C.Structure):
    _fields_ = [("first_invalid", C.c_uint64), ("reserved", C.c_uint64)]


TRIM_SPACE, TRIM_QUOTES = 1, 2
UTF8_VALID = 2**64 - 1


class Stitch(C.Structure):
    _fields_ = [
        ("in_quote_in", C.c_uint32),
        ("in_quote_final", C.c_uint32),
        ("count", C.c_uint64),
        ("tape_index_base", C.c_uint64),
        ("total_entries", C.c_uint64),
        ("error", C.c_uint32),
        ("reemit", C.c_uint32),   # this shard's pass ran with another entering state than the true one
    ]


class BatchItem(C.Structure):
    """csvsimd_batch_item: one buffer of csvsimd_stage1_index_batch_device_async."""
    _fields_ = [("dbuf", C.c_void_p), ("len", C.c_uint64), ("base_off", C.c_uint64), ("dtape", C.c_void_p),
                ("tape_cap", C.c_uint64), ("in_quote_in", C.c_uint32), ("reserved", C.c_uint32)]


class HostBatchItem(C.Structure):
    """csvsimd_host_batch_item: one host file of csvsimd_stage1_index_batch."""
    _fields_ = [("buf", C.c_void_p), ("len", C.c_uint64), ("tape", C.c_void_p), ("tape_cap", C.c_uint64),
                ("tape_len", C.c_uint64), ("in_quote_out", C.c_uint32), ("status", C.c_int32)]


class MultiShard(C.Structure):
    """csvsimd_multi_shard: one shard of csvsimd_stage1_index_multi (in: ctx, dbuf, dtape, tape_cap; out: the rest)."""
    _fields_ = [("ctx", C.c_void_p), ("dbuf", C.c_void_p), ("dtape", C.c_void_p), ("tape_cap", C.c_uint64),
                ("begin", C.c_uint64), ("end", C.c_uint64), ("result", ShardResult), ("stitch", Stitch)]


class _Boundary(C.Structure):
    _fields_ = [("start", C.c_uint64), ("len", C.c_uint64)]


class _Chunk(C.Structure):
    _fields_ = [("id", C.c_uint8), ("start", C.c_uint64), ("end", C.c_uint64), ("record_cnt", C.c_uint32)]


class FreqStatus(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_distinct", C.c_uint64), ("max_field_bytes", C.c_uint64),
                ("overflow", C.c_uint64)]


class ColFreqStatus(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_distinct", C.c_uint64), ("truncated", C.c_uint64),
                ("overflow", C.c_uint64)]


class IngestPhases(C.Structure):
    """csvsimd_ingest_phases: where the wall time of the thread's latest csvsimd_stage1_index call went (seconds)."""
    _fields_ = [("bytes", C.c_uint64), ("chunks", C.c_uint64), ("host_threads", C.c_uint32), ("reserved", C.c_uint32),
                ("wall", C.c_double), ("stage_copy", C.c_double), ("stage_wait", C.c_double), ("expand_copy", C.c_double),
                ("submit", C.c_double), ("wait_staged", C.c_double), ("wait_record", C.c_double),
                ("wait_expanded", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "reserved"}


SEARCH_EQUALS, SEARCH_STARTS_WITH, SEARCH_CONTAINS = 0, 1, 2
ABI_VERSION = 5   # what this binding was written against: checked when the library is loaded

# every symbol include/csvsimd.h declares: (restype, argtypes)
_u64p = C.POINTER(C.c_uint64)
_PROTOTYPES = {
    "csvsimd_strerror": (C.c_char_p, [C.c_int]),
    "csvsimd_last_error": (C.c_char_p, []),
    "csvsimd_device_count": (C.c_int, []),
    "csvsimd_abi_version": (C.c_uint32, []),
    "csvsimd_tile_bytes": (C.c_uint32, []),
    "csvsimd_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "csvsimd_ctx_destroy": (None, [C.c_void_p]),
    "csvsimd_ctx_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "csvsimd_stage1_index_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                                    C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "csvsimd_stage1_index_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                              C.c_void_p, C.c_uint64, C.POINTER(ShardResult), C.c_void_p]),
    "csvsimd_dialect_init": (C.c_int, [C.POINTER(Dialect)]),
    "csvsimd_stage1_index_device_dialect_async": (C.c_int, [C.c_void_p, C.POINTER(Dialect), C.c_void_p, C.c_uint64,
                                                            C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                                            C.c_void_p, C.c_void_p]),
    "csvsimd_stage1_index_dialect": (C.c_int, [C.c_void_p, C.POINTER(Dialect), C.c_void_p, C.c_uint64, C.c_void_p,
                                               C.c_uint64, _u64p, C.POINTER(C.c_uint32)]),
    "csvsimd_trim_spans_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8,
                                            C.c_void_p]),
    "csvsimd_utf8_validate_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "csvsimd_utf8_validate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Utf8Result),
                                               C.c_void_p]),
    "csvsimd_hbm_probe_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "csvsimd_copy_probe_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int,
                                            C.c_int, C.POINTER(C.c_float)]),
    "csvsimd_tape_record_spans_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64,
                                                   C.c_uint64, C.c_void_p, C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_ctx_hint_density": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "csvsimd_ctx_limit_workgroups": (C.c_int, [C.c_void_p, C.c_uint32]),
    "csvsimd_ctx_kernel_name": (C.c_char_p, [C.c_void_p, C.POINTER(Dialect)]),
    "csvsimd_stage1_bound": (C.c_int, [C.c_uint64, _u64p]),
    "csvsimd_stage1_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p,
                                       C.POINTER(C.c_uint32)]),
    "csvsimd_stage1_index_batch": (C.c_int, [C.c_void_p, C.POINTER(HostBatchItem), C.c_uint32]),
    "csvsimd_ingest_chunk_plan": (C.c_int, [C.c_uint64, _u64p, C.c_uint64, _u64p]),
    "csvsimd_ingest_last_phases": (C.c_int, [C.POINTER(IngestPhases)]),
    "csvsimd_stage1_index_batch_device_async": (C.c_int, [C.c_void_p, C.POINTER(BatchItem), C.c_uint32, C.c_void_p,
                                                          C.c_void_p]),
    "csvsimd_multi_shard_range": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, _u64p, _u64p]),
    "csvsimd_stage1_index_multi": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(MultiShard), C.c_uint32, C.c_uint32]),
    "csvsimd_stitch_shards": (C.c_int, [C.POINTER(ShardResult), C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(Stitch)]),
    "csvsimd_stitch_shards_device_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                     C.c_void_p]),
    "csvsimd_stage1_reemit_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                                     C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "csvsimd_stage1_kernel_name": (C.c_char_p, [C.c_int, C.POINTER(Dialect)]),
    "csvsimd_build_has_probes": (C.c_uint32, []),
    "csvsimd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "csvsimd_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "csvsimd_comm_destroy": (None, [C.c_void_p]),
    "csvsimd_stage1_index_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                               C.c_void_p, C.c_uint64, C.POINTER(ShardResult), C.POINTER(Stitch),
                                               C.c_void_p]),
    "csvsimd_tape_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "csvsimd_tape_destroy": (None, [C.c_void_p]),
    "csvsimd_tape_field_cnt": (C.c_uint32, [C.c_void_p]),
    "csvsimd_tape_record_cnt": (C.c_uint32, [C.c_void_p]),
    "csvsimd_tape_record_jump_size": (C.c_uint64, [C.c_void_p]),
    "csvsimd_tape_record_offset": (C.c_uint32, [C.c_void_p]),
    "csvsimd_tape_new_line": (C.c_int, [C.c_void_p]),
    "csvsimd_tape_header_name": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64]),
    "csvsimd_tape_seek_record": (C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u64p, C.POINTER(C.c_int)]),
    "csvsimd_tape_seek_field": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p, C.POINTER(C.c_int)]),
    "csvsimd_boundaries": (C.c_int, [C.c_uint32, C.c_uint8, C.POINTER(_Boundary), C.POINTER(C.c_uint32)]),
    "csvsimd_tape_chunks": (C.c_int, [C.c_void_p, C.c_uint8, C.POINTER(_Chunk), C.POINTER(C.c_uint32)]),
    "csvsimd_create": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "csvsimd_tape_index": (_u64p, [C.c_void_p, _u64p]),
    "csvsimd_tape_bytes": (C.POINTER(C.c_uint8), [C.c_void_p, _u64p]),
    "csvsimd_tape_field_spans_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64,
                                                  C.c_uint64, C.c_void_p, C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_gather_fields_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                               C.c_uint32, C.c_void_p, C.c_void_p]),
    "csvsimd_chunk_field_spans_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_Chunk),
                                                   C.c_uint32, C.c_void_p, C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_column_frequency_scratch_bytes": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint64]),
    "csvsimd_column_frequency_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int,
                                                  C.POINTER(_Chunk), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                                  C.c_void_p, C.c_uint64, C.POINTER(FreqStatus), C.c_void_p]),
    "csvsimd_column_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int,
                                               C.POINTER(_Chunk), C.c_uint32, C.c_char_p, C.c_uint32, C.c_int,
                                               C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_chunk_to_columns_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                                  C.c_int, C.POINTER(_Chunk), C.POINTER(C.c_uint32), C.c_uint32,
                                                  C.c_void_p, C.c_uint32, C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_columnar_frequency_scratch_bytes": (C.c_uint64, [C.c_uint64]),
    "csvsimd_columnar_frequency_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                                    C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                    C.POINTER(ColFreqStatus), C.c_void_p]),
    "csvsimd_columnar_frequency_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                                          C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                          C.c_void_p, C.c_void_p]),
    "csvsimd_columnar_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p,
                                                 C.c_uint32, C.c_int, C.c_void_p, _u64p, C.c_void_p]),
    "csvsimd_bitmap_select_scratch_bytes": (C.c_uint64, [C.c_uint64]),
    "csvsimd_bitmap_select_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                               _u64p, C.c_void_p]),
    "csvsimd_synth_fill_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                            C.c_uint64, C.c_uint32, C.c_void_p]),
    "csvsimd_tape_checksum_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "csvsimd_selftest_device": (C.c_int, [C.c_int]),
    "csvsimd_stage1_time_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Loads libcsvsimd_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError = the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        got = handle.csvsimd_abi_version()
        if got != ABI_VERSION:   # entry points changed their argument lists between versions: never call across them
            raise OSError(f"{LIB_PATH} reports C ABI version {got}, this binding is written against {ABI_VERSION}: rebuild "
                          "(__graft_entry__.build())")
        _lib = handle
    return _lib


def _check(rc: int) -> None:
    if rc != OK:
        raise StructureError(rc)


def tile_bytes() -> int:
    return lib().csvsimd_tile_bytes()


def device_count() -> int:
    return lib().csvsimd_device_count()


def selftest(device: int = 0) -> None:
    _check(lib().csvsimd_selftest_device(device))


class Context:
    """One per (thread, device): owns the look-back scratch (csvsimd_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib().csvsimd_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def hint_density(self, entries: int, nbytes: int) -> None:
        """Entries per byte of the data about to be indexed (nbytes 0: forget): chooses the kernel instantiation of the
        following launches (same tape either way)."""
        _check(lib().csvsimd_ctx_hint_density(self._h, entries, nbytes))

    def limit_workgroups(self, n: int) -> None:
        """Test knob: at most n workgroups per stage-1 launch of this context (0: the default grid)."""
        _check(lib().csvsimd_ctx_limit_workgroups(self._h, n))

    def kernel_name(self, dialect: "Dialect" = None) -> str:
        return lib().csvsimd_ctx_kernel_name(self._h, C.byref(dialect) if dialect is not None else None).decode()

    def reserve(self, max_len: int) -> None:
        _check(lib().csvsimd_ctx_reserve(self._h, max_len))

    def close(self) -> None:
        if self._h:
            lib().csvsimd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stage 1 on device-resident bytes (pointers are raw device addresses) ----------------
    def stage1_index_device(self, dbuf: int, length: int, base_off: int = 0, in_quote_in: int = 0,
                            dtape: int = 0, tape_cap: int = 0, stream: int = 0,
                            allow_overflow: bool = False) -> ShardResult:
        r = ShardResult()
        rc = lib().csvsimd_stage1_index_device(self._h, dbuf, length, base_off, in_quote_in, dtape or None,
                                               tape_cap, C.byref(r), stream or None)
        if rc == ERR_TAPE_CAPACITY and allow_overflow:
            return r
        _check(rc)
        return r

    def stage1_index_device_async(self, dbuf: int, length: int, base_off: int, in_quote_in: int, dtape: int,
                                  tape_cap: int, d_result: int, stream: int = 0) -> None:
        _check(lib().csvsimd_stage1_index_device_async(self._h, dbuf, length, base_off, in_quote_in,
                                                       dtape or None, tape_cap, d_result, stream or None))

    def stage1_index_batch_device_async(self, items, d_results: int, stream: int = 0) -> None:
        """ONE launch over many buffers: items = [(dbuf, length, base_off, dtape, tape_cap, in_quote_in), ...]; record i
        of d_results (device, 64 bytes each) is what stage1_index_device_async would have written for buffer i alone."""
        arr = (BatchItem * len(items))()
        for i, (dbuf, length, base_off, dtape, cap, inq) in enumerate(items):
            arr[i].dbuf, arr[i].len, arr[i].base_off = dbuf or None, length, base_off
            arr[i].dtape, arr[i].tape_cap, arr[i].in_quote_in = dtape or None, cap, inq
        _check(lib().csvsimd_stage1_index_batch_device_async(self._h, arr, len(items), d_results, stream or None))

    def stage1_reemit_device_async(self, dbuf: int, length: int, base_off: int, d_stitch: int, dtape: int,
                                   tape_cap: int, d_result: int, stream: int = 0) -> None:
        """Second launch of a sharded step: indexes the shard again, under the entering state d_stitch->in_quote_in, iff
        the DEVICE word d_stitch->reemit is 1 when the kernel starts (the first pass ran with another state than the
        true one — a wrong speculation or a wrong ENTER_GUESS, either way round); otherwise it returns at once."""
        _check(lib().csvsimd_stage1_reemit_device_async(self._h, dbuf, length, base_off, d_stitch, dtape or None,
                                                        tape_cap, d_result, stream or None))

    def stage1_index_device_dialect_async(self, dialect: Dialect, dbuf: int, length: int, base_off: int,
                                          in_quote_in: int, dtape: int, tape_cap: int, d_result: int,
                                          stream: int = 0) -> None:
        _check(lib().csvsimd_stage1_index_device_dialect_async(self._h, C.byref(dialect), dbuf, length, base_off,
                                                               in_quote_in, dtape or None, tape_cap, d_result,
                                                               stream or None))

    def hbm_probe_device(self, dbuf: int, length: int, dout: int, write_per16: int = 0, stream: int = 0,
                         warmup: int = 2, iters: int = 10, blocks_per_cu: int = 4) -> float:
        """ms per pass of the bare HBM stream: write_per16 bytes written per 16 read (0, 4 or 25)."""
        ms = C.c_float()
        _check(lib().csvsimd_hbm_probe_device(self._h, dbuf, length, dout, write_per16, blocks_per_cu, stream or None,
                                              warmup, iters, C.byref(ms)))
        return float(ms.value)

    def copy_probe_device(self, dsrc: int, ddst: int, length: int, mode: int = 1, stream: int = 0, warmup: int = 2,
                          iters: int = 10) -> float:
        """ms per plain copy of `length` bytes: mode 0 = hipMemcpyDtoDAsync, 1 = 16 B per thread, 2 = the same, non-temporal."""
        ms = C.c_float()
        _check(lib().csvsimd_copy_probe_device(self._h, dsrc, ddst, length, mode, stream or None, warmup, iters,
                                               C.byref(ms)))
        return float(ms.value)

    def utf8_validate_device(self, dbuf: int, length: int, stream: int = 0) -> Optional[int]:
        """None if dbuf[0..length) is valid UTF-8, else the offset of the first offending byte."""
        r = Utf8Result()
        _check(lib().csvsimd_utf8_validate_device(self._h, dbuf or None, length, C.byref(r), stream or None))
        return None if r.first_invalid == UTF8_VALID else int(r.first_invalid)

    def utf8_validate_device_async(self, dbuf: int, length: int, d_result: int, stream: int = 0) -> None:
        _check(lib().csvsimd_utf8_validate_device_async(self._h, dbuf or None, length, d_result, stream or None))

    def stage1_time_device(self, dbuf: int, length: int, dtape: int, tape_cap: int, d_result: int,
                           stream: int = 0, warmup: int = 2, iters: int = 10) -> float:
        ms = C.c_float()
        _check(lib().csvsimd_stage1_time_device(self._h, dbuf, length, dtape or None, tape_cap, d_result,
                                                stream or None, warmup, iters, C.byref(ms)))
        return float(ms.value)

    # ---- reader::read(&Mmap) -> StructureIndex (src/reader.rs:150) ---------------------------
    def read(self, data) -> np.ndarray:
        """Host bytes -> index (uint64 array, [0] == 0 sentinel).  One pass with a capacity guess
        (one entry per 8 bytes), one exact retry if the file is denser — the protocol of
        include/csvsimd.h, same as the Rust binding."""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a, dtype=np.uint8)
        n = C.c_uint64()
        ptr = a.ctypes.data if a.size else None
        out = np.empty(a.size // 8 + 64, dtype=np.uint64)
        rc = lib().csvsimd_stage1_index(self._h, ptr, a.size, out.ctypes.data, out.size, C.byref(n), None)
        if rc == ERR_TAPE_CAPACITY:
            out = np.empty(n.value, dtype=np.uint64)
            rc = lib().csvsimd_stage1_index(self._h, ptr, a.size, out.ctypes.data, out.size, C.byref(n), None)
        _check(rc)
        return out[: n.value].copy()

    def read_dialect(self, data, dialect: Dialect) -> np.ndarray:
        """read() for another delimiter / quote / escape byte (extension, see include/csvsimd.h)."""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a, dtype=np.uint8)
        n = C.c_uint64()
        ptr = a.ctypes.data if a.size else None
        out = np.empty(a.size // 8 + 64, dtype=np.uint64)
        fn = lib().csvsimd_stage1_index_dialect
        rc = fn(self._h, C.byref(dialect), ptr, a.size, out.ctypes.data, out.size, C.byref(n), None)
        if rc == ERR_TAPE_CAPACITY:
            out = np.empty(n.value, dtype=np.uint64)
            rc = fn(self._h, C.byref(dialect), ptr, a.size, out.ctypes.data, out.size, C.byref(n), None)
        _check(rc)
        return out[: n.value].copy()

    def read_into(self, data: np.ndarray, tape: np.ndarray) -> Tuple[int, int, int]:
        """Raw csvsimd_stage1_index: returns (rc, tape_len, in_quote_out)."""
        n, q = C.c_uint64(), C.c_uint32()
        has_tape = tape is not None and tape.size > 0
        rc = lib().csvsimd_stage1_index(self._h, data.ctypes.data if data.size else None, data.size,
                                        tape.ctypes.data if has_tape else None,
                                        tape.size if has_tape else 0, C.byref(n), C.byref(q))
        return rc, n.value, q.value

    def read_many_into(self, items) -> int:
        """Raw csvsimd_stage1_index_batch on a prepared (HostBatchItem * n) array: returns rc; outputs are in the items."""
        return lib().csvsimd_stage1_index_batch(self._h, items, len(items))

    def read_many(self, datas, caps=None):
        """reader::read for MANY host buffers in one call (csvsimd_stage1_index_batch): -> list of StructureIndex arrays
        (uint64, sentinel first), one per buffer.  caps: tape capacity per buffer (default: one that always fits)."""
        arrs = [d if isinstance(d, np.ndarray) else np.frombuffer(bytes(d), dtype=np.uint8) for d in datas]
        arrs = [np.ascontiguousarray(a) for a in arrs]
        tapes = [np.empty((a.size + 1) if caps is None else caps[i], dtype=np.uint64) for i, a in enumerate(arrs)]
        items = (HostBatchItem * len(arrs))()
        for it, a, t in zip(items, arrs, tapes):
            it.buf, it.len, it.tape, it.tape_cap = (a.ctypes.data if a.size else None), a.size, (t.ctypes.data if t.size else None), t.size
        rc = self.read_many_into(items)
        if rc != OK and rc != ERR_TAPE_CAPACITY:
            _check(rc)
        out = []
        for it, t in zip(items, tapes):
            if it.status != OK and not (caps is not None and it.status == ERR_TAPE_CAPACITY):
                _check(it.status)
            out.append(t[: min(it.tape_len, t.size)].copy())
        self.last_batch = [(it.tape_len, it.in_quote_out, it.status) for it in items]
        return out

    # ---- csv_simd::create(filename) -> Tape (src/lib.rs:61-74) ---------------------------------
    def create(self, filename: str) -> "Tape":
        h = C.c_void_p()
        _check(lib().csvsimd_create(self._h, os.fsencode(filename), C.byref(h)))
        return Tape(h)


class Comm:
    """Native RCCL communicator for csvsimd_stage1_index_sharded (one per rank)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().csvsimd_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        h = C.c_void_p()
        _check(lib().csvsimd_comm_create(C.create_string_buffer(unique_id, 128), rank, world, device, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    def close(self) -> None:
        if self._h:
            lib().csvsimd_comm_destroy(self._h)
            self._h = None

    def index_sharded(self, ctx: "Context", dbuf: int, length: int, base_off: int, dtape: int, tape_cap: int,
                      file_in_quote_in: int = 0, stream: int = 0) -> Tuple[ShardResult, Stitch]:
        r, st = ShardResult(), Stitch()
        _check(lib().csvsimd_stage1_index_sharded(ctx._h, self._h, dbuf, length, base_off, file_in_quote_in,
                                                  dtape or None, tape_cap, C.byref(r), C.byref(st), stream or None))
        return r, st


def ingest_chunk_plan(length: int):
    """The cuts csvsimd_stage1_index streams a buffer of `length` bytes in: [0, ..., length]."""
    n = C.c_uint64()
    lib().csvsimd_ingest_chunk_plan(length, None, 0, C.byref(n))
    cuts = (C.c_uint64 * n.value)()
    _check(lib().csvsimd_ingest_chunk_plan(length, cuts, n.value, C.byref(n)))
    return list(cuts)


def ingest_last_phases() -> dict:
    """Phase times of this thread's latest host-buffer call (Context.read / read_into / create)."""
    p = IngestPhases()
    _check(lib().csvsimd_ingest_last_phases(C.byref(p)))
    return p.as_dict()


def multi_shard_range(length: int, n_shards: int, i: int) -> Tuple[int, int]:
    b, e = C.c_uint64(), C.c_uint64()
    _check(lib().csvsimd_multi_shard_range(length, n_shards, i, C.byref(b), C.byref(e)))
    return b.value, e.value


def stage1_index_multi(data: np.ndarray, ctxs: Sequence["Context"], dbufs: Sequence[int], dtapes: Sequence[int],
                       tape_caps: Sequence[int], file_in_quote_in: int = 0):
    """One host buffer -> len(ctxs) shards, one per context (contexts may sit on different GPUs of this process):
    streamed to the devices concurrently, indexed, stitched; bytes and tapes stay on the devices.  Returns the filled
    MultiShard array (begin, end, result, stitch per shard)."""
    g = len(ctxs)
    arr = (MultiShard * g)()
    for i in range(g):
        arr[i].ctx, arr[i].dbuf, arr[i].dtape, arr[i].tape_cap = ctxs[i]._h, dbufs[i] or None, dtapes[i] or None, tape_caps[i]
    _check(lib().csvsimd_stage1_index_multi(data.ctypes.data if data.size else None, data.size, arr, g, file_in_quote_in))
    return arr


def stage1_bound(length: int) -> int:
    n = C.c_uint64()
    _check(lib().csvsimd_stage1_bound(length, C.byref(n)))
    return n.value


def stitch_shards(results: Sequence[ShardResult], rank: int, file_in_quote_in: int = 0) -> Stitch:
    arr = (ShardResult * len(results))(*results)
    out = Stitch()
    _check(lib().csvsimd_stitch_shards(arr, len(results), rank, file_in_quote_in, C.byref(out)))
    return out


def stitch_shards_device_async(d_results: int, n_shards: int, rank: int, file_in_quote_in: int, d_stitch: int,
                               stream: int = 0) -> None:
    """csvsimd_stitch_shards as a one-lane kernel: device records in, device csvsimd_stitch out."""
    _check(lib().csvsimd_stitch_shards_device_async(d_results, n_shards, rank, file_in_quote_in, d_stitch,
                                                    stream or None))


def stage1_kernel_name(emit: bool = True, dialect: Optional[Dialect] = None) -> str:
    return lib().csvsimd_stage1_kernel_name(1 if emit else 0, C.byref(dialect) if dialect is not None else None).decode()


def build_has_probes() -> bool:
    return bool(lib().csvsimd_build_has_probes())


def boundaries(task_size: int, job_count: int) -> Optional[List[Tuple[int, int]]]:
    """reference ``boundaries`` (src/tape.rs:385-428): list of (start, len) or None."""
    out = (_Boundary * max(job_count, 1))()
    n = C.c_uint32()
    rc = lib().csvsimd_boundaries(task_size, job_count, out, C.byref(n))
    if rc == ERR_INVALID_STATE:
        return None
    _check(rc)
    return [(out[i].start, out[i].len) for i in range(n.value)]


class Tape:
    """reference ``Tape`` (src/tape.rs:74-153) + ``RecordSource`` (src/record_source.rs)."""

    def __init__(self, handle: C.c_void_p, keepalive=None):
        self._h = handle
        self._keep = keepalive

    @classmethod
    def from_index(cls, data: np.ndarray, index: np.ndarray) -> "Tape":
        """TapeCore::create + Tape::from_core on a finished index (borrows both arrays)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        index = np.ascontiguousarray(index, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().csvsimd_tape_create(data.ctypes.data, data.size, index.ctypes.data, index.size, C.byref(h)))
        return cls(h, keepalive=(data, index))

    def close(self) -> None:
        if self._h:
            lib().csvsimd_tape_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def field_cnt(self) -> int:
        return lib().csvsimd_tape_field_cnt(self._h)

    @property
    def record_cnt(self) -> int:
        return lib().csvsimd_tape_record_cnt(self._h)

    @property
    def record_jump_size(self) -> int:
        return lib().csvsimd_tape_record_jump_size(self._h)

    @property
    def record_offset(self) -> int:
        return lib().csvsimd_tape_record_offset(self._h)

    @property
    def new_line(self) -> str:
        return "CRLF" if lib().csvsimd_tape_new_line(self._h) == NEWLINE_CRLF else "LF"

    def header(self) -> List[str]:
        names = []
        for i in range(self.field_cnt):
            n = lib().csvsimd_tape_header_name(self._h, i, None, 0)
            buf = C.create_string_buffer(int(n) + 1)
            lib().csvsimd_tape_header_name(self._h, i, buf, n)
            names.append(buf.raw[:n].decode("utf-8", "replace"))
        return names

    def index(self) -> np.ndarray:
        n = C.c_uint64()
        p = lib().csvsimd_tape_index(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def index_view(self) -> np.ndarray:
        """The tape's own index, not copied: valid until close()."""
        n = C.c_uint64()
        p = lib().csvsimd_tape_index(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def bytes(self) -> bytes:
        if getattr(self, "_bytes", None) is None:  # one copy, cached: seek_* slice it
            n = C.c_uint64()
            p = lib().csvsimd_tape_bytes(self._h, C.byref(n))
            self._bytes = C.string_at(p, n.value)
        return self._bytes

    def _span(self, fn, *idx) -> Optional[Tuple[int, int]]:
        b, e, f = C.c_uint64(), C.c_uint64(), C.c_int()
        _check(fn(self._h, *idx, C.byref(b), C.byref(e), C.byref(f)))
        return (b.value, e.value) if f.value else None

    def seek_record(self, record_idx: int) -> Optional[bytes]:
        s = self._span(lib().csvsimd_tape_seek_record, record_idx)
        return None if s is None else self.bytes()[s[0]: s[1]]

    def seek_field(self, record_idx: int, field_idx: int) -> Optional[bytes]:
        s = self._span(lib().csvsimd_tape_seek_field, record_idx, field_idx)
        return None if s is None else self.bytes()[s[0]: s[1]]

    def chunks(self, num: int) -> List[Tuple[int, int, int, int]]:
        """Tape::chunks (src/tape.rs:95-140): [(id, start, end, record_cnt)] in index-key units."""
        out = (_Chunk * max(num, 1))()
        n = C.c_uint32()
        _check(lib().csvsimd_tape_chunks(self._h, num, out, C.byref(n)))
        return [(out[i].id, out[i].start, out[i].end, out[i].record_cnt) for i in range(n.value)]


# ---- device-side consumers of a device-resident tape (raw device addresses) ----------------------
def tape_field_spans_device(dindex: int, index_len: int, field_cnt: int, new_line: str, field_idx: int,
                            first_record: int, n_records: int, d_begin: int, d_end: int, stream: int = 0) -> int:
    """Bulk seek_field on the GPU; returns how many of the requested records exist."""
    n = C.c_uint64()
    _check(lib().csvsimd_tape_field_spans_device(dindex, index_len, field_cnt,
                                                 NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, field_idx,
                                                 first_record, n_records, d_begin or None, d_end or None,
                                                 C.byref(n), stream or None))
    return n.value


def tape_record_spans_device(dindex: int, index_len: int, field_cnt: int, new_line: str, first_record: int,
                             n_records: int, d_begin: int, d_end: int, stream: int = 0) -> int:
    """Bulk seek_record on the GPU; returns how many of the requested records exist."""
    n = C.c_uint64()
    _check(lib().csvsimd_tape_record_spans_device(dindex, index_len, field_cnt,
                                                  NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, first_record,
                                                  n_records, d_begin or None, d_end or None, C.byref(n),
                                                  stream or None))
    return n.value


def gather_fields_device(dbytes: int, bytes_len: int, d_begin: int, d_end: int, n_records: int, d_dst: int,
                         stride: int, d_len: int = 0, stream: int = 0) -> None:
    _check(lib().csvsimd_gather_fields_device(dbytes, bytes_len, d_begin, d_end, n_records, d_dst, stride,
                                              d_len or None, stream or None))


def _chunk(c) -> _Chunk:
    """(id, start, end, record_cnt) as Tape.chunks returns it -> csvsimd_chunk"""
    return c if isinstance(c, _Chunk) else _Chunk(*c)


def chunk_field_spans_device(dindex: int, index_len: int, field_cnt: int, new_line: str, chunk, field_idx: int,
                             d_begin: int, d_end: int, stream: int = 0) -> int:
    """Bulk seek_field over one csvsimd_chunk (Tape::chunks' unit of parallel work); returns its record count."""
    n = C.c_uint64()
    ch = _chunk(chunk)
    _check(lib().csvsimd_chunk_field_spans_device(dindex, index_len, field_cnt,
                                                  NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, C.byref(ch),
                                                  field_idx, d_begin or None, d_end or None, C.byref(n),
                                                  stream or None))
    return n.value


def column_frequency_scratch_bytes(n_records: int, n_chunks: int = 1, max_field_bytes: int = 32) -> int:
    return lib().csvsimd_column_frequency_scratch_bytes(n_records, n_chunks, max_field_bytes)


def column_frequency_device(ctx: "Context", dbytes: int, dindex: int, index_len: int, field_cnt: int, new_line: str,
                            chunks, field_idx: int, d_scratch: int, scratch_bytes: int, d_entries: int,
                            entries_cap: int, stream: int = 0, allow_capacity: bool = False) -> FreqStatus:
    """Exact frequency count of a column of the row-major file over the given chunks; entries (first_record, begin, end,
    count as 4 x uint64) land in d_entries, status.n_distinct of them.  scratch_bytes: column_frequency_scratch_bytes(
    records, chunks, longest field) — a scratch too small for the longest field is reported (ERR_TAPE_CAPACITY) with
    status.max_field_bytes set."""
    arr = (_Chunk * len(chunks))(*[_chunk(c) for c in chunks])
    st = FreqStatus()
    rc = lib().csvsimd_column_frequency_device(ctx._h, dbytes, dindex, index_len, field_cnt,
                                               NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, arr, len(chunks),
                                               field_idx, d_scratch, scratch_bytes, d_entries or None, entries_cap,
                                               C.byref(st), stream or None)
    if not (rc == ERR_TAPE_CAPACITY and allow_capacity):
        _check(rc)
    return st


def column_search_device(ctx: "Context", dbytes: int, bytes_len: int, dindex: int, index_len: int, field_cnt: int,
                         new_line: str, chunk, field_idx: int, needle: bytes, mode: int, d_bitmap: int,
                         stream: int = 0) -> int:
    """Bitmap of the chunk's records whose field equals / starts with / contains `needle`; returns the match count."""
    n = C.c_uint64()
    ch = _chunk(chunk)
    _check(lib().csvsimd_column_search_device(ctx._h, dbytes, bytes_len, dindex, index_len, field_cnt,
                                              NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, C.byref(ch),
                                              field_idx, needle, len(needle), mode, d_bitmap or None, C.byref(n),
                                              stream or None))
    return n.value


def chunk_to_columns_device(ctx: "Context", dbytes: int, bytes_len: int, dindex: int, index_len: int, field_cnt: int,
                            new_line: str, chunk, fields, d_cols: int, stride: int, d_lens: int = 0,
                            stream: int = 0) -> int:
    """Row-major CSV -> columns in one pass over the chunk's bytes and tape: field fields[c] of the chunk's i-th record
    lands in d_cols[(c * n + i) * stride ...) (truncated, zero padded), its length in d_lens[c * n + i] (uint32).
    fields = list of field ids, or None for all columns.  Returns n = the chunk's record count.  Asynchronous."""
    n = C.c_uint64()
    ch = _chunk(chunk)
    if fields is None:
        arr, nf = None, 0
    else:
        nf = len(fields)
        arr = (C.c_uint32 * max(nf, 1))(*fields)
    _check(lib().csvsimd_chunk_to_columns_device(ctx._h, dbytes, bytes_len, dindex, index_len, field_cnt,
                                                 NEWLINE_CRLF if new_line == "CRLF" else NEWLINE_LF, C.byref(ch), arr, nf,
                                                 d_cols or None, stride, d_lens or None, C.byref(n), stream or None))
    return n.value


def columnar_frequency_scratch_bytes(n_records: int) -> int:
    return lib().csvsimd_columnar_frequency_scratch_bytes(n_records)


def columnar_frequency_device(ctx: "Context", d_col: int, d_len: int, n_records: int, stride: int, first_record: int,
                              d_scratch: int, scratch_bytes: int, d_entries: int, entries_cap: int, stream: int = 0,
                              allow_capacity: bool = False) -> ColFreqStatus:
    """Exact frequency count of one column of a columnar copy; entries (first_record, count as 2 x uint64) land in
    d_entries, status.n_distinct of them.  Synchronous (the status comes back)."""
    st = ColFreqStatus()
    rc = lib().csvsimd_columnar_frequency_device(ctx._h, d_col or None, d_len or None, n_records, stride, first_record,
                                                 d_scratch, scratch_bytes, d_entries or None, entries_cap, C.byref(st),
                                                 stream or None)
    if not (rc == ERR_TAPE_CAPACITY and allow_capacity):
        _check(rc)
    return st


def columnar_frequency_device_async(ctx: "Context", d_col: int, d_len: int, n_records: int, stride: int, first_record: int,
                                    d_scratch: int, scratch_bytes: int, d_entries: int, entries_cap: int, d_status: int,
                                    stream: int = 0) -> None:
    """The same as two launches on `stream` (three from 4 Mi records on) and nothing else: the 32-byte status record (n_records, n_distinct, truncated,
    overflow as 4 x uint64) is written to DEVICE memory at d_status; capturable into a graph."""
    _check(lib().csvsimd_columnar_frequency_device_async(ctx._h, d_col or None, d_len or None, n_records, stride,
                                                         first_record, d_scratch, scratch_bytes, d_entries or None,
                                                         entries_cap, d_status, stream or None))


def columnar_search_device(ctx: "Context", d_col: int, d_len: int, n_records: int, stride: int, needle: bytes, mode: int,
                           d_bitmap: int, stream: int = 0) -> int:
    """column_search_device on a column of a columnar copy; returns the match count."""
    n = C.c_uint64()
    _check(lib().csvsimd_columnar_search_device(ctx._h, d_col or None, d_len or None, n_records, stride, needle,
                                                len(needle), mode, d_bitmap or None, C.byref(n), stream or None))
    return n.value


def bitmap_select_scratch_bytes(n_rows: int) -> int:
    return lib().csvsimd_bitmap_select_scratch_bytes(n_rows)


def bitmap_select_device(d_bitmap: int, n_rows: int, first_record: int, d_scratch: int, d_out: int, out_cap: int,
                         stream: int = 0) -> int:
    n = C.c_uint64()
    _check(lib().csvsimd_bitmap_select_device(d_bitmap or None, n_rows, first_record, d_scratch or None, d_out or None,
                                              out_cap, C.byref(n), stream or None))
    return n.value


# ---- device utilities (raw device addresses; torch tensors' .data_ptr() fit) --------------------
def trim_spans_device(dbytes: int, d_begin: int, d_end: int, n_records: int, flags: int = TRIM_SPACE,
                      quote: int = 0x22, stream: int = 0) -> None:
    _check(lib().csvsimd_trim_spans_device(dbytes, d_begin, d_end, n_records, flags, quote, stream or None))


def synth_fill_device(dbuf: int, file_off: int, length: int, cols: int, width: int, seed: int,
                      quote_pct: int = 0, stream: int = 0) -> None:
    _check(lib().csvsimd_synth_fill_device(dbuf, file_off, length, cols, width, seed, quote_pct, stream or None))


def tape_checksum_device(dtape: int, n: int, first_index: int, d_out: int, stream: int = 0) -> None:
    _check(lib().csvsimd_tape_checksum_device(dtape or None, n, first_index, d_out, stream or None))


# the synthetic corpora of SURVEY.md §8d / BASELINE.md: name -> (cols, width, seed, quote_pct)
WORKLOADS = {
    "16x32_noquote": (16, 32, 0xC5F00002, 0),
    "16x32_q10": (16, 32, 0xC5F00003, 10),
    "64x31_noquote": (64, 31, 0xC5F00004, 0),
    "64x31_q10": (64, 31, 0xC5F00004, 10),
    "1024x4_dense": (1024, 4, 0xC5F00005, 0),
}


def workload_len(name: str, target_bytes: int) -> int:
    """Whole rows only: the largest multiple of the row size that fits target_bytes."""
    cols, width, _, _ = WORKLOADS[name]
    row = cols * (width + 1)
    return (target_bytes // row) * row
