"""Host-side tape / record access through the C ABI (reference src/tape.rs, src/record_source.rs).
CPU only: the index comes from the golden vectors, no stage-1 compute is invoked."""
import numpy as np
import pytest


def tape_of(pkg, golden, name):
    data, exp = golden[name]
    return pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), np.array(exp["index"], dtype=np.uint64)), data, exp


def test_boundaries_doc_test(pkg):
    # reference doc-test src/tape.rs:362-384
    r = pkg.boundaries(8, 3)
    assert r == [(0, 3), (3, 3), (6, 2)] and sum(l for _, l in r) == 8
    r = pkg.boundaries(1000, 12)
    assert r[0] == (0, 84) and r[1] == (84, 84) and r[11] == (917, 83) and sum(l for _, l in r) == 1000
    r = pkg.boundaries(8, 12)
    assert r[0] == (0, 8) and sum(l for _, l in r) == 8
    assert pkg.boundaries(0, 3) is None
    assert pkg.boundaries(5, 0) is None


def test_sample_csv_tape(pkg, golden):
    # field_cnt 3, LF, jump 3, record_cnt 15, record_offset 18
    t, data, exp = tape_of(pkg, golden, "sample.csv")
    assert t.field_cnt == 3 and t.new_line == "LF" and t.record_jump_size == 3
    assert t.record_cnt == 15 and t.record_offset == 18
    assert t.header() == ["Name", "Number", "Done"]
    assert t.seek_record(0) == b'Edm nd,3, "o"' == data[19:32]
    assert t.seek_field(0, 0) == b"Edm nd" and t.seek_field(0, 1) == b"3" and t.seek_field(0, 2) == b' "o"'
    # record_idx + 1 >= record_cnt -> Ok(None) (src/record_source.rs:77-81)
    assert t.seek_record(13) is not None and t.seek_record(14) is None
    assert t.seek_field(0, 3) is None  # field_idx >= field_cnt (src/record_source.rs:117-119)
    lines = data.split(b"\n")
    for r in range(14):
        assert t.seek_record(r) == lines[r + 1]


def test_sample_rx_tape_bom_crlf_quoted(pkg, golden):
    t, data, exp = tape_of(pkg, golden, "sample_rx.csv")
    assert t.field_cnt == 8 and t.new_line == "CRLF" and t.record_jump_size == 9 and t.record_cnt == 8
    assert t.header()[0] == "NPI Number" and t.header()[-1] == "NRx Count"  # BOM skipped (tape.rs:241-249)
    assert t.seek_field(1, 2) == b'"INTERNAL MED, CARD. ELECTROPHYSIOLOGY"'   # quoted comma stays inside
    assert t.seek_field(0, 0) == b"1003002813" and t.seek_field(0, 7) == b"2"
    assert t.seek_field(6, 5) == b'"CASH,IT"'
    rows = data.split(b"\r\n")
    for r in range(7):
        assert t.seek_record(r) == rows[r + 1]
    assert t.seek_record(7) is None


def test_ragged_file_is_invalid_csv_format(pkg, golden):
    # reader_test01.csv: 16 structurals, field_cnt 3 -> 16 % 3 != 0 (src/tape.rs:327,342-344)
    data, exp = golden["reader_test01.csv"]
    assert exp["ragged"]
    with pytest.raises(pkg.StructureError) as e:
        pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), np.array(exp["index"], dtype=np.uint64))
    assert e.value.code == pkg.ERR_INVALID_CSV_FORMAT


def test_chunks(pkg, golden):
    # Tape::chunks (src/tape.rs:95-140): first chunk skips the header row
    t, _, _ = tape_of(pkg, golden, "sample.csv")
    ch = t.chunks(4)
    b = pkg.boundaries(15, 4)
    assert len(ch) == 4
    assert ch[0] == (0, 3, (b[0][0] + b[0][1]) * 3, b[0][1] - 1)
    for i in range(1, 4):
        assert ch[i] == (i, b[i][0] * 3, (b[i][0] + b[i][1]) * 3, b[i][1])
    with pytest.raises(pkg.StructureError) as e:
        t.chunks(0)
    assert e.value.code == pkg.ERR_INVALID_STATE


def test_header_without_following_byte_is_invalid_state(pkg):
    # the Rust indexes memmap[header_end_idx + 1] and panics; the C ABI reports InvalidState
    with pytest.raises(pkg.StructureError) as e:
        pkg.Tape.from_index(np.frombuffer(b"a,b,c", dtype=np.uint8), np.array([0, 1, 3], dtype=np.uint64))
    assert e.value.code == pkg.ERR_INVALID_STATE


def test_header_quirks_are_mirrored(pkg):
    # Header::new decides CRLF by looking at the byte AFTER the first line end (src/tape.rs:235-238):
    # an LF file whose second line is empty is therefore taken for CRLF — mirrored, not fixed.
    data = np.frombuffer(b"a,b\n\nx,y\n" + b"p" * 64, dtype=np.uint8)
    idx = np.array([0, 1, 3, 4, 6, 8], dtype=np.uint64)     # ',', LF, LF, ',', LF
    with pytest.raises(pkg.StructureError) as e:           # jump = 3 (CRLF), 5 % 3 != 0
        pkg.Tape.from_index(data, idx)
    assert e.value.code == pkg.ERR_INVALID_CSV_FORMAT
    # names are trimmed, the delimiter is a plain ',' even inside quotes (src/tape.rs:259-262)
    data = np.frombuffer(b'  x , "a,b" ,z\n1,2,3,4\n' + b"p" * 64, dtype=np.uint8)
    idx = np.array([0, 4, 8, 12, 14, 16, 18, 20, 22], dtype=np.uint64)   # hand-made: every ',' and LF
    t = pkg.Tape.from_index(data, idx)
    assert t.header() == ["x", '"a', 'b"', "z"] and t.field_cnt == 4 and t.record_cnt == 2
    assert t.seek_field(0, 3) == b"4" and t.seek_record(0) == b"1,2,3,4" and t.seek_record(1) is None
    # BOM skipping takes ANY leading run of ef/bb/bf bytes (src/tape.rs:241-249)
    data = np.frombuffer(b"\xbf\xef\xbbn,m\r\n1,2\r\n" + b"p" * 64, dtype=np.uint8)
    idx = np.array([0, 4, 6, 7, 9, 11, 12], dtype=np.uint64)
    t = pkg.Tape.from_index(data, idx)
    assert t.header() == ["n", "m"] and t.new_line == "CRLF" and t.record_jump_size == 3 and t.record_cnt == 2
    assert t.seek_field(0, 1) == b"2"


def test_ingest_chunk_plan_has_no_degenerate_chunk(pkg):
    # the host side owns the chunking (csvsimd_stage1_index): every plan tiles [0, len) in order, no chunk exceeds the
    # 32-MiB slot, and a multi-chunk plan holds no stub (148 MiB + 1 byte used to end in a 1-byte chunk that
    # still paid two launches and an event wait).  Round 5: the plan starts small (len / 32, 1 ... 4 MiB), doubles
    # up to full slots and halves down to len / 16 (1 ... 8 MiB): a call waits for the staging of its first chunk and
    # for the way back of its last one, everything between them overlaps.  No chunk below 1 MiB where the file allows it: a
    # copy's fixed cost (~18 us) hides behind another copy's transfer only if that lasts as long
    KiB, MiB = 1 << 10, 1 << 20
    rng = np.random.default_rng(4)
    lens = [0, 1, 300, MiB, MiB + 1, 2 * MiB + 77, 4 * MiB, 4 * MiB + 1, 12 * MiB + 1, 16 * MiB + 1, 64 * MiB + 1, 128 * MiB - 1, 128 * MiB, 128 * MiB + 1,
            148 * MiB + 1, 152 * MiB + 1, 136 * MiB + 1, 2048 * MiB, 2048 * MiB + 777]
    lens += [int(x) for x in rng.integers(1, 600 * MiB, 300)] + [int(x) for x in rng.integers(MiB, 40 * MiB, 100)]
    for n in lens:
        cuts = pkg.ingest_chunk_plan(n)
        assert cuts[0] == 0 and cuts[-1] == n, n
        sizes = [b - a for a, b in zip(cuts, cuts[1:])]
        assert all(s > 0 for s in sizes) and all(s <= 32 * MiB for s in sizes), (n, sizes)
        if len(sizes) > 1:
            assert min(sizes) >= 512 * KiB, (n, sizes)
            assert sizes[0] <= max(MiB, n // 32 + 64 * KiB) + 512 * KiB and sizes[0] <= 4 * MiB + 512 * KiB, (n, sizes)   # a short fill ...
            assert sizes[-1] <= max(MiB, n // 16 + 64 * KiB) * 1.25 + 512 * KiB and sizes[-1] <= 10 * MiB, (n, sizes)   # ... and a short drain
        # about log2 chunks on the way up and down, full slots between them
        assert len(sizes) <= 16 + n // (32 * MiB), (n, len(sizes))
    # a large file ramps up and down around full slots
    sizes = [b - a for a, b in zip(*(lambda c: (c, c[1:]))(pkg.ingest_chunk_plan(2048 * MiB)))]
    assert sizes[:4] == [4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB] and sizes[-2:] == [16 * MiB, 8 * MiB]
    sizes = [b - a for a, b in zip(*(lambda c: (c, c[1:]))(pkg.ingest_chunk_plan(32 * MiB)))]
    assert sizes[0] == MiB and sizes[-1] == 2 * MiB and sum(sizes) == 32 * MiB
