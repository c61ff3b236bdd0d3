"""The oracle against the reference's golden material (CPU only).

Pins oracle/oracle.c: both restatements (scalar definition, SSE instruction-level) must
reproduce the reference's known-answer test (src/reader.rs:318-327), the full fixture vectors
(tests/golden/expected.json) and each other on randomized inputs."""
import numpy as np
import pytest

from conftest import random_csvish


def test_reference_known_answer_mk_index(oracle, golden):
    # reference src/reader.rs:325-326: index[1] == 4, index[last] == 95 on reader_test01.csv
    data, _ = golden["reader_test01.csv"]
    for idx in (oracle.sse_read(data), oracle.scalar_read(data)):
        assert idx[0] == 0 and idx[1] == 4 and idx[-1] == 95


@pytest.mark.parametrize("name", ["reader_test01.csv", "sample.csv", "sample_rx.csv"])
def test_golden_vectors(oracle, golden, name):
    data, exp = golden[name]
    want = np.array(exp["index"], dtype=np.uint64)
    assert len(data) == exp["len"]
    assert np.array_equal(oracle.sse_read(data), want)
    assert np.array_equal(oracle.scalar_read(data), want)


def test_class_table_all_bytes(oracle):
    # legend src/stage1.rs:41-48: newline 1 (0x0a, 0x0d), comma 2, space 4, escape 8, quote 16
    want = {0x0A: 1, 0x0D: 1, 0x2C: 2, 0x20: 4, 0x5C: 8, 0x22: 16}
    for b in range(256):
        assert oracle.lib().oracle_byte_class(b) == want.get(b, 0), hex(b)


# The reference holds exactly two more known answers for this path besides src/reader.rs:325-326 — both in
# prose, not in a test: the worked string-mask example and the two rows of the class table.
REF_QUOTES, REF_STRING_MASK = 0b100010000, 0b011110000   # src/avx/stage1.rs:350-352, design_notes_1.md:90-91
REF_LO_ROW = [4, 0, 16, 0, 0, 0, 0, 0, 0, 0, 1, 0, 10, 1, 0, 0]   # design_notes_1.md:130-132 "low nibble ... encode"
REF_HI_ROW = [1, 0, 22, 0, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]    # design_notes_1.md:136-138 "high nibble ... encode"


def known_answer_block():
    """64 bytes whose quote bits are REF_QUOTES (quotes at bytes 4 and 8) and whose every other byte is a
    comma: the tape shows the string mask directly — bytes 5, 6, 7 are inside the string, byte 9 is not."""
    b = np.full(64, 0x2C, dtype=np.uint8)
    b[4] = b[8] = 0x22
    want = np.array([i for i in range(64) if not (REF_STRING_MASK >> i) & 1 and i != 8], dtype=np.uint64)
    return b, want


def test_reference_worked_example_string_mask(oracle):
    L = oracle.lib()
    assert L.oracle_string_mask_clmul(REF_QUOTES) == REF_STRING_MASK      # what the reference executes
    assert L.oracle_string_mask_loop(REF_QUOTES) == REF_STRING_MASK       # the recipe in the notes
    rng = np.random.default_rng(3)
    for q in [0, 1, 1 << 63, (1 << 64) - 1] + [int(x) for x in rng.integers(0, 2**63, size=200, dtype=np.uint64)]:
        assert L.oracle_string_mask_clmul(q) == L.oracle_string_mask_loop(q)
    # ... and through both restatements of the whole path
    block, want = known_answer_block()
    assert np.array_equal(oracle.scalar_index(block)[0], want)
    assert np.array_equal(oracle.sse_read(block)[1:], want)


def test_reference_class_table_rows(oracle):
    for b in range(256):
        assert oracle.lib().oracle_byte_class(b) == (REF_LO_ROW[b & 15] & REF_HI_ROW[b >> 4]), hex(b)


def test_scalar_equals_sse_random(oracle):
    rng = np.random.default_rng(1234)
    lengths = list(range(64, 64 + 130)) + [255, 256, 257, 1000, 4095, 4096, 4097, 10000]
    for n in lengths:
        for pq in (None, 0.02, 0.3):
            d = random_csvish(rng, n, pq)
            assert np.array_equal(oracle.sse_read(d), oracle.scalar_read(d)), (n, pq)


def test_sse_short_inputs_policy(oracle):
    # < 64 bytes is outside the reference's defined domain (UB/panic); the oracle
    # (and the GPU library) define it by the blocking-independent semantics.
    rng = np.random.default_rng(5)
    for n in range(0, 64):
        d = random_csvish(rng, n)
        assert np.array_equal(oracle.sse_read(d), oracle.scalar_read(d)), n


def test_sse_ignores_unaligned_head(oracle):
    # reader.rs:180-181: align_to::<__m128>() head bytes are never processed and offsets are
    # relative to the aligned body.  mmap never has a head; the restatement still models it.
    rng = np.random.default_rng(6)
    d = random_csvish(rng, 500, 0.0)
    for head in (1, 7, 15):
        got = oracle.sse_read(d, head=head)
        skip = 16 - head
        want = oracle.scalar_read(d[skip:])
        assert np.array_equal(got, want), head


def test_quote_semantics_examples(oracle):
    # inclusive prefix-xor: opening quote is "inside", closing quote is "outside" (avx/stage1.rs:342-407)
    pad = b"x" * 64
    cases = {
        b'a,"b,c",d\n': [1, 7, 9],
        b'"a""b",c\n': [6, 8],          # doubled quote toggles twice
        b'a\\,b\n': [2, 4],              # backslash escapes nothing (class 8 unused)
        b'"\n,\r"\r\n': [5, 6],
    }
    for text, want in cases.items():
        idx = oracle.scalar_read(text + pad)
        assert idx[0] == 0 and list(idx[1:]) == want, text
        assert np.array_equal(oracle.sse_read(text + pad), idx)


def test_shard_descriptor_and_carry(oracle):
    rng = np.random.default_rng(7)
    d = random_csvish(rng, 5000, 0.05)
    full, inq = oracle.scalar_index(d)
    for cut in (1, 63, 64, 1000, 4999):
        a, qa = oracle.scalar_index(d[:cut])
        b, qb = oracle.scalar_index(d[cut:], base_off=cut, in_quote_in=qa)
        assert np.array_equal(np.concatenate([a, b]), full) and qb == inq
        p, c0, c1 = oracle.shard_descriptor(d[cut:])
        e0, _ = oracle.scalar_index(d[cut:], in_quote_in=0)
        e1, _ = oracle.scalar_index(d[cut:], in_quote_in=1)
        assert (c0, c1) == (e0.size, e1.size) and p == (qa ^ qb)


def test_checksum_is_order_sensitive(oracle):
    t = np.arange(1, 1000, dtype=np.uint64) * 33
    a = oracle.tape_checksum(t, 1)
    t2 = t.copy()
    t2[[10, 11]] = t2[[11, 10]]
    assert oracle.tape_checksum(t2, 1) != a
    assert oracle.tape_checksum(t, 2) != a
    # additive over splits
    x = oracle.tape_checksum(t[:400], 1)
    y = oracle.tape_checksum(t[400:], 401)
    assert ((x[0] + y[0]) % 2**64, (x[1] + y[1]) % 2**64) == a


def test_synth_shapes(oracle, pkg):
    for name, (cols, width, seed, q) in pkg.WORKLOADS.items():
        row = cols * (width + 1)
        n = 3 * row
        d = oracle.synth(0, n, cols, width, seed, q)
        assert d[row - 1] == 0x0A and d[n - 1] == 0x0A
        # any sub-range reproduces the same bytes (counter-based)
        sub = oracle.synth(row - 5, 40, cols, width, seed, q)
        assert np.array_equal(sub, d[row - 5: row + 35])
        idx = oracle.scalar_read(d)
        if q == 0:
            # analytic tape of the no-quote corpora: every (width+1)-th byte
            want = np.arange(1, n // (width + 1) + 1, dtype=np.uint64) * (width + 1) - 1
            assert np.array_equal(idx[1:], want), name
        else:
            assert idx.size - 1 == n // (width + 1), name  # quoted ',' and LF are hidden, field count unchanged
    # quoting really happens and hides a comma + LF
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    d = oracle.synth(0, 200 * cols * (width + 1), cols, width, seed, q)
    assert (d == 0x22).sum() > 0 and (d == 0x22).sum() % 2 == 0
    assert ((d == 0x2C) | (d == 0x0A)).sum() > oracle.scalar_read(d).size - 1


def test_mt_flavour_equals_scalar(oracle):
    # the multi-threaded baseline (not a reference behaviour) must produce the same tape
    rng = np.random.default_rng(77)
    for n in (64, 1000, 100_003):
        for pq in (0.0, 0.05):
            d = random_csvish(rng, n, pq)
            want = oracle.scalar_read(d)
            for threads in (1, 2, 3, 8):
                assert np.array_equal(oracle.sse_read_mt(d, threads), want), (n, pq, threads)


def test_dialect_default_equals_reference_definition(oracle):
    # the extension's scalar definition with (',', '"', no escape) is the reference's semantics
    rng = np.random.default_rng(99)
    for n in (0, 1, 64, 1000, 50000):
        d = random_csvish(rng, n, 0.1)
        for inq in (0, 1):
            a, qa = oracle.scalar_index(d, base_off=5, in_quote_in=inq)
            b, qb, e = oracle.dialect_index(d, base_off=5, in_quote_in=inq)
            assert np.array_equal(a, b) and qa == qb and e == 0


def test_dialect_against_python_csv_module(oracle):
    # independent check of the extension's semantics: files written by Python's csv module in a
    # ';' / "'" / backslash dialect; every record the csv module reads back must contribute
    # exactly len(row) structural bytes (one per field end), and the fields must sit between them
    import csv
    import io
    rng = np.random.default_rng(7)
    pool = ["a", "bc", ";", "'", "\\", "x;y", "it's", "", " ", "q\\r", "1;2;3", "''"]
    for quoting in (csv.QUOTE_MINIMAL, csv.QUOTE_NONE, csv.QUOTE_ALL):
        rows = [[pool[i] + (pool[j] if k % 3 == 0 else "")
                 for k, (i, j) in enumerate(rng.integers(0, len(pool), size=(int(rng.integers(1, 9)), 2)))]
                for _ in range(300)]
        buf = io.StringIO()
        kw = dict(delimiter=";", quotechar="'", escapechar="\\", doublequote=False, lineterminator="\n",
                  quoting=quoting)
        csv.writer(buf, **kw).writerows(rows)
        text = buf.getvalue()
        back = list(csv.reader(io.StringIO(text), **kw))
        assert back == rows                      # the csv module agrees with itself
        data = text.encode()
        idx, inq, esc = oracle.dialect_index(data, ord(";"), ord("'"), ord("\\"))
        assert (inq, esc) == (0, 0)
        assert idx.size == sum(len(r) for r in rows)
        # rows end where the csv module says they end
        ends = [i for i in idx.tolist() if data[i] == 0x0A]
        assert len(ends) == len(rows) and ends[-1] == len(data) - 1
        # and the raw text of each field, un-escaped and un-quoted, is the field
        prev = -1
        flat = [f for r in rows for f in r]
        for pos, want in zip(idx.tolist(), flat):
            raw = data[prev + 1: pos].decode()
            if raw[:1] == "'" and quoting != csv.QUOTE_NONE:
                raw = raw[1:-1]
            out, k = [], 0
            while k < len(raw):
                if raw[k] == "\\":
                    k += 1
                out.append(raw[k])
                k += 1
            assert "".join(out) == want
            prev = pos


def _py_utf8_first_invalid(b: bytes):
    try:
        b.decode("utf-8")
        return None
    except UnicodeDecodeError as e:
        return e.start


def test_utf8_oracle_against_cpython_decoder(oracle):
    # pins the extension's UTF-8 oracle on an independent implementation (CPython's decoder)
    cases = [b"", b"plain ascii, \"quoted\"\n", "héllo,世界,\U0001F600\n".encode(), b"\x80", b"\xbf",
             b"\xc0\x80", b"\xc1\xbf", b"\xc2", b"a\xc2", b"\xc2\x41", b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xe0\xa0\x80",
             b"\xed\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf", b"\xee\x80\x80", b"\xef\xbf\xbf", b"\xe2\x82",
             b"ab\xe2\x82", b"\xe2\x28\xa1", b"\xe2\x82\x28", b"\xf0\x80\x80\x80", b"\xf0\x8f\xbf\xbf",
             b"\xf0\x90\x80\x80", b"\xf4\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"\xff", b"\xfe",
             b"a\xf0\x90\x80a", b"\xf0\x90\x80", b"\xf0\x90", b"\xf0", b"\xf1\x80\x80\x80\x80", b"\xc2\x80\x80",
             b"\xef\xbb\xbfName,Number\r\n"]
    for c in cases:
        assert oracle.utf8_first_invalid(c) == _py_utf8_first_invalid(c), c
    rng = np.random.default_rng(11)
    pool = np.frombuffer(b"a,\n\"\x7f\x80\x8f\x90\x9f\xa0\xbf\xc0\xc1\xc2\xdf\xe0\xe1\xec\xed\xee\xef\xf0\xf1\xf3\xf4\xf5\xff",
                         dtype=np.uint8)
    for _ in range(4000):
        b = pool[rng.integers(0, pool.size, size=int(rng.integers(0, 12)))].tobytes()
        assert oracle.utf8_first_invalid(b) == _py_utf8_first_invalid(b), b
    # valid text with one corrupted byte
    text = ("naïve,東京,\U0001F680 café\n" * 50).encode()
    for _ in range(300):
        m = bytearray(text)
        m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        assert oracle.utf8_first_invalid(bytes(m)) == _py_utf8_first_invalid(bytes(m))


def test_trim_span_oracle(oracle):
    data = b'  "a b"  ,x,   ,"",\' q \''
    b, e = oracle.trim_spans(data, [0, 10, 12, 16, 19], [9, 11, 15, 18, 24], 3)
    assert [data[i:j] for i, j in zip(b.tolist(), e.tolist())] == [b"a b", b"x", b"", b"", b"' q '"]
    b, e = oracle.trim_spans(data, [0, 19], [9, 24], 1)
    assert [data[i:j] for i, j in zip(b.tolist(), e.tolist())] == [b'"a b"', b"' q '"]
    b, e = oracle.trim_spans(data, [19], [24], 2, quote=0x27)
    assert data[int(b[0]):int(e[0])] == b" q "
