"""Device-side consumers of the tape against their scalar definitions in oracle/:
the reference's stated use of the tape is "frequency counts, and function search" (design_notes_1.md:1-4) over
the Chunk records of Tape::chunks (src/tape.rs:12-19, 95-140).  Everything is checked against
collections.Counter / == / startswith / `in` over seek_field restated in the oracle (src/record_source.rs:106-140)
— never against the product's own host mirror."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def make_csv(rng, rows, line_end, quoted=True):
    """A file with a low-cardinality column, a high-cardinality one, empty fields, quoted fields holding commas,
    and fields of every length 0..40."""
    cities = [b"Oslo", b"Bergen", b"", b"New York", b'"Washington, D.C."', b"Os", b"Oslo ", b"S\xc3\xa3o Paulo"]
    out = [b"id,city,code,note"]
    for r in range(rows):
        city = cities[int(rng.integers(0, len(cities)))]
        if not quoted and city.startswith(b'"'):
            city = b"DC"
        code = bytes(rng.integers(97, 123, size=int(rng.integers(0, 41)), dtype=np.uint8))
        note = b"x" * int(rng.integers(0, 3)) + (b"needle" if rng.random() < 0.1 else b"hay") + b"y" * int(rng.integers(0, 3))
        out.append(b"%d,%s,%s,%s" % (r, city, code, note))
    return line_end.join(out) + line_end


class DeviceTape:
    def __init__(self, ctx, pkg, torch, data: bytes):
        self.data = data
        self.index = ctx.read(data)                      # host copy of the reference tape (sentinel first)
        self.tape = pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), self.index)
        self.dbytes = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        # the tape as stage 1 leaves it on the device, behind its sentinel
        self.dindex = torch.zeros(self.index.size + 2, dtype=torch.int64, device="cuda:0")
        r = ctx.stage1_index_device(self.dbytes.data_ptr(), len(data), 0, 0, self.dindex.data_ptr() + 8, self.index.size + 1)
        assert r.count + 1 == self.index.size
        self.field_cnt, self.new_line = self.tape.field_cnt, self.tape.new_line
        self.crlf = self.new_line == "CRLF"

    def args(self):
        return self.dbytes.data_ptr(), self.dindex.data_ptr(), self.index.size, self.field_cnt, self.new_line


@pytest.mark.parametrize("line_end", [b"\n", b"\r\n"])
def test_chunk_spans_frequency_and_search(ctx, pkg, oracle, torch_cuda, line_end):
    torch = torch_cuda
    rng = np.random.default_rng(2024)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 5000, line_end))
    dbytes, dindex, index_len, field_cnt, new_line = dt.args()
    nrec = dt.tape.record_cnt - 1
    for n_chunks in (1, 3, 7):
        chunks = dt.tape.chunks(n_chunks)
        assert sum(c[3] for c in chunks) == nrec
        # ---- spans of a column, chunk by chunk ----------------------------------------------------------
        for f in (1, 3):
            for ch in chunks:
                b = torch.full((ch[3] + 2,), -1, dtype=torch.int64, device="cuda:0")
                e = torch.full((ch[3] + 2,), -1, dtype=torch.int64, device="cuda:0")
                n = pkg.chunk_field_spans_device(dindex, index_len, field_cnt, new_line, ch, f, b.data_ptr(), e.data_ptr())
                assert n == ch[3] and bool((b[n:] == -1).all())
                bh, eh = b.cpu().tolist(), e.cpu().tolist()
                for k, rec in enumerate(oracle.chunk_record_ids(ch, field_cnt, dt.crlf)):
                    assert dt.data[bh[k]: eh[k]] == oracle.seek_field(dt.data, dt.index, field_cnt, dt.crlf, rec, f)
        # ---- frequency count over all chunks == Counter ------------------------------------------------
        for f in (1, 2, 3):
            want = oracle.column_frequency(dt.data, dt.index, field_cnt, dt.crlf, chunks, f)
            longest = max(len(v) for v in want)
            need = pkg.column_frequency_scratch_bytes(nrec, len(chunks), longest)
            scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
            ent = torch.zeros((len(want) + 3, 4), dtype=torch.int64, device="cuda:0")
            st = pkg.column_frequency_device(ctx, dbytes, dindex, index_len, field_cnt, new_line, chunks, f,
                                             scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
            assert (st.n_records, st.n_distinct, st.max_field_bytes, st.overflow) == (nrec, len(want), longest, 0)
            got = {}
            for first, b_, e_, cnt in ent[: st.n_distinct].cpu().tolist():
                text = dt.data[b_: e_]
                assert text not in got
                got[text] = cnt
                # the representative is the FIRST record holding the value
                assert oracle.seek_field(dt.data, dt.index, field_cnt, dt.crlf, first, f) == text
                assert all(oracle.seek_field(dt.data, dt.index, field_cnt, dt.crlf, r, f) != text for r in range(first))
            assert got == dict(want)
        # ---- search, chunk by chunk: bitmap, count, ascending record ids -----------------------------------
        for f, needle, mode in ((1, b"Oslo", pkg.SEARCH_EQUALS), (1, b"Os", pkg.SEARCH_STARTS_WITH),
                                (3, b"needle", pkg.SEARCH_CONTAINS), (1, b"", pkg.SEARCH_EQUALS),
                                (1, b"", pkg.SEARCH_CONTAINS), (2, b"qz", pkg.SEARCH_CONTAINS),
                                (1, b'"Washington, D.C."', pkg.SEARCH_EQUALS), (3, b"yy", pkg.SEARCH_CONTAINS),
                                (1, b"ashington, D.C", pkg.SEARCH_CONTAINS), (1, b'"Washington', pkg.SEARCH_STARTS_WITH),
                                (3, b"hayy", pkg.SEARCH_CONTAINS), (3, b"xneedleyy", pkg.SEARCH_CONTAINS)):
            for ch in chunks:
                want = oracle.column_search(dt.data, dt.index, field_cnt, dt.crlf, ch, f, needle, mode)
                words = (ch[3] + 63) // 64
                bm = torch.zeros(words + 1, dtype=torch.int64, device="cuda:0")
                n = pkg.column_search_device(ctx, dbytes, len(dt.data), dindex, index_len, field_cnt, new_line, ch, f, needle, mode,
                                             bm.data_ptr())
                assert n == len(want), (needle, mode, ch)
                first_record = ch[1] // dt.tape.record_jump_size - 1
                scratch = torch.empty(pkg.bitmap_select_scratch_bytes(ch[3]), dtype=torch.uint8, device="cuda:0")
                ids = torch.full((n + 2,), -1, dtype=torch.int64, device="cuda:0")
                got_n = pkg.bitmap_select_device(bm.data_ptr(), ch[3], first_record, scratch.data_ptr(), ids.data_ptr(), n + 2)
                assert got_n == n and ids[:n].cpu().tolist() == want and bool((ids[n:] == -1).all())


def test_frequency_capacity_protocol_and_argument_checks(ctx, pkg, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(7)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 3000, b"\n"))
    dbytes, dindex, index_len, field_cnt, new_line = dt.args()
    chunks = dt.tape.chunks(2)
    nrec = sum(c[3] for c in chunks)
    want = oracle.column_frequency(dt.data, dt.index, field_cnt, False, chunks, 2)   # ~3000 distinct codes
    longest = max(len(v) for v in want)
    ent = torch.zeros((8, 4), dtype=torch.int64, device="cuda:0")
    # a scratch sized for shorter fields than the column holds: refused, and the status says how long the longest one is
    small = pkg.column_frequency_scratch_bytes(nrec, len(chunks), 1)
    assert small < pkg.column_frequency_scratch_bytes(nrec, len(chunks), longest)
    scratch = torch.empty(small, dtype=torch.uint8, device="cuda:0")
    st = pkg.column_frequency_device(ctx, dbytes, dindex, index_len, field_cnt, new_line, chunks, 2, scratch.data_ptr(), small,
                                     ent.data_ptr(), 8, allow_capacity=True)
    assert st.max_field_bytes == longest and st.n_distinct == 0
    # ... sized from that answer it fits; too few output entries: the status says how many are needed, the first 8 are written
    need = pkg.column_frequency_scratch_bytes(nrec, len(chunks), st.max_field_bytes)
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.StructureError) as e:
        pkg.column_frequency_device(ctx, dbytes, dindex, index_len, field_cnt, new_line, chunks, 2, scratch.data_ptr(), need,
                                    ent.data_ptr(), 8)
    assert e.value.code == pkg.ERR_TAPE_CAPACITY
    st = pkg.column_frequency_device(ctx, dbytes, dindex, index_len, field_cnt, new_line, chunks, 2, scratch.data_ptr(), need,
                                     ent.data_ptr(), 8, allow_capacity=True)
    assert st.n_distinct == len(want) > 1024
    for first, b_, e_, cnt in ent.cpu().tolist():
        assert want[dt.data[b_: e_]] == cnt and oracle.seek_field(dt.data, dt.index, field_cnt, False, first, 2) == dt.data[b_: e_]
    # a chunk that is not whole rows of this tape, a field that does not exist, a scratch that is not even aligned
    bad = (0, chunks[0][1] + 1, chunks[0][2], chunks[0][3])
    for kwargs in (dict(chunks=[bad]), dict(field=field_cnt), dict(misalign=8)):
        with pytest.raises(pkg.StructureError) as e:
            pkg.column_frequency_device(ctx, dbytes, dindex, index_len, field_cnt, new_line, kwargs.get("chunks", chunks),
                                        kwargs.get("field", 1), scratch.data_ptr() + kwargs.get("misalign", 0), need - 256,
                                        ent.data_ptr(), 8)
        assert e.value.code == pkg.ERR_INVALID_ARG


def test_gather_wide_every_alignment(ctx, pkg, oracle, torch_cuda):
    # the 16-byte path: fields at every byte alignment, every length around the 16-byte steps, the last field of the
    # buffer (no read past the end), strides that are and are not multiples of 16
    torch = torch_cuda
    rng = np.random.default_rng(11)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 2000, b"\n", quoted=False))
    dbytes, dindex, index_len, field_cnt, new_line = dt.args()
    nrec = dt.tape.record_cnt - 1
    for f in (2, 3):                                   # 3 = the last field of every row (and of the file)
        b = torch.empty(nrec, dtype=torch.int64, device="cuda:0")
        e = torch.empty(nrec, dtype=torch.int64, device="cuda:0")
        assert pkg.tape_field_spans_device(dindex, index_len, field_cnt, new_line, f, 0, nrec, b.data_ptr(), e.data_ptr()) == nrec
        for stride in (16, 32, 48, 20):
            dst = torch.full((nrec * stride + 16,), 0xEE, dtype=torch.uint8, device="cuda:0")
            ln = torch.zeros(nrec, dtype=torch.int32, device="cuda:0")
            pkg.gather_fields_device(dbytes, len(dt.data), b.data_ptr(), e.data_ptr(), nrec, dst.data_ptr(), stride, ln.data_ptr())
            rows = dst[: nrec * stride].cpu().numpy().reshape(nrec, stride)
            assert bool((dst[nrec * stride:] == 0xEE).all())
            lh = ln.cpu().tolist()
            for rec in range(nrec):
                want = oracle.seek_field(dt.data, dt.index, field_cnt, False, rec, f)
                assert lh[rec] == len(want)
                assert bytes(rows[rec][: min(len(want), stride)]) == want[:stride]
                assert not rows[rec][len(want):].any()


def test_consumers_on_the_synthetic_corpus_at_scale(ctx, pkg, oracle, torch_cuda):
    # 64 MiB of the 16 x 32 corpus (127 k rows): all values distinct in a column — the worst case for the table —
    # checked by size-independent properties plus a sampled byte comparison against the oracle's seek_field
    torch = torch_cuda
    cols, width, seed, q = pkg.WORKLOADS["16x32_noquote"]
    n = pkg.workload_len("16x32_noquote", 64 << 20)
    dbytes = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbytes.data_ptr(), 0, n, cols, width, seed, q)
    entries = n // (width + 1)
    dindex = torch.zeros(entries + 2, dtype=torch.int64, device="cuda:0")
    r = ctx.stage1_index_device(dbytes.data_ptr(), n, 0, 0, dindex.data_ptr() + 8, entries + 1)
    assert r.count == entries
    index_len = entries + 1
    rows = entries // cols                              # header-shaped first row included
    jump = cols
    chunks = [(i, max(1, i * (rows // 4)) * jump, (rows if i == 3 else (i + 1) * (rows // 4)) * jump, 0) for i in range(4)]
    nrec = rows - 1
    need = pkg.column_frequency_scratch_bytes(nrec, len(chunks), width)
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    ent = torch.zeros((nrec + 8, 4), dtype=torch.int64, device="cuda:0")
    st = pkg.column_frequency_device(ctx, dbytes.data_ptr(), dindex.data_ptr(), index_len, cols, "LF", chunks, 5,
                                     scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
    assert (st.n_records, st.max_field_bytes, st.overflow) == (nrec, width, 0)
    e = ent[: st.n_distinct]
    assert int(e[:, 3].sum()) == nrec and bool((e[:, 2] - e[:, 1] == width).all())
    host = dbytes.cpu().numpy().tobytes()
    hindex = dindex[:index_len].cpu().numpy().view(np.uint64)
    want = oracle.column_frequency(host, hindex, cols, False, chunks, 5)
    assert st.n_distinct == len(want)
    for first, b_, e_, cnt in e[:: max(1, st.n_distinct // 500)].cpu().tolist():
        assert want[host[b_: e_]] == cnt and oracle.seek_field(host, hindex, cols, False, first, 5) == host[b_: e_]
    # search for a value that certainly exists: the text of record 1000, field 5
    needle = oracle.seek_field(host, hindex, cols, False, 1000, 5)
    bm = torch.zeros((nrec + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
    whole = (0, jump, rows * jump, nrec)
    assert pkg.column_search_device(ctx, dbytes.data_ptr(), n, dindex.data_ptr(), index_len, cols, "LF", whole, 5, needle,
                                    pkg.SEARCH_EQUALS, bm.data_ptr()) == want[needle]
    hits = oracle.column_search(host, hindex, cols, False, whole, 5, needle[3:9], pkg.SEARCH_CONTAINS)
    assert pkg.column_search_device(ctx, dbytes.data_ptr(), n, dindex.data_ptr(), index_len, cols, "LF", whole, 5, needle[3:9],
                                    pkg.SEARCH_CONTAINS, bm.data_ptr()) == len(hits) >= 1
    scratch2 = torch.empty(pkg.bitmap_select_scratch_bytes(nrec), dtype=torch.uint8, device="cuda:0")
    ids = torch.zeros(len(hits), dtype=torch.int64, device="cuda:0")
    assert pkg.bitmap_select_device(bm.data_ptr(), nrec, 0, scratch2.data_ptr(), ids.data_ptr(), len(hits)) == len(hits)
    assert ids.cpu().tolist() == hits


def test_bitmap_select_ignores_bits_past_n_rows(pkg, torch_cuda):
    """include/csvsimd.h: "set bits among the first n_rows".  A caller's bitmap (not one search_kernel wrote) may hold
    anything past n_rows in its last word: those bits are not records."""
    torch = torch_cuda
    for n_rows in (1, 63, 64, 65, 100, 4096 + 7):
        words = (n_rows + 63) // 64
        bm = torch.full((words,), -1, dtype=torch.int64, device="cuda:0")       # every bit set, incl. the stray ones
        scratch = torch.empty(pkg.bitmap_select_scratch_bytes(n_rows), dtype=torch.uint8, device="cuda:0")
        ids = torch.full((n_rows + 70,), -1, dtype=torch.int64, device="cuda:0")
        n = pkg.bitmap_select_device(bm.data_ptr(), n_rows, 500, scratch.data_ptr(), ids.data_ptr(), n_rows + 70)
        assert n == n_rows
        assert ids[:n].cpu().tolist() == list(range(500, 500 + n_rows)) and bool((ids[n:] == -1).all())
