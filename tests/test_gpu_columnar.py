"""Row-major CSV -> columns in one pass (csvsimd_chunk_to_columns_device) and the two consumers on a column, against
the scalar definitions in oracle/: every cell is seek_field's text (src/record_source.rs:106-140) truncated / zero padded,
the frequency count is collections.Counter, the search is == / startswith / `in` (reference design_notes_1.md:1-4,
src/tape.rs:12-19, 95-140).  Never checked against the product's own per-column consumers."""
import numpy as np
import pytest

from test_gpu_consumers import DeviceTape, make_csv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def transpose(ctx, pkg, torch, dt, chunk, fields, stride, with_lens=True):
    n = chunk[3]
    nf = dt.field_cnt if fields is None else len(fields)
    cols = torch.full((nf, n, stride), 0xEE, dtype=torch.uint8, device="cuda:0")
    lens = torch.full((nf, n), -1, dtype=torch.int32, device="cuda:0") if with_lens else None
    dbytes, dindex, index_len, field_cnt, new_line = dt.args()
    got_n = pkg.chunk_to_columns_device(ctx, dbytes, len(dt.data), dindex, index_len, field_cnt, new_line, chunk, fields,
                                        cols.data_ptr(), stride, lens.data_ptr() if with_lens else 0)
    torch.cuda.synchronize()
    assert got_n == n
    return cols, lens


def check_cells(oracle, dt, chunk, fields, stride, cols, lens):
    flist = list(range(dt.field_cnt)) if fields is None else list(fields)
    h = cols.cpu().numpy()
    hl = lens.cpu().numpy() if lens is not None else None
    for i, rec in enumerate(oracle.chunk_record_ids(chunk, dt.field_cnt, dt.crlf)):
        for c, f in enumerate(flist):
            text = oracle.seek_field(dt.data, dt.index, dt.field_cnt, dt.crlf, rec, f)
            want = text[:stride] + b"\0" * (stride - min(len(text), stride))
            assert h[c, i].tobytes() == want, (rec, f, text)
            if hl is not None:
                assert int(hl[c, i]) == len(text)


@pytest.mark.parametrize("line_end", [b"\n", b"\r\n"])
def test_every_cell_is_seek_field(ctx, pkg, oracle, torch_cuda, line_end):
    torch = torch_cuda
    rng = np.random.default_rng(31)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 1500, line_end))
    for n_chunks in (1, 3, 7):
        for ch in dt.tape.chunks(n_chunks):
            for fields, stride in ((None, 32), ([3, 1], 16), ([2, 2, 0], 48), ([1], 16)):
                cols, lens = transpose(ctx, pkg, torch, dt, ch, fields, stride)
                check_cells(oracle, dt, ch, fields, stride, cols, lens)
    # without the length array, and the first n_fields columns by count
    ch = dt.tape.chunks(1)[0]
    cols, _ = transpose(ctx, pkg, torch, dt, ch, None, 64, with_lens=False)
    check_cells(oracle, dt, ch, None, 64, cols, None)


def test_rows_that_do_not_fit_the_window_take_the_global_path(ctx, pkg, oracle, torch_cuda):
    """Rows longer than the 32-KiB window, and more columns than the 4096 staged tape entries: same cells."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    rows = [b"a,b,c"]
    for r in range(40):
        big = bytes(rng.integers(97, 123, size=int(rng.integers(30000, 70000)), dtype=np.uint8)) if r % 3 == 0 else b"s%d" % r
        rows.append(b"%d,%s,x%d" % (r, big, r))
    dt = DeviceTape(ctx, pkg, torch, b"\n".join(rows) + b"\n")
    for ch in dt.tape.chunks(2):
        cols, lens = transpose(ctx, pkg, torch, dt, ch, None, 64)
        check_cells(oracle, dt, ch, None, 64, cols, lens)
    wide = [b",".join(b"h%d" % c for c in range(5000))]
    for r in range(6):
        wide.append(b",".join(b"%d" % ((r * 7919 + c) % 1000) for c in range(5000)))
    dt = DeviceTape(ctx, pkg, torch, b"\n".join(wide) + b"\n")
    ch = dt.tape.chunks(1)[0]
    fields = [0, 1, 4095, 4096, 4097, 4999]
    cols, lens = transpose(ctx, pkg, torch, dt, ch, fields, 16)
    check_cells(oracle, dt, ch, fields, 16, cols, lens)


def test_any_alignment_of_the_byte_buffer(ctx, pkg, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(77)
    data = make_csv(rng, 300, b"\n")
    base = DeviceTape(ctx, pkg, torch, data)
    for off in (1, 5, 15):
        shifted = torch.zeros(len(data) + 32, dtype=torch.uint8, device="cuda:0")
        shifted[off: off + len(data)] = base.dbytes
        ch = base.tape.chunks(1)[0]
        n = ch[3]
        cols = torch.zeros((base.field_cnt, n, 32), dtype=torch.uint8, device="cuda:0")
        lens = torch.zeros((base.field_cnt, n), dtype=torch.int32, device="cuda:0")
        pkg.chunk_to_columns_device(ctx, shifted.data_ptr() + off, len(data), base.dindex.data_ptr(), base.index.size,
                                    base.field_cnt, base.new_line, ch, None, cols.data_ptr(), 32, lens.data_ptr())
        torch.cuda.synchronize()
        check_cells(oracle, base, ch, None, 32, cols, lens)


def test_columnar_frequency_is_counter(ctx, pkg, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(2025)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 6000, b"\n"))
    ch = dt.tape.chunks(1)[0]
    n = ch[3]
    stride = 48   # every field of make_csv is <= 40 bytes
    cols, lens = transpose(ctx, pkg, torch, dt, ch, None, stride)
    first_record = ch[1] // dt.tape.record_jump_size - 1
    need = pkg.columnar_frequency_scratch_bytes(n)
    scratch = torch.full((need,), 0xA5, dtype=torch.uint8, device="cuda:0")   # never cleared by anybody: the call must not care
    for f in (1, 2, 3, 0):
        want = oracle.column_frequency(dt.data, dt.index, dt.field_cnt, dt.crlf, [ch], f)
        ent = torch.zeros((len(want) + 3, 2), dtype=torch.int64, device="cuda:0")
        st = pkg.columnar_frequency_device(ctx, cols[f].data_ptr(), lens[f].data_ptr(), n, stride, first_record,
                                           scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
        assert (st.n_records, st.n_distinct, st.truncated, st.overflow) == (n, len(want), 0, 0)
        got = {}
        for first, cnt in ent[: st.n_distinct].cpu().tolist():
            text = oracle.seek_field(dt.data, dt.index, dt.field_cnt, dt.crlf, first, f)
            assert text not in got
            got[text] = cnt
            # the entry names the FIRST record holding the value
            assert all(oracle.seek_field(dt.data, dt.index, dt.field_cnt, dt.crlf, r, f) != text for r in range(first))
        assert got == dict(want)
    # capacity protocol: a scratch smaller than the call needs is refused before anything runs
    ent = torch.zeros((8, 2), dtype=torch.int64, device="cuda:0")
    with pytest.raises(pkg.StructureError) as e:
        pkg.columnar_frequency_device(ctx, cols[2].data_ptr(), lens[2].data_ptr(), n, stride, 0, scratch.data_ptr(), need - 1,
                                      ent.data_ptr(), 8)
    assert e.value.code == pkg.ERR_TAPE_CAPACITY
    # too few output entries: the status says how many are needed
    st = pkg.columnar_frequency_device(ctx, cols[1].data_ptr(), lens[1].data_ptr(), n, stride, 0, scratch.data_ptr(), need,
                                       ent.data_ptr(), 2, allow_capacity=True)
    assert st.n_distinct == len(oracle.column_frequency(dt.data, dt.index, dt.field_cnt, dt.crlf, [ch], 1)) > 2
    # a stride shorter than some values: the count would merge values that differ past it -> refused, and counted
    short, slens = transpose(ctx, pkg, torch, dt, ch, [2], 16)
    st = pkg.columnar_frequency_device(ctx, short[0].data_ptr(), slens[0].data_ptr(), n, 16, 0, scratch.data_ptr(), need,
                                       ent.data_ptr(), 8, allow_capacity=True)
    assert st.truncated == int((slens[0] > 16).sum())  and st.truncated > 0


def test_columnar_frequency_async_is_launches_only_and_graph_capturable(ctx, pkg, torch_cuda):
    """The asynchronous form: nothing waited for, allocated or copied — so it can be captured and replayed on new data; the
    status record is read from device memory by the caller.  Sizes around the slab (8 192 records) and partition geometry
    (above 1 Mi records there are more partitions than workgroups: later partitions are drawn from tickets, and neighbouring
    partitions are merged as one when pass 1 left few tuples), all-distinct, few-valued and in-between columns, an empty
    column."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    stride = 32
    for n in (1, 63, 8191, 8192, 8193, 100_000, 1_000_003, 2_100_000, 4_300_001):
        need = pkg.columnar_frequency_scratch_bytes(n)
        scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        col = torch.zeros((n, stride), dtype=torch.uint8, device="cuda:0")
        ent = torch.zeros((n + 4, 2), dtype=torch.int64, device="cuda:0")
        d_status = torch.full((4,), -1, dtype=torch.int64, device="cuda:0")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pkg.columnar_frequency_device_async(ctx, col.data_ptr(), 0, n, stride, 5, scratch.data_ptr(), need, ent.data_ptr(),
                                                ent.shape[0], d_status.data_ptr(), side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pkg.columnar_frequency_device_async(ctx, col.data_ptr(), 0, n, stride, 5, scratch.data_ptr(), need, ent.data_ptr(),
                                                ent.shape[0], d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for kind in ("distinct", "few", "mid", "one"):
            if kind == "distinct":
                keys = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            elif kind == "mid":
                keys = rng.integers(0, 3000, size=n).astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D)
            elif kind == "few":
                keys = rng.integers(0, 37, size=n).astype(np.uint64)
            else:
                keys = np.zeros(n, dtype=np.uint64)
            host = np.zeros((n, stride), dtype=np.uint8)
            host[:, :8] = keys.view(np.uint8).reshape(n, 8)
            host[:, 24] = 7
            col.copy_(torch.from_numpy(host))
            ent.fill_(-1)
            d_status.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            uniq, first, counts = np.unique(keys, return_index=True, return_counts=True)
            st = d_status.cpu().tolist()
            assert st == [n, uniq.size, 0, 0], (n, kind, st)
            got = {int(f) - 5: int(c) for f, c in ent[: uniq.size].cpu().tolist()}
            assert got == {int(f): int(c) for f, c in zip(first, counts)}, (n, kind)
            assert bool((ent[uniq.size:] == -1).all())
        del g
    # no records: the status is still written
    d_status = torch.full((4,), -1, dtype=torch.int64, device="cuda:0")
    pkg.columnar_frequency_device_async(ctx, 0, 0, 0, stride, 0, scratch.data_ptr(), need, 0, 0, d_status.data_ptr())
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("stride", [16, 32])
def test_columnar_frequency_long_columns_of_few_values(ctx, pkg, torch_cuda, stride):
    """From 4 Mi records on (two slabs per CU and more) a streaming kernel counts first — one workgroup per CU keeps one table
    over its whole share of slabs — and the general kernel only counts the shares it gave up on (round 5).  Columns that take
    each way and BOTH in one call: few values everywhere; few values in the first 60 % and distinct ones after (shares of
    either kind and one that changes its mind half way); as many values as a table just holds; with a lengths array, values
    that differ only in their length, and more over-long records in one share than a slab's counter holds; some thousand
    values (the second streaming kernel: a larger table of keys, rows compared in the column), with and without lengths."""
    torch = torch_cuda
    rng = np.random.default_rng(4100 + stride)
    n = 4_200_000 + 4321
    need = pkg.columnar_frequency_scratch_bytes(n)
    scratch = torch.full((need,), 0x5A, dtype=torch.uint8, device="cuda:0")
    col = torch.zeros((n, stride), dtype=torch.uint8, device="cuda:0")
    ent = torch.zeros((n + 4, 2), dtype=torch.int64, device="cuda:0")
    split = int(n * 0.6)
    for kind in ("few", "half", "edge", "lengths", "mid", "tenk", "near", "mid_lengths"):
        lens = None
        if kind in ("mid", "tenk", "near"):
            # some thousand values: the second streaming kernel's table of keys (12 288 values per share at most: "near" has
            # shares on either side of that)
            keys = rng.integers(0, {"mid": 5000, "tenk": 10_000, "near": 12_200}[kind], size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        elif kind == "mid_lengths":
            keys = rng.integers(0, 300, size=n).astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93)
            lens = rng.integers(8, stride + 1, size=n).astype(np.int32)      # x up to 25 lengths each: ~7 000 (value, length) pairs
        elif kind == "few":
            keys = rng.integers(0, 100, size=n).astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D)
        elif kind == "half":
            keys = rng.integers(0, 50, size=n).astype(np.uint64)
            keys[split:] = (np.arange(n - split, dtype=np.uint64) + np.uint64(1000)) * np.uint64(0x9E3779B97F4A7C15)
        elif kind == "edge":
            keys = rng.integers(0, 2300, size=n).astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93)   # (a share takes 2 304)
        else:
            keys = rng.integers(0, 20, size=n).astype(np.uint64)
            lens = rng.integers(8, stride + 1, size=n).astype(np.int32)
            lens[100_000:130_000] = stride + 5          # 30 000 over-long records in ONE share (a slab's counter holds 8 192)
            lens[n - 3] = stride + 1
        host = np.zeros((n, stride), dtype=np.uint8)
        host[:, :8] = keys.view(np.uint8).reshape(n, 8)
        host[:, stride - 1] = 9
        if lens is not None:
            host[lens < stride, stride - 1] = 0      # (a column is zero padded past a value's length)
        col.copy_(torch.from_numpy(host))
        dl = torch.from_numpy(lens).to("cuda:0") if lens is not None else None
        ent.fill_(-1)
        st = pkg.columnar_frequency_device(ctx, col.data_ptr(), dl.data_ptr() if dl is not None else 0, n, stride, 11,
                                           scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0],
                                           allow_capacity=lens is not None)   # (over-long records: counted, and reported by the code)
        if lens is None:
            ident = keys
        else:                                         # (value, length): the identity of a record
            _, kid = np.unique(keys, return_inverse=True)
            ident = kid.astype(np.uint64) * np.uint64(64) + lens.astype(np.uint64)
        uniq, first, counts = np.unique(ident, return_index=True, return_counts=True)
        trunc = 0 if lens is None else int((lens > stride).sum())
        assert (st.n_records, st.n_distinct, st.truncated, st.overflow) == (n, uniq.size, trunc, 0), (kind, stride)
        got = ent[: uniq.size].cpu().numpy()
        order = np.argsort(got[:, 0])
        want_order = np.argsort(first)
        assert np.array_equal(got[order, 0] - 11, first[want_order]) and np.array_equal(got[order, 1], counts[want_order]), (kind, stride)
        assert bool((ent[uniq.size:] == -1).all())


def test_columnar_frequency_same_hash_tag_never_merges(ctx, pkg, torch_cuda):
    """Fixed-width keys without a length array; many records, few values, and values that differ in one late byte."""
    torch = torch_cuda
    rng = np.random.default_rng(9)
    n, stride = 200_000, 32
    vocab = rng.integers(0, 256, size=(300, stride), dtype=np.uint8)
    vocab[150:] = vocab[:150]
    vocab[150:, 31] ^= 1                      # pairs that differ in the last byte only
    pick = rng.integers(0, 300, size=n)
    col = torch.from_numpy(vocab[pick]).cuda()
    need = pkg.columnar_frequency_scratch_bytes(n)
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    ent = torch.zeros((400, 2), dtype=torch.int64, device="cuda:0")
    st = pkg.columnar_frequency_device(ctx, col.data_ptr(), 0, n, stride, 1000, scratch.data_ptr(), need, ent.data_ptr(), 400)
    want = np.bincount(pick, minlength=300)
    first = {v: int(np.flatnonzero(pick == v)[0]) for v in range(300)}
    assert st.n_distinct == int((want > 0).sum()) and st.n_records == n
    got = {int(f) - 1000: int(c) for f, c in ent[: st.n_distinct].cpu().tolist()}
    assert got == {first[v]: int(want[v]) for v in range(300) if want[v]}


def test_columnar_search(ctx, pkg, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(404)
    dt = DeviceTape(ctx, pkg, torch, make_csv(rng, 5000, b"\r\n"))
    stride = 48
    for ch in dt.tape.chunks(3):
        n = ch[3]
        cols, lens = transpose(ctx, pkg, torch, dt, ch, None, stride)
        first_record = ch[1] // dt.tape.record_jump_size - 1
        for f, needle, mode in ((1, b"Oslo", pkg.SEARCH_EQUALS), (1, b"Os", pkg.SEARCH_STARTS_WITH),
                                (3, b"needle", pkg.SEARCH_CONTAINS), (1, b"", pkg.SEARCH_EQUALS),
                                (1, b"", pkg.SEARCH_CONTAINS), (2, b"qz", pkg.SEARCH_CONTAINS),
                                (1, b'"Washington, D.C."', pkg.SEARCH_EQUALS), (3, b"yy", pkg.SEARCH_CONTAINS),
                                (1, b"ashington, D.C", pkg.SEARCH_CONTAINS), (1, b'"Washington', pkg.SEARCH_STARTS_WITH),
                                (3, b"hayy", pkg.SEARCH_CONTAINS), (3, b"xneedleyy", pkg.SEARCH_CONTAINS),
                                (2, b"abcdefghijklmnopqrstuvwxyzabcdefghijklmnopqrstuvwxyz", pkg.SEARCH_CONTAINS)):
            want = oracle.column_search(dt.data, dt.index, dt.field_cnt, dt.crlf, ch, f, needle, mode)
            bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
            got_n = pkg.columnar_search_device(ctx, cols[f].data_ptr(), lens[f].data_ptr(), n, stride, needle, mode,
                                               bm.data_ptr())
            assert got_n == len(want), (needle, mode)
            scratch = torch.empty(pkg.bitmap_select_scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
            ids = torch.full((got_n + 2,), -1, dtype=torch.int64, device="cuda:0")
            k = pkg.bitmap_select_device(bm.data_ptr(), n, first_record, scratch.data_ptr(), ids.data_ptr(), got_n + 2)
            assert k == got_n and ids[:k].cpu().tolist() == want
    # a stride shorter than some records: refused (the bitmap only speaks for the first `stride` bytes)
    ch = dt.tape.chunks(1)[0]
    short, slens = transpose(ctx, pkg, torch, dt, ch, [2], 16)
    bm = torch.zeros((ch[3] + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
    with pytest.raises(pkg.StructureError) as e:
        pkg.columnar_search_device(ctx, short[0].data_ptr(), slens[0].data_ptr(), ch[3], 16, b"a", pkg.SEARCH_CONTAINS,
                                   bm.data_ptr())
    assert e.value.code == pkg.ERR_TAPE_CAPACITY


def test_synthetic_corpus_all_columns_at_once(ctx, pkg, torch_cuda):
    """127 k records of the 16x32 corpus, all 16 columns in one pass: every cell against plain slicing of the
    fixed-pitch rows (the corpus' closed form), lengths all 32."""
    torch = torch_cuda
    cols_, width, seed, q = pkg.WORKLOADS["16x32_noquote"]
    n = pkg.workload_len("16x32_noquote", 64 << 20)
    dbytes = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbytes.data_ptr(), 0, n, cols_, width, seed, q)
    entries = n // (width + 1)
    dindex = torch.zeros(entries + 2, dtype=torch.int64, device="cuda:0")
    r = ctx.stage1_index_device(dbytes.data_ptr(), n, 0, 0, dindex.data_ptr() + 8, entries + 1)
    rows = r.count // cols_
    nrec = rows - 1
    whole = (0, cols_, rows * cols_, nrec)
    out = torch.zeros((cols_, nrec, 32), dtype=torch.uint8, device="cuda:0")
    lens = torch.zeros((cols_, nrec), dtype=torch.int32, device="cuda:0")
    got = pkg.chunk_to_columns_device(ctx, dbytes.data_ptr(), n, dindex.data_ptr(), r.count + 1, cols_, "LF", whole, None,
                                      out.data_ptr(), 32, lens.data_ptr())
    torch.cuda.synchronize()
    assert got == nrec and bool((lens == width).all())
    table = dbytes.view(rows, cols_, width + 1)[1:, :, :width]           # row-major: record, column, byte
    assert torch.equal(out, table.permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("stride", [16, 32])
def test_columnar_search_on_16_and_32_byte_rows(ctx, pkg, torch_cuda, stride):
    """Rows of 16 / 32 bytes take the register-resident kernel (round 5): every mode, needles of every length 0 ... stride + 1
    at every position of a row, rows of every length 0 ... stride over a three-letter alphabet (partial matches everywhere),
    with and without a lengths array — bit for bit what Python's ==, startswith and `in` say about the same bytes
    (the definitions of oracle_py.column_search; the reference only states the goal: design_notes_1.md:1-4)."""
    torch = torch_cuda
    rng = np.random.default_rng(1600 + stride)
    n = 20_000 + 37
    host = np.zeros((n, stride), dtype=np.uint8)
    lens = rng.integers(0, stride + 1, size=n).astype(np.int32)
    lens[:200] = stride
    body = rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=(n, stride))
    for i in range(n):
        host[i, : lens[i]] = body[i, : lens[i]]
    rows = [host[i, : lens[i]].tobytes() for i in range(n)]
    col = torch.from_numpy(host).to("cuda:0")
    dl = torch.from_numpy(lens).to("cuda:0")
    bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
    needles = [b"", b"a", b"ab", b"abc", b"cab", b"bbbb", b"abcabcab", b"abcabcabc"]
    for m in (1, 2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33):
        if m <= stride + 1:
            i = int(rng.integers(0, 200))
            full = rows[i] + b"a"
            for at in {0, 1, max(0, stride - m), max(0, (stride - m) // 2)}:
                needles.append(full[at: at + m])
    for needle in needles:
        for mode, fn in ((pkg.SEARCH_EQUALS, lambda r: r == needle), (pkg.SEARCH_STARTS_WITH, lambda r: r.startswith(needle)),
                         (pkg.SEARCH_CONTAINS, lambda r: needle in r)):
            want = np.array([fn(r) for r in rows], dtype=bool)
            bm.zero_()
            got_n = pkg.columnar_search_device(ctx, col.data_ptr(), dl.data_ptr(), n, stride, needle, mode, bm.data_ptr())
            bits = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
            assert got_n == int(want.sum()) and np.array_equal(bits, want), (stride, needle, mode)
    # no lengths array: every row is its whole zero-padded stride (fixed-width keys)
    padded = [host[i].tobytes() for i in range(n)]
    for needle in (b"abc", padded[5][:7], padded[9], padded[3][stride - 3:], b"\0", b"c\0", b"\0\0\0\0", b"a\0\0\0\0",
                   padded[7][stride - 6:], padded[11][stride - 4:], padded[13][2:stride]):
        for mode, fn in ((pkg.SEARCH_EQUALS, lambda r: r == needle), (pkg.SEARCH_STARTS_WITH, lambda r: r.startswith(needle)),
                         (pkg.SEARCH_CONTAINS, lambda r: needle in r)):
            want = np.array([fn(r) for r in padded], dtype=bool)
            got_n = pkg.columnar_search_device(ctx, col.data_ptr(), 0, n, stride, needle, mode, bm.data_ptr())
            bits = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
            assert got_n == int(want.sum()) and np.array_equal(bits, want), (stride, needle, mode, "fixed width")
    # the packed result word counts matches in 36 bits: more records than that per call are refused before anything runs
    with pytest.raises(pkg.StructureError) as e:
        pkg.columnar_search_device(ctx, col.data_ptr(), 0, 1 << 36, stride, b"abc", pkg.SEARCH_CONTAINS, bm.data_ptr())
    assert e.value.code == pkg.ERR_INVALID_ARG
