"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on the same
bytes, against the committed golden vectors, and — at BASELINE.json's full sizes — through
size-independent properties (analytic tape of the quote-free corpora, sortedness, split
invariance, idempotence, order-sensitive checksums).  Bit-exact everywhere: this is integer work.
"""
import numpy as np
import pytest

from conftest import random_csvish

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def gpu_index(ctx, torch, host: np.ndarray, *, base_off=0, in_quote_in=0, misalign=0, cap=None):
    """Runs the device entry point on a copy of `host` placed `misalign` bytes past an aligned
    device address.  Returns (entries uint64[], ShardResult)."""
    n = host.size
    dbuf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda:0")
    if n:
        dbuf[misalign: misalign + n] = torch.from_numpy(host)
    # poison around the payload: stray reads of neighbouring bytes must not leak into the tape
    dbuf[:misalign] = 0x2C
    dbuf[misalign + n:] = 0x2C
    cap = (n + 1) if cap is None else cap
    dtape = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
    r = ctx.stage1_index_device(dbuf.data_ptr() + misalign, n, base_off, in_quote_in, dtape.data_ptr(), cap,
                                allow_overflow=True)
    torch.cuda.synchronize()
    assert (dtape[cap:] == -1).all(), "wrote past tape_cap"
    k = min(r.count, cap)
    assert r.written == k
    assert (dtape[k:] == -1).all(), "wrote more entries than counted"
    return dtape[:k].cpu().numpy().view(np.uint64), r


def test_wavefront_selftest(pkg, torch_cuda):
    pkg.selftest(0)


@pytest.mark.parametrize("name", ["reader_test01.csv", "sample.csv", "sample_rx.csv"])
def test_golden_fixtures_via_read(ctx, golden, oracle, name):
    # drop-in entry point == reader::read (src/reader.rs:150) on the reference's own fixtures
    data, exp = golden[name]
    got = ctx.read(data)
    assert np.array_equal(got, np.array(exp["index"], dtype=np.uint64))
    assert np.array_equal(got, oracle.sse_read(data))
    if name == "reader_test01.csv":  # the reference's own assertion, src/reader.rs:325-326
        assert got[1] == 4 and got[-1] == 95


def test_reference_worked_example_string_mask_on_gpu(ctx, pkg, torch_cuda, oracle):
    # "0b100010000 quotes -> 0b011110000 string mask" (src/avx/stage1.rs:350-352, design_notes_1.md:90-91),
    # at every position of a stripe / round / span the kernel treats differently, and through read()
    from test_oracle import known_answer_block
    block, want = known_answer_block()
    assert np.array_equal(ctx.read(block)[1:], want)
    T = pkg.tile_bytes()
    for pad in (0, 64, 4032, 4096, 32704, 32768, T - 64, T):
        buf = np.full(pad + 64 + 70, 0x61, dtype=np.uint8)
        buf[pad: pad + 64] = block
        got, r = gpu_index(ctx, torch_cuda, buf)
        assert np.array_equal(got, want + np.uint64(pad)) and r.in_quote_out == 0, pad


def test_config1_sample_csv_to_tape(ctx, pkg, golden, tmp_path):
    # BASELINE config 1: res/sample.csv -> stage 1 -> tape, end to end through csv_simd::create
    data, exp = golden["sample.csv"]
    p = tmp_path / "sample.csv"
    p.write_bytes(data)
    t = ctx.create(str(p))
    assert np.array_equal(t.index(), np.array(exp["index"], dtype=np.uint64))
    assert (t.field_cnt, t.record_cnt, t.record_jump_size, t.new_line) == (3, 15, 3, "LF")
    assert t.header() == ["Name", "Number", "Done"] and t.seek_record(0) == b'Edm nd,3, "o"'
    # ragged file -> InvalidCsvFormat, as the reference's TapeCore::init (src/tape.rs:342-344)
    data2, _ = golden["reader_test01.csv"]
    p2 = tmp_path / "ragged.csv"
    p2.write_bytes(data2)
    with pytest.raises(pkg.StructureError) as e:
        ctx.create(str(p2))
    assert e.value.code == pkg.ERR_INVALID_CSV_FORMAT
    with pytest.raises(pkg.StructureError) as e:
        ctx.create(str(tmp_path / "missing.csv"))
    assert e.value.code == pkg.ERR_IO


def test_random_small_all_residues(ctx, torch_cuda, oracle):
    # every len % 64 and len % 16, empty input included
    rng = np.random.default_rng(2024)
    for n in list(range(0, 200)) + [255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097]:
        d = random_csvish(rng, n, 0.1)
        got, r = gpu_index(ctx, torch_cuda, d)
        want, inq = oracle.scalar_index(d)
        assert np.array_equal(got, want), n
        assert (r.count, r.in_quote_out, r.error) == (want.size, inq, 0), n
        p, c0, c1 = oracle.shard_descriptor(d)
        assert (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1), n
        if n >= 64:  # the reference's defined domain: also equal to the SSE restatement
            assert np.array_equal(np.concatenate([[0], got]).astype(np.uint64), oracle.sse_read(d)), n


def test_misaligned_base_off_and_entering_state(ctx, torch_cuda, pkg, oracle):
    rng = np.random.default_rng(77)
    T = pkg.tile_bytes()
    for n in (1, 15, 16, 17, 100, 5000, T // 2 - 3, T // 2 + 5, T - 3, T + 5):
        d = random_csvish(rng, n, 0.05)
        for mis in (0, 1, 7, 15, 16, 63, 64, 65, 100, 127):
            for inq in (0, 1):
                got, r = gpu_index(ctx, torch_cuda, d, base_off=10**12 + 3, in_quote_in=inq, misalign=mis)
                want, q = oracle.scalar_index(d, base_off=10**12 + 3, in_quote_in=inq)
                assert np.array_equal(got, want), (n, mis, inq)
                assert r.in_quote_out == q and r.count == want.size


def test_tile_boundaries_and_lookback(ctx, torch_cuda, pkg, oracle):
    # sizes around multiples of the tile (pkg.tile_bytes(), 256 KiB); quotes make the in-string state cross tiles
    rng = np.random.default_rng(31337)
    T = pkg.tile_bytes()
    for n in (T - 1, T, T + 1, 2 * T - 16, 2 * T + 16, 5 * T + 12345, 37 * T + 1):
        for pq in (0.0, 0.001, 0.2):
            d = random_csvish(rng, n, pq)
            got, r = gpu_index(ctx, torch_cuda, d)
            want, q = oracle.scalar_index(d)
            assert r.count == want.size and r.in_quote_out == q, (n, pq)
            assert np.array_equal(got, want), (n, pq)


def test_adversarial_densities(ctx, torch_cuda, pkg, oracle):
    T = pkg.tile_bytes()
    n = 3 * T + 1000
    cases = {
        "all_commas": np.full(n, 0x2C, dtype=np.uint8),           # 1 entry per byte: compaction windows
        "all_quotes": np.full(n, 0x22, dtype=np.uint8),
        "all_lf": np.full(n, 0x0A, dtype=np.uint8),
        "no_struct": np.full(n, 0x61, dtype=np.uint8),
        "high_bytes": np.tile(np.array([0xAC, 0x8A, 0x8D, 0xA2, 0x2C, 0xFF], dtype=np.uint8), n // 6 + 1)[:n],
        "quote_comma": np.tile(np.frombuffer(b'",', dtype=np.uint8), n // 2 + 1)[:n],
        "one_open_quote": np.concatenate([np.frombuffer(b'"', dtype=np.uint8), np.full(n - 1, 0x2C, dtype=np.uint8)]),
        "crlf_rows": np.tile(np.frombuffer(b"ab,cd,ef\r\n", dtype=np.uint8), n // 10 + 1)[:n],
    }
    for name, d in cases.items():
        got, r = gpu_index(ctx, torch_cuda, d)
        want, q = oracle.scalar_index(d)
        assert r.count == want.size and r.in_quote_out == q, name
        assert np.array_equal(got, want), name


def test_speculative_scatter_guesses(ctx, torch_cuda, pkg, oracle):
    """The emit phase scatters a tile into its window BEFORE the look-back has resolved it, under the guess "the
    entering state with more entries" (stage1_kernels.hip: scatter_span_spec).  Tiles built so that the guess is
    right with either state and wrong with either state, sparse enough (<= one entry per 16 bytes) that every span
    fits one window, i.e. the speculation really happens; the tape must be the oracle's in every case."""
    T = pkg.tile_bytes()
    row = b"aaaaaaaaaaaaaaaaaaa,bbbbbbbbbbbbbbbbbbb\n"          # 40 bytes, 2 entries

    def rows(nbytes):
        return (row * (nbytes // len(row) + 1))[:nbytes]

    def plain():                 # entered outside, stays outside: guess "outside", right
        return rows(T)

    def opens_at_end():          # ... and ends inside a string
        b = bytearray(rows(T))
        b[T - 24:T] = b',"unterminated text .....'[:24]
        return bytes(b)

    def closes_at_start():       # truly entered INSIDE; the string closes at once: guess "inside", right
        b = bytearray(rows(T))
        b[0:12] = b'still text",'
        return bytes(b)

    def quoted_body():           # entered outside; nearly the whole tile is ONE quoted field full of commas:
        b = bytearray(b"y" * T)  # "outside" has 3 entries, "inside" thousands -> guess "inside", WRONG
        b[0:6] = b"ab,cd\n"
        b[6] = 0x22
        for i in range(8, T - 32, 24):
            b[i] = 0x2C
        b[T - 2] = 0x22
        b[T - 1] = 0x0A
        return bytes(b)

    def quoted_body_entered_inside():   # truly entered inside and stays inside a comma-rich text until the very end:
        b = bytearray(b"z" * T)         # "inside" has no entries, "outside" thousands -> guess "outside", WRONG
        for i in range(8, T - 32, 24):
            b[i] = 0x2C
        b[T - 1] = 0x22                 # the string closes with the tile
        return bytes(b)

    seq = [plain(), opens_at_end(), closes_at_start(), quoted_body(), plain(), opens_at_end(),
           quoted_body_entered_inside(), plain(), quoted_body(), opens_at_end(), closes_at_start(), plain()]
    for tail in (0, 1, 4097):
        d = np.frombuffer(b"".join(seq) + rows(tail), dtype=np.uint8).copy()
        for inq in (0, 1):
            got, r = gpu_index(ctx, torch_cuda, d, in_quote_in=inq, base_off=3)
            want, q = oracle.scalar_index(d, base_off=3, in_quote_in=inq)
            assert r.count == want.size and r.in_quote_out == q, (tail, inq)
            assert np.array_equal(got, want), (tail, inq)


def test_enter_guess_chooses_the_state_of_the_first_tiles(ctx, torch_cuda, pkg, oracle):
    """CSVSIMD_ENTER_GUESS: a shard whose entering state nobody knows is indexed under the state for which its first
    EIGHT tiles (2 MiB), composed in order, hold more entries; the record says which one was used, and count / tape /
    leaving state are the oracle's for THAT state — whether the guess is the truth (shards cut out of a quoted CSV
    anywhere; a first tile that lies, outvoted by the tiles behind it) or not (2 MiB of lying tiles)."""
    T = pkg.tile_bytes()
    rng = np.random.default_rng(808)
    # a quoted CSV: 10 % of the 30-byte fields are quoted and hold a comma and a line break
    fields = []
    for i in range(3 * T // 31 + 50):
        if rng.random() < 0.1:
            fields.append(b'"' + b"q" * 9 + b"," + b"q" * 9 + b"\n" + b"q" * 8 + b'"')
        else:
            fields.append(b"f" * 30)
        fields.append(b"\n" if i % 16 == 15 else b",")
    text = np.frombuffer(b"".join(fields), dtype=np.uint8).copy()
    _, _ = oracle.scalar_index(text)
    quotes = np.flatnonzero(text == 0x22)
    cuts = [0, 1, 31 * 5 + 7, int(quotes[10]) + 3, int(quotes[10]) + 1, int(quotes[11]) + 1, T + 12345,
            int(quotes[len(quotes) // 2]) + 5, int(quotes[len(quotes) // 2 + 1]) + 5]
    for cut in cuts:
        d = text[cut:]
        truth = int(np.count_nonzero(text[:cut] == 0x22) & 1)
        got, r = gpu_index(ctx, torch_cuda, d, base_off=cut, in_quote_in=pkg.ENTER_GUESS)
        want, q = oracle.scalar_index(d, base_off=cut, in_quote_in=r.in_quote_in_used)
        assert r.in_quote_in_used == truth, cut          # on a real quoted CSV the guess is the truth
        assert (r.count, r.in_quote_out, r.error) == (want.size, q, 0) and np.array_equal(got, want), cut
        p, c0, c1 = oracle.shard_descriptor(d)
        assert (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1)

    def lying_tiles(k, truly_inside):
        """k tiles that are ONE long quoted field full of commas (a comma every 24 bytes), then plain rows.
        truly_inside False: the field opens at byte 6 -> entered OUTSIDE, yet those tiles have ~0 entries "outside" and
        thousands "inside".  True: the shard starts in the middle of that field (entered INSIDE): the tiles have no
        entries "inside" and thousands "outside"."""
        body = np.full(k * T, ord("y"), dtype=np.uint8)
        body[8:k * T - 32:24] = 0x2C
        if not truly_inside:
            body[6] = 0x22
        body[k * T - 2] = 0x22
        rows = np.frombuffer((b"aaaa,bbbb\n" * ((9 * T) // 10 + 40))[: 9 * T + 333], dtype=np.uint8)
        return np.concatenate([body, rows])

    # tile 0 lies, the tiles behind it tell the truth: round 2's one-tile vote chose wrong here, eight tiles do not
    for truly_inside in (False, True):
        for k in (1, 3):
            b = lying_tiles(k, truly_inside)
            got, r = gpu_index(ctx, torch_cuda, b, in_quote_in=pkg.ENTER_GUESS)
            assert r.in_quote_in_used == int(truly_inside), (k, truly_inside)
            want, q = oracle.scalar_index(b, in_quote_in=int(truly_inside))
            assert (r.count, r.in_quote_out) == (want.size, q) and np.array_equal(got, want), (k, truly_inside)
    # the same with only two tiles behind the lying one (a shard of three tiles: all three vote)
    b = lying_tiles(1, False)[: 2 * T + 333]
    got, r = gpu_index(ctx, torch_cuda, b, in_quote_in=pkg.ENTER_GUESS)
    assert r.in_quote_in_used == 0
    want, q = oracle.scalar_index(b, in_quote_in=0)
    assert (r.count, r.in_quote_out) == (want.size, q) and np.array_equal(got, want)
    # adversarial for good: ALL eight voting tiles lie (2 MiB inside one quoted field, truly entered outside): the guess
    # says "inside", says so in the record, and everything is the oracle's for THAT state — the stitch then re-emits
    b = lying_tiles(8, False)
    got, r = gpu_index(ctx, torch_cuda, b, in_quote_in=pkg.ENTER_GUESS)
    assert r.in_quote_in_used == 1
    want, q = oracle.scalar_index(b, in_quote_in=1)
    assert (r.count, r.in_quote_out) == (want.size, q) and np.array_equal(got, want)
    # explicit states still mean what they say, and say so
    for inq in (0, 1):
        got, r = gpu_index(ctx, torch_cuda, b, in_quote_in=inq)
        want, q = oracle.scalar_index(b, in_quote_in=inq)
        assert r.in_quote_in_used == inq and r.count == want.size and np.array_equal(got, want)
    # count-only launch (no tape): the same choice, the same counts
    dbuf = torch_cuda.from_numpy(b).cuda()
    r0 = ctx.stage1_index_device(dbuf.data_ptr(), b.size, 0, pkg.ENTER_GUESS, 0, 0)
    want, q = oracle.scalar_index(b, in_quote_in=1)
    assert (r0.in_quote_in_used, r0.count, r0.in_quote_out, r0.written) == (1, want.size, q, 0)
    # an empty shard and a shard smaller than a tile
    got, r = gpu_index(ctx, torch_cuda, np.zeros(0, dtype=np.uint8), in_quote_in=pkg.ENTER_GUESS)
    assert (r.count, r.in_quote_in_used, r.in_quote_out) == (0, 0, 0)
    small = np.frombuffer(b'tail of a quoted field",x,y\n1,2,3\n', dtype=np.uint8).copy()
    got, r = gpu_index(ctx, torch_cuda, small, in_quote_in=pkg.ENTER_GUESS)
    want, q = oracle.scalar_index(small, in_quote_in=1)
    assert r.in_quote_in_used == 1 and np.array_equal(got, want)
    # back-to-back launches: the vote counter is reset by every launch (the second shard has other tiles)
    for _ in range(3):
        for truly_inside in (False, True):
            b = lying_tiles(1, truly_inside)
            got, r = gpu_index(ctx, torch_cuda, b, in_quote_in=pkg.ENTER_GUESS)
            assert r.in_quote_in_used == int(truly_inside)


def test_tape_capacity_and_count_only(ctx, torch_cuda, pkg, oracle):
    rng = np.random.default_rng(8)
    d = random_csvish(rng, 300000, 0.01)
    want, q = oracle.scalar_index(d)
    for cap in (0, 1, 1000, want.size - 1, want.size):
        got, r = gpu_index(ctx, torch_cuda, d, cap=cap)
        assert r.count == want.size and np.array_equal(got, want[:cap])
    # count-only (dtape NULL)
    dbuf = torch_cuda.from_numpy(d).cuda()
    r = ctx.stage1_index_device(dbuf.data_ptr(), d.size)
    assert (r.count, r.in_quote_out) == (want.size, q)
    # host entry point: exact retry protocol of include/csvsimd.h
    rc, n, _ = ctx.read_into(d, np.zeros(10, dtype=np.uint64))
    assert rc == pkg.ERR_TAPE_CAPACITY and n == want.size + 1
    tape = np.zeros(n, dtype=np.uint64)
    rc, n2, q2 = ctx.read_into(d, tape)
    assert rc == 0 and n2 == n and q2 == q and tape[0] == 0 and np.array_equal(tape[1:], want)


def test_synth_generator_matches_oracle(ctx, torch_cuda, pkg, oracle):
    for name, (cols, width, seed, q) in pkg.WORKLOADS.items():
        for off, n in ((0, 100000), (123457, 70001)):
            d = torch_cuda.empty(n + 3, dtype=torch_cuda.uint8, device="cuda:0")
            pkg.synth_fill_device(d.data_ptr(), off, n, cols, width, seed, q)
            assert np.array_equal(d[:n].cpu().numpy(), oracle.synth(off, n, cols, width, seed, q)), name


def test_checksum_kernel_matches_oracle(ctx, torch_cuda, pkg, oracle):
    rng = np.random.default_rng(3)
    t = np.sort(rng.integers(0, 2**40, size=100003).astype(np.uint64))
    dt = torch_cuda.from_numpy(t.view(np.int64)).cuda()
    out = torch_cuda.zeros(2, dtype=torch_cuda.int64, device="cuda:0")
    pkg.tape_checksum_device(dt.data_ptr(), t.size, 17, out.data_ptr())
    got = tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist())
    assert got == oracle.tape_checksum(t, 17)


def test_split_invariance_and_idempotence(ctx, torch_cuda, pkg, oracle):
    # a shard cut anywhere, with the carried state, reproduces the single-pass tape
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    n = 6 * 1024 * 1024 + 999
    dbuf = torch_cuda.empty(n, dtype=torch_cuda.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // 16
    full = torch_cuda.empty(cap, dtype=torch_cuda.int64, device="cuda:0")
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, full.data_ptr(), cap)
    again = torch_cuda.empty(cap, dtype=torch_cuda.int64, device="cuda:0")
    r2 = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, again.data_ptr(), cap)
    assert r2.count == r.count and torch_cuda.equal(full[: r.count], again[: r.count])
    want, _ = oracle.scalar_index(dbuf.cpu().numpy())
    assert np.array_equal(full[: r.count].cpu().numpy().view(np.uint64), want)
    for cut in (777, 2 * 1024 * 1024 + 777, n - 5):
        a = torch_cuda.empty(cap, dtype=torch_cuda.int64, device="cuda:0")
        b = torch_cuda.empty(cap, dtype=torch_cuda.int64, device="cuda:0")
        ra = ctx.stage1_index_device(dbuf.data_ptr(), cut, 0, 0, a.data_ptr(), cap)
        rb = ctx.stage1_index_device(dbuf.data_ptr() + cut, n - cut, cut, ra.in_quote_out, b.data_ptr(), cap)
        assert ra.count + rb.count == r.count and rb.in_quote_out == r.in_quote_out
        cat = torch_cuda.cat([a[: ra.count], b[: rb.count]])
        assert torch_cuda.equal(cat, full[: r.count]), cut
        # and the stitch arithmetic agrees with the carried run
        spec = ctx.stage1_index_device(dbuf.data_ptr() + cut, n - cut, cut, 0)
        st = pkg.stitch_shards([ra, spec], 1)
        assert (st.in_quote_in, st.count, st.tape_index_base) == (ra.in_quote_out, rb.count, 1 + ra.count)


# ---- BASELINE.json configs at full size ------------------------------------------------------------
def run_workload(ctx, torch, pkg, name, target):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, target)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 16
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap)
    assert r.error == 0
    return n, dbuf, dtape, r, (cols, width, seed, q)


@pytest.mark.parametrize("name", ["16x32_noquote", "1024x4_dense", "64x31_noquote"])
def test_full_size_noquote_analytic_tape(ctx, torch_cuda, pkg, oracle, name):
    # configs 2, 5 and the 1-GPU shape of config 4 at 1 GiB: the tape of a quote-free corpus is
    # known in closed form (every (width+1)-th byte), so the whole tape is checked bit for bit.
    torch = torch_cuda
    n, dbuf, dtape, r, (cols, width, seed, q) = run_workload(ctx, torch, pkg, name, 1 << 30)
    S = n // (width + 1)
    assert r.count == S and r.in_quote_out == 0 and r.quote_parity == 0
    want = torch.arange(1, S + 1, dtype=torch.int64, device="cuda:0") * (width + 1) - 1
    assert torch.equal(dtape[:S], want)
    # oracle (the SSE restatement) on a 64 MiB window of the same bytes, entry by entry
    win = 64 << 20
    host = dbuf[:win].cpu().numpy()
    o = oracle.sse_read(host)
    assert np.array_equal(o[1:], dtape[: o.size - 1].cpu().numpy().view(np.uint64))
    # checksum of checksums: device kernel == oracle formula on the analytic tape's first window
    out = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    pkg.tape_checksum_device(dtape.data_ptr(), o.size - 1, 1, out.data_ptr())
    assert tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist()) == oracle.tape_checksum(o[1:], 1)


def test_full_size_quoted_config3(ctx, torch_cuda, pkg, oracle):
    # config 3: 1 GiB, 10 % quoted fields hiding ',' and LF: the quote-carry path end to end
    torch = torch_cuda
    n, dbuf, dtape, r, (cols, width, seed, q) = run_workload(ctx, torch, pkg, "16x32_q10", 1 << 30)
    S = n // (width + 1)
    assert r.count == S and r.in_quote_out == 0   # hidden delimiters are not counted
    t = dtape[:S]
    assert bool((t[1:] > t[:-1]).all())           # strictly ascending
    b = dbuf[t]                                    # every entry points at a structural byte
    assert bool(((b == 0x2C) | (b == 0x0A)).all())
    rows = n // (cols * (width + 1))
    assert int((b == 0x0A).sum()) == rows
    # full compare with the oracle on the first 256 MiB (entry by entry) and a checksum on all of it
    host = dbuf.cpu().numpy()
    win = 256 << 20
    o = oracle.sse_read(host[:win])
    assert np.array_equal(o[1:], t[: o.size - 1].cpu().numpy().view(np.uint64))
    want, inq = oracle.scalar_index(host)
    assert want.size == S and inq == 0
    out = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    pkg.tape_checksum_device(dtape.data_ptr(), S, 1, out.data_ptr())
    assert tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist()) == oracle.tape_checksum(want, 1)


def test_config4_shard_8gib_analytic(ctx, torch_cuda, pkg):
    # config 4's per-GPU shard: 8 GiB of the 64-col corpus taken from the MIDDLE of the 64 GiB
    # file (rank 3 of 8), with the deliberately misaligned start (+777 bytes, mid-row).
    torch = torch_cuda
    cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
    shard = 1 << 33
    lo = 3 * shard + 777
    dbuf = torch.empty(shard, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), lo, shard, cols, width, seed, q)
    first = lo // (width + 1) * (width + 1) + width  # first delimiter at or after lo
    S = (lo + shard - 1 - first) // (width + 1) + 1
    dtape = torch.empty(S + 8, dtype=torch.int64, device="cuda:0")
    ctx.reserve(shard)
    r = ctx.stage1_index_device(dbuf.data_ptr(), shard, lo, 0, dtape.data_ptr(), S + 8)
    assert r.count == S and r.error == 0 and r.in_quote_out == 0
    want = torch.arange(0, S, dtype=torch.int64, device="cuda:0") * (width + 1) + first
    assert torch.equal(dtape[:S], want)


def test_repeatability_stress(ctx, torch_cuda, pkg):
    # The look-back is a concurrent protocol: one green run proves little.  Re-run the 1 GiB quoted
    # corpus many times (emitting and count-only kernels) and demand identical counts and tapes.
    # (This test is what exposed a stale-tile-id race at ~1 in 10^5 tiles during development.)
    torch = torch_cuda
    n, dbuf, dtape, r0, (cols, width, seed, q) = run_workload(ctx, torch, pkg, "16x32_q10", 1 << 30)
    S = n // (width + 1)
    assert r0.count == S
    ref = dtape[:S].clone()
    cap = dtape.numel()
    for rep in range(25):
        dtape.fill_(-1)
        r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap)
        assert (r.count, r.in_quote_out, r.error) == (S, 0, 0), rep
        assert torch.equal(dtape[:S], ref), rep
        assert bool((dtape[S:] == -1).all())
    for rep in range(50):
        r = ctx.stage1_index_device(dbuf.data_ptr(), n)
        assert (r.count, r.count_enter_outside, r.in_quote_out) == (S, S, 0), rep


def test_host_ingest_pipeline_multi_chunk(ctx, pkg, oracle, tmp_path, monkeypatch):
    # csvsimd_stage1_index streams the file in chunks (pinned here to the 32-MiB maximum; smaller files
    # use ~len/4) over a two-slot pipeline: quoted regions and tape bases must carry across chunk
    # boundaries; a dense chunk takes the exact-capacity retry; csvsimd_create runs the same path from a file.
    monkeypatch.setenv("CSVSIMD_INGEST_CHUNK_MIB", "32")
    rng = np.random.default_rng(4711)
    n = (32 << 20) * 2 + 12345                      # 3 chunks, ragged tail
    d = random_csvish(rng, n, 0.002)                # long quoted stretches cross the boundaries
    d[(32 << 20) - 3: (32 << 20) + 3] = np.frombuffer(b',"\n,",', dtype=np.uint8)
    got = ctx.read(d)
    want = oracle.scalar_read(d)
    assert got.size == want.size and np.array_equal(got, want)
    dense = np.full((32 << 20) + 77, 0x2C, dtype=np.uint8)   # every byte structural: > 1 entry / 4 B
    got = ctx.read(dense)
    assert got.size == dense.size + 1 and got[0] == 0 and np.array_equal(got[1:], np.arange(dense.size, dtype=np.uint64))
    # count-only call and capacity protocol across chunks
    rc, cnt, q = ctx.read_into(d, None)
    assert rc == 0 and cnt == want.size
    small = np.zeros(1000, dtype=np.uint64)
    rc, cnt, _ = ctx.read_into(d, small)
    assert rc == pkg.ERR_TAPE_CAPACITY and cnt == want.size and np.array_equal(small, want[:1000])
    # from a file: a regular 3-column CSV of ~40 MiB, adaptive chunk size (about a quarter of the file)
    monkeypatch.delenv("CSVSIMD_INGEST_CHUNK_MIB")
    rows = 1_500_000
    body = b"Name,Number,Done\n" + b"".join(b"abc%07d,%09d,\"y,n\"\n" % (i, i * 7) for i in range(rows))
    p = tmp_path / "big.csv"
    p.write_bytes(body)
    t = ctx.create(str(p))
    assert (t.field_cnt, t.record_cnt, t.new_line) == (3, rows + 1, "LF")
    assert t.seek_field(rows - 1, 2) == b'"y,n"' and t.seek_field(123456, 0) == b"abc0123456"
    assert np.array_equal(t.index(), oracle.scalar_read(body))


def test_host_ingest_chained_chunks_ramped_plan_and_mid_pipeline_retry(ctx, pkg, oracle):
    """A file large enough for the ramped chunk plan (4, 8, 16, 32 ... 16, 8 MiB).  Chunks are chained ON THE DEVICE
    (chunk i + 1 reads its entering state from chunk i's record; the host reads records one chunk late), so: quoted
    stretches that cross every kind of boundary, a chunk in the MIDDLE that is denser than the capacity guess (its
    re-run happens after the chunk behind it was already enqueued from its first record), an odd number of quotes in
    that dense chunk (the state it hands on must be right the first time), and the capacity protocol at this size."""
    rng = np.random.default_rng(90210)
    n = (168 << 20) + 54321
    d = random_csvish(rng, n, 0.001)
    cuts = [4 << 20, 12 << 20, 28 << 20, 60 << 20, 92 << 20]
    for c in cuts:                                                     # a quote right at, before and after each cut
        d[c - 2: c + 3] = np.frombuffer(b',"\n",', dtype=np.uint8)[:5]
    dense0 = 61 << 20                                                  # inside the chunk [60, 92) MiB
    d[dense0: dense0 + (24 << 20)] = 0x2C                              # 24 MiB of commas: > 1 entry per 4 bytes
    d[dense0 + 12345] = 0x22                                           # ... with ONE quote in it: parity flips here
    d[dense0 + (23 << 20)] = 0x22                                      # ... and back, 11 MiB of commas are quoted
    d[dense0 + (23 << 20) + 5] = 0x22                                  # ... and once more: the chunk leaves INSIDE a string
    want = oracle.scalar_read(d)
    got = ctx.read(d)
    assert got.size == want.size and np.array_equal(got, want)
    rc, cnt, q = ctx.read_into(d, None)
    assert rc == 0 and cnt == want.size
    assert q == int(np.count_nonzero(d == 0x22) & 1)
    small = np.zeros(100_000, dtype=np.uint64)
    rc, cnt, _ = ctx.read_into(d, small)
    assert rc == pkg.ERR_TAPE_CAPACITY and cnt == want.size and np.array_equal(small, want[:100_000])


def test_host_ingest_three_threads_many_slot_reuses(pkg, oracle, monkeypatch):
    """The round-4 pipeline: a stager thread (up to three chunks ahead), the submitting caller, an expander thread, four
    slots.  Forty-odd 4-MiB chunks reuse every slot ten times; several chunks — neighbours among them, the first and the
    last — are denser than the capacity guess and take the exact-capacity re-run while the chunks behind them are already
    enqueued; quoted stretches cross chunk boundaries.  Then the same context reads a two-chunk file (everything on the
    caller's thread) and a tiny one, and the phase record says which pipeline ran."""
    monkeypatch.setenv("CSVSIMD_INGEST_CHUNK_MIB", "4")
    rng = np.random.default_rng(31337)
    chunk = 4 << 20
    n = 43 * chunk + 4321
    d = random_csvish(rng, n, 0.001)
    for c in (0, 7, 8, 9, 21, 42):                                     # dense chunks: > 1 entry per 4 bytes
        d[c * chunk + 100: (c + 1) * chunk - 100] = 0x2C
        d[c * chunk + 5000] = 0x22                                     # an odd number of quotes inside: the state handed on flips
    c = pkg.Context(0)
    try:
        want = oracle.scalar_read(d)
        for rep in range(2):                                           # the second call finds every slot allocated
            got = c.read(d)
            assert got.size == want.size and np.array_equal(got, want), rep
            ph = pkg.ingest_last_phases()
            assert ph["host_threads"] == 3 and ph["chunks"] == 43 and ph["bytes"] == n
            assert ph["wall"] > 0 and ph["stage_copy"] > 0 and ph["expand_copy"] > 0
        rc, cnt, q = c.read_into(d, None)                              # count only: no expander
        assert rc == 0 and cnt == want.size and q == int(np.count_nonzero(d == 0x22) & 1)
        small = np.zeros(12345, dtype=np.uint64)
        rc, cnt, _ = c.read_into(d, small)
        assert rc == pkg.ERR_TAPE_CAPACITY and cnt == want.size and np.array_equal(small, want[:12345])
        two = d[: 2 * chunk + 17]                                        # two chunks (the 17-byte stub is folded): everything
        assert np.array_equal(c.read(two), oracle.scalar_read(two))      # in turn on the caller's thread
        ph = pkg.ingest_last_phases()
        assert ph["host_threads"] == 1 and ph["chunks"] == 2
        monkeypatch.delenv("CSVSIMD_INGEST_CHUNK_MIB")
        mid = d[7 * chunk - 5: 12 * chunk + 99]                          # 20 MiB, the library's own plan (0.75 ... 8 ... 1.5 MiB):
        assert np.array_equal(c.read(mid), oracle.scalar_read(mid))      # dense chunks among small ones, three threads
        ph = pkg.ingest_last_phases()
        assert ph["host_threads"] == 3 and ph["chunks"] == len(pkg.ingest_chunk_plan(mid.size)) - 1 >= 6
        for lo, n_ in ((0, (1 << 20) + 1), (3 * chunk - 77, (2 << 20) + 4097), (8 * chunk - 5000, (4 << 20) + 1),
                       (6 * chunk + 33, 8 << 20)):                       # a few MiB: 3 ... 8 chunks of 256 KiB ... 2 MiB
            part = d[lo: lo + n_]
            assert np.array_equal(c.read(part), oracle.scalar_read(part)), (lo, n_)
            rc, cnt, q = c.read_into(part, None)
            assert rc == 0 and cnt == oracle.scalar_read(part).size
        tiny = d[9 * chunk - 150: 9 * chunk + 150]
        assert np.array_equal(c.read(tiny), oracle.scalar_read(tiny))
        assert np.array_equal(c.read(d), want)                         # and the large file again after the small ones
    finally:
        c.close()


def test_async_entry_point_is_graph_capturable(ctx, torch_cuda, pkg, oracle):
    # include/csvsimd.h promises: no allocation and no synchronisation inside
    # csvsimd_stage1_index_device_async once the scratch is reserved -> it can be captured into a
    # hipGraph (one kernel node per launch: the launch clears nothing and reduces nothing outside itself) and replayed on
    # new contents of the same buffers.
    torch = torch_cuda
    rng = np.random.default_rng(12)
    n = 3 * pkg.tile_bytes() + 4321
    dbuf = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    cap = n + 1
    dtape = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    ctx.reserve(n)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ctx.stage1_index_device_async(dbuf.data_ptr(), n, 7, 0, dtape.data_ptr(), cap, dres.data_ptr(), side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        ctx.stage1_index_device_async(dbuf.data_ptr(), n, 7, 0, dtape.data_ptr(), cap, dres.data_ptr(), s)
    from csv_simd_amd import sharded
    for rep in range(3):
        d = random_csvish(rng, n, 0.05)
        dbuf.copy_(torch.from_numpy(d))
        dtape.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        r = sharded.result_from_words(dres.cpu().tolist())
        want, q = oracle.scalar_index(d, base_off=7)
        assert (r.count, r.in_quote_out, r.error) == (want.size, q, 0), rep
        assert np.array_equal(dtape[: r.count].cpu().numpy().view(np.uint64), want), rep


def test_misaligned_tape_pointer(ctx, torch_cuda, oracle):
    # the 16-byte tape stores peel one entry when the tape pointer / running index is odd
    torch = torch_cuda
    rng = np.random.default_rng(13)
    d = random_csvish(rng, 500_000, 0.01)
    want, _ = oracle.scalar_index(d)
    dbuf = torch.from_numpy(d).cuda()
    backing = torch.full((want.size + 16,), -1, dtype=torch.int64, device="cuda:0")
    for shift in (0, 1):   # 8-byte shift = 16-byte misalignment of the shard's tape
        backing.fill_(-1)
        view = backing[shift:]
        r = ctx.stage1_index_device(dbuf.data_ptr(), d.size, 0, 0, view.data_ptr(), want.size)
        assert r.count == want.size
        assert np.array_equal(view[: want.size].cpu().numpy().view(np.uint64), want)
        assert bool((backing[:shift] == -1).all()) and bool((view[want.size:] == -1).all())


def test_device_field_spans_and_gather(ctx, torch_cuda, pkg, golden, oracle):
    # RecordSource::seek_field for all records at once on the device must agree
    # with the host-side seek_field (reference src/record_source.rs:106-140)
    torch = torch_cuda

    def check(data: bytes, fields_to_check):
        host_index = ctx.read(data)
        t = pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), host_index)
        dbytes = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        # device-resident tape with its sentinel: entries land at dindex[1:]
        dindex = torch.zeros(host_index.size + 4, dtype=torch.int64, device="cuda:0")
        r = ctx.stage1_index_device(dbytes.data_ptr(), len(data), 0, 0, dindex.data_ptr() + 8, host_index.size + 3)
        assert r.count + 1 == host_index.size
        nrec = t.record_cnt - 1
        for f in fields_to_check:
            b = torch.full((nrec + 3,), -1, dtype=torch.int64, device="cuda:0")
            e = torch.full((nrec + 3,), -1, dtype=torch.int64, device="cuda:0")
            nv = pkg.tape_field_spans_device(dindex.data_ptr(), host_index.size, t.field_cnt, t.new_line, f, 0, nrec + 3,
                                             b.data_ptr(), e.data_ptr())
            assert nv == nrec and bool((b[nrec:] == -1).all())
            stride = 48
            dst = torch.zeros(nrec * stride, dtype=torch.uint8, device="cuda:0")
            ln = torch.zeros(nrec, dtype=torch.int32, device="cuda:0")
            pkg.gather_fields_device(dbytes.data_ptr(), len(data), b.data_ptr(), e.data_ptr(), nrec, dst.data_ptr(), stride,
                                     ln.data_ptr())
            bh, eh, dh, lh = b.cpu().tolist(), e.cpu().tolist(), dst.cpu().numpy().reshape(nrec, stride), ln.cpu().tolist()
            step = max(1, nrec // 500)
            for rec in list(range(0, nrec, step)) + [nrec - 1]:
                # the definition: seek_field restated line by line in the oracle (src/record_source.rs:106-140);
                # the product's own host mirror must say the same
                want = oracle.seek_field(data, host_index, t.field_cnt, t.new_line == "CRLF", rec, f)
                assert want == t.seek_field(rec, f)
                assert data[bh[rec]: eh[rec]] == want, (rec, f)
                assert lh[rec] == len(want)
                row = bytes(dh[rec][: min(len(want), stride)])
                assert row == want[:stride] and not dh[rec][len(want):].any()
        # whole records: seek_record for all rows at once
        rb = torch.full((nrec + 2,), -1, dtype=torch.int64, device="cuda:0")
        re_ = torch.full((nrec + 2,), -1, dtype=torch.int64, device="cuda:0")
        assert pkg.tape_record_spans_device(dindex.data_ptr(), host_index.size, t.field_cnt, t.new_line, 0, nrec + 2,
                                            rb.data_ptr(), re_.data_ptr()) == nrec
        assert bool((rb[nrec:] == -1).all())
        rbh, reh = rb.cpu().tolist(), re_.cpu().tolist()
        for rec in list(range(0, nrec, max(1, nrec // 300))) + [nrec - 1]:
            assert data[rbh[rec]: reh[rec]] == t.seek_record(rec), rec
        assert pkg.tape_record_spans_device(dindex.data_ptr(), host_index.size, t.field_cnt, t.new_line, nrec, 3,
                                            rb.data_ptr(), re_.data_ptr()) == 0
        # the Ok(None) cases of seek_field
        assert pkg.tape_field_spans_device(dindex.data_ptr(), host_index.size, t.field_cnt, t.new_line, t.field_cnt, 0, 5, b.data_ptr(), e.data_ptr()) == 0
        assert pkg.tape_field_spans_device(dindex.data_ptr(), host_index.size, t.field_cnt, t.new_line, 0, nrec, 5, b.data_ptr(), e.data_ptr()) == 0
        return t

    check(golden["sample.csv"][0], [0, 1, 2])
    check(golden["sample_rx.csv"][0], [0, 2, 5, 7])       # BOM, CRLF, quoted commas
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    data = bytes(oracle.synth(0, 4000 * cols * (width + 1), cols, width, seed, 0))   # LF rows, 16 fields
    check(data, [0, 7, 15])
    # a ragged index is rejected like TapeCore::init does
    rag = golden["reader_test01.csv"][0]
    idx = ctx.read(rag)
    d = torch.from_numpy(idx.view(np.int64)).cuda()
    with pytest.raises(pkg.StructureError) as ei:
        pkg.tape_field_spans_device(d.data_ptr(), idx.size, 3, "LF", 0, 0, 1, d.data_ptr(), d.data_ptr())
    assert ei.value.code == pkg.ERR_INVALID_CSV_FORMAT


def test_native_rccl_sharded_step_single_rank(ctx, torch_cuda, pkg, oracle):
    # csvsimd_stage1_index_sharded with a real RCCL communicator (world = 1 is all a 1-GPU box
    # allows): speculative pass, ncclAllGather of the result record, stitch, forced re-emit.
    torch = torch_cuda
    rng = np.random.default_rng(21)
    d = random_csvish(rng, 2 * pkg.tile_bytes() + 999, 0.03)
    dbuf = torch.from_numpy(d).cuda()
    cap = d.size
    dtape = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
    comm = pkg.Comm(pkg.Comm.unique_id(), 0, 1, 0)
    try:
        for file_inq in (0, 1):   # 1: the shard really starts inside a string -> second pass
            dtape.fill_(-1)
            r, st = comm.index_sharded(ctx, dbuf.data_ptr(), d.size, 1000, dtape.data_ptr(), cap, file_inq)
            want, q = oracle.scalar_index(d, base_off=1000, in_quote_in=file_inq)
            assert (st.in_quote_in, st.count, st.tape_index_base, st.total_entries) == (file_inq, want.size, 1, want.size + 1)
            assert (r.count, r.in_quote_out, st.in_quote_final) == (want.size, q, q)
            assert np.array_equal(dtape[: r.count].cpu().numpy().view(np.uint64), want)
    finally:
        comm.close()


def test_config4_q10_shard_with_carried_state(ctx, torch_cuda, pkg, oracle):
    # the quoted variant of config 4 (SURVEY §8d "64x31_q10 exercises the parity stitch"): a 2 GiB
    # shard cut mid-row (+777) out of the middle of the file, entered with whatever state the
    # preceding bytes leave, checked against the oracle by count, state and order-sensitive checksum
    torch = torch_cuda
    cols, width, seed, q = pkg.WORKLOADS["64x31_q10"]
    shard = 2 << 30
    lo = 5 * shard + 777
    dbuf = torch.empty(shard, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), lo, shard, cols, width, seed, q)
    host = dbuf.cpu().numpy()
    cap = shard // 24
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    for inq in (0, 1):
        want, q_out = oracle.scalar_index(host, base_off=lo, in_quote_in=inq)
        r = ctx.stage1_index_device(dbuf.data_ptr(), shard, lo, inq, dtape.data_ptr(), cap)
        assert (r.count, r.in_quote_out, r.error) == (want.size, q_out, 0)
        p, c0, c1 = oracle.shard_descriptor(host)
        assert (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1)
        out = torch.zeros(2, dtype=torch.int64, device="cuda:0")
        pkg.tape_checksum_device(dtape.data_ptr(), r.count, 1, out.data_ptr())
        assert tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist()) == oracle.tape_checksum(want, 1)
        t = dtape[: r.count]
        assert bool((t[1:] > t[:-1]).all())
        head = 1 << 20
        assert np.array_equal(t[:head].cpu().numpy().view(np.uint64), want[:head])
        assert np.array_equal(t[-head:].cpu().numpy().view(np.uint64), want[-head:])
    # the same shard with nobody telling it how it is entered: the kernel's own choice must be the truth (the bytes
    # before the cut, from the CPU generator), and the tape the one of that state
    row = cols * (width + 1)
    r0 = (lo // row) * row
    truth = int(np.count_nonzero(oracle.synth(r0, lo - r0, cols, width, seed, q) == 0x22)) & 1
    r = ctx.stage1_index_device(dbuf.data_ptr(), shard, lo, pkg.ENTER_GUESS, dtape.data_ptr(), cap)
    want, q_out = oracle.scalar_index(host, base_off=lo, in_quote_in=truth)
    assert (r.in_quote_in_used, r.count, r.in_quote_out, r.error) == (truth, want.size, q_out, 0)
    out = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    pkg.tape_checksum_device(dtape.data_ptr(), r.count, 1, out.data_ptr())
    assert tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist()) == oracle.tape_checksum(want, 1)


def test_fuzz_many_shapes(ctx, torch_cuda, pkg, oracle):
    # randomized sizes (0 .. ~5 tiles), alignments, entering states, offsets and quote densities
    torch = torch_cuda
    rng = np.random.default_rng(20261003)
    T = pkg.tile_bytes()
    for case in range(120):
        n = int(rng.choice([rng.integers(0, 300), rng.integers(300, 70000), rng.integers(T - 70, T + 70),
                            rng.integers(T, 5 * T)]))
        pq = float(rng.choice([0.0, 0.0005, 0.01, 0.2, 0.5]))
        d = random_csvish(rng, n, pq)
        if n and rng.random() < 0.3:   # long runs of one byte class
            a, b = sorted(rng.integers(0, n + 1, size=2))
            d[a:b] = rng.choice(np.frombuffer(b',"\na', dtype=np.uint8))
        mis = int(rng.integers(0, 128))
        inq = int(rng.integers(0, 2))
        base = int(rng.integers(0, 2**40))
        got, r = gpu_index(ctx, torch, d, base_off=base, in_quote_in=inq, misalign=mis)
        want, q = oracle.scalar_index(d, base_off=base, in_quote_in=inq)
        assert (r.count, r.in_quote_out, r.error) == (want.size, q, 0), (case, n, pq, mis, inq)
        assert np.array_equal(got, want), (case, n, pq, mis, inq)


def test_more_entries_than_the_reference_u32_index(ctx, torch_cuda, pkg):
    # Maximum sizes: the reference keeps its tape position in a u32 (`array_idx`, src/reader.rs:218) and
    # record_cnt in a u32 (src/tape.rs:329), so it stops being defined at 2^32 entries.  This path
    # counts and indexes in 64 bits: one launch over 21.5 GB of the dense corpus = 2^32 + 1024 entries
    # (34 GB of tape), every entry checked against the closed form 5 j + 4.
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~60 GB of free HBM")
    cols, width, seed, q = pkg.WORKLOADS["1024x4_dense"]
    row = cols * (width + 1)
    rows = ((2**32 + 1024) * (width + 1) + row - 1) // row
    n = rows * row
    S = n // (width + 1)
    assert S > 2**32
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    dtape = torch.empty(S + 8, dtype=torch.int64, device="cuda:0")
    dtape[S:] = -1
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), S)
    assert (r.count, r.written, r.in_quote_out, r.error) == (S, S, 0, 0)
    assert r.count_enter_outside == S and r.count_enter_inside == 0   # no quote ever closes a string entered inside
    assert bool((dtape[S:] == -1).all())
    step = 1 << 28
    for lo in range(0, S, step):
        hi = min(S, lo + step)
        want = torch.arange(lo, hi, dtype=torch.int64, device="cuda:0") * (width + 1) + width
        assert torch.equal(dtape[lo:hi], want), lo
        del want
    # a shard that starts beyond 2^32 entries of tape: base offsets are 64-bit too
    r2 = ctx.stage1_index_device(dbuf.data_ptr(), 1 << 20, 2**45 + 5, 0, dtape.data_ptr(), S)
    assert r2.count == (1 << 20) // (width + 1)
    assert int(dtape[0]) == 2**45 + 5 + width and int(dtape[r2.count - 1]) == 2**45 + 5 + (r2.count - 1) * 5 + width


@pytest.mark.parametrize("line_end", ["\n", "\r\n"])
def test_end_to_end_against_python_csv_module(ctx, pkg, tmp_path, line_end):
    # An independent parser as the judge of the whole chain: files written by Python's csv module
    # (RFC 4180 quoting: fields with commas, quotes — doubled — and line breaks inside quotes) go
    # through csv_simd::create -> seek_field; un-quoted, every field must equal what was written.
    import csv
    import io
    rng = np.random.default_rng(4180)
    pool = ["plain", "", "with,comma", 'say "hi"', "two\nlines", "cr\rlf\r\n inside", " padded ", "x" * 70, '"', ",",
            "é世界", "a,b,\"c\"\n,d"]
    n_fields = 7
    rows = [[pool[i] for i in rng.integers(0, len(pool), size=n_fields)] for _ in range(5000)]
    buf = io.StringIO()
    w = csv.writer(buf, lineterminator=line_end, quoting=csv.QUOTE_MINIMAL)
    w.writerow([f"col{i}" for i in range(n_fields)])
    w.writerows(rows)
    data = buf.getvalue().encode()
    assert list(csv.reader(io.StringIO(buf.getvalue(), newline="")))[1:] == rows   # the csv module agrees with itself
    p = tmp_path / "rfc4180.csv"
    p.write_bytes(data)
    t = ctx.create(str(p))
    assert t.field_cnt == n_fields and t.record_cnt == len(rows) + 1
    assert t.new_line == ("CRLF" if line_end == "\r\n" else "LF")
    for r in list(range(0, len(rows), 37)) + [len(rows) - 1]:
        for f in range(n_fields):
            raw = t.seek_field(r, f).decode()
            if raw[:1] == '"':
                assert raw[-1] == '"'
                raw = raw[1:-1].replace('""', '"')
            assert raw == rows[r][f], (r, f)
    # the whole record too
    assert t.seek_record(0).decode().count(",") >= n_fields - 1


def test_distinct_contexts_on_concurrent_host_threads(pkg, torch_cuda, oracle):
    # include/csvsimd.h: "thread-safe for distinct contexts".  Four host threads, one context each,
    # run the host-buffer entry point (private streams, staging pool) and the device entry point at
    # the same time on the same GPU; every result must equal the oracle's.
    import threading
    torch = torch_cuda
    rng = np.random.default_rng(99)
    inputs = [random_csvish(rng, int(n), 0.03) for n in (3_000_001, 70_000_000, 1_234_567, 40_000_003)]
    wants = [oracle.scalar_index(d)[0] for d in inputs]
    errors = []

    def worker(k):
        try:
            c = pkg.Context(0)
            d, want = inputs[k], wants[k]
            dbuf = torch.from_numpy(d.copy()).cuda()
            dtape = torch.empty(want.size + 8, dtype=torch.int64, device="cuda:0")
            for it in range(6):
                got = c.read(d)
                assert got[0] == 0 and np.array_equal(got[1:], want), (k, it, "host path")
                r = c.stage1_index_device(dbuf.data_ptr(), d.size, 0, 0, dtape.data_ptr(), want.size + 8)
                assert r.count == want.size, (k, it, "device path")
                assert np.array_equal(dtape[: r.count].cpu().numpy().view(np.uint64), want), (k, it)
            c.close()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_launch_epochs_wrap_and_shard_sizes_change(pkg, torch_cuda, oracle):
    # round 2: a launch is ONE kernel — the look-back words are tagged with a launch epoch instead of
    # being zeroed, and the last workgroup leaves the control block ready for the next launch.  Drive one
    # context through more than two wraps of the 10-bit epoch while the shard size keeps changing, so that
    # words left behind by larger earlier launches (same epoch value, one wrap earlier) would be picked up
    # if the wrap did not clear them.  Every launch: count, state and order-sensitive checksum vs the oracle.
    torch = torch_cuda
    c = pkg.Context(0)
    try:
        T = pkg.tile_bytes()
        rng = np.random.default_rng(77)
        big = random_csvish(rng, 9 * T + 4321, 0.03)
        sizes = [big.size, 3 * T + 17, T // 2 + 5, 5 * T, 64, 0, 7 * T + 1]
        dbuf = torch.from_numpy(big).cuda()
        dtape = torch.empty(big.size + 8, dtype=torch.int64, device="cuda:0")
        dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        want = {}
        for n in sizes:
            for inq in (0, 1):
                e, q = oracle.scalar_index(big[:n], base_off=5, in_quote_in=inq)
                want[(n, inq)] = (e.size, q, oracle.tape_checksum(e, 1))
        from csv_simd_amd import sharded
        launches = 2300
        sums = torch.zeros((launches, 2), dtype=torch.int64, device="cuda:0")
        recs = torch.zeros((launches, 8), dtype=torch.int64, device="cuda:0")
        s = torch.cuda.current_stream().cuda_stream
        plan = []
        for i in range(launches):
            n = sizes[(i * 5 + i // 7) % len(sizes)]
            inq = (i // 3) & 1
            plan.append((n, inq))
            c.stage1_index_device_async(dbuf.data_ptr(), n, 5, inq, dtape.data_ptr(), dtape.numel(),
                                        recs[i].data_ptr(), s)
            if i % 23 == 0 or i > launches - 40:   # the checksum reads the record, so it synchronises: sampled
                torch.cuda.synchronize()
                cnt = int(recs[i, 0])
                pkg.tape_checksum_device(dtape.data_ptr(), cnt, 1, sums[i].data_ptr(), s)
        torch.cuda.synchronize()
        recs_h, sums_h = recs.cpu().tolist(), sums.cpu().tolist()
        for i, (n, inq) in enumerate(plan):
            r = sharded.result_from_words(recs_h[i])
            cnt, q, chk = want[(n, inq)]
            assert (r.count, r.in_quote_out, r.error, r.written) == (cnt, q, 0, cnt), (i, n, inq)
            if i % 23 == 0 or i > launches - 40:
                assert tuple(int(x) & (2**64 - 1) for x in sums_h[i]) == chk, (i, n, inq)
    finally:
        c.close()


@pytest.mark.parametrize("world,shard_tiles", [(3, 1), (8, 9)])
@pytest.mark.parametrize("guess", [False, True])
def test_device_stitch_and_reemit_three_shards_one_gpu(pkg, torch_cuda, oracle, guess, world, shard_tiles):
    # the N > 1 step as bench.py / csvsimd_stage1_index_sharded drive it, with three (or, BASELINE config 4's shape,
    # eight) shards on ONE GPU and a device-to-device copy standing in for the all-gather: speculative pass -> records
    # side by side in device memory -> stitch kernel -> re-emit launch that reads its entering state from device
    # memory.  No host value is used between the first launch and the final copy-out.  Nine tiles per shard: the
    # eight-tile vote of CSVSIMD_ENTER_GUESS sees a full window and a tile it does not look at.
    torch = torch_cuda
    from csv_simd_amd import sharded
    rng = np.random.default_rng(5150)
    T = pkg.tile_bytes()
    for trial, p_quote in enumerate((0.0, 0.02, 0.11, 0.5)):
        n = world * shard_tiles * T + 1000 * trial + 99
        d = random_csvish(rng, n, p_quote)
        cuts = [0] + [r * shard_tiles * T + int(rng.integers(-T // 2, T // 2)) for r in range(1, world)] + [n]
        dbuf = torch.from_numpy(d).cuda()
        ctxs = [pkg.Context(0) for _ in range(world)]
        try:
            tapes = [torch.full((cuts[r + 1] - cuts[r] + 8,), -1, dtype=torch.int64, device="cuda:0") for r in range(world)]
            d_all = torch.zeros(8 * world, dtype=torch.int64, device="cuda:0")
            d_st = torch.zeros((world, sharded.STITCH_WORDS), dtype=torch.int64, device="cuda:0")
            d_fin = torch.zeros((world, 8), dtype=torch.int64, device="cuda:0")
            s = torch.cuda.current_stream().cuda_stream
            for file_inq in (0, 1):
                for r in range(world):
                    lo, hi = cuts[r], cuts[r + 1]
                    tapes[r].fill_(-1)
                    # first pass: speculate "entered outside", or (guess) let ranks > 0 choose from their first eight tiles
                    first = pkg.ENTER_GUESS if (guess and r > 0) else 0
                    ctxs[r].stage1_index_device_async(dbuf.data_ptr() + lo, hi - lo, lo, first, tapes[r].data_ptr(),
                                                      tapes[r].numel(), d_fin[r].data_ptr(), s)
                    d_all[8 * r: 8 * r + 8].copy_(d_fin[r])          # "all-gather"
                spec = [t.clone() for t in tapes]
                for r in range(world):
                    lo, hi = cuts[r], cuts[r + 1]
                    pkg.stitch_shards_device_async(d_all.data_ptr(), world, r, file_inq, d_st[r].data_ptr(), s)
                    ctxs[r].stage1_reemit_device_async(dbuf.data_ptr() + lo, hi - lo, lo, d_st[r].data_ptr(),
                                                       tapes[r].data_ptr(), tapes[r].numel(), d_fin[r].data_ptr(), s)
                torch.cuda.synchronize()
                want, q_final = oracle.scalar_index(d, in_quote_in=file_inq)
                got, state, base = [], file_inq, 1
                recs = [sharded.result_from_words(d_all[8 * r: 8 * r + 8].cpu().tolist()) for r in range(world)]
                for r in range(world):
                    lo, hi = cuts[r], cuts[r + 1]
                    st = sharded.stitch_from_words(d_st[r].cpu().tolist())
                    host_st = pkg.stitch_shards(recs, r, file_inq)     # device stitch == host stitch
                    for f in ("in_quote_in", "in_quote_final", "count", "tape_index_base", "total_entries", "error",
                              "reemit"):
                        assert getattr(st, f) == getattr(host_st, f), (f, r)
                    assert recs[r].in_quote_in_used in (0, 1) and (guess or recs[r].in_quote_in_used == 0)
                    assert st.reemit == int(recs[r].in_quote_in_used != state)
                    e, q = oracle.scalar_index(d[lo:hi], base_off=lo, in_quote_in=state)
                    fin = sharded.result_from_words(d_fin[r].cpu().tolist())
                    assert (st.in_quote_in, st.count, st.tape_index_base) == (state, e.size, base)
                    assert (fin.count, fin.in_quote_out, fin.error) == (e.size, q, 0)
                    if not st.reemit:   # the re-emit launch must have been a no-op: tape untouched, bit for bit
                        assert torch.equal(tapes[r], spec[r])
                    # (entries of the speculative pass beyond a shorter re-emitted tape stay where they were)
                    assert bool((tapes[r][max(fin.count, recs[r].count):] == -1).all())
                    got.append(tapes[r][: fin.count].cpu().numpy().view(np.uint64))
                    state, base = q, base + e.size
                assert np.array_equal(np.concatenate(got), want)
                assert st.total_entries == want.size + 1 and st.in_quote_final == q_final
        finally:
            for c in ctxs:
                c.close()


def test_sharded_device_chain_is_graph_capturable(pkg, torch_cuda, oracle):
    # the part of a sharded step that runs on the device — speculative launch, stitch kernel, re-emit launch — holds no
    # allocation, no synchronisation and no host decision, so it can be captured into ONE hipGraph and replayed on new
    # bytes (the all-gather between launch and stitch is the caller's: here the record is simply shard 0 of 1, and the
    # file-level entering state is what makes the re-emit fire or not)
    torch = torch_cuda
    from csv_simd_amd import sharded
    rng = np.random.default_rng(808)
    n = 2 * pkg.tile_bytes() + 777
    c = pkg.Context(0)
    try:
        c.reserve(n)
        dbuf = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        dtape = torch.full((n + 8,), -1, dtype=torch.int64, device="cuda:0")
        d_rec = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        d_st = torch.zeros(sharded.STITCH_WORDS, dtype=torch.int64, device="cuda:0")
        for file_inq in (0, 1):
            def chain(stream):
                c.stage1_index_device_async(dbuf.data_ptr(), n, 11, 0, dtape.data_ptr(), dtape.numel(), d_rec.data_ptr(), stream)
                pkg.stitch_shards_device_async(d_rec.data_ptr(), 1, 0, file_inq, d_st.data_ptr(), stream)
                c.stage1_reemit_device_async(dbuf.data_ptr(), n, 11, d_st.data_ptr(), dtape.data_ptr(), dtape.numel(),
                                             d_rec.data_ptr(), stream)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(side.cuda_stream)          # warm-up outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain(torch.cuda.current_stream().cuda_stream)
            for rep in range(4):
                d = random_csvish(rng, n, 0.04)
                dbuf.copy_(torch.from_numpy(d))
                dtape.fill_(-1)
                g.replay()
                torch.cuda.synchronize()
                want, q = oracle.scalar_index(d, base_off=11, in_quote_in=file_inq)
                rec = sharded.result_from_words(d_rec.cpu().tolist())
                st = sharded.stitch_from_words(d_st.cpu().tolist())
                assert (st.in_quote_in, st.count, st.in_quote_final, st.error) == (file_inq, want.size, q, 0), (file_inq, rep)
                assert (rec.count, rec.in_quote_out, rec.error) == (want.size, q, 0)
                assert np.array_equal(dtape[: rec.count].cpu().numpy().view(np.uint64), want), (file_inq, rep)
            del g
    finally:
        c.close()
