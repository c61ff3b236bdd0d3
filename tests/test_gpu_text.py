"""UTF-8 validation and span trimming on the device (SURVEY.md §8f rank 4 extensions) against the
oracle's sequential definitions (which tests/test_oracle.py pins on CPython's decoder)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAD_AND_GOOD = [
    b"\x80", b"\xbf", b"\xc0\x80", b"\xc1\xbf", b"\xc2", b"\xc2\x41", b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xe0\xa0\x80",
    b"\xed\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf", b"\xee\x80\x80", b"\xef\xbf\xbf", b"\xe2\x82", b"\xe2\x28\xa1",
    b"\xe2\x82\x28", b"\xf0\x80\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xf0\x90\x80\x80", b"\xf4\x8f\xbf\xbf",
    b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"\xff", b"\xf0\x90\x80", b"\xf0\x90", b"\xf0", b"\xf1\x80\x80\x80\x80",
    b"\xc2\x80\x80", "é".encode(), "世".encode(), "\U0001F600".encode(),
]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def gpu_utf8(ctx, torch, host: np.ndarray, misalign=0, poison=0xFF):
    n = host.size
    dbuf = torch.full((n + 256,), poison, dtype=torch.uint8, device="cuda:0")
    if n:
        dbuf[misalign: misalign + n] = torch.from_numpy(host.copy())
    return ctx.utf8_validate_device(dbuf.data_ptr() + misalign, n)


@pytest.mark.parametrize("filler", ["a", "é", "世", "ก", "한", "\U0001F600"])   # ASCII; C3; E4; E0; ED; F0 leads
def test_utf8_sequences_at_every_boundary(ctx, torch_cuda, oracle, filler):
    # each sequence, valid or not, ending at / straddling 16-byte chunk, 1-KiB wave-load and 4-KiB
    # wave-iteration boundaries, and at the very start / end of the buffer; two poisons around it.
    # The text around it is ASCII, 2-, 3- or 4-byte characters: a 1-KiB chunk takes the kernel's basic-rule path, the
    # one with the E0 / ED second-byte test, or the full rules, by the filler's lead and by what the inserted sequence
    # brings (the insert also cuts characters of the filler in two: the oracle says where the first error then is)
    n = 3 * 4096 + 100
    clen = len(filler.encode())
    fill = np.frombuffer((filler * (n // clen + 1)).encode(), dtype=np.uint8)[:n]
    for mis in (0, 5, 15, 77, 127):
        for seq in BAD_AND_GOOD:
            s = np.frombuffer(seq, dtype=np.uint8)
            for edge in (0, 16, 1024, 4096, 8192, n):
                for shift in (-4, -3, -2, -1, 0, 1):
                    at = edge + shift - mis
                    if at < 0 or at + s.size > n:
                        continue
                    # as it falls, and moved back to a character boundary of the filler (the inserted sequence is then
                    # the first thing that can be wrong)
                    for pos in {at, at - at % clen}:
                        d = fill.copy()
                        d[pos: pos + s.size] = s
                        want = oracle.utf8_first_invalid(d)
                        for poison in (0xFF, 0x80):
                            assert gpu_utf8(ctx, torch_cuda, d, mis, poison) == want, (mis, seq, edge, shift, pos, poison)


def test_utf8_small_and_empty(ctx, torch_cuda, oracle):
    rng = np.random.default_rng(21)
    pool = np.frombuffer(b"a,\n\"\x7f\x80\x8f\x90\x9f\xa0\xbf\xc0\xc1\xc2\xdf\xe0\xe1\xec\xed\xee\xef\xf0\xf1\xf3\xf4\xf5\xff",
                         dtype=np.uint8)
    assert gpu_utf8(ctx, torch_cuda, np.zeros(0, dtype=np.uint8)) is None
    for _ in range(400):
        d = pool[rng.integers(0, pool.size, size=int(rng.integers(1, 40)))]
        assert gpu_utf8(ctx, torch_cuda, d, int(rng.integers(0, 128))) == oracle.utf8_first_invalid(d), d.tobytes()


def test_utf8_multilingual_text_and_corruptions(ctx, torch_cuda, oracle):
    rng = np.random.default_rng(22)
    row = "id,名前,città,emoji\n42,東京都,Perché no,\U0001F680\U0001F600\n7,Ünïcödé,naïve café,ok\n"
    text = np.frombuffer((row * 6000).encode(), dtype=np.uint8)           # ~0.6 MiB, mixed
    cjk = np.frombuffer(("漢字仮名交じり文" * 40000).encode(), dtype=np.uint8)  # ~0.9 MiB, no ASCII at all
    ascii_only = np.frombuffer((b"plain,ascii,row\n" * 70000), dtype=np.uint8)
    for base in (text, cjk, ascii_only):
        assert oracle.utf8_first_invalid(base) is None
        for mis in (0, 3):
            assert gpu_utf8(ctx, torch_cuda, base, mis) is None
        assert gpu_utf8(ctx, torch_cuda, base[:-1], 0) == oracle.utf8_first_invalid(base[:-1])  # maybe truncated
        for _ in range(25):
            d = base.copy()
            for p in rng.integers(0, d.size, size=int(rng.integers(1, 4))):
                d[p] = rng.integers(0, 256)
            assert gpu_utf8(ctx, torch_cuda, d, int(rng.integers(0, 128))) == oracle.utf8_first_invalid(d)


def test_utf8_large_buffer(ctx, torch_cuda, oracle):
    # 256 MiB of valid text, then one bad byte far in: the grid-stride loop and the atomic minimum
    torch = torch_cuda
    row = np.frombuffer("k,värde,値\n".encode(), dtype=np.uint8)
    reps = (256 << 20) // row.size
    d = torch.from_numpy(row.copy()).cuda().repeat(reps)
    n = d.numel()
    assert ctx.utf8_validate_device(d.data_ptr(), n) is None
    first = row.size * 1000000            # a row start: the ASCII 'k' becomes 0xFF
    for p in (n - 7, n // 2 + 3, first):
        d[p] = 0xFF
    assert ctx.utf8_validate_device(d.data_ptr(), n) == first
    d[first] = ord("k")
    d[first + 4] = ord("a")               # second byte of the 2-byte 'ä': its lead is the error
    assert ctx.utf8_validate_device(d.data_ptr(), n) == first + 3


def test_trim_spans(ctx, torch_cuda, pkg, golden, oracle):
    torch = torch_cuda
    # the reference's own fixture has space-padded fields: `Eliot    ,     2, Yes`
    data, _ = golden["sample.csv"]
    idx = ctx.read(data)
    t = pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), idx)
    dbytes = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    dindex = torch.from_numpy(idx.view(np.int64)).cuda()
    nrec = t.record_cnt - 1
    for f in range(t.field_cnt):
        b = torch.zeros(nrec, dtype=torch.int64, device="cuda:0")
        e = torch.zeros(nrec, dtype=torch.int64, device="cuda:0")
        assert pkg.tape_field_spans_device(dindex.data_ptr(), idx.size, t.field_cnt, t.new_line, f, 0, nrec,
                                           b.data_ptr(), e.data_ptr()) == nrec
        b0, e0 = b.cpu().numpy().view(np.uint64), e.cpu().numpy().view(np.uint64)
        for flags in (pkg.TRIM_SPACE, pkg.TRIM_QUOTES, pkg.TRIM_SPACE | pkg.TRIM_QUOTES):
            bb, ee = b.clone(), e.clone()
            pkg.trim_spans_device(dbytes.data_ptr(), bb.data_ptr(), ee.data_ptr(), nrec, flags)
            wb, we = oracle.trim_spans(data, b0, e0, flags)
            assert np.array_equal(bb.cpu().numpy().view(np.uint64), wb)
            assert np.array_equal(ee.cpu().numpy().view(np.uint64), we)
            if flags == 3:
                got = [data[i:j] for i, j in zip(wb.tolist(), we.tolist())]
                assert all(g == t.seek_field(r, f).strip(b" ") or g == t.seek_field(r, f).strip(b" ")[1:-1]
                           for r, g in enumerate(got))
                if f == 0:
                    assert got[0] == b"Edm nd" and b"Eliot" in got
    # random spans over space-heavy bytes
    rng = np.random.default_rng(23)
    pool = np.frombuffer(b'  "ab', dtype=np.uint8)
    raw = pool[rng.integers(0, pool.size, size=20000)]
    cuts = np.sort(rng.integers(0, raw.size + 1, size=3000))
    b0, e0 = cuts[:-1].astype(np.uint64), cuts[1:].astype(np.uint64)
    d = torch.from_numpy(raw.copy()).cuda()
    for flags in (0, 1, 2, 3):
        bb = torch.from_numpy(b0.view(np.int64).copy()).cuda()
        ee = torch.from_numpy(e0.view(np.int64).copy()).cuda()
        pkg.trim_spans_device(d.data_ptr(), bb.data_ptr(), ee.data_ptr(), b0.size, flags)
        wb, we = oracle.trim_spans(raw, b0, e0, flags)
        assert np.array_equal(bb.cpu().numpy().view(np.uint64), wb) and np.array_equal(ee.cpu().numpy().view(np.uint64), we)
    with pytest.raises(pkg.StructureError):
        pkg.trim_spans_device(d.data_ptr(), bb.data_ptr(), ee.data_ptr(), 1, 8)
