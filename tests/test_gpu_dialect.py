"""Dialect extension: other delimiter / quote bytes and an escape byte.

Not reference behaviour (the reference hard-wires ',' and '"', src/avx/stage1.rs:392-394), so the
checker is the scalar definition oracle_dialect_index; the default dialect through the same entry
point must still equal the reference semantics, and a re-delimited copy of the reference's own
fixtures must reproduce their golden indexes.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import random_csvish

pytestmark = pytest.mark.gpu

BS = 0x5C


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def gpu_dialect(ctx, pkg, torch, host, dialect, *, base_off=0, in_quote_in=0, misalign=0):
    n = host.size
    dbuf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda:0")
    if n:
        dbuf[misalign: misalign + n] = torch.from_numpy(host)
    # poison around the payload with bytes that are special in every dialect used here
    dbuf[:misalign] = dialect.escape or dialect.delimiter
    dbuf[misalign + n:] = dialect.escape or dialect.delimiter
    cap = n + 1
    dtape = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    ctx.stage1_index_device_dialect_async(dialect, dbuf.data_ptr() + misalign, n, base_off, in_quote_in,
                                          dtape.data_ptr(), cap, dres.data_ptr())
    torch.cuda.synchronize()
    r = pkg.ShardResult.from_buffer_copy(dres.cpu().numpy().tobytes())
    assert r.error == 0 and r.written == r.count
    assert (dtape[r.count:] == -1).all()
    return dtape[: r.count].cpu().numpy().view(np.uint64), r


def check(ctx, pkg, torch, oracle, d, dialect, **kw):
    got, r = gpu_dialect(ctx, pkg, torch, d, dialect, **kw)
    want, q, e = oracle.dialect_index(d, dialect.delimiter, dialect.quote, dialect.escape,
                                      base_off=kw.get("base_off", 0), in_quote_in=kw.get("in_quote_in", 0),
                                      escape_in=dialect.escape_in)
    assert r.count == want.size and np.array_equal(got, want)
    assert r.in_quote_out == q
    if dialect.escape:
        assert r.escape_out == e
    return got, r


def test_default_dialect_is_the_reference_path(ctx, pkg, torch_cuda, oracle):
    rng = np.random.default_rng(1)
    d = random_csvish(rng, 300000, 0.05)
    got, r = gpu_dialect(ctx, pkg, torch_cuda, d, pkg.Dialect())
    want, q = oracle.scalar_index(d)
    assert np.array_equal(got, want) and r.in_quote_out == q


@pytest.mark.parametrize("name", ["reader_test01.csv", "sample.csv", "sample_rx.csv"])
@pytest.mark.parametrize("delim,quote", [("\t", '"'), ("^", "'"), ("|", '"'), (0x01, 0x02)])
def test_redelimited_golden_fixtures(ctx, pkg, golden, name, delim, quote):
    # the reference's fixtures with ',' / '"' swapped for another pair keep their golden offsets
    data, exp = golden[name]
    dl = delim if isinstance(delim, int) else ord(delim)
    q = quote if isinstance(quote, int) else ord(quote)
    assert (dl == 0x2C or bytes([dl]) not in data) and (q == 0x22 or bytes([q]) not in data)
    conv = data.replace(b",", bytes([dl])).replace(b'"', bytes([q]))
    got = ctx.read_dialect(conv, pkg.Dialect(delim, quote))
    assert np.array_equal(got, np.array(exp["index"], dtype=np.uint64))


def test_other_delimiters_random(ctx, pkg, torch_cuda, oracle):
    rng = np.random.default_rng(2)
    T = pkg.tile_bytes()
    alphabet = np.frombuffer(b",;\t|'\"\n\ra \\\x00\xff", dtype=np.uint8)
    for dialect in (pkg.Dialect(";", "'"), pkg.Dialect("\t", '"'), pkg.Dialect("|", None),
                    pkg.Dialect(0xFF, 0x00 + 0x61)):
        for n in (0, 1, 63, 64, 65, 4097, T - 1, T + 1, 3 * T + 777):
            d = alphabet[rng.integers(0, alphabet.size, size=n)].astype(np.uint8)
            for mis, inq in ((0, 0), (5, 1), (15, 0), (64, 1), (100, 0), (127, 1)):
                check(ctx, pkg, torch_cuda, oracle, d, dialect, base_off=7 + 2**40, in_quote_in=inq, misalign=mis)


def escapey(rng, n, p_esc):
    alphabet = np.frombuffer(b',"\n\ra\\', dtype=np.uint8)
    w = np.array([1, 0.3, 1, 0.2, 3, 0], dtype=float)
    w = w / w.sum() * (1 - p_esc)
    w[5] = p_esc
    return alphabet[rng.choice(alphabet.size, size=n, p=w)].astype(np.uint8)


def test_escape_random_densities(ctx, pkg, torch_cuda, oracle):
    rng = np.random.default_rng(3)
    T = pkg.tile_bytes()
    for n in (0, 1, 2, 63, 64, 65, 128, 4096, 4097, 32768 + 1, T, T + 3, 2 * T + 12345):
        for p in (0.02, 0.3, 0.7, 0.97):
            d = escapey(rng, n, p)
            for mis in (0, 1, 9, 15, 63, 64, 65, 127):
                for esc_in in (0, 1):
                    dia = pkg.Dialect(",", '"', "\\", escape_in=esc_in)
                    check(ctx, pkg, torch_cuda, oracle, d, dia, in_quote_in=int(rng.integers(0, 2)), misalign=mis)


def test_escape_with_other_delimiters_and_without_quoting(ctx, pkg, torch_cuda, oracle):
    rng = np.random.default_rng(33)
    T = pkg.tile_bytes()
    alphabet = np.frombuffer(b",;\t'\"\n\ra ^\\", dtype=np.uint8)
    for dia in (pkg.Dialect(",", None, "\\"), pkg.Dialect(";", "'", "^"), pkg.Dialect("\t", '"', "\\", escape_in=1),
                pkg.Dialect("^", None, ";")):
        for n in (0, 1, 65, 4097, T + 1, 2 * T + 4321):
            d = alphabet[rng.integers(0, alphabet.size, size=n)].astype(np.uint8)
            for mis in (0, 13, 100):
                check(ctx, pkg, torch_cuda, oracle, d, dia, in_quote_in=int(rng.integers(0, 2)), misalign=mis)


def test_escape_hashed_and_compare_classification_agree_with_the_definition(ctx, pkg, torch_cuda, oracle):
    """An escape dialect runs one of two classifications: the hashed LUT (kernel <..., 3>: the special bytes have a
    collision-free 3-bit hash, most triples) or the direct compares (<..., 2>: the fallback).  The library says which
    (csvsimd_stage1_kernel_name); triples of BOTH kinds — and all 256 byte values in the data — against the scalar
    definition, with and without a quote byte, across tile boundaries and misaligned."""
    torch = torch_cuda
    rng = np.random.default_rng(2718)
    T = pkg.tile_bytes()
    cand = [(d, q, e) for d in (0x2C, 0x3B, 0x09, 0x7C, 0x20, 0x3A, 0x41, 0xE9) for q in (0x22, 0x27, 0x60, 0, 0xAB)
            for e in (0x5C, 0x5E, 0x7E, 0x25, 0x2F, 0xB1) if len({d, q, e, 0x0A, 0x0D}) == 5 or (q == 0 and len({d, e, 0x0A, 0x0D}) == 4)]
    kinds = {2: [], 3: []}
    for d, q, e in cand:
        name = pkg.stage1_kernel_name(True, pkg.Dialect(d, q or None, e))
        kinds[int(name.split(",")[2])].append((d, q, e))          # <emit, DBG, DIALECT, BATCH, DENSE>
    assert len(kinds[3]) > len(kinds[2]) > 0, {k: len(v) for k, v in kinds.items()}     # both paths exist in this sample
    for kind in (2, 3):
        for d, q, e in kinds[kind][:6]:
            dia = pkg.Dialect(d, q or None, e)
            special = np.array([d, e, 0x0A, 0x0D] + ([q] if q else []), dtype=np.uint8)
            n = 2 * T + 777
            data = rng.integers(0, 256, size=n, dtype=np.uint8)                          # every byte value occurs
            hit = rng.random(n) < 0.15
            data[hit] = special[rng.integers(0, special.size, size=int(hit.sum()))]      # ... and the special ones often
            for mis in (0, 5):
                check(ctx, pkg, torch, oracle, data, dia, misalign=mis, in_quote_in=mis & 1)
            # neighbours of every special byte (one bit away): never classified as special
            near = np.array(sorted({int(b) ^ (1 << k) for b in special for k in range(8)} - set(int(b) for b in special)),
                            dtype=np.uint8)
            check(ctx, pkg, torch, oracle, np.tile(near, 300), dia)


def test_escape_runs_across_every_boundary(ctx, pkg, torch_cuda, oracle):
    # runs of escape bytes that end at / straddle stripe (64 B), round (4 KiB), wave-span (32 KiB)
    # and tile boundaries, including runs longer than one and two whole stripes
    rng = np.random.default_rng(4)
    T = pkg.tile_bytes()
    n = 2 * T + 5000
    for mis in (0, 3, 67):
        for run_len in (1, 2, 3, 63, 64, 65, 127, 128, 129, 191, 192, 4096, 4097, 32768 + 1):
            d = escapey(rng, n, 0.0)
            d[d == BS] = ord("a")
            for edge in (64, 4096, 32768, T, T + 32768, 2 * T):
                for end_shift in (-1, 0, 1):
                    end = edge + end_shift - mis   # run occupies [end - run_len, end)
                    if end - run_len < 0 or end + 2 > n:
                        continue
                    d[end - run_len: end] = BS
                    d[end] = ord(",")           # the byte whose fate the run parity decides
                    d[end + 1] = ord('"')
            for esc_in in (0, 1):
                dia = pkg.Dialect(",", '"', "\\", escape_in=esc_in)
                check(ctx, pkg, torch_cuda, oracle, d, dia, misalign=mis)


def test_escape_whole_buffer_of_escapes(ctx, pkg, torch_cuda, oracle):
    for n in (1, 2, 63, 64, 65, 128, 4096, 40000):
        d = np.full(n, BS, dtype=np.uint8)
        for esc_in in (0, 1):
            for mis in (0, 7, 64, 71):
                dia = pkg.Dialect(",", '"', "\\", escape_in=esc_in)
                _, r = check(ctx, pkg, torch_cuda, oracle, d, dia, misalign=mis)
                assert r.escape_out == (n - esc_in) % 2


def test_escape_split_invariance(ctx, pkg, torch_cuda, oracle):
    # shards chained through (in_quote_out, escape_out) reproduce the unsplit tape
    rng = np.random.default_rng(5)
    T = pkg.tile_bytes()
    d = escapey(rng, 3 * T + 999, 0.35)
    whole, rw = check(ctx, pkg, torch_cuda, oracle, d, pkg.Dialect(",", '"', "\\"))
    for cut in (1, 64, 4095, T - 1, T, T + 17, 2 * T + 333):
        a, ra = gpu_dialect(ctx, pkg, torch_cuda, d[:cut], pkg.Dialect(",", '"', "\\"))
        b, rb = gpu_dialect(ctx, pkg, torch_cuda, d[cut:], pkg.Dialect(",", '"', "\\", escape_in=ra.escape_out),
                            base_off=cut, in_quote_in=ra.in_quote_out, misalign=cut % 128)
        assert np.array_equal(np.concatenate([a, b]), whole), cut
        assert (rb.in_quote_out, rb.escape_out) == (rw.in_quote_out, rw.escape_out)


def test_escape_host_path_chains_chunks(ctx, pkg, oracle, monkeypatch):
    # csvsimd_stage1_index_dialect streams the buffer in chunks (pinned here to 4 MiB): odd escape
    # runs straddle the first two boundaries, the escape state must carry from chunk to chunk
    monkeypatch.setenv("CSVSIMD_INGEST_CHUNK_MIB", "4")
    rng = np.random.default_rng(6)
    n = (12 << 20) + 70000
    d = escapey(rng, n, 0.05)
    for edge in (4 << 20, 8 << 20):
        d[edge - 4] = ord("a")
        d[edge - 3: edge] = BS
        d[edge] = ord(",")
    got = ctx.read_dialect(d, pkg.Dialect(",", '"', "\\"))
    want, _, _ = oracle.dialect_index(d, escape=BS)
    assert got[0] == 0 and np.array_equal(got[1:], want)
    assert not np.isin(np.array([4 << 20, 8 << 20], dtype=np.uint64), got).any()


def test_enter_guess_in_the_dialect_variants(ctx, pkg, torch_cuda, oracle):
    # CSVSIMD_ENTER_GUESS is part of the shared kernel body: the dialect instantiations choose their entering state
    # the same way, and the tape is the scalar definition's for the state the record says was used
    T = pkg.tile_bytes()
    for dialect in (pkg.Dialect(";", "'"), pkg.Dialect(",", '"', "\\")):
        dl, qt = bytes([dialect.delimiter]), bytes([dialect.quote])
        row = b"aaaa" + dl + qt + b"x" + dl + b"y\nz" + qt + dl + b"bbbb\n"    # one quoted field with a delimiter and an LF
        text = np.frombuffer(row * (2 * T // len(row) + 3), dtype=np.uint8).copy()
        for cut in (0, 7, 9, len(row) * 1000 + 8, T + len(row) * 3 + 2):       # cuts 7 / 9 / +8 lie inside the quoted field
            d = text[cut:]
            truth = int(np.count_nonzero(text[:cut] == dialect.quote) & 1)
            got, r = gpu_dialect(ctx, pkg, torch_cuda, d, dialect, base_off=cut, in_quote_in=pkg.ENTER_GUESS)
            assert r.in_quote_in_used == truth, (dialect.delimiter, cut)
            want, q, _ = oracle.dialect_index(d, dialect.delimiter, dialect.quote, dialect.escape, base_off=cut,
                                              in_quote_in=truth, escape_in=0)
            assert r.count == want.size and r.in_quote_out == q and np.array_equal(got, want)


def test_dialect_argument_checks(ctx, pkg, torch_cuda):
    torch = torch_cuda
    dbuf = torch.zeros(256, dtype=torch.uint8, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    for bad in (pkg.Dialect(0, '"'), pkg.Dialect("\n", '"'), pkg.Dialect(",", ","), pkg.Dialect(",", '"', '"'),
                pkg.Dialect(",", '"', ","), pkg.Dialect(",", "\r")):
        with pytest.raises(pkg.StructureError) as e:
            ctx.stage1_index_device_dialect_async(bad, dbuf.data_ptr(), 256, 0, 0, 0, 0, dres.data_ptr())
        assert e.value.code == pkg.ERR_INVALID_ARG
    d = pkg.Dialect("x", "y", "z", 1)
    assert pkg.lib().csvsimd_dialect_init(C.byref(d)) == 0
    assert (d.delimiter, d.quote, d.escape, d.escape_in) == (0x2C, 0x22, 0, 0)
