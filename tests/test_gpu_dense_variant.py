"""The second instantiation of the stage-1 kernel — the one a context runs on delimiter-dense data (entries per byte above
~0.1: csvsimd_ctx_hint_density, or what the synchronous entry points learn by themselves) — against the oracle and against
the default instantiation, bit for bit, on EVERY BASELINE configuration and on the edge cases the default one is tested on:
its geometry (2 rounds per wave: 64-KiB tiles) and its emit path (an 8-KiB window per wave, no per-entry capacity tests)
are different code, its results must not be.
Reference: crush_set_bits is one routine for every density (src/stage1.rs:162-296)."""
import numpy as np
import pytest

from conftest import random_csvish

pytestmark = pytest.mark.gpu

DENSE = "void csvsimd_dense::stage1_kernel<true, 0, 0, false, true>(csvsimd_dense::KernelArgs)"
DEFAULT = "void csvsimd::stage1_kernel<true, 0, 0, false, false>(csvsimd::KernelArgs)"


@pytest.fixture()
def dctx(pkg):
    """A context told that its data is dense: every emitting launch of the reference dialect runs the dense instantiation."""
    import torch
    assert torch.cuda.is_available()
    c = pkg.Context(0)
    c.hint_density(1, 2)
    assert c.kernel_name() == DENSE
    yield c
    c.close()


def index_both(pkg, torch, dctx, ctx, host, *, base_off=0, in_quote_in=0, misalign=0, cap=None):
    n = host.size
    dbuf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda:0")
    if n:
        dbuf[misalign: misalign + n] = torch.from_numpy(host)
    dbuf[:misalign] = 0x2C
    dbuf[misalign + n:] = 0x2C
    cap = (n + 1) if cap is None else cap
    out = []
    for c in (dctx, ctx):
        c.hint_density(1, 2) if c is dctx else c.hint_density(1, 1000)      # (a synchronous call re-learns the density: pin it)
        dtape = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
        r = c.stage1_index_device(dbuf.data_ptr() + misalign, n, base_off, in_quote_in, dtape.data_ptr(), cap, allow_overflow=True)
        torch.cuda.synchronize()
        assert (dtape[cap:] == -1).all(), "wrote past tape_cap"
        k = min(r.count, cap)
        assert r.written == k and (dtape[k:] == -1).all()
        out.append((dtape[:k].cpu().numpy().view(np.uint64), r))
    return out


def same_record(a, b):
    return all(getattr(a, f) == getattr(b, f) for f in ("count", "count_enter_outside", "count_enter_inside", "quote_parity",
                                                        "in_quote_out", "error", "written", "in_quote_in_used"))


def test_the_hint_selects_the_instantiation_and_sync_calls_learn_it(pkg, ctx):
    import torch
    c = pkg.Context(0)
    try:
        assert c.kernel_name() == DEFAULT                                  # nothing known: the default
        c.hint_density(1, 5)
        assert c.kernel_name() == DENSE
        c.hint_density(1, 32)
        assert c.kernel_name() == DEFAULT
        c.hint_density(0, 0)
        # a synchronous call reads its record: the context knows the data from then on
        for fill, want in ((0x2C, DENSE), (0x61, DEFAULT)):
            d = torch.full((1 << 20,), fill, dtype=torch.uint8, device="cuda:0")
            t = torch.empty((1 << 20) + 8, dtype=torch.int64, device="cuda:0")
            r = c.stage1_index_device(d.data_ptr(), d.numel(), 0, 0, t.data_ptr(), t.numel())
            assert r.count == (d.numel() if fill == 0x2C else 0) and c.kernel_name() == want
        # dialects and count-only launches are not affected
        assert "stage1_kernel<true, 0, 1, false, false>" in c.kernel_name(pkg.Dialect(";", '"'))
    finally:
        c.close()


@pytest.mark.parametrize("name", ["reader_test01.csv", "sample.csv", "sample_rx.csv"])
def test_golden_fixtures(pkg, dctx, ctx, golden, oracle, name):
    import torch
    data, exp = golden[name]
    host = np.frombuffer(data, dtype=np.uint8)
    (got, r), (ref, r0) = index_both(pkg, torch, dctx, ctx, host)
    want = oracle.sse_read(data)
    assert np.array_equal(got, want[1:]) and np.array_equal(got, ref) and same_record(r, r0)
    assert got.tolist() == exp["index"][1:]


def test_every_size_residue_alignment_and_entering_state(pkg, dctx, ctx, oracle):
    import torch
    rng = np.random.default_rng(404)
    T = pkg.tile_bytes()
    D = 64 << 10                                                      # the dense instantiation's tile
    sizes = (list(range(0, 140)) + [4095, 4096, 4097, 8191, 8192, 8193, 32767, 32768, 32769, D - 1, D, D + 1, 2 * D - 17, 9 * D + 5,
                                    33 * D + 4097, T - 1, T, T + 1, 2 * T + 63, 3 * T + 4097])
    for i, n in enumerate(sizes):
        p_quote = (None, 0.0, 0.02, 0.3)[i % 4]
        d = random_csvish(rng, n, p_quote)
        mis, inq, base = int(rng.integers(0, 128)), int(rng.integers(0, 2)), int(rng.integers(0, 1 << 40))
        (got, r), (ref, r0) = index_both(pkg, torch, dctx, ctx, d, base_off=base, in_quote_in=inq, misalign=mis)
        want, q = oracle.scalar_index(d, base_off=base, in_quote_in=inq)
        assert np.array_equal(got, want) and r.in_quote_out == q, (n, mis, inq)
        assert np.array_equal(got, ref) and same_record(r, r0), (n, mis, inq)


def test_adversarial_densities_and_capacity_protocol(pkg, dctx, ctx, oracle):
    import torch
    rng = np.random.default_rng(405)
    T = pkg.tile_bytes()
    n = 2 * T + 12345
    cases = {
        "all structural": np.full(n, 0x2C, dtype=np.uint8),
        "alternating": np.tile(np.frombuffer(b",a", dtype=np.uint8), n // 2 + 1)[:n].copy(),
        "every 5th (the dense corpus)": np.tile(np.frombuffer(b"abcd,", dtype=np.uint8), n // 5 + 1)[:n].copy(),
        "dense then empty": np.concatenate([np.full(T, 0x0A, dtype=np.uint8), np.full(n - T, 0x61, dtype=np.uint8)]),
        "one long quoted stretch of commas": np.concatenate([np.frombuffer(b'a,"', dtype=np.uint8), np.full(n - 6, 0x2C, dtype=np.uint8),
                                                             np.frombuffer(b'",b', dtype=np.uint8)]),
        "random dense": rng.choice(np.frombuffer(b',,,\n"a', dtype=np.uint8), size=n),
    }
    for label, d in cases.items():
        want, q = oracle.scalar_index(d)
        for cap in (None, 0, 1, 17, 2047, 2048, 2049, 4095, 4096, 4097, want.size - 1, want.size, want.size + 1):
            if cap is not None and cap < 0:
                continue
            (got, r), (ref, r0) = index_both(pkg, torch, dctx, ctx, d, cap=cap)
            k = want.size if cap is None else min(cap, want.size)
            assert r.count == want.size and r.in_quote_out == q, (label, cap)
            assert np.array_equal(got, want[:k]) and np.array_equal(got, ref) and same_record(r, r0), (label, cap)


@pytest.mark.parametrize("name", ["16x32_noquote", "16x32_q10", "1024x4_dense", "64x31_noquote", "64x31_q10"])
def test_baseline_configurations_at_1_gib(pkg, dctx, ctx, oracle, name):
    """configs 2, 3, 5 and the corpus of config 4 (both variants) at 1 GiB: the dense instantiation's tape == the default
    instantiation's == the closed form (quote-free) / the oracle's checksum + 64-MiB window (quoted)."""
    import torch
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 1 << 30)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    S = n // (width + 1)
    cap = S + 16
    tapes = []
    for c in (dctx, ctx):
        c.hint_density(1, 2) if c is dctx else c.hint_density(1, 1000)
        want_kernel = DENSE if c is dctx else DEFAULT
        assert c.kernel_name() == want_kernel
        t = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
        r = c.stage1_index_device(dbuf.data_ptr(), n, 0, 0, t.data_ptr(), cap)
        assert (r.count, r.in_quote_out, r.error) == (S, 0, 0)
        tapes.append(t)
    assert torch.equal(tapes[0], tapes[1])
    t = tapes[0][:S]
    if not q:
        assert torch.equal(t, torch.arange(1, S + 1, dtype=torch.int64, device="cuda:0") * (width + 1) - 1)
    win = 64 << 20
    host = dbuf[:win].cpu().numpy()
    o = oracle.sse_read(host)
    assert np.array_equal(o[1:], t[: o.size - 1].cpu().numpy().view(np.uint64))
    out = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    pkg.tape_checksum_device(t.data_ptr(), o.size - 1, 1, out.data_ptr())
    assert tuple(int(x) & (2**64 - 1) for x in out.cpu().tolist()) == oracle.tape_checksum(o[1:], 1)


def test_config4_shard_cut_mid_row_with_guess_and_reemit_flow(pkg, dctx, oracle):
    """config 4's shape through the dense instantiation: an 8-GiB shard of the quoted corpus cut at +777, entered by the
    kernel's own guess, then re-emitted under the other state through the stitch record (the sharded step's launches)."""
    import torch
    from csv_simd_amd import sharded
    cols, width, seed, q = pkg.WORKLOADS["64x31_q10"]
    shard = 1 << 33
    lo = 3 * shard + 777
    dbuf = torch.empty(shard, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), lo, shard, cols, width, seed, q)
    cap = int(shard // 32 * 1.25) + 1024
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    dctx.reserve(shard)
    row = cols * (width + 1)
    r0 = (lo // row) * row
    truth = int(np.count_nonzero(oracle.synth(r0, lo - r0, cols, width, seed, q) == 0x22)) & 1
    dctx.hint_density(1, 2)
    r = dctx.stage1_index_device(dbuf.data_ptr(), shard, lo, pkg.ENTER_GUESS, dtape.data_ptr(), cap)
    assert (r.in_quote_in_used, r.error) == (truth, 0)
    head = 32 << 20
    want, _ = oracle.scalar_index(dbuf[:head].cpu().numpy(), base_off=lo, in_quote_in=truth)
    assert np.array_equal(dtape[: want.size - 8].cpu().numpy().view(np.uint64), want[:-8])
    count_true = r.count
    # the stitch says "you were entered in the OTHER state": the re-emit launch indexes the shard again under it
    dctx.hint_density(1, 2)
    d_st = torch.zeros(sharded.STITCH_WORDS, dtype=torch.int64, device="cuda:0")
    d_st[0] = truth ^ 1
    d_st[4] = 1 << 32                                             # reemit
    d_rec = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    dctx.stage1_reemit_device_async(dbuf.data_ptr(), shard, lo, d_st.data_ptr(), dtape.data_ptr(), cap, d_rec.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = sharded.result_from_words(d_rec.cpu().tolist())
    want2, _ = oracle.scalar_index(dbuf[:head].cpu().numpy(), base_off=lo, in_quote_in=truth ^ 1)
    assert rec.in_quote_in_used == (truth ^ 1) and rec.error == 0 and rec.count != count_true
    k = min(want2.size - 8, rec.count)
    assert np.array_equal(dtape[:k].cpu().numpy().view(np.uint64), want2[:k])


# ---- round 5: the dense geometry wherever it applies (VERDICT r4 missing #4, #5) ------------------------------------------------
DENSE_D1 = "void csvsimd_dense::stage1_kernel<true, 0, 1, false, true>(csvsimd_dense::KernelArgs)"


def test_a_fresh_context_looks_at_the_data_before_its_first_synchronous_launch(pkg):
    """No hint, no history: the first synchronous call on >= 8 MiB samples sixteen 64-KiB windows and chooses.  The
    asynchronous entry point never does (it may not synchronise) and stays on the default geometry until told."""
    import torch
    for name, want in (("1024x4_dense", DENSE), ("64x31_noquote", DEFAULT), ("16x32_q10", DEFAULT)):
        cols, width, seed, q = pkg.WORKLOADS[name]
        n = pkg.workload_len(name, 64 << 20)
        dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
        cap = n // (width + 1) + 64
        t = torch.empty(int(cap * 1.3), dtype=torch.int64, device="cuda:0")
        dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        c = pkg.Context(0)
        try:
            assert c.kernel_name() == DEFAULT
            c.stage1_index_device_async(dbuf.data_ptr(), n, 0, 0, t.data_ptr(), t.numel(), dres.data_ptr(), 0)
            torch.cuda.synchronize()
            assert c.kernel_name() == DEFAULT                      # an asynchronous launch teaches the host nothing
            r = c.stage1_index_device(dbuf.data_ptr(), n, 0, 0, t.data_ptr(), t.numel())
            assert r.error == 0 and c.kernel_name() == want, name
        finally:
            c.close()


@pytest.mark.parametrize("name", ["16x32_noquote", "16x32_q10", "1024x4_dense", "64x31_noquote", "64x31_q10"])
def test_dense_batch_on_every_configuration(pkg, dctx, ctx, oracle, name):
    """DENSE x BATCH: 24 slices (whole rows, 1 ... 9 MiB, every other one entered inside a string, odd base offsets) of
    each BASELINE corpus in ONE batched launch of the dense geometry == each slice alone through the default
    instantiation == the oracle (a 2-MiB head of every slice, the count of all of it)."""
    import torch
    cols, width, seed, q = pkg.WORKLOADS[name]
    row = cols * (width + 1)
    rng = np.random.default_rng(17)
    n = pkg.workload_len(name, 160 << 20)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    items, singles, off = [], [], 0
    for i in range(24):
        ln = int(rng.integers(1 << 20, 9 << 20)) // row * row + (i % 3) * 7     # (not always whole rows: ragged ends too)
        ln = min(ln, n - off)
        cap = ln // (width + 1) * 2 + 64
        tb = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
        ts = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
        items.append((dbuf.data_ptr() + off, ln, 1000 * i + 1, tb.data_ptr(), cap, i & 1))
        singles.append((off, ln, 1000 * i + 1, ts, cap, i & 1, tb))
        off += ln
    dres = torch.zeros((24, 8), dtype=torch.int64, device="cuda:0")
    dctx.hint_density(1, 2)
    dctx.stage1_index_batch_device_async(items, dres.data_ptr())
    torch.cuda.synchronize()
    for i, (o, ln, base, ts, cap, st, tb) in enumerate(singles):
        ctx.hint_density(1, 1000)
        r0 = ctx.stage1_index_device(dbuf.data_ptr() + o, ln, base, st, ts.data_ptr(), cap)
        assert r0.error == 0
        r = pkg.ShardResult.from_buffer_copy(dres[i].cpu().numpy().tobytes())
        assert same_record(r, r0) and r.error == 0, (name, i)
        assert torch.equal(tb, ts), (name, i)
        head = min(ln, 2 << 20)
        want, _ = oracle.scalar_index(dbuf[o: o + head].cpu().numpy(), base_off=base, in_quote_in=st)
        k = max(want.size - 4, 0)
        assert np.array_equal(tb[:k].cpu().numpy().view(np.uint64), want[:k]), (name, i)


@pytest.mark.parametrize("name", ["16x32_noquote", "16x32_q10", "1024x4_dense", "64x31_noquote", "64x31_q10"])
def test_dense_geometry_with_another_delimiter_and_quote_byte(pkg, oracle, name):
    """DENSE x dialect 1 (delimiter ';', quote "'"): 256 MiB of each BASELINE corpus with its ',' and '"' bytes replaced,
    through the dense geometry == through the default geometry of the same dialect == the reference-dialect tape of the
    untouched corpus (the replacement is a bijection on the special bytes) == the oracle's dialect index on a head."""
    import torch
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 256 << 20)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 64
    ref_tape = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
    c0 = pkg.Context(0)
    r_ref = c0.stage1_index_device(dbuf.data_ptr(), n, 5, 0, ref_tape.data_ptr(), cap)
    alt = dbuf.clone()
    alt[dbuf == 0x2C] = ord(";")
    alt[dbuf == 0x22] = ord("'")
    dia = pkg.Dialect(";", "'")
    from csv_simd_amd import sharded
    tapes = []
    try:
        for dense in (True, False):
            c0.hint_density(1, 2) if dense else c0.hint_density(1, 1000)
            assert c0.kernel_name(dia) == (DENSE_D1 if dense else "void csvsimd::stage1_kernel<true, 0, 1, false, false>(csvsimd::KernelArgs)")
            t = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
            dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            c0.stage1_index_device_dialect_async(dia, alt.data_ptr(), n, 5, 0, t.data_ptr(), cap, dres.data_ptr(), 0)
            torch.cuda.synchronize()
            r = sharded.result_from_words(dres.cpu().tolist())
            assert (r.count, r.in_quote_out, r.error) == (r_ref.count, r_ref.in_quote_out, 0), (name, dense)
            tapes.append(t)
        assert torch.equal(tapes[0], tapes[1]) and torch.equal(tapes[0], ref_tape)
        head = alt[: 4 << 20].cpu().numpy()
        want, _, _ = oracle.dialect_index(head, ord(";"), ord("'"), 0, base_off=5)
        assert np.array_equal(tapes[0][: want.size - 4].cpu().numpy().view(np.uint64), want[:-4])
        # quoting switched off in the dialect (quote = 0): every delimiter, CR and LF counts, also in the dense geometry
        nq = pkg.Dialect(";", None)
        c0.hint_density(1, 2)
        capq = int(cap * 1.4)
        t = torch.full((capq,), -1, dtype=torch.int64, device="cuda:0")
        dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        c0.stage1_index_device_dialect_async(nq, alt.data_ptr(), n, 0, 0, t.data_ptr(), capq, dres.data_ptr(), 0)
        torch.cuda.synchronize()
        r = sharded.result_from_words(dres.cpu().tolist())
        want, _, _ = oracle.dialect_index(head, ord(";"), 0, 0)
        assert r.error == 0 and np.array_equal(t[: want.size - 4].cpu().numpy().view(np.uint64), want[:-4])
        assert r.count == int(((alt == ord(";")) | (alt == 0x0A) | (alt == 0x0D)).sum().item())
    finally:
        c0.close()
