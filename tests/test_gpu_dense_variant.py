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
        c.hint_density(1, 2) if c is dctx else c.hint_density(0, 0)      # (a synchronous call re-learns the density: pin it)
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
        c.hint_density(1, 2) if c is dctx else c.hint_density(0, 0)
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
