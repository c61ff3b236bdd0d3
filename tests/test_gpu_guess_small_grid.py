"""CSVSIMD_ENTER_GUESS needs no particular number of resident workgroups (round 5).

The shard's first 2 MiB vote on its entering state (8 tiles of the default geometry, 32 of the dense one).  Until round 4
a workgroup counted at most two tiles before it waited for that choice, so the vote could only complete with 4 (dense: 16)
workgroups of the launch resident at once; a launch squeezed in beside other contexts' persistent grids ended in the spin
bound (CSVSIMD_ERR_INTERNAL).  Now a workgroup that cannot resolve its held tile gives it up (to be counted again later)
and draws the next voter: a grid of ONE workgroup completes the vote alone.  csvsimd_ctx_limit_workgroups caps the grid.

Results are compared with the oracle's index under the TRUE entering state (reference reader::read semantics,
src/reader.rs:150-306, src/avx/stage1.rs:337-407), which on a quoted CSV is what the vote chooses."""
import numpy as np
import pytest

from conftest import random_csvish

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def quoted_csv(rng, nbytes, dense=False):
    """10 % of the fields quoted, holding a comma and a line break (dense: 3-byte fields, so > 0.125 entries per byte).
    A block of ~300 KiB is generated field by field and repeated (whole quoted fields: the parity carries over)."""
    out, size, i = [], 0, 0
    while size < min(nbytes, 300_000):
        if rng.random() < 0.1:
            f = b'"qq,q\nq"' if dense else b'"' + b"q" * 9 + b"," + b"q" * 9 + b"\n" + b"q" * 8 + b'"'
        else:
            f = b"abc" if dense else b"f" * 30
        sep = b"\n" if i % 16 == 15 else b","
        out.append(f + sep)
        size += len(f) + 1
        i += 1
    block = np.frombuffer(b"".join(out), dtype=np.uint8)
    return np.tile(block, nbytes // block.size + 1)[:nbytes].copy()


def run_guess(pkg, torch, c, shard, base_off, dense):
    n = shard.size
    dbuf = torch.from_numpy(shard).to("cuda:0")
    dtape = torch.full((n + 9,), -1, dtype=torch.int64, device="cuda:0")
    c.hint_density(1, 2) if dense else c.hint_density(1, 1000)
    r = c.stage1_index_device(dbuf.data_ptr(), n, base_off, pkg.ENTER_GUESS, dtape.data_ptr(), n + 1)
    torch.cuda.synchronize()
    return dtape[: r.count].cpu().numpy().view(np.uint64), r


@pytest.mark.parametrize("dense", [False, True], ids=["default-geometry", "dense-geometry"])
@pytest.mark.parametrize("limit", [1, 2, 3, 5, 9, 0])
def test_guess_completes_on_any_grid(pkg, oracle, torch_cuda, dense, limit):
    T = pkg.tile_bytes()
    rng = np.random.default_rng(7000 + limit + (100 if dense else 0))
    text = quoted_csv(rng, 4 * (1 << 20) + 2 * T + 12345, dense)
    quotes = np.flatnonzero(text == 0x22)
    c = pkg.Context(0)
    try:
        c.limit_workgroups(limit)
        want_name = "csvsimd_dense" if dense else "csvsimd::"
        # a 4-MiB shard (16 tiles / 64 dense tiles: twice the vote), entered outside and inside a quoted field; shards of
        # fewer tiles than voters (1, 3); a shard that ends right behind the vote
        for cut, n in ((0, 4 << 20), (int(quotes[11]) + 1, 4 << 20), (int(quotes[40]) + 2, (4 << 20) + 777), (T + 5, T - 9),
                       (int(quotes[21]) + 1, 3 * T - 100), (64, 2 << 20), (int(quotes[31]) + 3, (2 << 20) + 1)):
            shard = text[cut: cut + n]
            truth = int(np.count_nonzero(text[:cut] == 0x22) & 1)
            got, r = run_guess(pkg, torch_cuda, c, shard, cut, dense)
            assert want_name in c.kernel_name()
            assert r.error == 0 and r.in_quote_in_used == truth, (limit, cut, n)
            want, q = oracle.scalar_index(shard, base_off=cut, in_quote_in=truth)
            assert (r.count, r.in_quote_out) == (want.size, q) and np.array_equal(got, want), (limit, cut, n)
            p, c0, c1 = oracle.shard_descriptor(shard)
            assert (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1)
        # random bytes: whatever the vote chooses, the record says so and count / tape / leaving state are that state's
        d = random_csvish(rng, (3 << 20) + 4321, 0.01)
        got, r = run_guess(pkg, torch_cuda, c, d, 0, dense)
        want, q = oracle.scalar_index(d, in_quote_in=r.in_quote_in_used)
        assert r.error == 0 and (r.count, r.in_quote_out) == (want.size, q) and np.array_equal(got, want)
    finally:
        c.close()


def test_guess_on_one_workgroup_escape_dialect(pkg, oracle, torch_cuda):
    # the instantiations that park part of the held tile in LDS (escape dialects) give it up the same way
    torch = torch_cuda
    rng = np.random.default_rng(515)
    T = pkg.tile_bytes()
    text = quoted_csv(rng, 3 * (1 << 20) + 999).copy()
    text[rng.integers(0, text.size, 2000)] = ord("\\")
    dia = pkg.Dialect(",", '"', "\\")
    quotes = np.flatnonzero(text == 0x22)
    c = pkg.Context(0)
    try:
        c.limit_workgroups(1)
        for cut in (0, int(quotes[15]) + 1, T + 3):
            shard = text[cut:]
            n = shard.size
            dbuf = torch.from_numpy(shard).to("cuda:0")
            dtape = torch.full((n + 9,), -1, dtype=torch.int64, device="cuda:0")
            dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            c.stage1_index_device_dialect_async(dia, dbuf.data_ptr(), n, cut, pkg.ENTER_GUESS, dtape.data_ptr(), n + 1,
                                                dres.data_ptr(), 0)
            torch.cuda.synchronize()
            from csv_simd_amd import sharded
            r = sharded.result_from_words(dres.cpu().tolist())
            want, q, _ = oracle.dialect_index(shard, 0x2C, 0x22, 0x5C, base_off=cut, in_quote_in=r.in_quote_in_used)
            assert r.error == 0 and (r.count, r.in_quote_out) == (want.size, q), cut
            assert np.array_equal(dtape[: r.count].cpu().numpy().view(np.uint64), want), cut
    finally:
        c.close()


@pytest.mark.parametrize("dense", [False, True], ids=["default-geometry", "dense-geometry"])
def test_three_guess_launches_share_the_gpu(pkg, oracle, torch_cuda, dense):
    """Three contexts, three streams, three GUESS launches in flight at once, each over a shard large enough for a full
    persistent grid (so the three grids compete for the chip's wave slots), repeated: every launch ends without the spin
    bound and with the oracle's tape.  (the dense geometry needed 16 resident workgroups per launch.)"""
    torch = torch_cuda
    from csv_simd_amd import sharded
    rng = np.random.default_rng(99)
    n = 160 << 20
    text = quoted_csv(rng, 3 * n + 100, dense)
    cuts = [0, n + 7, 2 * n + 13]
    truths = [int(np.count_nonzero(text[:c] == 0x22) & 1) for c in cuts]
    ctxs = [pkg.Context(0) for _ in cuts]
    streams = [torch.cuda.Stream() for _ in cuts]
    try:
        shards = [text[c: c + n] for c in cuts]
        wants = [oracle.scalar_index(s, base_off=c, in_quote_in=t) for s, c, t in zip(shards, cuts, truths)]
        dbufs = [torch.from_numpy(s).to("cuda:0") for s in shards]
        dtapes = [torch.empty(n // 2 + 64, dtype=torch.int64, device="cuda:0") for _ in cuts]
        dres = [torch.zeros(8, dtype=torch.int64, device="cuda:0") for _ in cuts]
        for c in ctxs:
            c.reserve(n)
            c.hint_density(1, 2) if dense else c.hint_density(1, 1000)
        torch.cuda.synchronize()
        for rep in range(6):
            for c, s, db, dt, dr, cut in zip(ctxs, streams, dbufs, dtapes, dres, cuts):
                c.stage1_index_device_async(db.data_ptr(), n, cut, pkg.ENTER_GUESS, dt.data_ptr(), dt.numel(), dr.data_ptr(),
                                            s.cuda_stream)
            torch.cuda.synchronize()
            for i in range(len(cuts)):
                r = sharded.result_from_words(dres[i].cpu().tolist())
                want, q = wants[i]
                assert r.error == 0 and r.in_quote_in_used == truths[i], (rep, i)
                assert (r.count, r.in_quote_out) == (want.size, q), (rep, i)
                if rep in (0, 5):
                    assert np.array_equal(dtapes[i][: r.count].cpu().numpy().view(np.uint64), want), (rep, i)
    finally:
        for c in ctxs:
            c.close()
