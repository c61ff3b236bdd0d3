"""csvsimd_stage1_index_batch_device_async: many independent buffers in ONE persistent launch.  Record i and tape i must
be exactly what the single-buffer entry point produces for buffer i alone — i.e. the oracle's index of that buffer
(reference reader::read per file, src/reader.rs:150-306; csv_simd::create per file, src/lib.rs:61-74)."""
import numpy as np
import pytest

from conftest import random_csvish

pytestmark = pytest.mark.gpu


def run_batch(ctx, pkg, torch, bufs, states, base_offs, misaligns, caps=None):
    n = len(bufs)
    dbufs, dtapes, items = [], [], []
    for i, b in enumerate(bufs):
        t = torch.full((b.size + 256,), 0x2C, dtype=torch.uint8, device="cuda:0")     # poison around the payload
        if b.size:
            t[misaligns[i]: misaligns[i] + b.size] = torch.from_numpy(b)
        cap = b.size + 1 if caps is None else caps[i]
        tape = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
        dbufs.append(t)
        dtapes.append(tape)
        items.append((t.data_ptr() + misaligns[i], b.size, base_offs[i], tape.data_ptr(), cap, states[i]))
    dres = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
    ctx.stage1_index_batch_device_async(items, dres.data_ptr())
    torch.cuda.synchronize()
    recs = [pkg.ShardResult.from_buffer_copy(dres[i].cpu().numpy().tobytes()) for i in range(n)]
    return recs, dtapes


def check_batch(ctx, pkg, torch, oracle, bufs, states=None, base_offs=None, misaligns=None):
    n = len(bufs)
    states = states or [0] * n
    base_offs = base_offs or [0] * n
    misaligns = misaligns or [0] * n
    recs, tapes = run_batch(ctx, pkg, torch, bufs, states, base_offs, misaligns)
    for i, b in enumerate(bufs):
        want, q = oracle.scalar_index(b, base_off=base_offs[i], in_quote_in=states[i])
        r = recs[i]
        assert (r.count, r.in_quote_out, r.error, r.written, r.in_quote_in_used) == (want.size, q, 0, want.size, states[i]), i
        p, c0, c1 = oracle.shard_descriptor(b)
        assert (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1), i
        got = tapes[i][: r.count].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), i
        assert bool((tapes[i][r.count:] == -1).all()), i          # nothing of a neighbour's tape, nothing past the count


def test_batch_of_mixed_buffers_equals_each_alone(ctx, pkg, oracle):
    import torch
    rng = np.random.default_rng(606)
    T = pkg.tile_bytes()
    sizes = [0, 1, 63, 64, 65, 4096, T - 1, T, T + 1, 3 * T + 777, 0, 5 * T, 17, 2 * T - 16]
    bufs = [random_csvish(rng, n, pq) for n, pq in zip(sizes, [0.05, 0.0, 0.2, 0.01] * 4)]
    states = [int(rng.integers(0, 2)) for _ in bufs]
    base_offs = [0, 7, 10**12, 3, 0, 1, 2, 5, 8, 13, 21, 34, 55, 89]
    misaligns = [0, 1, 15, 16, 0, 3, 64, 0, 127, 5, 0, 0, 9, 100]
    check_batch(ctx, pkg, torch, oracle, bufs, states, base_offs, misaligns)
    # one buffer, and the same batch again (the table and the scratch are reused; launch epochs advance)
    check_batch(ctx, pkg, torch, oracle, bufs[9:10], [1], [42], [7])
    check_batch(ctx, pkg, torch, oracle, bufs, states, base_offs, misaligns)
    # only empty buffers
    check_batch(ctx, pkg, torch, oracle, [np.zeros(0, dtype=np.uint8)] * 3, [0, 1, 0])


def test_batch_lookback_stops_at_buffer_boundaries(ctx, pkg, oracle):
    """Buffers that END inside a quoted string next to buffers that would look entirely different if that state leaked
    into them; more than 64 buffers (the tile -> buffer search works 64 table lines at a time); buffers of many tiles."""
    import torch
    rng = np.random.default_rng(99)
    T = pkg.tile_bytes()
    open_ended = np.frombuffer((b'a,b,"unterminated ' + b"x,y\n" * 100), dtype=np.uint8).copy()
    plain = np.frombuffer(b"1,2,3\n" * 3000, dtype=np.uint8).copy()
    bufs = []
    for i in range(150):
        bufs.append(open_ended if i % 3 == 0 else (plain if i % 3 == 1 else random_csvish(rng, int(rng.integers(1, 3 * T)), 0.02)))
    check_batch(ctx, pkg, torch, oracle, bufs, [i % 2 for i in range(150)])
    big = [random_csvish(rng, 40 * T + 123, 0.001), random_csvish(rng, 33 * T, 0.3), np.full(7 * T + 5, 0x2C, dtype=np.uint8)]
    check_batch(ctx, pkg, torch, oracle, big, [0, 1, 0])


def test_batch_capacity_protocol_and_argument_checks(ctx, pkg, oracle):
    import torch
    rng = np.random.default_rng(5)
    bufs = [random_csvish(rng, 100_000, 0.01) for _ in range(3)]
    wants = [oracle.scalar_index(b)[0] for b in bufs]
    caps = [wants[0].size, 10, 0]
    recs, tapes = run_batch(ctx, pkg, torch, bufs, [0, 0, 0], [0, 0, 0], [0, 0, 0], caps)
    for i in range(3):
        assert recs[i].count == wants[i].size and recs[i].written == min(caps[i], wants[i].size)
        k = recs[i].written
        assert np.array_equal(tapes[i][:k].cpu().numpy().view(np.uint64), wants[i][:k]) and bool((tapes[i][k:] == -1).all())
    dres = torch.zeros((2, 8), dtype=torch.int64, device="cuda:0")
    d = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.StructureError):   # ENTER_GUESS is not a batch state
        ctx.stage1_index_batch_device_async([(d.data_ptr(), 64, 0, 0, 0, pkg.ENTER_GUESS)], dres.data_ptr())
    with pytest.raises(pkg.StructureError):   # a capacity without a tape
        ctx.stage1_index_batch_device_async([(d.data_ptr(), 64, 0, 0, 5, 0)], dres.data_ptr())
    with pytest.raises(pkg.StructureError):   # no buffers
        ctx.stage1_index_batch_device_async([], dres.data_ptr())


def test_batch_across_an_epoch_wrap(pkg, oracle):
    """The look-back words carry a 10-bit launch epoch; when it wraps (every 1024 launches of a context) the last
    workgroup clears the words used since the last wrap — AFTER it has read every buffer's last word.  (Round 3's first
    batched kernel cleared first: every 1024th batch came back with the spin-bound error flag; found by scripts/soak.py.)
    1 100 batched launches on a fresh context, mixed with single-buffer launches: every record equals the first's."""
    import torch
    rng = np.random.default_rng(1024)
    T = pkg.tile_bytes()
    ctx = pkg.Context(0)
    bufs = [random_csvish(rng, n, 0.05) for n in (2 * T + 5, 100, 3 * T - 1)]
    dbufs = [torch.from_numpy(b).cuda() for b in bufs]
    tapes = [torch.full((b.size + 1,), -1, dtype=torch.int64, device="cuda:0") for b in bufs]
    items = [(d.data_ptr(), b.size, 0, t.data_ptr(), t.numel(), i & 1) for i, (d, b, t) in enumerate(zip(dbufs, bufs, tapes))]
    dres = torch.zeros((3, 8), dtype=torch.int64, device="cuda:0")
    dsingle = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    first = None
    for it in range(1100):
        ctx.stage1_index_batch_device_async(items, dres.data_ptr())
        got = dres.cpu()
        if first is None:
            first = got.clone()
            for i, b in enumerate(bufs):
                want, q = oracle.scalar_index(b, in_quote_in=i & 1)
                assert int(got[i, 0]) == want.size and (int(got[i, 4]) & 0xFFFFFFFF) == 0
                assert np.array_equal(tapes[i][: want.size].cpu().numpy().view(np.uint64), want)
        assert torch.equal(got, first), it
        if it % 3 == 0:   # single-buffer launches share the context's epoch counter
            ctx.stage1_index_device_async(dbufs[0].data_ptr(), bufs[0].size, 0, 0, tapes[0].data_ptr(), tapes[0].numel(),
                                          dsingle.data_ptr())
    torch.cuda.synchronize()
    for i, b in enumerate(bufs[1:], start=1):
        want, _ = oracle.scalar_index(b, in_quote_in=i & 1)
        assert np.array_equal(tapes[i][: want.size].cpu().numpy().view(np.uint64), want)
    ctx.close()
