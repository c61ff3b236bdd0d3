"""csvsimd_stage1_index_multi: ONE host buffer -> G shards on the GPUs of this process (here: G contexts on the one GPU
of the box), bytes streamed to the devices concurrently, tapes left sharded in order.  The concatenation of the shard
tapes behind the sentinel must be the oracle's index of the whole buffer (reference reader::read, src/reader.rs:150-306),
for every G, with shards that start inside quoted fields."""
import numpy as np
import pytest

from conftest import random_csvish


def test_multi_shard_ranges_tile_the_file(pkg):
    for n in (0, 1, 63, 64, 1000, (1 << 33) + 12345):
        for g in (1, 2, 3, 8):
            cuts = [pkg.multi_shard_range(n, g, i) for i in range(g)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert all(c[0] % 64 == 0 for c in cuts)
            assert all(c[1] >= c[0] for c in cuts)
    with pytest.raises(pkg.StructureError):
        pkg.multi_shard_range(100, 0, 0)
    with pytest.raises(pkg.StructureError):
        pkg.multi_shard_range(100, 2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("g", [1, 2, 3])
def test_one_host_buffer_to_g_shards(pkg, oracle, g):
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1000 + g)
    n = (70 << 20) + 4242                       # > 2 staging chunks per shard at g = 1, ragged
    d = random_csvish(rng, n, 0.002)            # long quoted stretches: shards 1.. usually start inside or near one
    ctxs = [pkg.Context(0) for _ in range(g)]
    ranges = [pkg.multi_shard_range(n, g, i) for i in range(g)]
    # make sure at least one interior cut lies INSIDE a quoted stretch
    if g > 1:
        c = ranges[1][0]
        d[c - 100] = 0x22
        d[c - 99: c + 100] = 0x2C               # 199 commas right across the cut, inside the string opened at c - 100 ...
        state = int(np.count_nonzero(d[: c - 100] == 0x22) & 1)
        if state:                               # ... unless that quote CLOSED a string: open another one
            d[c - 101] = 0x22
    want = oracle.scalar_read(d)
    dbufs = [torch.zeros(max(e - b, 1), dtype=torch.uint8, device="cuda:0") for b, e in ranges]
    caps = [(e - b) + 8 for b, e in ranges]
    dtapes = [torch.full((c,), -1, dtype=torch.int64, device="cuda:0") for c in caps]
    # the call works on the contexts' private streams: dbuf / dtape must be idle when it starts (include/csvsimd.h) —
    # the fills above run on torch's stream
    torch.cuda.synchronize()
    dev_before = torch.cuda.current_device()
    sh = pkg.stage1_index_multi(d, ctxs, [t.data_ptr() for t in dbufs], [t.data_ptr() for t in dtapes], caps, 0)
    assert torch.cuda.current_device() == dev_before          # the caller's current device is the caller's
    torch.cuda.synchronize()
    parts, base, state = [np.zeros(1, dtype=np.uint64)], 1, 0
    for i in range(g):
        b, e = ranges[i]
        assert (sh[i].begin, sh[i].end) == (b, e)
        assert bytes(dbufs[i][: e - b].cpu().numpy()) == d[b:e].tobytes()          # the bytes are on the device
        assert sh[i].stitch.tape_index_base == base and sh[i].stitch.in_quote_in == state
        assert sh[i].result.in_quote_in_used == state and sh[i].result.count == sh[i].stitch.count
        k = sh[i].result.count
        parts.append(dtapes[i][:k].cpu().numpy().view(np.uint64))
        # (behind its k entries a tape may hold leftovers of a first pass that guessed the other entering state)
        assert k <= caps[i] and sh[i].result.written == k
        base += k
        state = sh[i].result.in_quote_out
    got = np.concatenate(parts)
    assert sh[g - 1].stitch.total_entries == want.size == got.size
    assert np.array_equal(got, want)
    if g > 1:
        truth = int(np.count_nonzero(d[: ranges[1][0]] == 0x22) & 1)
        assert truth == 1 and sh[1].stitch.in_quote_in == 1                        # shard 1 really starts inside a string
    # capacity protocol: a tape that is too small is reported, nothing is written past it
    small = torch.full((16,), -1, dtype=torch.int64, device="cuda:0")
    with pytest.raises(pkg.StructureError) as err:
        pkg.stage1_index_multi(d, ctxs, [t.data_ptr() for t in dbufs], [small.data_ptr()] + [t.data_ptr() for t in dtapes[1:]],
                               [8] + caps[1:], 0)
    assert err.value.code == pkg.ERR_TAPE_CAPACITY and bool((small[8:] == -1).all())
    # the same context twice is refused (shards run concurrently)
    if g > 1:
        with pytest.raises(pkg.StructureError) as err:
            pkg.stage1_index_multi(d, [ctxs[0]] * g, [t.data_ptr() for t in dbufs], [t.data_ptr() for t in dtapes], caps, 0)
        assert err.value.code == pkg.ERR_INVALID_ARG
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_one_host_buffer_to_shards_on_distinct_devices(pkg, oracle):
    """The same flow with every shard on ITS OWN GPU (device ordinals 0 .. G - 1): what csvsimd_stage1_index_multi is for.
    Skipped on a one-GPU box; on the driver's 8-GPU node it runs with G = device_count (until then the
    entry point had only ever seen `cuda:0`).  Tapes stay on their devices; the concatenation behind the sentinel is the
    oracle's index of the whole buffer (reference reader::read, src/reader.rs:150-306)."""
    import torch
    g = min(pkg.device_count(), 8)
    if g < 2:
        pytest.skip("needs at least two GPUs in this process")
    rng = np.random.default_rng(4242)
    n = (96 << 20) + 1717
    d = random_csvish(rng, n, 0.002)
    ranges = [pkg.multi_shard_range(n, g, i) for i in range(g)]
    c = ranges[1][0]
    d[c - 100] = 0x22
    d[c - 99: c + 100] = 0x2C
    if int(np.count_nonzero(d[: c - 100] == 0x22) & 1):
        d[c - 101] = 0x22                       # shard 1 starts inside a quoted field full of commas
    want = oracle.scalar_read(d)
    ctxs = [pkg.Context(i) for i in range(g)]
    dbufs = [torch.zeros(max(e - b, 1), dtype=torch.uint8, device=f"cuda:{i}") for i, (b, e) in enumerate(ranges)]
    caps = [(e - b) + 8 for b, e in ranges]
    dtapes = [torch.full((cap,), -1, dtype=torch.int64, device=f"cuda:{i}") for i, cap in enumerate(caps)]
    for i in range(g):
        torch.cuda.synchronize(i)
    dev_before = torch.cuda.current_device()
    sh = pkg.stage1_index_multi(d, ctxs, [t.data_ptr() for t in dbufs], [t.data_ptr() for t in dtapes], caps, 0)
    assert torch.cuda.current_device() == dev_before
    parts, base, state = [np.zeros(1, dtype=np.uint64)], 1, 0
    for i in range(g):
        torch.cuda.synchronize(i)
        b, e = ranges[i]
        assert bytes(dbufs[i][: e - b].cpu().numpy()) == d[b:e].tobytes()
        assert sh[i].stitch.tape_index_base == base and sh[i].stitch.in_quote_in == state
        k = sh[i].result.count
        assert sh[i].result.in_quote_in_used == state and k == sh[i].stitch.count and sh[i].result.written == k
        parts.append(dtapes[i][:k].cpu().numpy().view(np.uint64))
        base += k
        state = sh[i].result.in_quote_out
    got = np.concatenate(parts)
    assert sh[g - 1].stitch.total_entries == want.size == got.size and np.array_equal(got, want)
    assert sh[1].stitch.in_quote_in == 1
    for c_ in ctxs:
        c_.close()
