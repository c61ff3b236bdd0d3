"""Multi-rank stitch: host arithmetic (csvsimd_stitch_shards) and the N>1 control flow over a
real torch.distributed group (gloo, CPU, world_size 2, 3, 4 and 8)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import random_csvish

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dist_worker  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partition(pkg):
    from csv_simd_amd import sharded
    for n in (0, 1, 63, 64, 1000, 12345, 2**20 + 17):
        for world in (1, 2, 3, 8):
            for skew in (0, 777):
                cuts = [sharded.shard_range(n, r, world, 64, skew) for r in range(world)]
                assert cuts[0][0] == 0 and cuts[-1][1] == n
                for a, b in zip(cuts, cuts[1:]):
                    assert a[1] == b[0] and a[0] <= a[1]


def test_stitch_host_arithmetic(pkg, oracle):
    rng = np.random.default_rng(99)
    for trial in range(20):
        n = int(rng.integers(200, 5000))
        d = random_csvish(rng, n, 0.08)
        world = int(rng.integers(1, 9))
        cuts = sorted(int(x) for x in rng.integers(0, n + 1, size=world - 1))
        bounds = [0] + cuts + [n]
        results = []
        for i in range(world):
            p, c0, c1 = oracle.shard_descriptor(d[bounds[i]: bounds[i + 1]])
            r = pkg.ShardResult()
            r.quote_parity, r.count_enter_outside, r.count_enter_inside = p, c0, c1
            r.in_quote_in_used = int(rng.integers(0, 2))   # whatever state each first pass happened to run with
            results.append(r)
        full, inq = oracle.scalar_index(d)
        state, base = 0, 1
        for i in range(world):
            st = pkg.stitch_shards(results, i)
            e, q = oracle.scalar_index(d[bounds[i]: bounds[i + 1]], in_quote_in=state)
            assert (st.in_quote_in, st.count, st.tape_index_base) == (state, e.size, base)
            assert st.total_entries == full.size + 1 and st.in_quote_final == inq
            assert st.reemit == int(results[i].in_quote_in_used != state)   # re-emit iff the pass used the wrong state
            state, base = q, base + e.size


@pytest.mark.parametrize("world,p_quote,skew,device_flow,guess",
                         [(2, 0.0, 0, False, False), (2, 0.06, 0, False, False), (2, 0.06, 777, False, False),
                          (3, 0.1, 13, False, False), (2, 0.06, 777, True, False), (4, 0.1, 13, True, False),
                          (2, 0.1, 777, 2, False), (3, 0.06, 13, 2, False),
                          # ranks > 0 let the pass choose its entering state (CSVSIMD_ENTER_GUESS)
                          (3, 0.1, 13, False, True), (4, 0.06, 777, True, True), (3, 0.1, 13, 2, True),
                          # BASELINE config 4's rank count: 8 shards, cut mid-row, device flow, two steps in flight
                          (8, 0.1, 777, 2, True)])
def test_gloo_sharded_stitch(pkg, oracle, tmp_path, world, p_quote, skew, device_flow, guess):
    import torch.multiprocessing as mp
    n, seed = 40000, 4242
    mp.spawn(_dist_worker.worker, args=(world, free_port(), n, seed, p_quote, skew, str(tmp_path), device_flow, guess),
             nprocs=world, join=True)
    data = _dist_worker.make_data(n, seed, p_quote)
    want = oracle.scalar_read(data)
    shards = [np.load(tmp_path / f"shard{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    got = np.concatenate([np.zeros(1, dtype=np.uint64)] + shards)
    assert np.array_equal(got, want)
    base = 1
    for r in range(world):
        assert metas[r][1] == base and metas[r][2] == want.size
        base += shards[r].size
    if p_quote > 0:
        # the stitch must have been non-trivial for at least one rank in these seeds
        assert any(int(m[0]) == 1 for m in metas) or world == 2
