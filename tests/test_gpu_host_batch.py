"""csvsimd_stage1_index_batch: reader::read for MANY host files in one call (round 5; VERDICT r4 missing #3 / next #4).

The reference works per file (csv_simd::create, /root/reference/src/lib.rs:61-74) and its own inputs are 96-623 bytes
(res/*.csv): per item the call must produce exactly what csvsimd_stage1_index does for that buffer alone, i.e. the
oracle's index (oracle.sse_read = the restatement of reader::read, src/reader.rs:150-306, for len >= 64; the scalar
definition for shorter buffers, where the reference itself is undefined)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, random_csvish

pytestmark = pytest.mark.gpu


def want_of(oracle, d):
    return oracle.sse_read(d) if d.size >= 64 else oracle.scalar_read(d)


def test_golden_fixtures_inside_one_batch(pkg, ctx, oracle, golden):
    rng = np.random.default_rng(5)
    files = []
    for name, (data, exp) in golden.items():
        files.append(np.frombuffer(data, dtype=np.uint8))
    fixtures = len(files)
    for _ in range(61):
        files.insert(int(rng.integers(0, len(files) + 1)), random_csvish(rng, int(rng.integers(64, 3000)), 0.02))
    got = ctx.read_many(files)
    assert len(got) == len(files)
    seen = 0
    for d, g in zip(files, got):
        assert np.array_equal(g, want_of(oracle, d))
    for name, (data, exp) in golden.items():      # and the committed expected indexes themselves
        idx = next(i for i, f in enumerate(files) if f.size == len(data) and f.tobytes() == data)
        assert got[idx].tolist() == exp["index"], name
        seen += 1
    assert seen == fixtures == 3


def test_many_files_every_kind(pkg, ctx, oracle):
    """2 500 files of 0 ... 24 KiB in several groups: empty files, files shorter than 64 bytes, files that leave a quote open
    (the next file must still be entered outside a string), all-comma files (denser than their share of the group's tape
    block: they take the single-file path inside the call), one file of 3 MiB (too large to pack), count-only items."""
    rng = np.random.default_rng(77)
    files = []
    for i in range(2500):
        kind = i % 25
        if kind == 0:
            d = np.zeros(0, dtype=np.uint8)
        elif kind == 1:
            d = random_csvish(rng, int(rng.integers(1, 64)), 0.05)
        elif kind == 2:
            d = np.full(int(rng.integers(100, 9000)), 0x2C, dtype=np.uint8)          # every byte structural
        elif kind == 3:
            d = random_csvish(rng, int(rng.integers(64, 5000)), 0.0)
            d[int(rng.integers(0, d.size))] = 0x22                                     # exactly one quote: left open
        else:
            d = random_csvish(rng, int(rng.integers(64, 24 << 10)), 0.01)
        files.append(d)
    big = random_csvish(rng, (3 << 20) + 17, 0.001)
    files.insert(1234, big)
    got = ctx.read_many(files)
    for i, (d, g) in enumerate(zip(files, got)):
        assert np.array_equal(g, want_of(oracle, d)), (i, d.size)
    for (tape_len, q, status), d in zip(ctx.last_batch, files):
        assert status == 0 and q == int(np.count_nonzero(d == 0x22) & 1)
    # count-only items and tapes that are too small, mixed with ordinary ones
    import ctypes as C
    sub = files[100:400]
    tapes = [np.full(d.size + 1, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64) for d in sub]
    items = (pkg.HostBatchItem * len(sub))()
    for j, (it, d, t) in enumerate(zip(items, sub, tapes)):
        it.buf, it.len = (d.ctypes.data if d.size else None), d.size
        if j % 3 == 0:
            it.tape, it.tape_cap = None, 0                                            # count only
        elif j % 3 == 1:
            it.tape, it.tape_cap = t.ctypes.data, min(5, t.size)                     # too small (unless the file is tiny)
        else:
            it.tape, it.tape_cap = t.ctypes.data, t.size
    rc = ctx.read_many_into(items)
    assert rc in (0, pkg.ERR_TAPE_CAPACITY)
    saw_capacity = False
    for j, (it, d, t) in enumerate(zip(items, sub, tapes)):
        want = want_of(oracle, d)
        assert it.tape_len == want.size, j
        if j % 3 == 0:
            assert it.status == 0
        elif j % 3 == 1:
            cap = min(5, t.size)
            if want.size > cap:
                saw_capacity = True
                assert it.status == pkg.ERR_TAPE_CAPACITY
                assert np.array_equal(t[:cap], want[:cap]) and bool((t[cap:] == 0xFFFFFFFFFFFFFFFF).all()), j
            else:
                assert it.status == 0 and np.array_equal(t[: want.size], want)
        else:
            assert it.status == 0 and np.array_equal(t[: want.size], want)
    assert saw_capacity and rc == pkg.ERR_TAPE_CAPACITY


def test_ten_thousand_4k_files_of_the_quoted_corpus(pkg, ctx, oracle):
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    row = cols * (width + 1)
    per = (4096 // row) * row or row
    whole = oracle.synth(0, per * 10_000, cols, width, seed, q)
    whole = np.frombuffer(bytes(whole), dtype=np.uint8) if not isinstance(whole, np.ndarray) else whole
    files = [whole[i * per: (i + 1) * per] for i in range(10_000)]
    got = ctx.read_many(files)
    rng = np.random.default_rng(1)
    for i in list(rng.integers(0, 10_000, 400)) + [0, 1, 9_999]:
        assert np.array_equal(got[i], oracle.sse_read(np.ascontiguousarray(files[i]))), i
    # every file: the count and the last entry (a file of whole rows ends with LF)
    for f, g in zip(files, got):
        assert g[-1] == f.size - 1
    total = sum(g.size - 1 for g in got)
    assert total == oracle.scalar_read(whole).size - 1       # files are whole rows: nothing is lost or doubled at the cuts
    # an empty batch and a batch of one
    assert ctx.read_many([]) == []
    one = ctx.read_many([files[7]])
    assert np.array_equal(one[0], got[7])


@pytest.mark.timeout(300)
def test_hot_pipeline_threads_finish_in_any_order(pkg, oracle):
    """Round 5's pipelines keep their stager / expander threads hot between calls, and the CALLER stages chunk (group) 0 while
    the stager is on chunk 1 already: the two finish in either order.  The first version counted both in one word (the
    stager's "2 staged" was overwritten by the caller's "1") and a call whose second chunk was short waited for ever — found
    by scripts/fuzz_gpu.py, pinned here: thousands of back-to-back calls whose second chunk / group is a fraction of the
    first, single files and batches alternating on one context."""
    rng = np.random.default_rng(2)
    c = pkg.Context(0)
    try:
        MiB = 1 << 20
        sizes = [2 * MiB + 570_000, 2 * MiB + 600_001, 2 * MiB + 800_064, 2 * MiB + 524_289]     # plans [1 MiB, a short chunk, 1 MiB]
        datas = [random_csvish(rng, n, 0.01) for n in sizes]
        wants = [oracle.scalar_read(d) for d in datas]
        for d in datas:
            sz = [b - a for a, b in zip(pkg.ingest_chunk_plan(d.size), pkg.ingest_chunk_plan(d.size)[1:])]
            assert len(sz) >= 3 and min(sz[1:-1] or sz) <= sz[0]
        tapes = [np.empty(w.size + 8, dtype=np.uint64) for w in wants]
        small = [random_csvish(rng, int(rng.integers(100, 9000)), 0.02) for _ in range(300)]
        small_want = [oracle.scalar_read(f) for f in small]
        for rep in range(1500):
            i = rep % len(datas)
            rc, tl, _ = c.read_into(datas[i], tapes[i])
            assert rc == 0 and tl == wants[i].size
            if rep % 100 == 0:
                assert np.array_equal(tapes[i][:tl], wants[i])
            if rep % 25 == 0:
                got = c.read_many(small)
                assert all(np.array_equal(g, w) for g, w in zip(got, small_want))
    finally:
        c.close()
