"""BASELINE.json configs[3] AS config 4, at full size, on one GPU: the 64 GiB 64-column file cut into EIGHT contiguous
shards of 8 GiB at i * N / 8 + 777 (mid-row; SURVEY.md §8d's "deliberately misaligned variant"), one context per shard,
and the sharded step exactly as the ranks of an 8-GPU job run it — first pass (shard 0 knows how the file starts, every
other shard lets the kernel choose its entering state: CSVSIMD_ENTER_GUESS), the eight 64-byte records side by side in
device memory (a device copy stands in for the all-gather: one GPU has no peers), the stitch kernel, the re-emit launch
that reads its flag and state from device memory.  64 GiB of input + ~21 GiB of tape fit one 288 GB MI355X.

Checked against the oracle the way SURVEY §8d "Parity at scale" prescribes: the order-sensitive checksum of the WHOLE
tape (device kernel per shard == the CPU oracle over 1-GiB windows of the same bytes) and 1 Mi entries either side of
every shard cut, entry for entry; the quote-free variant also against the closed form.  What is left untested after
this is RCCL itself at N > 1 (the transport of eight 64-byte records), nothing of the path's arithmetic.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD = 8
SKEW = 777
WINDOW_ENTRIES = 1 << 20
MASK64 = (1 << 64) - 1


def _run_step(pkg, torch, sharded, ctxs, dbuf, cuts, tapes, d_fin, d_all, d_st, file_inq, guess=True):
    """One sharded step over all shards on one stream; nothing between the first launch and the synchronise reads a
    host value."""
    s = torch.cuda.current_stream().cuda_stream
    for r in range(WORLD):
        lo, hi = cuts[r], cuts[r + 1]
        first = file_inq if r == 0 else (pkg.ENTER_GUESS if guess else 0)
        ctxs[r].stage1_index_device_async(dbuf.data_ptr() + lo, hi - lo, lo, first, tapes[r].data_ptr(), tapes[r].numel(),
                                          d_fin[r].data_ptr(), s)
        d_all[8 * r: 8 * r + 8].copy_(d_fin[r])                       # "all-gather"
    for r in range(WORLD):
        lo, hi = cuts[r], cuts[r + 1]
        pkg.stitch_shards_device_async(d_all.data_ptr(), WORLD, r, file_inq, d_st[r].data_ptr(), s)
        ctxs[r].stage1_reemit_device_async(dbuf.data_ptr() + lo, hi - lo, lo, d_st[r].data_ptr(), tapes[r].data_ptr(),
                                           tapes[r].numel(), d_fin[r].data_ptr(), s)
    torch.cuda.synchronize()
    recs = [sharded.result_from_words(d_all[8 * r: 8 * r + 8].cpu().tolist()) for r in range(WORLD)]
    fins = [sharded.result_from_words(d_fin[r].cpu().tolist()) for r in range(WORLD)]
    sts = [sharded.stitch_from_words(d_st[r].cpu().tolist()) for r in range(WORLD)]
    return recs, fins, sts


def _entries_between(torch, tapes, counts, cuts, lo, hi):
    """The tape entries with lo <= value < hi, from the sharded device tapes (ascending within and across shards)."""
    parts = []
    for r in range(WORLD):
        if cuts[r + 1] <= lo or cuts[r] >= hi:
            continue
        t = tapes[r][: counts[r]]
        a = int(torch.searchsorted(t, torch.tensor([lo], dtype=torch.int64, device=t.device)).item())
        b = int(torch.searchsorted(t, torch.tensor([hi], dtype=torch.int64, device=t.device)).item())
        parts.append(t[a:b].cpu().numpy().view(np.uint64))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)


@pytest.mark.parametrize("name,file_inq", [("64x31_q10", 0), ("64x31_q10", 1), ("64x31_noquote", 0)])
def test_config4_full_size_eight_shards_on_one_gpu(pkg, oracle, name, file_inq):
    import torch
    from csv_simd_amd import sharded
    assert torch.cuda.is_available()
    free, _ = torch.cuda.mem_get_info()
    if free < 100 << 30:
        pytest.skip("needs ~90 GiB of free HBM (64 GiB file + tapes)")
    cols, width, seed, q = pkg.WORKLOADS[name]
    row, pitch = cols * (width + 1), width + 1
    N = 1 << 36                                                       # 64 GiB: BASELINE config 4 (whole rows: 2048 | 2^36)
    assert N % row == 0
    cuts = [0] + [i * (N // WORLD) + SKEW for i in range(1, WORLD)] + [N]
    dev = torch.device("cuda", 0)
    dbuf = torch.empty(N, dtype=torch.uint8, device=dev)
    pkg.synth_fill_device(dbuf.data_ptr(), 0, N, cols, width, seed, q)
    cap = int((N // WORLD + SKEW) // pitch * (1.25 if q else 1.0)) + 1024
    tapes = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(WORLD)]
    d_fin = torch.zeros((WORLD, 8), dtype=torch.int64, device=dev)
    d_all = torch.zeros(8 * WORLD, dtype=torch.int64, device=dev)
    d_st = torch.zeros((WORLD, sharded.STITCH_WORDS), dtype=torch.int64, device=dev)
    ctxs = [pkg.Context(0) for _ in range(WORLD)]
    try:
        for c, (lo, hi) in zip(ctxs, zip(cuts, cuts[1:])):
            c.reserve(hi - lo)
        recs, fins, sts = _run_step(pkg, torch, sharded, ctxs, dbuf, cuts, tapes, d_fin, d_all, d_st, file_inq)

        # ---- the stitch: entering states from the CPU generator alone, bases and totals as a chain --------------------
        state, base = file_inq, 1
        for r in range(WORLD):
            lo = cuts[r]
            r0 = (lo // row) * row                                    # a row starts in the file's own state
            truth = file_inq ^ (int(np.count_nonzero(oracle.synth(r0, lo - r0, cols, width, seed, q) == 0x22)) & 1)
            st, fin, rec = sts[r], fins[r], recs[r]
            assert (st.error, fin.error, rec.error) == (0, 0, 0)
            assert st.in_quote_in == truth == state, (r, st.in_quote_in, truth, state)
            assert st.tape_index_base == base and st.count == fin.count and fin.count <= cap, r
            assert st.reemit == int((rec.in_quote_in_used & 1) != truth), r
            assert fin.in_quote_in_used == truth, r
            if r > 0 and file_inq == 0:
                # the first eight tiles speak for the true state: nobody re-emits on CSV read the way it was written
                assert rec.in_quote_in_used == truth and st.reemit == 0, r
            if r > 0 and file_inq == 1:
                # the file is DECLARED to start inside a string: every shard's natural reading is the wrong one, so
                # every rank > 0 goes through the re-emit launch for real, at full size
                assert st.reemit == 1, r
            state, base = fin.in_quote_out, base + fin.count
        counts = [f.count for f in fins]
        assert all(s_.total_entries == base and s_.in_quote_final == state for s_ in sts)
        assert state == file_inq                                      # whole rows: the file ends in the state it began in

        # ---- closed form (quote-free): entry k of the file is byte k * pitch + width -----------------------------------
        if not q:
            assert base - 1 == N // pitch
            k = 0
            for r in range(WORLD):
                for a in range(0, counts[r], 64 << 20):
                    b = min(counts[r], a + (64 << 20))
                    want = torch.arange(k + a, k + b, dtype=torch.int64, device=dev) * pitch + width
                    assert torch.equal(tapes[r][a:b], want), (r, a)
                k += counts[r]

        # ---- every shard's tape is ascending and stays inside its byte range --------------------------------------------
        for r in range(WORLD):
            t = tapes[r][: counts[r]]
            if counts[r]:
                assert bool((t[1:] > t[:-1]).all()) and int(t[0]) >= cuts[r] and int(t[-1]) < cuts[r + 1], r

        # ---- 1 Mi entries either side of every cut, entry for entry against the oracle ----------------------------------
        # (a window starts on a row boundary, where the state is the file's; under file_inq = 1 entries are sparse, so
        # the windows are bounded in bytes and hold what they hold)
        wbytes = (WINDOW_ENTRIES + 4096) * pitch
        for c in cuts[1:-1]:
            w0 = ((c - wbytes) // row) * row
            w1 = min(N, w0 + 2 * wbytes + row)
            host = dbuf[w0:w1].cpu().numpy()
            want, _ = oracle.scalar_index(host, base_off=w0, in_quote_in=file_inq)
            got = _entries_between(torch, tapes, counts, cuts, w0, w1)
            assert np.array_equal(got, want), c
            if file_inq == 0:
                below = int(np.count_nonzero(want < c))
                assert below >= WINDOW_ENTRIES and want.size - below >= WINDOW_ENTRIES
            # the device bytes are the file both sides mean
            assert np.array_equal(host[: 1 << 20], oracle.synth(w0, 1 << 20, cols, width, seed, q))

        # ---- the whole tape: order-sensitive checksum, device kernel == CPU oracle over 1-GiB windows ------------------
        # (one variant: the CPU side reads all 64 GiB back and indexes them, ~20 s of host work)
        if q and file_inq == 0:
            out = torch.zeros((WORLD, 2), dtype=torch.int64, device=dev)
            for r in range(WORLD):
                pkg.tape_checksum_device(tapes[r].data_ptr(), counts[r], sts[r].tape_index_base, out[r].data_ptr())
            got_sum = [sum(int(v) & MASK64 for v in out[:, j].cpu().tolist()) & MASK64 for j in (0, 1)]
            import ctypes as C
            import os
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            win = 1 << 30                                             # a multiple of the row: windows start outside strings
            pin = torch.empty(win, dtype=torch.uint8).pin_memory()   # page aligned: what the SSE restatement's mmap is
            wtape = np.empty(win // 16, dtype=np.uint64)
            cnt = C.c_uint64()
            lib = oracle.lib()
            s1 = s2 = 0
            idx = 1
            for w0 in range(0, N, win):
                pin.copy_(dbuf[w0: w0 + win])
                rc = lib.oracle_sse_read_mt(pin.data_ptr(), win, threads, wtape.ctypes.data, wtape.size, C.byref(cnt))
                assert rc == 0
                e = wtape[1: cnt.value]                               # behind the sentinel; offsets relative to the window
                e += np.uint64(w0)
                a, b = oracle.tape_checksum(e, idx)
                s1, s2, idx = (s1 + a) & MASK64, (s2 + b) & MASK64, idx + e.size
            assert idx == base
            assert (s1, s2) == tuple(got_sum)
    finally:
        for c in ctxs:
            c.close()
