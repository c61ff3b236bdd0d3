"""world_size-N gloo worker for test_sharded.py (CPU).  The shard pass is a stand-in backed by
the oracle (allowed here: tests/ may use the oracle; the product path is GPU only) so that the
N>1 control flow — speculative pass, one all-gather, stitch, conditional re-emit — is exercised
exactly as bench.py / the GPU path drive it."""
import os
import sys

import numpy as np


def make_data(n, seed, p_quote):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import random_csvish
    return random_csvish(np.random.default_rng(seed), n, p_quote)


def worker(rank, world, port, n, seed, p_quote, skew, outdir, device_flow=False, guess=False):
    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from csv_simd_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = make_data(n, seed, p_quote)
        lo, hi = sharded.shard_range(n, rank, world, align=64, skew=skew)
        shard = data[lo:hi]
        passes = []

        d_result = torch.zeros(8, dtype=torch.int64)  # the result record, as the GPU path leaves it

        def run_pass(in_quote_in):
            passes.append(in_quote_in)
            if in_quote_in == pkg.ENTER_GUESS:
                # what the kernel does: the state under which the shard's first eight tiles have more entries
                _, a0, b0 = oracle.shard_descriptor(shard[: 8 * pkg.tile_bytes()])
                in_quote_in = int(b0 > a0)
            entries, inq_out = oracle.scalar_index(shard, base_off=lo, in_quote_in=in_quote_in)
            p, c0, c1 = oracle.shard_descriptor(shard)
            r = pkg.ShardResult()
            r.count, r.count_enter_outside, r.count_enter_inside = entries.size, c0, c1
            r.quote_parity, r.in_quote_out, r.written = p, inq_out, entries.size
            r.in_quote_in_used = in_quote_in
            d_result.copy_(torch.tensor(sharded.words_from_result(r), dtype=torch.int64))
            run_pass.entries = entries

        first = pkg.ENTER_GUESS if (guess and rank > 0) else 0   # rank 0 knows how the file starts
        if device_flow:
            # the control flow bench.py and the native C entry point drive on GPUs (stitch in "device"
            # memory, re-emit launch that reads its entering state from there), rehearsed on host tensors
            depth = 2 if device_flow == 2 else 1
            step = sharded.ShardedStep(torch.device("cpu"), depth=depth)

            def callbacks(slot):
                sl = step.slots[slot]

                def launch(inq):
                    nonlocal d_result
                    d_result = sl.d_result
                    run_pass(inq)

                def reemit(d_stitch_ptr):
                    nonlocal d_result
                    assert d_stitch_ptr == sl.d_stitch.data_ptr()
                    if (int(sl.d_stitch[4]) >> 32) & 1:    # csvsimd_stitch.reemit: what the kernel reads when it starts
                        d_result = sl.d_result
                        run_pass(int(sl.d_stitch[0]) & 1)   # ... and the true entering state next to it
                return launch, reemit

            if depth == 1:
                st, final, records = step.run(*callbacks(0), rehearsal=True, first_state=first)
            else:
                # two steps in flight, as bench.py drives them: both enqueued before either is collected
                step.enqueue(*callbacks(0), rehearsal=True, slot=0, first_state=first)
                step.enqueue(*callbacks(1), rehearsal=True, slot=1, first_state=first)
                try:
                    step.enqueue(*callbacks(0), rehearsal=True, slot=0, first_state=first)
                    raise AssertionError("a slot must not be enqueued again before it is collected")
                except RuntimeError:
                    pass
                st, final, records = step.collect(0)
                st1, final1, records1 = step.collect(1)
                assert (st1.in_quote_in, st1.count, st1.tape_index_base, st1.total_entries, st1.in_quote_final) == \
                    (st.in_quote_in, st.count, st.tape_index_base, st.total_entries, st.in_quote_final)
                assert final1.count == final.count and len(records1) == len(records)
                passes[:] = passes[:len(passes) // 2]   # the second step repeated the first one's passes
            re_emitted = bool(st.reemit)
            assert len(records) == world and records[rank].count_enter_outside + records[rank].count_enter_inside \
                == int(np.isin(shard, (0x2C, 0x0A, 0x0D)).sum())
        else:
            st, final, re_emitted = sharded.index_sharded(run_pass, d_result, first_state=first)
        assert final.count == st.count and final.in_quote_in_used == st.in_quote_in
        assert re_emitted == bool(st.reemit) and passes == ([first, st.in_quote_in] if re_emitted else [first])
        np.save(os.path.join(outdir, f"shard{rank}.npy"), run_pass.entries)
        np.save(os.path.join(outdir, f"meta{rank}.npy"),
                np.array([st.in_quote_in, st.tape_index_base, st.total_entries, st.in_quote_final, lo, hi],
                         dtype=np.uint64))
    finally:
        dist.destroy_process_group()
