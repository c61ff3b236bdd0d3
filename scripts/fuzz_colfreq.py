#!/usr/bin/env python3
"""dev tool (round 5): the frequency count on LONG columns (the streaming kernel + the general passes behind it) against
numpy, on random shapes: 4.2 ... 9 Mi records, strides 16 / 32, 1 ... 6 000 values or all distinct, uniform / skewed / values that
only appear late in a share / few values up to some record and distinct ones after it, with and without a lengths array
(over-long records included).  usage: fuzz_colfreq.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(0)
    t_end = time.time() + seconds
    cases = 0
    kinds = {}
    while time.time() < t_end:
        n = int(rng.integers(4_200_000, 9_000_000))
        stride = int(rng.choice([16, 32]))
        nvals = int(rng.choice([1, 2, 17, 100, 700, 1400, 2000, 2290, 2304, 2315, 2500, 3000, 6000]))
        shape = str(rng.choice(["uniform", "skewed", "late", "switch", "distinct"]))
        if shape == "distinct":
            keys = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        else:
            if shape == "skewed":
                idx = np.minimum((rng.pareto(1.2, size=n) * 3).astype(np.int64), nvals - 1)
            else:
                idx = rng.integers(0, nvals, size=n)
            if shape == "late":       # half of the values only appear in the last tenth of every run of 100 000 records
                pos = np.arange(n) % 100_000
                early = pos < 90_000
                idx = np.where(early, idx % max(1, nvals // 2), idx)
            keys = idx.astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93) + np.uint64(1)
            if shape == "switch":
                at = int(rng.integers(0, n))
                keys[at:] = (np.arange(n - at, dtype=np.uint64) + np.uint64(7)) * np.uint64(0x9E3779B97F4A7C15)
        with_len = bool(rng.integers(0, 2))
        lens = None
        if with_len:
            lens = rng.integers(8, stride + 1, size=n).astype(np.int32)
            if rng.integers(0, 2):
                a = int(rng.integers(0, n - 50_000))
                lens[a: a + int(rng.integers(1, 50_000))] = stride + 3
        host = np.zeros((n, stride), dtype=np.uint8)
        host[:, :8] = keys.view(np.uint8).reshape(n, 8)
        col = torch.from_numpy(host).to("cuda:0")
        dl = torch.from_numpy(lens).to("cuda:0") if lens is not None else None
        need = pkg.columnar_frequency_scratch_bytes(n)
        scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        if lens is None:
            ident = keys
        else:
            _, kid = np.unique(keys, return_inverse=True)
            ident = kid.astype(np.uint64) * np.uint64(64) + lens.astype(np.uint64)
        uniq, first, counts = np.unique(ident, return_index=True, return_counts=True)
        ent = torch.full((uniq.size + 4, 2), -1, dtype=torch.int64, device="cuda:0")
        st = pkg.columnar_frequency_device(ctx, col.data_ptr(), dl.data_ptr() if dl is not None else 0, n, stride, 3,
                                           scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0], allow_capacity=True)
        trunc = 0 if lens is None else int((lens > stride).sum())
        tag = (n, stride, nvals, shape, with_len)
        assert (st.n_records, st.n_distinct, st.truncated, st.overflow) == (n, uniq.size, trunc, 0), (tag, st.n_distinct, uniq.size, st.truncated, trunc)
        got = ent[: uniq.size].cpu().numpy()
        order = np.argsort(got[:, 0])
        want_order = np.argsort(first)
        assert np.array_equal(got[order, 0] - 3, first[want_order]) and np.array_equal(got[order, 1], counts[want_order]), tag
        assert bool((ent[uniq.size:] == -1).all()), tag
        cases += 1
        kinds[shape] = kinds.get(shape, 0) + 1
        del col, dl, scratch, ent
        if cases % 5 == 0:
            print(f"{cases} cases ok {kinds}", flush=True)
    print(f"fuzz_colfreq: {cases} cases identical to numpy (seed {seed}) {kinds}")


if __name__ == "__main__":
    main()
