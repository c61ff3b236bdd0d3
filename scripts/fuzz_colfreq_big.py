#!/usr/bin/env python3
"""dev tool: time-bounded fuzz of the columnar frequency count at sizes the differential fuzz (fuzz_columnar.py) does not
reach: up to 6 M records, so that pass 2 has more partitions than workgroups (tickets), few tuples (merged neighbouring
partitions), several rounds per partition (skew), strides of 16 / 32 (rows cached in LDS) and 48 / 64 (not), with and
without a length array.  Values are equal iff their ids are: expected = numpy.unique(ids).  usage: fuzz_colfreq_big.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    n = int(rng.choice([rng.integers(1, 70_000), rng.integers(900_000, 1_200_000), rng.integers(1_048_577, 6_000_000)]))
    stride = int(rng.choice([16, 32, 32, 32, 48, 64]))
    kind = rng.choice(["one", "few", "mid", "many", "distinct", "skew"])
    if kind == "one": ids = np.zeros(n, dtype=np.uint64)
    elif kind == "few": ids = rng.integers(0, int(rng.integers(2, 400)), size=n).astype(np.uint64)
    elif kind == "mid": ids = rng.integers(0, int(rng.integers(400, 20_000)), size=n).astype(np.uint64)
    elif kind == "many": ids = rng.integers(0, max(2, n // int(rng.integers(2, 30))), size=n).astype(np.uint64)
    elif kind == "distinct": ids = rng.permutation(n).astype(np.uint64)
    else:  # a few very heavy values among many light ones: heavy partitions, several rounds
        ids = np.where(rng.random(n) < 0.7, rng.integers(0, 5, size=n), rng.integers(5, max(6, n // 3), size=n)).astype(np.uint64)
    ids = ids * np.uint64(0x9E3779B97F4A7C15)  # spread over all 8 bytes
    host = np.zeros((n, stride), dtype=np.uint8)
    host[:, :8] = ids.view(np.uint8).reshape(n, 8)
    host[:, stride - 8:] = (ids ^ np.uint64(0xA5A5A5A5A5A5A5A5)).view(np.uint8).reshape(n, 8)  # values differ in late bytes too
    col = torch.from_numpy(host).to(dev)
    with_len = bool(rng.integers(0, 2))
    d_len = 0
    if with_len:  # the same id always has the same length; some longer than the stride (counted as truncated)
        lens_h = (stride - (ids % np.uint64(3)).astype(np.int64)).astype(np.int32)
        over = (ids % np.uint64(11)) == 0
        lens_h[over] = stride + 5
        lens = torch.from_numpy(lens_h).to(dev)
        d_len = lens.data_ptr()
    first_record = int(rng.integers(0, 1 << 40))
    need = pkg.columnar_frequency_scratch_bytes(n)
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    uniq, first, counts = np.unique(ids, return_index=True, return_counts=True)
    ent = torch.full((uniq.size + 3, 2), -1, dtype=torch.int64, device=dev)
    st = pkg.columnar_frequency_device(ctx, col.data_ptr(), d_len, n, stride, first_record, scratch.data_ptr(), need,
                                       ent.data_ptr(), ent.shape[0], allow_capacity=True)
    exp_trunc = int((lens_h > stride).sum()) if with_len else 0
    assert (st.n_records, st.n_distinct, st.truncated, st.overflow) == (n, uniq.size, exp_trunc, 0), (n, stride, kind, with_len, st.n_records, st.n_distinct, st.truncated, st.overflow, uniq.size, exp_trunc)
    got = ent[: uniq.size].cpu().numpy()
    order = np.argsort(got[:, 0])
    exp_first = np.sort(first.astype(np.int64) + first_record)
    assert np.array_equal(got[order, 0], exp_first), (n, stride, kind, with_len, "first records")
    assert np.array_equal(got[order, 1], counts[np.argsort(first)].astype(np.int64)), (n, stride, kind, with_len, "counts")
    assert bool((ent[uniq.size:] == -1).all())
    cases += 1
    if cases % 10 == 0: print("cases", cases, "last", n, stride, kind, with_len, uniq.size, flush=True)
print("fuzz_colfreq_big: %d cases in %.0f s, seed %d: clean" % (cases, budget, seed))
