#!/usr/bin/env python3
"""dev tool (round 5): csvsimd_columnar_search_device against Python's ==, startswith and `in` on random columns: strides 16 /
32 / 48, 1 ... 300 000 records, alphabets of 2 ... 26 letters (and zero bytes), with and without a lengths array (over-long
records included), needles drawn from the data and at random, every mode; the count, the bitmap and the return code.
usage: fuzz_colsearch.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(0)
    t_end = time.time() + seconds
    cases = calls = 0
    while time.time() < t_end:
        stride = int(rng.choice([16, 32, 32, 32, 48]))
        n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 70_000, 300_000]))
        letters = int(rng.choice([2, 3, 5, 26]))
        alphabet = np.frombuffer(bytes(range(97, 97 + letters)), dtype=np.uint8)
        if rng.integers(0, 4) == 0:
            alphabet = np.concatenate([alphabet, np.zeros(1, dtype=np.uint8)])
        with_len = bool(rng.integers(0, 2))
        body = rng.choice(alphabet, size=(n, stride))
        host = body.copy()
        lens = None
        if with_len:
            lens = rng.integers(0, stride + 1, size=n).astype(np.int32)
            mask = np.arange(stride)[None, :] >= lens[:, None]
            host[mask] = 0
            if rng.integers(0, 3) == 0:
                lens[rng.integers(0, n, size=max(1, n // 50))] = stride + int(rng.integers(1, 9))   # over-long records
        rows = [host[i, : min(int(lens[i]), stride)].tobytes() for i in range(n)] if with_len else [host[i].tobytes() for i in range(n)]
        col = torch.from_numpy(host).to("cuda:0")
        dl = torch.from_numpy(lens).to("cuda:0") if with_len else None
        bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
        needles = [b""]
        for _ in range(6):
            m = int(rng.integers(1, stride + 2))
            r = rows[int(rng.integers(0, n))] + bytes(rng.choice(alphabet, size=stride + 2))
            at = int(rng.integers(0, max(1, stride - m + 1)))
            needles.append(bytes(r[at: at + m]))
            needles.append(bytes(rng.choice(alphabet, size=int(rng.integers(1, 6)))))
        trunc = bool(with_len and (lens > stride).any())
        for needle in needles:
            for mode, fn in ((pkg.SEARCH_EQUALS, lambda r: r == needle), (pkg.SEARCH_STARTS_WITH, lambda r: r.startswith(needle)),
                             (pkg.SEARCH_CONTAINS, lambda r: needle in r)):
                want = np.fromiter((fn(r) for r in rows), dtype=bool, count=n)
                bm.zero_()
                try:
                    got_n = pkg.columnar_search_device(ctx, col.data_ptr(), dl.data_ptr() if with_len else 0, n, stride, needle, mode,
                                                       bm.data_ptr())
                    rc_trunc = False
                except pkg.StructureError as e:
                    assert e.code == pkg.ERR_TAPE_CAPACITY, e.code
                    rc_trunc = True
                    got_n = None
                assert rc_trunc == trunc, (stride, n, needle, mode, rc_trunc, trunc)
                bits = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
                assert np.array_equal(bits, want), (stride, n, letters, with_len, needle, mode)
                assert got_n is None or got_n == int(want.sum()), (stride, n, needle, mode, got_n, int(want.sum()))
                calls += 1
        cases += 1
        if cases % 10 == 0:
            print(f"{cases} columns, {calls} calls ok", flush=True)
    print(f"fuzz_colsearch: {cases} columns, {calls} calls identical to Python (seed {seed})")


if __name__ == "__main__":
    main()
