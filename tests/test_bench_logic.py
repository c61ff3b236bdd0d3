"""The host-side logic bench.py's "verified" rests on (CPU; no GPU needed): the cuts it makes for the quoted
mid-row leg, the job-level stitch check, the settle count.  A bug here would print a green line for a wrong tape."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_mid_row_cuts_are_what_they_claim(bench, pkg, oracle):
    cols, width, seed, q = pkg.WORKLOADS["64x31_q10"]
    row = cols * (width + 1)
    for world in (2, 3, 4, 8):
        shard = 64 << 20
        cuts = bench.mid_row_cuts(oracle, pkg, "64x31_q10", shard, world)
        per = (shard // row) * row
        assert cuts[0] == 0 and cuts[-1] == world * per and len(cuts) == world + 1
        assert all(a < b for a, b in zip(cuts, cuts[1:]))
        for i in range(1, world):
            c = cuts[i]
            r0 = (c // row) * row
            assert c != r0, "interior cuts lie in the middle of a row"
            before = oracle.synth(r0, c - r0, cols, width, seed, q)       # rows end outside a string
            state = int(np.count_nonzero(before == 0x22)) & 1
            assert state == i % 2, (world, i)      # odd cuts: the shard starts INSIDE a quoted field; even: outside
        # identical on every rank: a pure function of its arguments
        assert cuts == bench.mid_row_cuts(oracle, pkg, "64x31_q10", shard, world)


def test_verify_job_accepts_the_truth_and_nothing_else(bench):
    def facts(rank, lo, hi, count, state_in, state_out, base, total, final, reemit=0):
        return {"rank": rank, "lo": lo, "hi": hi, "count": count, "ref_count": count, "state_in": state_in,
                "state_out": state_out, "base": base, "total": total, "final": final, "reemit": reemit}
    good = [facts(0, 0, 100, 10, 0, 1, 1, 36, 0), facts(1, 100, 200, 20, 1, 1, 11, 36, 0, reemit=1),
            facts(2, 200, 300, 5, 1, 0, 31, 36, 0)]
    ok, reemits, total, inside = bench.verify_job(list(reversed(good)))    # order of arrival must not matter
    assert (ok, reemits, total, inside) == (True, 1, 36, 2)
    for field, rank, bad in (("base", 1, 12), ("state_in", 2, 0), ("count", 0, 11), ("total", 2, 37), ("final", 1, 1),
                             ("lo", 2, 201)):
        broken = [dict(f) for f in good]
        broken[rank][field] = bad
        assert bench.verify_job(broken)[0] is False, field


def test_settle_count_depends_on_bytes_only(bench):
    assert bench.settle_count(8 << 30) == bench.settle_count(8 << 30)
    assert 10 <= bench.settle_count(8 << 30) <= 20          # ~25 ms of 1.7-ms launches
    assert bench.settle_count(1 << 30) > bench.settle_count(8 << 30)
    assert bench.settle_count(1) == 400 and bench.settle_count(1 << 50) == 1
