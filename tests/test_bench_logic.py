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


class _StubShard:
    """What the verification functions read of a ShardBench, on CPU tensors."""

    def __init__(self, torch, data, lo, tape, width=3):
        self.dbuf = torch.from_numpy(data)
        self.n, self.lo, self.hi = data.size, lo, lo + data.size
        self.dtape = torch.from_numpy(tape.astype(np.int64))
        self.width, self.device = width, torch.device("cpu")


def test_torch_restatement_agrees_with_the_oracle_and_can_fail(bench, oracle):
    """bench.py's independent restatement of the definition (eq / cumsum / nonzero) must agree with the oracle on a
    correct tape — for both entering states, across its chunk boundaries — and must REJECT a tape with one entry
    changed, dropped or added."""
    import torch
    from conftest import random_csvish
    rng = np.random.default_rng(77)
    for inq in (0, 1):
        d = random_csvish(rng, 300_000, 0.05)
        lo = 12345
        want, q = oracle.scalar_index(d, base_off=lo, in_quote_in=inq)
        ok, n, state = bench.torch_reference_compare(_StubShard(torch, d, lo, want), inq, want.size)
        assert (ok, n, state) == (True, want.size, q)
        bad = want.copy()
        bad[want.size // 2] += 1
        assert bench.torch_reference_compare(_StubShard(torch, d, lo, bad), inq, bad.size)[0] is False
        assert bench.torch_reference_compare(_StubShard(torch, d, lo, want[:-1]), inq, want.size - 1)[0] is False
        extra = np.append(want, want[-1] + 1)
        assert bench.torch_reference_compare(_StubShard(torch, d, lo, extra), inq, extra.size)[0] is False
        # the wrong entering state is not the same tape
        assert bench.torch_reference_compare(_StubShard(torch, d, lo, want), inq ^ 1, want.size)[0] is False


def test_closed_form_check_can_fail(bench):
    import torch
    width, rows = 3, 1000
    data = np.frombuffer((b"abc," * 7 + b"abc\n") * rows, dtype=np.uint8).copy()
    pitch = width + 1
    lo = 8 * pitch * 5                                        # a shard that starts at a row boundary of a bigger file
    tape = np.arange(lo // pitch, (lo + data.size) // pitch, dtype=np.int64) * pitch + width
    sb = _StubShard(torch, data, lo, tape, width)
    assert bench.closed_form_compare(sb, tape.size) is True
    assert bench.closed_form_compare(sb, tape.size - 1) is False
    tape2 = tape.copy()
    tape2[17] += 4
    assert bench.closed_form_compare(_StubShard(torch, data, lo, tape2, width), tape2.size) is False


def test_cpu_baseline_variants_time_the_same_bytes(bench, pkg, oracle):
    """The CPU legs of the bench line (rank 0 at every N): every variant of BASELINE.md §2 runs on the same host
    sample and finds the same entries; the multi-threaded one states the cores it used."""
    cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
    n = pkg.workload_len("64x31_noquote", 8 << 20)
    host = oracle.synth(0, n, cols, width, seed, q)
    base = bench.cpu_baseline(oracle, host, budget_s=0.05)
    assert base["cores"] == 1 and base["kind"] == "port" and base["entries"] == n // (width + 1) + 1
    v = bench.cpu_baseline_variants(oracle, host, width, budget_scale=0.01)
    assert "native_build_error" not in v, v
    assert v["cpu_baseline_mt"]["cores"] == bench.cpu_threads() >= 1
    assert v["cpu_baseline_mt"]["entries"] == v["ref_sse_1t_native"]["entries"] == v["scalar_1t"]["entries"] == base["entries"]
    assert all(v[k]["value"] > 0 for k in ("cpu_baseline_mt", "ref_sse_1t_native", "scalar_1t"))


@pytest.mark.parametrize("n", [2, 4, 8])
def test_self_launch_command(bench, n):
    # `python bench.py --gpus N` (no launcher, the form the driver uses at N = 1) must become the driver's own N > 1
    # form: torch.distributed.run, one node, N ranks, rendezvous on 127.0.0.1, the same arguments
    argv = ["--gpus", str(n), "--steps", "7", "--warmup", "2"]
    cmd = bench.self_launch_command(n, argv, port=29123)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                       # the script, then exactly the caller's arguments
    assert all(not a.startswith("--gpus") for a in cmd[:i])   # --gpus belongs to bench.py, not to the launcher
    free = bench.self_launch_command(n, argv)        # no port given: a free one is picked
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


@pytest.mark.parametrize("n", [2, 4, 8])
def test_main_without_a_launcher_starts_the_ranks_as_a_child_before_any_gpu_call(bench, monkeypatch, n):
    import torch
    calls = []

    def no_gpu(*a, **k):
        raise AssertionError("the launching process must not touch the GPU")
    monkeypatch.setattr(torch.cuda, "is_available", no_gpu)
    monkeypatch.setattr(torch.cuda, "set_device", no_gpu)
    monkeypatch.setattr(bench.graft, "load_package", no_gpu)
    monkeypatch.setattr(bench, "self_launch", lambda k, argv: calls.append((k, list(argv))) or 7)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(n), "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                         # the child's exit code is this process's
    assert calls == [(n, ["--gpus", str(n), "--steps", "3"])]


def test_self_launch_relays_the_childs_exit_code(bench, monkeypatch):
    # the real subprocess path, with a child that is not torchrun: `python -c "exit 5"`
    monkeypatch.setattr(bench, "self_launch_command", lambda n, argv, port=None: [sys.executable, "-c", "raise SystemExit(5)"])
    assert bench.self_launch(2, []) == 5
