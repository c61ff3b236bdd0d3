#!/usr/bin/env python3
"""bench.py — stage-1 CSV structural indexing throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--gib-per-gpu G] [--skew B]

A "step" is one pass of the hot path over one batch of synthetic CSV already resident in HBM:
each rank indexes its contiguous byte shard of the file (first stage-1 launch through the C ABI: rank 0
knows how the file starts, every other rank lets the kernel choose the entering state its first eight tiles
speak for, CSVSIMD_ENTER_GUESS), the ranks exchange their shard descriptors with ONE all-gather over
RCCL (N > 1), a one-lane kernel stitches quote parity / tape bases ON THE DEVICE, and a second launch
re-emits the shard iff its first pass turns out to have used the wrong entering state (flag read from
device memory).  Nothing between the first
launch and the final copy-out waits for the host; the step ends when the tape and its length are
final on every rank.  Weak scaling: every rank always holds the same number of bytes (default 8 GiB =
one GPU's shard of BASELINE config 4, "64 GiB synthetic CSV, 64 cols, chunk-sharded 8xMI355X").

The timed path is VERIFIED at every N (the line carries "verified", the process exits non-zero on a
mismatch): each rank compares its whole tape shard, entry for entry, with an independent restatement of
the definition in plain torch ops on the same device bytes (and with the closed form on quote-free
corpora), compares 1 Mi entries either side of its shard boundaries and an order-sensitive checksum
with the CPU oracle, and rank 0 checks the stitched bases / totals / final state against the per-rank
truths.  On the default workload a second, untimed-for-`value` leg runs the QUOTED corpus cut mid-row
(--skew 777 semantics: shards start inside rows, some inside quoted fields) so that the re-emit path is
exercised and verified whenever this file runs with N > 1 ("q10_skew_check").

Timing hygiene: every timed region is preceded by 25 ms of the same, untimed, load (an idle MI355X runs its first
~10 ms of work through a power-management transient, see SETTLE_MS) in addition to the W warm-up steps; steps are
enqueued two deep over alternating tape buffers (step i + 1 is enqueued before step i's record is read back and
checked; all K records are checked inside the timed region, which is bracketed by barrier + synchronise).

Rank 0 prints ONE JSON line.  value = whole-job GiB/s (all ranks' bytes / max-over-ranks time).
roofline: HBM-bound, algorithmic bytes = 1 byte read per CSV byte scanned;
duration = the stage-1 kernel's average launch time from HIP events recorded on its own stream
(a launch is exactly one kernel), on every rank: frac is the slowest rank's.  The CPU beside it, on rank 0's host cores,
same bytes, same run, at EVERY N: cpu_baseline = the oracle's faithful SSE restatement of the reference loop
("ref_sse_1t": 1 thread like the reference, growing Vec) over a bounded sample; cpu_baseline_mt (chunked over the
stated number of threads), ref_sse_1t_native, scalar_1t.

Scaling: weak by default (every rank holds --gib-per-gpu bytes at every N).  --scaling strong cuts ONE --total-gib file
into one shard per rank; the default line also carries that leg (strong_scaling_check: the same 64 GiB file — BASELINE
config 4 — indexed and verified at this N: all of it on one GPU at N = 1, 8 GiB each at N = 8).

N = 1 only: other_workloads (the 1-GiB configs 2, 3, 5, timed and verified — config 5 by the library's dense instantiation,
which the context chooses from the corpus' density; an independent plain-copy yardstick next to the dense corpus'
bare-stream probe), batch_many_files (8 and 64 x 128 MiB: that many launches vs ONE batched launch), consumers (file +
tape -> columns in one pass, frequency count and search on a column, PMC traffic of the committed profile), latency (the
host-buffer drop-in on 300 B ... 32 MiB with one kept context, one host core beside it), ingest (the same entry point on
2 GiB, PCIe-inclusive, next to the probed H2D rate, with the call's per-thread phase times — never part of `value`).

`python bench.py --gpus N` (N > 1) without a launcher starts the N ranks itself: torch.distributed.run as a child process,
before this process has touched the GPU; the child's rank 0 prints the line.  Sharded steps keep three steps in flight:
the tail of a step (all-gather, stitch, re-emit, copy-out) runs on a second, high-priority stream beside the next step's
first pass.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable (copy)
WINDOW_ENTRIES = 1 << 20  # entries compared with the CPU oracle either side of every shard boundary


def build_if_needed():
    """Rank 0, before the process group's first barrier: the other ranks load the libraries only after it."""
    so = os.path.join(ROOT, "csv-simd_amd", "csrc", "libcsvsimd_hip.so")
    checker = os.path.join(ROOT, "oracle", "liboracle.so")   # the verification legs' CPU oracle
    if not (os.path.exists(so) and os.path.exists(checker)):
        graft.build()


def refuse_probe_environment(pkg):
    """The number this file prints must come from the product library and nothing else."""
    bad = sorted(k for k in os.environ if k.startswith("CSVSIMD_PROBE"))
    if bad:
        sys.exit(f"bench.py refuses to run with development probe variables set: {bad}")
    if os.environ.get("CSVSIMD_LIB"):
        sys.exit("bench.py refuses to run with CSVSIMD_LIB set: it times csv-simd_amd/csrc/libcsvsimd_hip.so only")
    if pkg.build_has_probes():
        sys.exit("bench.py refuses a library built with -DCSVSIMD_DEV_PROBES")


def self_launch_command(n_gpus, argv, port=None):
    """`python bench.py --gpus N` without a launcher: the command line of the child that runs this file under
    torch.distributed.run, one rank per GPU of this node (the same form the driver uses for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as s:      # a port nobody listens on right now
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Starts the N-rank job as a CHILD process and returns its exit code.  Called before this process has touched
    the GPU in any way (no torch.cuda call, no library load): on this pool a process that has initialised the GPU must
    not be replaced, and need not be — the child's rank 0 prints the JSON line on the stdout it inherits."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")               # what torchrun would set (and warn about) anyway
    return subprocess.run(self_launch_command(n_gpus, argv), env=env).returncode


PIPELINE_DEPTH = 2   # steps in flight: step i + 1 is enqueued before step i's records are read (see run_sharded)
# sharded steps: the tail of step i (all-gather, stitch, re-emit, copy-out) runs on a second stream and completes while
# step i + 1's stage-1 kernel drains, by which time step i + 2 must already be enqueued: three in flight
PIPELINE_DEPTH_SHARDED = 3


class ShardBench:
    """One rank's shard: device buffer, tape shard, context.  Shards are contiguous byte ranges of one
    file of world * n bytes (n = whole rows); with skew every interior cut moves `skew` bytes to the right,
    i.e. into the middle of a row ("deliberately misaligned variant")."""

    def __init__(self, pkg, device, workload, shard_bytes, rank, world, skew=0, cuts=None, depth=PIPELINE_DEPTH):
        self.pkg, self.device, self.workload = pkg, device, workload
        cols, width, seed, q = pkg.WORKLOADS[workload]
        self.cols, self.width, self.seed, self.q = cols, width, seed, q
        self.row = cols * (width + 1)
        per = (shard_bytes // self.row) * self.row
        self.total = world * per
        if cuts is None:
            cuts = [0] + [i * per + skew for i in range(1, world)] + [self.total]
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == self.total
        self.lo, self.hi = cuts[rank], cuts[rank + 1]
        self.n = self.hi - self.lo
        self.rank, self.world, self.skew = rank, world, skew
        self.dbuf = torch.empty(self.n, dtype=torch.uint8, device=device)
        pkg.synth_fill_device(self.dbuf.data_ptr(), self.lo, self.n, cols, width, seed, q)
        # the bench sizes the tape from the known shape: no retry, no count pre-pass in a step (a quoted
        # corpus holds up to three comma/LF bytes per quoted field; a wrong speculation may count them all)
        self.cap = int(self.n // (width + 1) * (1.25 if q else 1.0)) + 1024
        # two tape buffers: consecutive steps alternate between them, as a caller that indexes batch after batch
        # would, so that a step can be enqueued while the previous step's tape is still being consumed
        # (three for sharded steps, whose tails complete a step late: PIPELINE_DEPTH_SHARDED)
        self.dtapes = [torch.empty(self.cap, dtype=torch.int64, device=device) for _ in range(depth)]
        self.dtape = self.dtapes[0]                       # what launch() writes and the verification reads
        self.d_results = torch.zeros(depth, 8, dtype=torch.int64, device=device)
        self.h_results = torch.zeros(depth, 8, dtype=torch.int64).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.d_result, self.h_result = self.d_results[0], self.h_results[0]
        self.h_words = self.h_results.numpy()             # the same pinned bytes, cheap to read per step
        self.ctx = pkg.Context(device.index)
        self.ctx.reserve(self.n)
        # The timed launches are asynchronous: no record reaches the host before the next one is enqueued, and an asynchronous
        # launch never looks at the data first.  So the context gets to know its data the way a caller's would: ONE
        # synchronous call up front (untimed) — a fresh context samples sixteen 64-KiB windows of the buffer before that
        # call's launch and keeps what the call's record says; above ~0.125 entries per byte (the 1024 x 4 corpus: 0.2) every
        # later launch runs the dense instantiation.  Nothing about the generator's shape is passed to the library
        # (VERDICT r4 next #6); the bench only sizes its own tape from it.
        torch.cuda.synchronize(device)
        self.ctx.stage1_index_device(self.dbuf.data_ptr(), self.n, self.lo, 0, self.dtape.data_ptr(), self.cap, allow_overflow=True)
        self.ctx_tail = None    # sharded steps with an overlapped tail: the re-emit launch's own context (see reemit)
        torch.cuda.synchronize(device)

    def tail_context(self):
        """The re-emit launch of step i runs beside the first pass of step i + 1: it needs its own scratch (look-back
        words, ticket, epoch), i.e. its own context."""
        if self.ctx_tail is None:
            self.ctx_tail = self.pkg.Context(self.device.index)
            self.ctx_tail.reserve(self.n)
            # (the tail context only ever re-emits this shard: it is told what the first context LEARNED about it)
            if "csvsimd_dense" in self.ctx.kernel_name():
                self.ctx_tail.hint_density(1, 2)
            else:
                self.ctx_tail.hint_density(1, 1000)
        return self.ctx_tail

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def release(self):
        """Gives the device memory back (the legs of a run recycle HBM one after the other)."""
        torch.cuda.synchronize(self.device)
        self.dbuf = self.dtape = self.dtapes = None
        if self.ctx_tail is not None:
            self.ctx_tail.close()
            self.ctx_tail = None
        torch.cuda.empty_cache()

    def launch(self, in_quote_in, d_result=None):
        """Enqueue stage 1 over this rank's shard (asynchronous; result record -> d_result)."""
        d_result = self.d_result if d_result is None else d_result
        self.ctx.stage1_index_device_async(self.dbuf.data_ptr(), self.n, self.lo, in_quote_in,
                                           self.dtape.data_ptr(), self.cap, d_result.data_ptr(), self.stream())

    def reemit(self, d_stitch_ptr, d_result, ctx=None):
        (ctx or self.ctx).stage1_reemit_device_async(self.dbuf.data_ptr(), self.n, self.lo, d_stitch_ptr,
                                                     self.dtape.data_ptr(), self.cap, d_result.data_ptr(), self.stream())

    def check(self, r):
        if r.error or r.count > self.cap:
            raise RuntimeError(f"stage1 failed: error={r.error} count={r.count} cap={self.cap}")
        return r

    def use_slot(self, slot):
        """Steps alternate between the tape buffers; launch() / reemit() / the verification use the current one."""
        self.dtape = self.dtapes[slot]

    def enqueue_pass(self, in_quote_in, slot=0):
        """Single-GPU step, first half: launch into tape buffer `slot`, request the record's copy-out."""
        self.use_slot(slot)
        self.launch(in_quote_in, self.d_results[slot])
        self.h_results[slot].copy_(self.d_results[slot], non_blocking=True)
        self.events[slot].record(torch.cuda.current_stream(self.device))

    def collect_pass(self, slot=0):
        """Second half: wait for that step (its only synchronisation) and check its record."""
        from csv_simd_amd import sharded
        self.events[slot].synchronize()
        w = self.h_words[slot]
        if (int(w[4]) & 0xFFFFFFFF) or int(w[0]) > self.cap:   # error flag / more entries than the tape holds
            return self.check(sharded.result_from_words(self.h_results[slot].tolist()))
        return w

    def run_pass(self, in_quote_in):
        """Single-GPU step, unpipelined: launch, then read the result record back."""
        self.enqueue_pass(in_quote_in, 0)
        return self.collect_pass(0)


# ------------------------------------------------------------------------------------------------
# verification of the timed path (never inside the timed region)
# ------------------------------------------------------------------------------------------------
def torch_reference_compare(sb, in_quote_in, count):
    """Entry-for-entry comparison of the whole tape shard with the DEFINITION restated in torch ops on the
    same device bytes: byte i is structural iff it is ',', CR or LF and the number of '"' at positions <= i
    (plus the entering state) is even (src/avx/stage1.rs:342-407).  Independent of the HIP
    kernels (eq / cumsum / nonzero), chunked so that it runs at 8 GiB.  Returns (ok, entries, state_out)."""
    chunk = 256 << 20
    state, pos, ok = int(in_quote_in), 0, True
    for c0 in range(0, sb.n, chunk):
        x = sb.dbuf[c0: c0 + chunk]
        isq = x == 0x22
        par = torch.cumsum(isq, 0, dtype=torch.int32)
        par &= 1
        if state:
            par ^= 1
        st = ((x == 0x2C) | (x == 0x0A) | (x == 0x0D)) & (par == 0)
        idx = torch.nonzero(st).flatten()
        idx += sb.lo + c0
        k = idx.numel()
        if pos + k > count or not torch.equal(sb.dtape[pos: pos + k], idx):
            ok = False
        pos += k
        state ^= int(isq.sum().item()) & 1
        del x, isq, par, st, idx
    return ok and pos == count, pos, state


def closed_form_compare(sb, count):
    """Quote-free corpora: entry k of the file is the byte k * (width + 1) + width."""
    pitch = sb.width + 1
    k0, k1 = sb.lo // pitch, sb.hi // pitch
    if count != k1 - k0:
        return False
    ok, step = True, 64 << 20
    for a in range(k0, k1, step):
        b = min(k1, a + step)
        want = torch.arange(a, b, dtype=torch.int64, device=sb.device) * pitch + sb.width
        ok = ok and torch.equal(sb.dtape[a - k0: b - k0], want)
    return ok


def oracle_windows_compare(oracle, sb, in_quote_in, count):
    """The CPU oracle on the head and the tail of the shard: WINDOW_ENTRIES entries either side of the
    shard boundaries, entry for entry, plus the order-sensitive checksum of the head window computed by
    the device checksum kernel and by the oracle.  The entering state of the tail window is derived on
    the CPU from the synthetic corpus itself (a row start is never inside a quoted field)."""
    pkg = sb.pkg
    per_entry = sb.width + 1
    wbytes = min(sb.n, (WINDOW_ENTRIES + 4096) * per_entry)
    head = sb.dbuf[:wbytes].cpu().numpy()
    want, _ = oracle.scalar_index(head, base_off=sb.lo, in_quote_in=in_quote_in)
    k = min(want.size, WINDOW_ENTRIES, count)
    got = sb.dtape[:k].cpu().numpy().view(np.uint64)
    ok = bool(np.array_equal(got, want[:k]))
    # the same bytes regenerated by the CPU generator: the device corpus is the file both sides mean
    small = min(wbytes, 1 << 20)
    ok = ok and bool(np.array_equal(head[:small], oracle.synth(sb.lo, small, sb.cols, sb.width, sb.seed, sb.q)))
    out = torch.zeros(2, dtype=torch.int64, device=sb.device)
    pkg.tape_checksum_device(sb.dtape.data_ptr(), k, 1, out.data_ptr(), sb.stream())
    ok = ok and tuple(int(v) & (2**64 - 1) for v in out.cpu().tolist()) == oracle.tape_checksum(want[:k], 1)
    # tail: start at the last row boundary that leaves >= wbytes before the end of the shard
    t0 = ((sb.hi - wbytes) // sb.row) * sb.row
    if t0 <= sb.lo:
        return ok
    tail = sb.dbuf[t0 - sb.lo:].cpu().numpy()
    want_t, _ = oracle.scalar_index(tail, base_off=t0, in_quote_in=0)
    kt = min(want_t.size, WINDOW_ENTRIES, count)
    got_t = sb.dtape[count - kt: count].cpu().numpy().view(np.uint64)
    return ok and bool(np.array_equal(got_t, want_t[want_t.size - kt:]))


def true_entering_state(oracle, sb):
    """In-quote state at byte `lo` of the synthetic file, from the CPU generator alone."""
    r0 = (sb.lo // sb.row) * sb.row
    if r0 == sb.lo or not sb.q:
        return 0
    part = oracle.synth(r0, sb.lo - r0, sb.cols, sb.width, sb.seed, sb.q)
    return int(np.count_nonzero(part == 0x22)) & 1


def verify_rank(oracle, sb, st_in_quote_in, count, tape_index_base, total_entries, in_quote_final, reemit=0):
    """Everything one rank can check about its own shard.  Returns a dict of booleans + facts."""
    t = sb.dtape[:count]
    truth_in = true_entering_state(oracle, sb)
    ok_ref, ref_count, state_out = torch_reference_compare(sb, truth_in, count)
    res = {
        "entering_state_matches_generator": truth_in == int(st_in_quote_in),
        "tape_equals_torch_reference": bool(ok_ref),
        "ascending_in_range": bool(count == 0 or ((t[1:] > t[:-1]).all() and int(t[0]) >= sb.lo and int(t[-1]) < sb.hi)),
        "oracle_windows": bool(oracle_windows_compare(oracle, sb, truth_in, count)),
    }
    if not sb.q:
        res["closed_form"] = bool(closed_form_compare(sb, count))
    facts = {"rank": sb.rank, "lo": sb.lo, "hi": sb.hi, "count": int(count), "ref_count": int(ref_count),
             "state_in": truth_in, "state_out": int(state_out), "base": int(tape_index_base),
             "total": int(total_entries), "final": int(in_quote_final), "reemit": int(reemit)}
    return res, facts


def verify_job(all_facts):
    """Rank 0: the stitched bases, totals and states against the per-rank truths."""
    all_facts = sorted(all_facts, key=lambda f: f["rank"])
    base, state, ok = 1, 0, True
    for f in all_facts:
        ok = ok and f["base"] == base and f["state_in"] == state and f["count"] == f["ref_count"]
        base += f["ref_count"]
        state = f["state_out"]
    ok = ok and all(f["total"] == base and f["final"] == state for f in all_facts)
    for a, b in zip(all_facts, all_facts[1:]):
        ok = ok and a["hi"] == b["lo"]
    return ok, sum(f["reemit"] for f in all_facts), base, sum(f["state_in"] for f in all_facts)


def mid_row_cuts(oracle, pkg, workload, shard_bytes, world):
    """Shard cuts for the quoted corpus that are GUARANTEED to exercise both outcomes of the stitch: every
    interior cut lies in the middle of a row; odd cuts lie inside a quoted field (that rank's speculation is
    wrong and it re-emits), even cuts 777 bytes into the row or the next byte outside a quoted field (SURVEY
    §8d's "+777").  Computed from the CPU generator, identically on every rank."""
    cols, width, seed, q = pkg.WORKLOADS[workload]
    row = cols * (width + 1)
    per = (shard_bytes // row) * row
    cuts = [0]
    for i in range(1, world):
        r0 = i * per
        inside = np.zeros(row, dtype=bool)
        k = 0
        while not inside.any():                      # a row without a quoted field: take the next one
            line = oracle.synth(r0 + k * row, row, cols, width, seed, q)
            inside = (np.cumsum(line == 0x22) & 1).astype(bool)   # inclusive prefix parity: in-string bytes
            k += 1
        r0 += (k - 1) * row
        if i % 2 == 1:
            off = int(np.flatnonzero(inside)[min(5, int(inside.sum()) - 1)])
        else:
            off = 777
            while inside[off - 1]:                   # the state entering byte `off` is that of byte off - 1
                off += 1
        cuts.append(r0 + off)
    return cuts + [world * per]


# An MI355X that has been idle runs its first ~10 ms of load through a power-management transient: launches issued 2-10 ms
# after the load starts take 5-8 % longer than the ones before and after (scripts/per_launch_times.py: 8 GiB launches
# 1.66 1.71 1.80 1.76 1.70 then 1.645 flat; 1 GiB launches 0.225 x9, 0.25 x25, then 0.226 flat).  Every timed region
# below is therefore preceded by SETTLE_MS of the same, untimed, load — in addition to the W warm-up steps.
SETTLE_MS = 25.0


def settle_count(bytes_per_launch, cap=400):
    """Launches that make up SETTLE_MS of load; from the byte count alone, so every rank gets the same number."""
    return max(1, min(cap, int(math.ceil(SETTLE_MS * 1e-3 / (bytes_per_launch / 4.5e12)))))


def kernel_time_ms(sb, iters):
    """Average duration of `iters` back-to-back launches of the product kernel (HIP events on its stream), in steady
    state (see SETTLE_MS)."""
    return sb.ctx.stage1_time_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), sb.cap, sb.d_result.data_ptr(),
                                     sb.stream(), warmup=max(2, settle_count(sb.n)), iters=iters)


def time_steps(step, steps, warmup, device, dist_on, settle=0, drain=None):
    """`drain` (pipelined steps): collects whatever is still in flight — before the clock starts and, for the timed
    steps, before it stops: all K steps are complete, and their records checked, inside the timed region."""
    import torch.distributed as dist
    for _ in range(settle + warmup):
        step()
    if drain:
        drain()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if drain:
        drain()
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def load_traffic(workload, launch_bytes):
    """HBM bytes per launch from the committed PMC profile (separate --pmc passes), if it was taken
    on this workload at this size; None otherwise."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                e = json.load(f).get(workload)
            if e and e.get("launch_bytes") == launch_bytes:
                return e.get("hbm_bytes")
        except Exception:
            return None
    return None


def host_sample(sb, sample_bytes):
    """The first `sample_bytes` of this rank's shard, copied back from HBM into a 64-byte-aligned host array:
    the SAME bytes the GPU scans, for the CPU baselines and the ingest leg."""
    import __graft_entry__ as g
    n = min(sb.n, sample_bytes)
    return g.load_oracle().aligned_copy(sb.dbuf[:n].cpu().numpy())


def cpu_threads():
    """Threads the multi-threaded CPU variant uses: the cores this process may run on, at most 16 per GPU of the
    job (a GPU box of this pool gives a container 16 of the host's hardware threads per GPU; neither
    os.cpu_count() nor the affinity mask shows that share, both report the whole host)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(oracle, host, budget_s=10.0):
    """ref_sse_1t on a bounded sample of the SAME bytes (copied back from HBM): the oracle's faithful SSE
    restatement of reference reader::read (src/reader.rs:229-290), growing Vec, ONE thread like the reference."""
    n = host.size
    oracle.sse_read_growing_timed(host[: 1 << 24])  # warm the code path / page in
    times, entries = [], 0
    t_total = 0.0
    while len(times) < 3 or (t_total < budget_s and len(times) < 64):   # ~10 s of CPU work
        entries, dt = oracle.sse_read_growing_timed(host)
        times.append(dt)
        t_total += dt
    best, median = min(times), sorted(times)[len(times) // 2]
    return {"value": round(n / best / 2**30, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
            "variant": "ref_sse_1t (SSE restatement of reference reader::read, growing Vec, 1 thread)",
            "sample": f"first {n / 2**30:.2f} GiB of rank 0's shard, best of {len(times)} passes "
                      f"({t_total:.1f} s of CPU work; median {n / median / 2**30:.2f} GiB/s)",
            "entries": entries, "host_cpus": os.cpu_count()}


def cpu_baseline_variants(oracle, host, width, budget_scale=1.0):
    """The other three CPU variants of BASELINE.md §2 on the same sample, in the same run (VERDICT r2 missing #4):
    ref_sse_mt (chunked over `cores` threads with a quote-parity / count stitch, -march=native), ref_sse_1t_native
    (same loop, -march=native, pre-reserved output) and scalar_1t (the byte loop).  liboracle_native.so is built here,
    for THIS host, from oracle/oracle.c (test infrastructure: the checker's own source with other compiler flags)."""
    import ctypes as C
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    n = host.size
    tape = np.zeros(n // (width + 1) + 4096, dtype=np.uint64)
    cnt = C.c_uint64()
    u64p = C.POINTER(C.c_uint64)
    out = {}

    def best_of(fn, budget_s, min_reps=2, max_reps=32):
        times, t_total = [], 0.0
        while len(times) < min_reps or (t_total < budget_s and len(times) < max_reps):
            t0 = time.perf_counter()
            rc = fn()
            dt = time.perf_counter() - t0
            assert rc == 0
            times.append(dt)
            t_total += dt
        return min(times), len(times), t_total

    try:
        # -B: always rebuilt — the file is -march=native code, and a copy built on another machine (the repository travels
        # with its built .so files) could hold instructions this host's CPU does not have
        subprocess.run(["make", "-s", "-B", "-C", odir, "liboracle_native.so"], check=True, capture_output=True, timeout=120)
        nat = C.CDLL(os.path.join(odir, "liboracle_native.so"))
        nat.oracle_sse_read.restype = C.c_int
        nat.oracle_sse_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        nat.oracle_sse_read_mt.restype = C.c_int
        nat.oracle_sse_read_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, u64p]
        threads = cpu_threads()
        b, k, tt = best_of(lambda: nat.oracle_sse_read_mt(host.ctypes.data, n, threads, tape.ctypes.data, tape.size,
                                                          C.byref(cnt)), 4.0 * budget_scale)
        entries_mt = int(cnt.value)
        out["cpu_baseline_mt"] = {
            "value": round(n / b / 2**30, 3), "unit": "GiB/s", "cores": threads, "kind": "port",
            "variant": f"ref_sse_mt (the same SSE block loop on {threads} contiguous chunks, quote-parity / count "
                       "stitch, -march=native; NOT a reference behaviour: the reference is single-threaded)",
            "sample": f"first {n / 2**30:.2f} GiB of rank 0's shard, best of {k} passes ({tt:.1f} s)",
            "entries": entries_mt, "host_cpus": os.cpu_count()}
        b, k, tt = best_of(lambda: nat.oracle_sse_read(host.ctypes.data, n, tape.ctypes.data, tape.size, C.byref(cnt)), 4.0 * budget_scale)
        out["ref_sse_1t_native"] = {"value": round(n / b / 2**30, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
                                    "variant": "ref_sse_1t_native (-march=native, output pre-reserved)",
                                    "sample": f"the same {n / 2**30:.2f} GiB, best of {k}", "entries": int(cnt.value)}
    except Exception as e:   # no compiler on the host: the portable variants below still run
        out["native_build_error"] = repr(e)[:200]
    base = oracle.lib()
    m = min(n, 256 << 20)   # the byte loop runs at < 1 GiB/s: a quarter GiB is seconds
    b, k, tt = best_of(lambda: base.oracle_scalar_read(host.ctypes.data, m, tape.ctypes.data, tape.size, C.byref(cnt)), 2.0 * budget_scale)
    out["scalar_1t"] = {"value": round(m / b / 2**30, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
                        "variant": "scalar_1t (byte-at-a-time definition)",
                        "sample": f"first {m / 2**30:.2f} GiB, best of {k}", "entries": int(cnt.value)}
    return out


def ingest_leg(pkg, device, host, width, closed_form):
    """The host-buffer drop-in for reader::read (csvsimd_stage1_index: pageable host bytes in, host tape
    out) on a bounded sample of the same corpus, next to what this box's PCIe link moves host -> device
    (one pinned hipMemcpy).  PCIe-inclusive by construction; reported beside `value`, never in it."""
    n = host.size
    tape = np.empty(n // (width + 1) + 64, dtype=np.uint64)
    ctx = pkg.Context(device.index)
    rc, tl, _ = ctx.read_into(host[: 64 << 20], tape)      # allocates the pipeline, pages everything in
    best, ok, phases = None, rc == 0, None
    for _ in range(3):
        t0 = time.perf_counter()
        rc, tl, _ = ctx.read_into(host, tape)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, phases = dt, pkg.ingest_last_phases()     # where that call's wall time went, per pipeline thread
        ok = ok and rc == 0
    pitch = width + 1
    if closed_form:
        want = np.arange(n // pitch, dtype=np.uint64) * pitch + width
        ok = ok and tl == want.size + 1 and tape[0] == 0 and bool(np.array_equal(tape[1:tl], want))
    ctx.close()
    # probed link rate: pinned host -> device, the same number of bytes, best of 4
    pin = torch.empty(min(n, 1 << 30), dtype=torch.uint8).pin_memory()
    dst = torch.empty_like(pin, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h2d = None
    for _ in range(4):
        e0.record()
        dst.copy_(pin, non_blocking=True)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        h2d = ms if h2d is None else min(h2d, ms)
    h2d_gib = pin.numel() / (h2d * 1e-3) / 2**30
    gib = n / best / 2**30
    return {"value": round(gib, 2), "unit": "GiB/s", "bytes": n, "tape_entries": int(tl), "verified": bool(ok),
            "h2d_probe_GiB_s": round(h2d_gib, 2), "frac_of_h2d_probe": round(gib / h2d_gib, 3),
            "phases_ms_of_the_best_call": {k: (round(v * 1e3, 3) if isinstance(v, float) else v) for k, v in phases.items()},
            "note": "csvsimd_stage1_index: pageable host buffer -> pinned staging (a stager thread, up to three chunks ahead) "
                    "-> H2D -> kernel (chunks chained on the device: no host round trip between them) -> tape to a pinned "
                    "slot -> caller's tape (an expander thread); PCIe-inclusive, never part of `value`"}


def file_ingest_leg(pkg, device, host, width, closed_form, buffer_ingest):
    """csv_simd::create(filename) end to end — the reference's actual entry point (src/lib.rs:61-74: open, mmap,
    Header::new, reader::read, TapeCore::create) — on the SAME 2 GiB the `ingest` leg reads from a buffer, written to a file
    first (page cache hot: the link, not the disk, is what this measures), with a tape the library allocates itself.
    csvsimd_create maps the file, reads the header, runs the ingest pipeline on the mapping (the stager's slices fault the
    page-cache pages in, in parallel, ahead of the link) into an index block that is address space until written, and builds
    the tape object.  Reported next to the buffer ingest of the same bytes; never part of `value`."""
    import tempfile
    n = host.size
    d = os.environ.get("CSVSIMD_BENCH_TMPDIR") or tempfile.gettempdir()
    path = os.path.join(d, f"csvsimd_bench_{os.getpid()}.csv")
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        f.write(host.data)
    write_s = time.perf_counter() - t0
    out = {"bytes": n, "file": path, "write_s": round(write_s, 2)}
    try:
        ctx = pkg.Context(device.index)
        t0 = time.perf_counter()
        tp = ctx.create(path)                       # a fresh context: pipeline slots are pinned here
        first_ms = (time.perf_counter() - t0) * 1e3
        pitch = width + 1
        ok = True
        if closed_form:
            idx = tp.index_view()
            ok = bool(idx.size == n // pitch + 1 and idx[0] == 0 and
                      np.array_equal(idx[1:], np.arange(n // pitch, dtype=np.uint64) * pitch + width))
            del idx
        fields, records = tp.field_cnt, tp.record_cnt
        tp.close()
        best, phases, destroy_ms = None, None, None
        for _ in range(4):
            t0 = time.perf_counter()
            tp = ctx.create(path)
            dt = time.perf_counter() - t0
            ph = pkg.ingest_last_phases()
            ok = ok and tp.record_cnt == records
            t1 = time.perf_counter()
            tp.close()
            dd = (time.perf_counter() - t1) * 1e3
            if best is None or dt < best:
                best, phases, destroy_ms = dt, ph, dd
        ctx.close()
        gib = n / best / 2**30
        out.update({"value": round(gib, 2), "unit": "GiB/s", "ms": round(best * 1e3, 2), "verified": bool(ok),
                    "first_call_ms_fresh_context": round(first_ms, 1), "tape_destroy_ms": round(destroy_ms, 1),
                    "fields": fields, "records": records,
                    "buffer_ingest_GiB_s": buffer_ingest["value"],
                    "frac_of_buffer_ingest": round(gib / buffer_ingest["value"], 3),
                    "frac_of_h2d_probe": round(gib / buffer_ingest["h2d_probe_GiB_s"], 3),
                    "phases_ms_of_the_best_call": {k: (round(v * 1e3, 3) if isinstance(v, float) else v) for k, v in phases.items()},
                    "note": "csvsimd_create(path): open + mmap + Header::new + the ingest pipeline on the mapping + a tape whose "
                            "index the library allocates (address space until written) + Tape::from_core; best of 4 on a kept "
                            "context, the file's pages in the page cache, every call with a NEW index block"})
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    return out


def small_files_leg(pkg, oracle, device, n_files=10_000):
    """Many small HOST files in one call (csvsimd_stage1_index_batch): 10 000 files of ~4 KiB of the quoted 16x32 corpus
    (whole rows), host bytes in, host tapes out, next to (a) the same files through csvsimd_stage1_index one by one on a kept
    context and (b) ref_sse_1t over the same files on one host core (the reference's way: csv_simd::create per file,
    src/lib.rs:61-74; a Vec grown from [0] per file).  Every tape is checked against the oracle."""
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    row = cols * (width + 1)
    per = (4096 // row) * row
    whole = oracle.synth(0, per * n_files, cols, width, seed, q)
    files = [whole[i * per: (i + 1) * per] for i in range(n_files)]
    total = per * n_files
    tapes = [np.empty(per // 8 + 64, dtype=np.uint64) for _ in files]
    items = (pkg.HostBatchItem * n_files)()
    for it, a, t in zip(items, files, tapes):
        it.buf, it.len, it.tape, it.tape_cap = a.ctypes.data, a.size, t.ctypes.data, t.size
    ctx = pkg.Context(device.index)
    rc = ctx.read_many_into(items)                  # allocates the pipeline
    ok = rc == 0
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        rc = ctx.read_many_into(items)
        ts.append(time.perf_counter() - t0)
        ok = ok and rc == 0
    counts, _ = oracle.sse_read_growing_many_timed(files)
    cpu = min(oracle.sse_read_growing_many_timed(files)[1] for _ in range(5))
    for i in range(n_files):
        ok = ok and items[i].status == 0 and items[i].tape_len == counts[i]
    for i in list(range(0, n_files, 97)) + [n_files - 1]:
        ok = ok and bool(np.array_equal(tapes[i][: items[i].tape_len], oracle.sse_read(files[i])))
    # one by one through the single-file entry point (a sample: 1 000 files)
    k = min(1000, n_files)
    t0 = time.perf_counter()
    for i in range(k):
        ctx.read_into(files[i], tapes[i])
    one_by_one = (time.perf_counter() - t0) / k
    ctx.close()
    best = min(ts)
    gib = total / best / 2**30
    cpu_gib = total / cpu / 2**30
    return {"files": n_files, "bytes_per_file": per, "bytes": total, "ms": round(best * 1e3, 3),
            "ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 3), "GiB/s": round(gib, 2),
            "us_per_file": round(best / n_files * 1e6, 3),
            "cpu_ref_sse_1t": {"GiB/s": round(cpu_gib, 2), "ms": round(cpu * 1e3, 3), "us_per_file": round(cpu / n_files * 1e6, 3),
                               "cores": 1, "kind": "port",
                               "variant": "oracle_sse_read_growing_many: the SSE restatement of reader::read per file, a Vec "
                                          "grown from [0] per file, one thread"},
            "gpu_over_cpu": round(gib / cpu_gib, 2),
            "one_call_per_file_us": round(one_by_one * 1e6, 1),
            "batch_over_one_call_per_file": round(one_by_one * n_files / best, 1),
            "verified": bool(ok),
            "note": "csvsimd_stage1_index_batch: files packed into pinned groups (256 KiB ... 4 MiB), one H2D copy and ONE "
                    "batched launch per group, tapes and records written to pinned memory by the kernel, three host "
                    "threads; wall time of the call, host bytes in, host tapes out"}


def latency_leg(pkg, oracle, device):
    """The drop-in on SMALL inputs (the reference's only real inputs are 96-623 bytes: res/*.csv, src/lib.rs:52-74):
    end-to-end wall time of csvsimd_stage1_index — host bytes in, host tape out — with ONE context kept across calls
    (rust/reader_hip.rs holds it per thread), next to ref_sse_1t (the oracle's restatement of reader::read, one thread,
    growing Vec) on the same bytes on this host, and the size where the GPU path starts to win.  Up to 1 MiB the call is
    one launch over pinned, GPU-mapped memory (no copy engine); above, the chunked pipeline."""
    t0 = time.perf_counter()
    ctx = pkg.Context(device.index)
    create_ms = (time.perf_counter() - t0) * 1e3
    fixture = os.path.join(ROOT, "tests", "golden", "sample.csv")       # the reference's res/sample.csv (300 bytes)
    cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
    cases = [("res/sample.csv", np.frombuffer(open(fixture, "rb").read(), dtype=np.uint8))]
    for nbytes in (4 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 32 << 20):
        cases.append((f"16x32_q10 {nbytes >> 10} KiB", oracle.synth(0, nbytes, cols, width, seed, q)))
    rows, ok, first_us = [], True, None
    for name, data in cases:
        host = oracle.aligned_copy(data)
        tape = np.zeros(host.size // 8 + 64, dtype=np.uint64)
        t0 = time.perf_counter()
        rc, tl, _ = ctx.read_into(host, tape)                             # first call at this size: allocations included
        first = (time.perf_counter() - t0) * 1e6
        first_us = first if first_us is None else first_us
        want = oracle.sse_read(host)
        ok = ok and rc == 0 and tl == want.size and bool(np.array_equal(tape[:tl], want))
        reps = 300 if host.size <= (1 << 20) else 30
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            rc, tl, _ = ctx.read_into(host, tape)
            ts.append(time.perf_counter() - t0)
            ok = ok and rc == 0
        cs = []
        for _ in range(min(reps, 100)):
            _, dt = oracle.sse_read_growing_timed(host)
            cs.append(dt)
        g_best, g_med, c_best = min(ts) * 1e6, sorted(ts)[len(ts) // 2] * 1e6, min(cs) * 1e6
        rows.append({"input": name, "bytes": int(host.size), "gpu_us_best": round(g_best, 1), "gpu_us_median": round(g_med, 1),
                     "first_call_us": round(first, 1), "cpu_ref_sse_1t_us_best": round(c_best, 1),
                     "gpu_over_cpu": round(g_best / c_best, 2)})
    ctx.close()
    cross = next((r for r in rows if r["gpu_us_best"] < r["cpu_ref_sse_1t_us_best"]), None)
    below = [r for r in rows if cross and r["bytes"] < cross["bytes"]]
    return {"context_create_ms": round(create_ms, 2), "first_read_us": round(first_us, 1), "sizes": rows,
            "crossover": (f"the GPU path is faster from {cross['bytes']} bytes up" +
                          (f" (slower at {below[-1]['bytes']} bytes and below)" if below else "")) if cross
                         else "the CPU path is faster at every size measured",
            "verified": bool(ok),
            "note": "wall time of csvsimd_stage1_index (pageable host bytes -> host tape) with a context kept across calls; "
                    "cpu = the oracle's ref_sse_1t on the same bytes, same host, one thread like the reference"}


def consumer_traffic(key):
    """HBM bytes per launch of a consumer kernel from the committed PMC profile (separate --pmc passes over
    `bench.py --only-consumers`, scripts/collect_profiles_consumers.sh), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("consumers", {}).get(key)
    except Exception:
        return None


def consumers_leg(pkg, oracle, device):
    """§8f rank 3 — the reference's stated use of the tape ("frequency counts, and function search",
    design_notes_1.md:1-4) on a device-resident tape of the 16x32 corpus (1 GiB, 2.03 M records).  ONE pass turns the
    row-major file + tape into columns (csvsimd_chunk_to_columns_device: every byte of the file and of the tape read
    once, all 16 columns written); frequency count and search then run on a column with contiguous 16-byte loads.
    Wall time of the calls incl. their synchronisation, best of 3; algorithmic bytes next to the PMC traffic of the
    committed profile.  The per-column path of round 2 (a column of the ROW-MAJOR file per call) is timed beside it."""
    name = "16x32_noquote"
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 1 << 30)
    dbytes = torch.empty(n, dtype=torch.uint8, device=device)
    pkg.synth_fill_device(dbytes.data_ptr(), 0, n, cols, width, seed, q)
    entries = n // (width + 1)
    dindex = torch.zeros(entries + 2, dtype=torch.int64, device=device)
    ctx = pkg.Context(device.index)
    r = ctx.stage1_index_device(dbytes.data_ptr(), n, 0, 0, dindex.data_ptr() + 8, entries + 1)
    index_len, rows = r.count + 1, r.count // cols
    nrec, field = rows - 1, 5
    whole = (0, cols, rows * cols, nrec)
    args = (dbytes.data_ptr(), dindex.data_ptr(), index_len, cols, "LF")

    def best(fn, reps=3):
        t = None
        for _ in range(reps):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            t = dt if t is None else min(t, dt)
        return t, out

    res = {"workload": f"{name} 1 GiB: {nrec} records x {cols} columns ({width}-byte fields)"}
    stride = 32
    # ---- the whole file -> 16 columns, one pass ---------------------------------------------------------------------
    ccols = torch.empty((cols, nrec, stride), dtype=torch.uint8, device=device)
    clens = torch.empty((cols, nrec), dtype=torch.int32, device=device)
    to_cols = lambda: pkg.chunk_to_columns_device(ctx, dbytes.data_ptr(), n, args[1], index_len, cols, "LF", whole, None,
                                                  ccols.data_ptr(), stride, clens.data_ptr())
    t_wall, _ = best(to_cols, reps=5)
    # the call is asynchronous: its device time = torch events on the launch stream around 10 back-to-back calls
    t = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            to_cols()
        e1.record()
        e1.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / 10
        t = dt if t is None else min(t, dt)
    table = dbytes[: rows * cols * (width + 1)].view(rows, cols, width + 1)[1:, :, :width]
    ok = bool((clens == width).all()) and torch.equal(ccols, table.permute(1, 0, 2).contiguous())
    del table
    alg_read = (n - cols * (width + 1)) + 8 * (index_len - cols)          # the data rows' bytes + their tape entries
    alg_write = cols * nrec * (stride + 4)
    traffic = consumer_traffic("to_columns_kernel")
    res["to_columns"] = {"ms": round(t * 1e3, 3), "wall_ms_one_call_plus_sync": round(t_wall * 1e3, 3), "columns": cols, "stride": stride,
                         "algorithmic_bytes": {"read": alg_read, "written": alg_write},
                         "read_plus_write_GBps": round((alg_read + alg_write) / t / 1e9, 1),
                         "hbm_traffic_bytes_profiled": traffic,
                         "traffic_over_algorithmic": round(traffic / (alg_read + alg_write), 3) if traffic else None,
                         "note": "row-major file + tape read once (rows staged through LDS), every column written with "
                                 "1-KiB wave stores; ms = device time per call (events around 10 back-to-back asynchronous "
                                 "calls)"}
    # ---- frequency count of one column (all values distinct: the worst case) ----------------------------------------
    need = pkg.columnar_frequency_scratch_bytes(nrec)
    scratch = torch.empty(need, dtype=torch.uint8, device=device)
    ent = torch.empty((nrec + 8, 2), dtype=torch.int64, device=device)
    d_status = torch.zeros(4, dtype=torch.int64, device=device)
    col_ptr, len_ptr = ccols[field].data_ptr(), clens[field].data_ptr()

    def device_time(fn, reps=10):
        """the call is asynchronous (two or three launches): events on the launch stream around `reps` back-to-back calls"""
        t_ = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            dt_ = e0.elapsed_time(e1) * 1e-3 / reps
            t_ = dt_ if t_ is None else min(t_, dt_)
        return t_

    s_ = torch.cuda.current_stream(device).cuda_stream
    t_wall, st = best(lambda: pkg.columnar_frequency_device(ctx, col_ptr, len_ptr, nrec, stride, 0, scratch.data_ptr(), need,
                                                            ent.data_ptr(), ent.shape[0]))
    t = device_time(lambda: pkg.columnar_frequency_device_async(ctx, col_ptr, len_ptr, nrec, stride, 0, scratch.data_ptr(), need,
                                                                ent.data_ptr(), ent.shape[0], d_status.data_ptr(), s_))
    ok = ok and st.n_records == nrec and st.truncated == 0 and st.overflow == 0 and int(ent[: st.n_distinct, 1].sum()) == nrec
    ok = ok and d_status.cpu().tolist() == [nrec, st.n_distinct, 0, 0]
    # the CPU definition on a bounded sample: first 50 k records
    host = dbytes[: 50001 * cols * (width + 1)].cpu().numpy().tobytes()
    hidx = dindex[: 50001 * cols + 1].cpu().numpy().view(np.uint64)
    sample = (0, cols, 50001 * cols, 50000)
    want = oracle.column_frequency(host, hidx, cols, False, [sample], field)
    st_s = pkg.columnar_frequency_device(ctx, col_ptr, len_ptr, 50000, stride, 0, scratch.data_ptr(), need, ent.data_ptr(),
                                         ent.shape[0])
    got = {oracle.seek_field(host, hidx, cols, False, f_, field): c for f_, c in ent[: st_s.n_distinct].cpu().tolist()}
    ok = ok and got == dict(want)
    alg_f = nrec * (stride + 4) + int(st.n_distinct) * 16
    tr_rec = consumer_traffic("colfreq")
    # two launches, two access shapes: FETCH_SIZE counts pass 1's narrow requests in full and pass 2's wide ones at half
    # (profiles/pmc_traffic.json: note); the per-kernel reading is the figure, the uniform readings are its bounds
    tr_f = tr_rec.get("per_kernel_best_reading", tr_rec.get("fetch_x2_plus_write")) if isinstance(tr_rec, dict) else tr_rec
    res["frequency_count"] = {"ms": round(t * 1e3, 4), "wall_ms_one_call_plus_sync": round(t_wall * 1e3, 3),
                              "distinct": int(st.n_distinct),
                              "algorithmic_bytes": alg_f, "GBps_algorithmic": round(alg_f / t / 1e9, 1),
                              "hbm_traffic_bytes_profiled": tr_f,
                              "traffic_over_algorithmic": round(tr_f / alg_f, 3) if tr_f else None,
                              "traffic_over_algorithmic_bounds": ({"fetch_as_counted": round(tr_rec["fetch_as_counted_plus_write"] / alg_f, 3),
                                                                   "fetch_doubled": round(tr_rec["fetch_x2_plus_write"] / alg_f, 3)}
                                                                  if isinstance(tr_rec, dict) and "fetch_as_counted_plus_write" in tr_rec else None),
                              "scratch_bytes": need,
                              "note": "on the column, two launches, no table in device memory and nothing cleared per call: slabs "
                                      "of 8 192 records aggregate in LDS and leave (first record, count, hash) tuples partitioned "
                                      "by hash, partitions merge in LDS (bytes compared: exact) and write their entries; ms = "
                                      "device time per call (events around 10 back-to-back asynchronous calls); algorithmic "
                                      "bytes = the column + its lengths read, 16 B per distinct value written"}
    # ... and of a column of FEW distinct values (100 of them over the same 2.03 M records): the slabs' LDS tables collapse
    # them to ~100 tuples per workgroup
    vocab = ccols[field][:100].clone()
    pick = torch.randint(0, 100, (nrec,), device=device)
    few = vocab[pick].contiguous()
    t_few_wall, st_few = best(lambda: pkg.columnar_frequency_device(ctx, few.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(), need,
                                                                    ent.data_ptr(), ent.shape[0]))
    t_few = device_time(lambda: pkg.columnar_frequency_device_async(ctx, few.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(),
                                                                    need, ent.data_ptr(), ent.shape[0], d_status.data_ptr(), s_))
    cnt_few = torch.bincount(pick, minlength=100)
    got_few = ent[: st_few.n_distinct].cpu()
    ok = ok and st_few.n_distinct == 100 and int(got_few[:, 1].sum()) == nrec
    ok = ok and sorted(got_few[:, 1].tolist()) == sorted(cnt_few.cpu().tolist())
    res["frequency_count"]["few_distinct_values"] = {"ms": round(t_few * 1e3, 4), "wall_ms_one_call_plus_sync": round(t_few_wall * 1e3, 3),
                                                     "distinct": int(st_few.n_distinct)}
    del few, pick
    # ---- search -----------------------------------------------------------------------------------------------------
    needle = host[int(hidx[1000 * cols + field]) + 4: int(hidx[1000 * cols + field]) + 10]
    bm = torch.zeros((nrec + 63) // 64 + 1, dtype=torch.int64, device=device)
    t, hits = best(lambda: pkg.columnar_search_device(ctx, col_ptr, len_ptr, nrec, stride, needle, pkg.SEARCH_CONTAINS,
                                                      bm.data_ptr()))
    hits_s = pkg.columnar_search_device(ctx, col_ptr, len_ptr, 50000, stride, needle, pkg.SEARCH_CONTAINS, bm.data_ptr())
    ok = ok and hits >= 1 and hits_s == len(oracle.column_search(host, hidx, cols, False, sample, field, needle, 2))
    alg_s = nrec * (stride + 4)
    tr_s = consumer_traffic("colsearch_kernel")
    res["search_contains"] = {"ms": round(t * 1e3, 3), "matches": int(hits), "algorithmic_bytes": alg_s,
                              "GBps_algorithmic": round(alg_s / t / 1e9, 1), "hbm_traffic_bytes_profiled": tr_s,
                              "traffic_over_algorithmic": round(tr_s / alg_s, 3) if tr_s else None}
    # ---- round 2's per-column path on the row-major file, for comparison --------------------------------------------
    b = torch.empty(nrec, dtype=torch.int64, device=device)
    e = torch.empty(nrec, dtype=torch.int64, device=device)
    dst = torch.empty(nrec * 32, dtype=torch.uint8, device=device)

    def spans_gather():
        pkg.chunk_field_spans_device(args[1], index_len, cols, "LF", whole, field, b.data_ptr(), e.data_ptr())
        pkg.gather_fields_device(args[0], n, b.data_ptr(), e.data_ptr(), nrec, dst.data_ptr(), 32)
    t_g, _ = best(spans_gather)
    ok = ok and torch.equal(dst.view(nrec, 32), ccols[field])
    need_o = pkg.column_frequency_scratch_bytes(nrec, 1, width)
    scratch_o = torch.empty(need_o, dtype=torch.uint8, device=device)
    ent_o = torch.empty((nrec + 8, 4), dtype=torch.int64, device=device)
    t_f, st_o = best(lambda: pkg.column_frequency_device(ctx, *args, [whole], field, scratch_o.data_ptr(), need_o,
                                                         ent_o.data_ptr(), ent_o.shape[0]))
    ok = ok and st_o.n_distinct == st.n_distinct and st_o.max_field_bytes == width and int(ent_o[: st_o.n_distinct, 3].sum()) == nrec
    t_s, hits_o = best(lambda: pkg.column_search_device(ctx, args[0], n, *args[1:], whole, field, needle, pkg.SEARCH_CONTAINS,
                                                        bm.data_ptr()))
    ok = ok and hits_o == hits
    res["per_column_on_row_major_file"] = {"spans_plus_gather_ms": round(t_g * 1e3, 3), "frequency_count_ms": round(t_f * 1e3, 3),
                                           "search_contains_ms": round(t_s * 1e3, 3),
                                           "note": "one column per call straight from the row-major file (a 32-byte field of a "
                                                   "528-byte row costs 1-2 sectors + a slice of tape per record); the frequency "
                                                   "count = the column gathered straight from the tape + the columnar count above (one "
                                                   "implementation), one synchronisation"}
    res["verified"] = bool(ok)
    ctx.close()
    return res


def consumers_large_leg(pkg, device):
    """The column consumers at a size where a roofline fraction means something: a column of 32 Mi
    records x 32 bytes (1 GiB; random lower-case rows, fixed width), search in every mode and the frequency count with 100 /
    1 000 / 10 000 / all-distinct values (the first two are counted by the streaming kernel, the others by the general passes); and the whole 8-GiB 16x32 file + its tape -> 16 columns in one pass.  Device times by
    events on the launch stream; `frac` = algorithmic bytes / time / 8 TB/s (search: the column read once; count: the
    column read + 16 B per distinct value written; to_columns: file + tape read, columns + lengths written).  Checked:
    counts add up and match torch.unique / bincount, `contains` against a torch restatement on a 1-Mi-record slice, the
    columns against plain slicing of the fixed-pitch rows."""
    nrec, stride = 32 << 20, 32
    ctx = pkg.Context(device.index)
    s_ = torch.cuda.current_stream(device).cuda_stream
    out = {"records": nrec, "stride": stride, "column_bytes": nrec * stride}
    ok = True

    def device_time(fn, reps=5):
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3 / reps
            best = dt if best is None else min(best, dt)
        return best

    g = torch.Generator(device=device)
    g.manual_seed(1)
    col = torch.randint(97, 123, (nrec, stride), dtype=torch.uint8, device=device, generator=g)
    need = pkg.columnar_frequency_scratch_bytes(nrec)
    scratch = torch.empty(need, dtype=torch.uint8, device=device)
    ent = torch.empty((nrec + 8, 2), dtype=torch.int64, device=device)
    d_status = torch.zeros(4, dtype=torch.int64, device=device)
    alg = nrec * stride
    freq = {}
    for label, k in (("all_distinct", 0), ("100_values", 100), ("1000_values", 1000), ("10000_values", 10000)):
        if k:
            pick = torch.randint(0, k, (nrec,), device=device, generator=g)
            c = col[:k][pick].contiguous()
        else:
            c = col
        st = pkg.columnar_frequency_device(ctx, c.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
        t = device_time(lambda: pkg.columnar_frequency_device_async(ctx, c.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(), need,
                                                                    ent.data_ptr(), ent.shape[0], d_status.data_ptr(), s_))
        good = int(ent[: st.n_distinct, 1].sum()) == nrec and st.overflow == 0 and st.truncated == 0
        if k:
            cnt = torch.bincount(pick, minlength=k)
            good = good and st.n_distinct == int((cnt > 0).sum()) and \
                sorted(ent[: st.n_distinct, 1].cpu().tolist()) == sorted(cnt[cnt > 0].cpu().tolist())
            del c, pick
        else:
            good = good and st.n_distinct == nrec      # (26^32 rows: a repeat among 2^25 is not going to happen)
        ok = ok and good
        a = alg + int(st.n_distinct) * 16
        freq[label] = {"ms": round(t * 1e3, 4), "distinct": int(st.n_distinct), "algorithmic_bytes": a,
                       "GBps_algorithmic": round(a / t / 1e9, 1), "frac": round(a / t / 1e9 / HBM_PEAK_GBPS, 4), "verified": bool(good)}
    freq["note"] = ("three launches on the caller's stream, asynchronous: colfreq_stream_kernel first (one workgroup per CU keeps one LDS "
                    "table — 3 072 slots with copies of the rows — over its whole share of the column; 100 and 1 000 values end "
                    "there), then the general passes for the shares it gave up on (more than 2 304 values per share: the 10 000-value "
                    "and the all-distinct column, whose attempt ends within the first batch): slabs of 8 192 records -> tuples "
                    "partitioned by hash -> partitions merged in LDS, bytes compared everywhere; ms = device time per call "
                    "(events around 5 back-to-back calls, best of 3); algorithmic bytes = the column read + 16 B per distinct value")
    out["frequency_count"] = freq
    del ent, scratch
    bm = torch.zeros((nrec + 63) // 64 + 1, dtype=torch.int64, device=device)
    row = bytes(col[1000].cpu().numpy())
    search = {}
    for label, needle, mode in (("contains_6_bytes", row[4:10], pkg.SEARCH_CONTAINS), ("contains_1_byte", row[4:5], pkg.SEARCH_CONTAINS),
                                ("contains_12_bytes", row[14:26], pkg.SEARCH_CONTAINS), ("equals", row, pkg.SEARCH_EQUALS),
                                ("starts_with_5_bytes", row[:5], pkg.SEARCH_STARTS_WITH)):
        hits = pkg.columnar_search_device(ctx, col.data_ptr(), 0, nrec, stride, needle, mode, bm.data_ptr())
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            pkg.columnar_search_device(ctx, col.data_ptr(), 0, nrec, stride, needle, mode, bm.data_ptr())
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        ok = ok and hits >= 1
        search[label] = {"ms_wall_one_synchronous_call": round(t * 1e3, 4), "matches": int(hits), "algorithmic_bytes": alg,
                         "GBps_algorithmic": round(alg / t / 1e9, 1), "frac": round(alg / t / 1e9 / HBM_PEAK_GBPS, 4)}
    sl = col[: 1 << 20]
    nt = torch.tensor(list(row[4:10]), dtype=torch.uint8, device=device)
    m = torch.zeros(sl.shape[0], dtype=torch.bool, device=device)
    for s0 in range(stride - 6 + 1):
        m |= (sl[:, s0: s0 + 6] == nt).all(dim=1)
    hits = pkg.columnar_search_device(ctx, col.data_ptr(), 0, 1 << 20, stride, row[4:10], pkg.SEARCH_CONTAINS, bm.data_ptr())
    bits = bm[: (1 << 20) // 64].cpu().numpy().view(np.uint64)
    want_bits = np.packbits(m.cpu().numpy(), bitorder="little").view(np.uint64)
    ok = ok and int(m.sum()) == hits and bool(np.array_equal(bits, want_bits))
    search["note"] = ("rows of 32 bytes in registers, contiguous wave loads (a lane holds half rows), two batches in flight: equals / "
                      "starts-with / needles of up to three bytes search the halves where they are (colsearch32_kernel), longer needles "
                      "swap halves into whole rows and filter with v_mqsad_u32_u8 (colsearch_small_kernel); ms = wall time of one "
                      "synchronous call (launch, kernel, the count's copy), best of 7; algorithmic bytes = the column")
    out["search"] = search
    del col, bm, sl, m
    torch.cuda.empty_cache()
    # ---- the 8-GiB file + its tape -> 16 columns ------------------------------------------------------------------------
    name = "16x32_noquote"
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 8 << 30)
    dbytes = torch.empty(n, dtype=torch.uint8, device=device)
    pkg.synth_fill_device(dbytes.data_ptr(), 0, n, cols, width, seed, q)
    entries = n // (width + 1)
    dindex = torch.zeros(entries + 2, dtype=torch.int64, device=device)
    ctx.reserve(n)
    r = ctx.stage1_index_device(dbytes.data_ptr(), n, 0, 0, dindex.data_ptr() + 8, entries + 1)
    rows = r.count // cols
    nrec2 = rows - 1
    whole = (0, cols, rows * cols, nrec2)
    ccols = torch.empty((cols, nrec2, stride), dtype=torch.uint8, device=device)
    clens = torch.empty((cols, nrec2), dtype=torch.int32, device=device)
    to_cols = lambda: pkg.chunk_to_columns_device(ctx, dbytes.data_ptr(), n, dindex.data_ptr(), r.count + 1, cols, "LF", whole, None,
                                                  ccols.data_ptr(), stride, clens.data_ptr())
    to_cols()
    t = device_time(to_cols, reps=3)
    good = bool((clens == width).all())
    for cidx in (0, 7, 15):       # three of the sixteen columns against plain slicing of the fixed-pitch rows
        tbl = dbytes[: rows * cols * (width + 1)].view(rows, cols, width + 1)[1:, cidx, :width]
        good = good and torch.equal(ccols[cidx], tbl)
    ok = ok and good
    alg_read = (n - cols * (width + 1)) + 8 * (r.count + 1 - cols)
    alg_write = cols * nrec2 * (stride + 4)
    out["to_columns_8GiB"] = {"ms": round(t * 1e3, 3), "records": nrec2, "columns": cols,
                              "algorithmic_bytes": {"read": alg_read, "written": alg_write},
                              "read_plus_write_GBps": round((alg_read + alg_write) / t / 1e9, 1),
                              "frac": round((alg_read + alg_write) / t / 1e9 / HBM_PEAK_GBPS, 4), "verified": bool(good)}
    out["verified"] = bool(ok)
    ctx.close()
    return out


def batch_leg(pkg, device, k=8):
    """Many files: k buffers of 128 MiB (the 16x32 corpus, whole rows each) indexed by k launches back to back and by
    ONE batched launch (csvsimd_stage1_index_batch_device_async: the tiles of all buffers share one ticket, a look-back
    stops at its buffer's first tile).  A launch's fixed cost (fill + drain, ~20 us) is paid once instead of k times.
    k = 8 is 1 GiB per launch: the one fixed cost still weighs 9 % there, exactly as for ONE 1-GiB buffer; k = 64 is
    8 GiB per launch, where the batch runs at the rate of a single 8-GiB buffer.
    Times = torch events on the launch stream around 20 repetitions after a settle; every tape checked against the closed form."""
    name = "16x32_noquote"
    cols, width, seed, q = pkg.WORKLOADS[name]
    per = pkg.workload_len(name, 128 << 20)
    pitch = width + 1
    dbuf = torch.empty(k * per, dtype=torch.uint8, device=device)
    pkg.synth_fill_device(dbuf.data_ptr(), 0, k * per, cols, width, seed, q)
    cap = per // pitch + 64
    tapes = [torch.empty(cap, dtype=torch.int64, device=device) for _ in range(k)]
    dres = torch.zeros((k, 8), dtype=torch.int64, device=device)
    ctx = pkg.Context(device.index)
    ctx.reserve(k * per + (k << 18))
    stream = torch.cuda.current_stream(device).cuda_stream
    items = [(dbuf.data_ptr() + i * per, per, i * per, tapes[i].data_ptr(), cap, 0) for i in range(k)]

    def separate():
        for i in range(k):
            ctx.stage1_index_device_async(items[i][0], per, i * per, 0, items[i][3], cap, dres[i].data_ptr(), stream)

    def batched():
        ctx.stage1_index_batch_device_async(items, dres.data_ptr(), stream)

    def timed(fn, reps=20):
        for _ in range(settle_count(k * per)):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def verify():
        torch.cuda.synchronize(device)
        ok = True
        for i in range(k):
            cnt = int(dres[i, 0])
            k0 = (i * per) // pitch
            want = torch.arange(k0, k0 + per // pitch, dtype=torch.int64, device=device) * pitch + width
            ok = ok and cnt == per // pitch and torch.equal(tapes[i][:cnt], want)
        return bool(ok)

    ms_sep = timed(separate)
    ok = verify()
    for t in tapes:
        t.fill_(-1)
    ms_bat = timed(batched)
    ok = ok and verify()
    ctx.close()
    frac = lambda ms: round(k * per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    word = {8: "eight", 64: "sixty_four"}.get(k, str(k))
    return {"workload": f"{k} buffers x {per / 2**20:.0f} MiB of the {name} corpus ({k * per / 2**30:.2f} GiB per batch)",
            f"{word}_launches_ms": round(ms_sep, 4), f"{word}_launches_hbm_read_frac": frac(ms_sep),
            "one_batched_launch_ms": round(ms_bat, 4), "one_batched_launch_hbm_read_frac": frac(ms_bat),
            "verified": ok}


def back_to_back_leg(pkg, device, k=8, reps=6):
    """k consecutive 1-GiB files (the 16x32 corpus, config 2's shape), one stage-1 launch each, WITHOUT the batch API:
    (a) one context, one stream — every launch pays its ~20 us of fill and drain alone (the 1-GiB configurations' 60 %);
    (b) two contexts (two control blocks, two look-back scratch areas) alternating on two streams of DIFFERENT priority, i.e.
    two hardware queues: launch i + 1 is resident-ready while launch i drains, its workgroups take the wave slots the
    drain frees.  Times = events around reps x k launches after a settle; every tape checked against the closed form."""
    name = "16x32_noquote"
    cols, width, seed, q = pkg.WORKLOADS[name]
    per = pkg.workload_len(name, 1 << 30)
    pitch = width + 1
    dbuf = torch.empty(k * per, dtype=torch.uint8, device=device)
    pkg.synth_fill_device(dbuf.data_ptr(), 0, k * per, cols, width, seed, q)
    cap = per // pitch + 64
    tapes = [torch.empty(cap, dtype=torch.int64, device=device) for _ in range(k)]
    dres = torch.zeros((k, 8), dtype=torch.int64, device=device)
    ctxs = [pkg.Context(device.index), pkg.Context(device.index)]
    for c in ctxs:
        c.reserve(per)
    s_main = torch.cuda.current_stream(device)
    s_hi = torch.cuda.Stream(device=device, priority=-1)

    def one_queue():
        for i in range(k):
            ctxs[0].stage1_index_device_async(dbuf.data_ptr() + i * per, per, 0, 0, tapes[i].data_ptr(), cap, dres[i].data_ptr(),
                                              s_main.cuda_stream)

    def two_queues():
        for i in range(k):
            st = s_hi if (i & 1) else s_main
            ctxs[i & 1].stage1_index_device_async(dbuf.data_ptr() + i * per, per, 0, 0, tapes[i].data_ptr(), cap, dres[i].data_ptr(),
                                                  st.cuda_stream)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_hi.wait_stream(s_main)
            e0.record(s_main)
            for _ in range(reps):
                fn()
            s_main.wait_stream(s_hi)
            e1.record(s_main)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / (reps * k)
            best = ms if best is None else min(best, ms)
        return best

    def verify():
        torch.cuda.synchronize(device)
        want = torch.arange(1, per // pitch + 1, dtype=torch.int64, device=device) * pitch - 1
        ok = True
        for i in range(k):
            w = dres[i].cpu().tolist()
            ok = ok and w[0] == per // pitch and (w[4] & 0xFFFFFFFF) == 0 and torch.equal(tapes[i][: per // pitch], want)
        return bool(ok)

    ms_one = timed(one_queue)
    ok = verify()
    for t in tapes:
        t.zero_()
    ms_two = timed(two_queues)
    ok = ok and verify()
    for c in ctxs:
        c.close()
    frac = lambda ms: round(per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    return {"files": k, "bytes_per_file": per, "workload": name,
            "one_context_one_queue": {"ms_per_file": round(ms_one, 4), "hbm_read_frac": frac(ms_one)},
            "two_contexts_two_queues": {"ms_per_file": round(ms_two, 4), "hbm_read_frac": frac(ms_two)},
            "gain": round(ms_one / ms_two, 4), "verified": ok,
            "note": "aggregate over consecutive launches (events around %d x %d launches): what a caller indexing file after file "
                    "sees; two contexts on two hardware queues let a launch's fill overlap its predecessor's drain" % (reps, k)}


def dense_ceiling(sb):
    """What bare streams reach with the dense corpus's write share (1.6 B of tape per byte read), next to an
    INDEPENDENT yardstick: a plain copy — hipMemcpyDtoD and the textbook one-16-byte-element-per-thread
    kernel, the shape behind the guide's "6.29 TB/s float4 copy" — over the same buffers.  A copy moves 2 bytes per byte
    read; if it sustains clearly more read + write TB/s than the probe's mixes, the probe is not the ceiling."""
    ctx, n, s_ = sb.ctx, sb.n, sb.stream()
    src, dst = sb.dbuf.data_ptr(), sb.dtape.data_ptr()
    pm = ctx.hbm_probe_device(src, n, dst, 25, s_, 1, 5)
    pm2 = ctx.hbm_probe_device(src, n, dst, 25, s_, 1, 5, blocks_per_cu=2)
    ncopy = min(n, sb.cap * 8) // 16 * 16
    copies = {}
    for mode, name in ((0, "hipMemcpyDtoD"), (1, "kernel_16B_per_thread"), (2, "kernel_16B_per_thread_nontemporal")):
        ms = ctx.copy_probe_device(src, dst, ncopy, mode, s_, 2, 10)
        copies[name + "_read_plus_write_GBps"] = round(2 * ncopy / (ms * 1e-3) / 1e9, 1)
    return {"hbm_read_frac_16_waves_per_cu": round(n / (pm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "hbm_read_frac_8_waves_per_cu": round(n / (pm2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "read_plus_write_GBps_16_waves_per_cu": round((n + n * 25 // 16) / (pm * 1e-3) / 1e9, 1),
            "read_plus_write_GBps_8_waves_per_cu": round((n + n * 25 // 16) / (pm2 * 1e-3) / 1e9, 1),
            "plain_copy_yardstick": copies, "copy_bytes": ncopy,
            "note": "bare nt stream writing 25/16 B per byte read, no work at all, and plain copies of the same buffers "
                    "(2 B moved per byte read): what this GPU's memory system sustains for write-heavy mixes"}


def run_sharded(pkg, sb, device, dist_on, rehearsal, comm, steps, warmup):
    """Times `steps` steps of the workload in `sb`; returns (dt, state of the last step).

    Steps are pipelined PIPELINE_DEPTH deep: a step's launches, its collective and its copy-out are enqueued, and only
    then is the PREVIOUS step's record waited for and checked — the host's per-step work (launch latency, the
    read-back, Python) overlaps the GPU's, as in a caller that indexes one batch after another.  Consecutive steps
    write alternate tape buffers.  Every step's record is checked inside the timed region; CSVSIMD_BENCH_PIPELINE=0
    restores one step at a time.  (The native-RCCL entry point is one blocking C call per step: not pipelined.)"""
    from csv_simd_amd import sharded
    state = {}
    pipelined = os.environ.get("CSVSIMD_BENCH_PIPELINE", "1") != "0" and comm is None
    # CSVSIMD_BENCH_TAIL_OVERLAP=0: the sharded step's tail in stream order behind its first pass, as in round 3
    overlap = pipelined and dist_on and os.environ.get("CSVSIMD_BENCH_TAIL_OVERLAP", "1") != "0"
    depth = min(len(sb.dtapes), PIPELINE_DEPTH_SHARDED if overlap else PIPELINE_DEPTH) if pipelined else 1
    overlap = overlap and depth >= PIPELINE_DEPTH_SHARDED
    stepper = (sharded.ShardedStep(device, gather_via_host=rehearsal, depth=depth, overlap_tail=overlap)
               if (dist_on and comm is None) else None)
    ctx_re = sb.tail_context() if overlap else None
    state["tail"] = ("all-gather, stitch kernel, re-emit launch and copy-out on a second stream (the re-emit in its own "
                     "context): they overlap the next step's first pass" if overlap else "in stream order behind the first pass")
    # rank 0 knows how the file starts; every other rank lets the kernel choose the entering state its first eight tiles speak
    # for (CSVSIMD_ENTER_GUESS) — only a rank that chose wrong re-emits.  CSVSIMD_BENCH_NO_GUESS=1: speculate "outside"
    # everywhere, as round 1 did (then every rank that really starts inside a string re-emits).
    guess = os.environ.get("CSVSIMD_BENCH_NO_GUESS") != "1"
    first_state = pkg.ENTER_GUESS if (guess and sb.rank > 0) else 0
    state["first_pass"] = "rank 0: entering state known; ranks > 0: " + ("the kernel's guess from its first eight tiles"
                                                                        if guess else "speculated 'outside'")
    inflight = []          # slots enqueued and not yet collected, oldest first
    counter = [0]

    def enqueue(slot):
        if dist_on:
            d_res = stepper.slots[slot].d_result
            sb.use_slot(slot)
            tape = sb.dtape   # bound now: the re-emit must hit the same buffer as the speculative pass
            stepper.enqueue(lambda inq: sb.launch(inq, d_res), lambda p: sb.reemit(p, d_res, ctx_re), slot=slot,
                            first_state=first_state)
            assert sb.dtape is tape
        else:
            sb.enqueue_pass(0, slot)

    def collect(slot):
        if dist_on:
            st, final, _ = stepper.collect(slot)
            sb.check(final)
            assert final.count == st.count
            state.update(count=st.count, inq=st.in_quote_in, base=st.tape_index_base, total=st.total_entries,
                         final=st.in_quote_final, slot=slot, reemit=st.reemit)
        else:
            w = sb.collect_pass(slot)
            state.update(count=int(w[0]), inq=0, base=1, total=int(w[0]) + 1, final=(int(w[3]) >> 32) & 1, slot=slot)

    def drain():
        while inflight:
            collect(inflight.pop(0))

    def step():
        if comm is not None:
            r, st = comm.index_sharded(sb.ctx, sb.dbuf.data_ptr(), sb.n, sb.lo, sb.dtape.data_ptr(), sb.cap, 0,
                                       sb.stream())
            sb.check(r)
            state.update(count=st.count, inq=st.in_quote_in, base=st.tape_index_base, total=st.total_entries,
                         final=st.in_quote_final, slot=0, reemit=st.reemit)
            return
        slot = counter[0] % depth
        counter[0] += 1
        if len(inflight) == depth:      # the slot about to be reused holds the oldest step
            collect(inflight.pop(0))
        enqueue(slot)
        inflight.append(slot)
        if depth == 1:
            drain()

    dt = time_steps(step, steps, warmup, device, dist_on, settle=settle_count(sb.total // max(sb.world, 1)),
                    drain=drain)
    sb.use_slot(state.get("slot", 0))   # the tape the last step wrote is the one the verification reads
    state["pipeline_depth"] = depth
    return dt, state


def verify_everything(oracle, sb, state, dist_on, rank, world):
    """Per-rank checks, then the job-level stitch check on rank 0; raises SystemExit on any mismatch."""
    import torch.distributed as dist
    if os.environ.get("CSVSIMD_BENCH_REHEARSAL") == "1" and dist_on:
        # development rehearsal only (all ranks share ONE GPU): the ranks verify one after the other — three or
        # more processes running torch's synchronising ops (nonzero, equal) on one card at once were seen to stall
        for r in range(world):
            if r == rank:
                res, facts = verify_rank(oracle, sb, state["inq"], state["count"], state["base"], state["total"],
                                         state["final"], state.get("reemit", 0))
                torch.cuda.synchronize()
            dist.barrier()
    else:
        res, facts = verify_rank(oracle, sb, state["inq"], state["count"], state["base"], state["total"], state["final"],
                                 state.get("reemit", 0))
    gathered = [None] * world
    if dist_on:
        dist.all_gather_object(gathered, (res, facts))
    else:
        gathered = [(res, facts)]
    job_ok, reemits, total, inside = verify_job([g[1] for g in gathered])
    tape_ok = all(all(g[0].values()) for g in gathered)
    out = {"tape": bool(tape_ok), "stitch": bool(job_ok), "reemits": int(reemits),
           "ranks_entered_inside_a_string": int(inside), "total_entries": int(total),
           "how": "every rank: whole tape shard == torch restatement of the definition on the same bytes"
                  + ("" if sb.q else " == closed form")
                  + f"; {WINDOW_ENTRIES} entries either side of each shard boundary + checksum == CPU oracle; "
                    "entering state == CPU generator; rank 0: bases / totals / final state chain"}
    if not (tape_ok and job_ok):
        out["failed"] = [{"rank": g[1]["rank"], **{k: v for k, v in g[0].items() if not v}} for g in gathered
                         if not all(g[0].values())]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="64x31_noquote",
                    choices=["64x31_noquote", "64x31_q10", "16x32_noquote", "16x32_q10", "1024x4_dense"])
    ap.add_argument("--gib-per-gpu", type=float, default=8.0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --gib-per-gpu bytes on every rank; strong: ONE file of --total-gib bytes cut "
                         "into one shard per rank")
    ap.add_argument("--total-gib", type=float, default=64.0,
                    help="size of the one file of the strong-scaling legs (BASELINE config 4: 64 GiB)")
    ap.add_argument("--no-strong-check", action="store_true",
                    help="skip the strong-scaling leg of the default (weak) line")
    ap.add_argument("--skew", type=int, default=0,
                    help="move every interior shard cut this many bytes to the right (SURVEY §8d: 777)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE shapes at N=1")
    ap.add_argument("--no-verify", action="store_true", help="development only: the line then says verified: null")
    ap.add_argument("--no-q10-check", action="store_true", help="skip the quoted, mid-row-cut verification leg")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--native-rccl", action="store_true",
                    help="do the sharded step inside the C ABI (csvsimd_stage1_index_sharded: ncclAllGather "
                         "from C++) instead of torch.distributed.all_gather_into_tensor")
    ap.add_argument("--only-batch", action="store_true", help="development: run the `batch_many_files` leg alone")
    ap.add_argument("--batch-k", type=int, default=0, help="with --only-batch: run this one batch size (8 or 64) alone")
    ap.add_argument("--only-latency", action="store_true", help="development: run the `latency` leg alone")
    ap.add_argument("--only-small-files", action="store_true", help="development: run the `small_files` leg alone")
    ap.add_argument("--only-consumers-large", action="store_true", help="development / profiling: run the `consumers_at_1GiB` leg alone")
    ap.add_argument("--only-back-to-back", action="store_true", help="development / profiling: run the `back_to_back_1GiB` leg alone")
    ap.add_argument("--only-consumers", action="store_true",
                    help="development / profiling: run the `consumers` leg alone and print its record (not the "
                         "contract line)")
    args = ap.parse_args()
    if os.environ.get("CSVSIMD_BENCH_WATCHDOG"):   # development: where is a stuck run stuck?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CSVSIMD_BENCH_WATCHDOG"]), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the launcher.  Nothing has touched the GPU yet (importing torch does not),
        # so the ranks start as ordinary children; their rank 0 prints the line, this process relays the exit code.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    if rank == 0:
        build_if_needed()
    # under torch.distributed.run the collective path is exercised even with a single rank
    dist_on = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("CSVSIMD_BENCH_FORCE_DIST") == "1"
    # rehearsal of the N > 1 control flow on a ONE-GPU box (dev only, never used by the driver): all
    # ranks share cuda:0 and the records travel over gloo through the host instead of RCCL; the stitch
    # kernel and the re-emit launch are the real ones
    rehearsal = os.environ.get("CSVSIMD_BENCH_REHEARSAL") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if dist_on:
        # RCCL prints a version banner on the process's stdout when its first communicator comes up; this file's stdout is
        # ONE JSON line, so file descriptor 1 points at stderr until the group's first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                if "MASTER_ADDR" not in os.environ:   # CSVSIMD_BENCH_FORCE_DIST=1 without a launcher
                    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
                # the collective's own stream (torch keeps one per communicator) on a high-priority hardware queue as well:
                # on a queue it shares with the stage-1 launches its wait for the tail's event would hold those up
                os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
                dist.init_process_group("nccl", device_id=device)
            dist.barrier()
            if not rehearsal:
                torch.cuda.synchronize(device)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    pkg = graft.load_package()
    refuse_probe_environment(pkg)
    from csv_simd_amd import sharded
    oracle = graft.load_oracle() if not (args.no_verify and args.no_cpu_baseline) else None
    if args.only_consumers:
        print(json.dumps({"consumers": consumers_leg(pkg, oracle or graft.load_oracle(), device)}))
        return
    if args.only_batch:
        if args.batch_k:     # profiling: ONE batch size per process, so that a kernel-stats row is one launch shape
            print(json.dumps({f"batch_of_{args.batch_k}": batch_leg(pkg, device, k=args.batch_k)}))
        else:
            print(json.dumps({"batch_many_files": batch_leg(pkg, device), "at_8_GiB_per_batch": batch_leg(pkg, device, k=64)}))
        return
    if args.only_latency:
        print(json.dumps({"latency": latency_leg(pkg, oracle or graft.load_oracle(), device)}))
        return
    if args.only_consumers_large:
        print(json.dumps({"consumers_at_1GiB": consumers_large_leg(pkg, device)}))
        return
    if args.only_back_to_back:
        print(json.dumps({"back_to_back_1GiB": back_to_back_leg(pkg, device)}))
        return
    if args.only_small_files:
        print(json.dumps({"small_files": small_files_leg(pkg, oracle or graft.load_oracle(), device)}))
        return

    strong = args.scaling == "strong"
    # weak (default): every rank holds --gib-per-gpu bytes at every N.  strong: ONE file of --total-gib bytes cut into
    # `world` contiguous shards — N = 1 indexes the whole of BASELINE config 4 (64 GiB: 64 + 16 GiB of tape fit one
    # 288 GB GPU), N = 8 takes 8 GiB each: the same file getting faster, not more bytes.
    shard_bytes = int(args.total_gib * 2**30) // world if strong else int(args.gib_per_gpu * 2**30)
    sb_depth = PIPELINE_DEPTH_SHARDED if (dist_on and not args.native_rccl) else PIPELINE_DEPTH
    sb = ShardBench(pkg, device, args.workload, shard_bytes, rank, world, args.skew, depth=sb_depth)

    comm = None
    if dist_on and args.native_rccl:
        uid = [pkg.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)   # the 128-byte RCCL id travels over the existing group
        comm = pkg.Comm(uid[0], rank, world, device.index)

    dt, state = run_sharded(pkg, sb, device, dist_on, rehearsal, comm, args.steps, args.warmup)
    total_bytes = sb.total
    gib_s = total_bytes * args.steps / dt / 2**30
    rows = total_bytes // sb.row

    verified = None if args.no_verify else verify_everything(oracle, sb, state, dist_on, rank, world)

    # ---- roofline leg: the stage-1 kernel alone, HIP events on its own stream, on EVERY rank -------
    kern_ms_mine = kernel_time_ms(sb, max(5, min(args.steps, 50)))
    per_rank_ms = [kern_ms_mine]
    if dist_on:
        gathered_ms = [None] * world
        dist.all_gather_object(gathered_ms, float(kern_ms_mine))
        per_rank_ms = [float(x) for x in gathered_ms]
    # the slowest rank's kernel prices the job (at N = 1 that is the only one)
    kern_ms = max(per_rank_ms)
    achieved = sb.n / (kern_ms * 1e-3) / 1e9
    entries = state["count"]
    # what this GPU's HBM actually streams with the same traffic shape and no work at all (the tape
    # shard doubles as the probe's output buffer: it is rewritten by the next launch anyway); rank 0's GPU only
    probed = None
    if rank == 0 and sb.cap * 8 >= sb.n // 4:
        s_ = sb.stream()
        ms_r = sb.ctx.hbm_probe_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), 0, s_, 1, 5)
        ms_rw = sb.ctx.hbm_probe_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), 4, s_, 1, 5)
        ms_rw2 = sb.ctx.hbm_probe_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), 4, s_, 1, 5, blocks_per_cu=2)
        probed = {"read_only_GBps": round(sb.n / (ms_r * 1e-3) / 1e9, 1),
                  "read_with_quarter_written_GBps": round(sb.n / (ms_rw * 1e-3) / 1e9, 1),
                  "read_with_quarter_written_8_waves_per_cu_GBps": round(sb.n / (ms_rw2 * 1e-3) / 1e9, 1),
                  "kernel_vs_probe": round(sb.n / (kern_ms_mine * 1e-3) / 1e9 / (sb.n / (ms_rw * 1e-3) / 1e9), 3),
                  "gpu": "rank 0's",
                  "note": "bare nt stream of the same buffer with none of the work (csvsimd_hbm_probe_device), at the "
                          "kernel's 16 waves/CU and at the 8 waves/CU where a bare stream peaks; the 64x31 corpus "
                          "writes 8 B of tape per 32 B read"}
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": load_traffic(args.workload, sb.n),
        "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
        "kernel": sb.ctx.kernel_name(), "kernels_per_launch": 1, "kernel_ms": round(kern_ms, 4),
        "kernel_ms_per_rank": {"min": round(min(per_rank_ms), 4), "max": round(max(per_rank_ms), 4),
                               "all": [round(x, 4) for x in per_rank_ms],
                               "note": "every rank times its own launches (HIP events on its stream); `kernel_ms`, "
                                       "`achieved` and `frac` are the SLOWEST rank's"},
        "entry_point": "csvsimd_stage1_time_device -> launch_stage1 (the product library; probe builds refused)",
        "algorithmic_bytes_per_launch": sb.n,
        "read_plus_tape_write_GBps": round((sb.n + 8 * entries) / (kern_ms * 1e-3) / 1e9, 1),
        "probed_stream": probed,
    }

    if dist_on:
        backend = dist.get_backend()
        collective = (f"one all-gather of the 64-byte shard records per step ({backend}"
                      + (", ncclAllGather from the C ABI" if comm is not None else ", torch.distributed.all_gather_into_tensor")
                      + f", {dist.get_world_size()} ranks), stitch kernel + conditional re-emit launch on the device")
    else:
        backend = None
        collective = ("none: a single GPU holds the whole file, so the timed step is ONE stage-1 launch + its record's "
                      "copy-out — no all-gather, no stitch kernel, no re-emit launch (with a world-1 RCCL communicator "
                      "the same step costs +8 us: profiles/r02_bench_dist_world1_torch.json)")
    out = {
        "metric": "csv_bytes_scanned_per_s", "value": round(gib_s, 3), "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {sb.cols} cols x {sb.width}-byte fields, LF rows, "
                               f"{sb.n / 2**30:.3f} GiB per GPU"
                               + (f" = 1/{world} of ONE {total_bytes / 2**30:.0f} GiB file (BASELINE config 4, strong scaling)"
                                  if strong else " (BASELINE config 4's per-GPU shard shape)"),
                   "bytes_per_gpu": sb.n, "total_bytes": total_bytes, "tape_entries": int(state["total"]),
                   "skew": args.skew,
                   "steps_in_flight": state["pipeline_depth"],   # step i+1 (sharded: and i+2) enqueued before step i's record is read
                   "sharded_step_tail": state["tail"] if dist_on else None,
                   "first_pass": state["first_pass"] if world > 1 else "one rank: the file's entering state is known",
                   "rccl_world": dist.get_world_size() if (dist_on and backend == "nccl") else None,
                   "collective_world": dist.get_world_size() if dist_on else None,   # ranks the process group reports
                   "collective_backend": backend,
                   "parallelism": f"chunk-sharded x{world}; collective: {collective}"
                                  + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearsal else "")},
        "rows_indexed_per_s": round(rows * args.steps / dt, 1),
        "gib_per_s_per_gpu": round(gib_s / world, 3),
        "verified": verified,
        "roofline": roofline,
    }

    # the first 2 GiB of rank 0's shard, on the host, for the CPU baselines and the ingest leg (taken before the
    # device buffers are recycled by the legs below)
    sample = host_sample(sb, 2 << 30) if (rank == 0 and not (args.no_cpu_baseline and args.no_ingest)) else None
    main_width, main_q, main_lo = sb.width, sb.q, sb.lo

    # ---- the quoted corpus, cut mid-row: the stitch is non-trivial and ranks re-emit -------------
    failed = verified is not None and not (verified["tape"] and verified["stitch"])
    if not args.no_q10_check and args.workload == "64x31_noquote" and not args.no_verify and not strong:
        sb.release()
        cuts = mid_row_cuts(oracle, pkg, "64x31_q10", shard_bytes, world)
        sbq = ShardBench(pkg, device, "64x31_q10", shard_bytes, rank, world, cuts=cuts, depth=sb_depth)
        k = max(3, min(args.steps, 5))
        dtq, stq = run_sharded(pkg, sbq, device, dist_on, rehearsal, comm, k, 1)
        vq = verify_everything(oracle, sbq, stq, dist_on, rank, world)
        failed = failed or not (vq["tape"] and vq["stitch"])
        out["q10_skew_check"] = {"workload": "64x31_q10, interior cuts mid-row: odd ones inside a quoted field (that rank "
                                             "really starts inside a string: it re-emits unless its first pass chose "
                                             "that state itself), even ones at +777 outside a string",
                                 "cuts": cuts, "first_pass": stq["first_pass"],
                                 "ms_per_step": round(dtq / k * 1e3, 4), "steps": k,
                                 "GiB/s": round(sbq.total * k / dtq / 2**30, 2), "verified": vq}
        if rank == 0 and world == 1:
            ms = kernel_time_ms(sbq, 10)
            out["q10_skew_check"]["kernel_ms"] = round(ms, 4)
            out["q10_skew_check"]["hbm_read_frac"] = round(sbq.n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        sbq.release()
        del sbq

    # ---- strong scaling on the config-4 file, in the same run: 64 GiB cut into `world` shards -------------------
    # (the default line is weak scaling: 8 GiB per GPU at every N.  This leg indexes the SAME 64 GiB file at every N —
    # all of it on one GPU at N = 1, 8 GiB each at N = 8 — so the driver's N = 1, 2, 4, 8 runs also hold the
    # speed-up on one file, verified like the main leg.)
    if not args.no_strong_check and args.workload == "64x31_noquote" and not strong and not args.no_verify:
        strong_total = int(args.total_gib * 2**30)
        if strong_total // world == shard_bytes:
            out["strong_scaling_check"] = {"total_GiB": args.total_gib, "identical_to_main_leg": True,
                                           "ms_per_step": out["ms_per_step"], "GiB/s": out["value"],
                                           "note": f"at N = {world} the {args.total_gib:g} GiB file's shards ARE the "
                                                   "main leg's shards"}
        else:
            sb.release()
            sbs = ShardBench(pkg, device, args.workload, strong_total // world, rank, world, depth=sb_depth)
            k = max(3, min(args.steps, 5))
            dts, sts = run_sharded(pkg, sbs, device, dist_on, rehearsal, comm, k, 1)
            vs = verify_everything(oracle, sbs, sts, dist_on, rank, world)
            failed = failed or not (vs["tape"] and vs["stitch"])
            ms = kernel_time_ms(sbs, 5)
            out["strong_scaling_check"] = {
                "workload": f"ONE {sbs.total / 2**30:.0f} GiB file of the 64x31 corpus (BASELINE config 4) cut into "
                            f"{world} contiguous shard(s) of {sbs.n / 2**30:.2f} GiB",
                "total_bytes": sbs.total, "bytes_per_gpu": sbs.n, "ms_per_step": round(dts / k * 1e3, 4), "steps": k,
                "GiB/s": round(sbs.total * k / dts / 2**30, 2), "kernel_ms_this_rank": round(ms, 4),
                "hbm_read_frac_this_rank": round(sbs.n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "verified": vs}
            sbs.release()
            del sbs

    if rank == 0 and world == 1:
        sb.release()
        if not args.no_extra:
            extra = {}
            for name in ("16x32_noquote", "16x32_q10", "1024x4_dense"):
                del_sb = ShardBench(pkg, device, name, 1 << 30, 0, 1)
                r = sharded.result_from_words(del_sb.run_pass(0).tolist())
                ms = kernel_time_ms(del_sb, 20)
                extra[name] = {"bytes": del_sb.n, "entries": r.count, "kernel": del_sb.ctx.kernel_name(), "kernel_ms": round(ms, 4),
                               "GiB/s": round(del_sb.n / (ms * 1e-3) / 2**30, 2),
                               "hbm_read_frac": round(del_sb.n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "read_plus_tape_write_GBps": round((del_sb.n + 8 * r.count) / (ms * 1e-3) / 1e9, 1)}
                if name == "1024x4_dense":
                    extra[name]["probed_stream"] = dense_ceiling(del_sb)
                if not args.no_verify:
                    del_sb.run_pass(0)
                    ok, _, _ = torch_reference_compare(del_sb, 0, r.count)
                    extra[name]["verified"] = bool(ok and oracle_windows_compare(oracle, del_sb, 0, r.count))
                    failed = failed or not extra[name]["verified"]
                del_sb.release()
                del del_sb
            out["other_workloads"] = extra
            out["batch_many_files"] = batch_leg(pkg, device)
            out["batch_many_files"]["at_8_GiB_per_batch"] = batch_leg(pkg, device, k=64)
            failed = failed or not (out["batch_many_files"]["verified"] and out["batch_many_files"]["at_8_GiB_per_batch"]["verified"])
            out["back_to_back_1GiB"] = back_to_back_leg(pkg, device)
            failed = failed or not out["back_to_back_1GiB"]["verified"]
            out["consumers"] = consumers_leg(pkg, oracle, device)
            failed = failed or not out["consumers"]["verified"]
            out["consumers_at_1GiB"] = consumers_large_leg(pkg, device)
            failed = failed or not out["consumers_at_1GiB"]["verified"]
        if not args.no_ingest:
            out["latency"] = latency_leg(pkg, oracle, device)
            failed = failed or not out["latency"]["verified"]
            out["ingest"] = ingest_leg(pkg, device, sample, main_width, closed_form=(not main_q and main_lo == 0))
            failed = failed or not out["ingest"]["verified"]
            out["file_ingest"] = file_ingest_leg(pkg, device, sample, main_width, closed_form=(not main_q and main_lo == 0),
                                                 buffer_ingest=out["ingest"])
            failed = failed or not out["file_ingest"]["verified"]
            out["small_files"] = small_files_leg(pkg, oracle, device)
            failed = failed or not out["small_files"]["verified"]
    # ---- the CPU beside it, same bytes, same run, at EVERY N (rank 0's host cores) -------------------------------
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(oracle, sample)
        out.update(cpu_baseline_variants(oracle, sample, main_width))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: VERIFICATION FAILED (see `verified` in the JSON line)")


if __name__ == "__main__":
    main()
