"""Chunk-sharded stage 1 across ranks (one process per GPU, torch.distributed over RCCL/xGMI).

New relative to the reference, which is single-threaded (its README.md:24 lists "splitting work
without first knowing record breaks" as a TODO).  The file is cut into contiguous byte ranges,
one per rank.  Every rank indexes its range speculatively as if it were entered outside a quoted
string and obtains the composable shard descriptor

    (quote_parity, count_enter_outside, count_enter_inside)

— the same two quantities the reference carries between 64-byte blocks (`inside_str`,
`array_idx`: src/reader.rs:217-218).  ONE all-gather of the 64-byte result record per rank stitches
them (RCCL has no exclusive scan; the payload is latency-bound, so the 7 x ~153 GB/s xGMI links
are irrelevant).  Rank 0 knows how the file starts; the other ranks either speculate (first_state 0) or let the kernel
choose the entering state its first eight tiles (2 MiB) speak for (first_state ENTER_GUESS).  Only a rank whose first pass turns out to
have used the wrong state re-emits its shard (never on quote-free corpora; with ENTER_GUESS not on quoted CSV either).  The tape stays sharded in rank order with
absolute offsets: concatenating the shards, after the sentinel 0, is the reference's tape.
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist

from . import ShardResult, Stitch, stitch_shards


def shard_range(total_len: int, rank: int, world: int, align: int = 64, skew: int = 0) -> Tuple[int, int]:
    """Contiguous byte range [begin, end) of `rank`: i*N/world rounded down to `align`, optionally
    displaced by `skew` bytes (the deliberately misaligned variant of SURVEY.md §8d)."""
    def cut(i: int) -> int:
        if i <= 0:
            return 0
        if i >= world:
            return total_len
        c = (i * total_len // world) // align * align + skew
        return min(max(c, 0), total_len)
    return cut(rank), cut(rank + 1)


def result_from_words(h) -> ShardResult:
    """csvsimd_shard_result from its 8 x int64 image (the layout of include/csvsimd.h)."""
    h = [int(x) for x in h]
    r = ShardResult()
    r.count, r.count_enter_outside, r.count_enter_inside = h[0], h[1], h[2]
    r.quote_parity = h[3] & 0xFFFFFFFF
    r.in_quote_out = (h[3] >> 32) & 0xFFFFFFFF
    r.error = h[4] & 0xFFFFFFFF
    r.escape_out = (h[4] >> 32) & 0xFFFFFFFF
    r.written = h[5]
    r.in_quote_in_used = h[6] & 0xFFFFFFFF
    return r


def words_from_result(r: ShardResult) -> List[int]:
    return [r.count, r.count_enter_outside, r.count_enter_inside, r.quote_parity | (r.in_quote_out << 32),
            r.error | (r.escape_out << 32), r.written, r.in_quote_in_used, 0]


STITCH_WORDS = 8  # csvsimd_stitch as int64 words (5 used: in_quote_in | in_quote_final << 32, count, base, total,
                  # error), padded so that every part of a step's device block starts on a 64-byte boundary


def stitch_from_words(h) -> Stitch:
    h = [int(x) for x in h]
    st = Stitch()
    st.in_quote_in = h[0] & 0xFFFFFFFF
    st.in_quote_final = (h[0] >> 32) & 0xFFFFFFFF
    st.count, st.tape_index_base, st.total_entries = h[1], h[2], h[3]
    st.error = h[4] & 0xFFFFFFFF
    st.reemit = (h[4] >> 32) & 0xFFFFFFFF
    return st


class _StepSlot:
    """Device block + pinned host image + completion event of one step in flight."""

    def __init__(self, device: torch.device, world: int):
        # one device block so a single copy carries everything the host wants: [mine 8 | stitch 8 | all 8w]
        self.d_block = torch.zeros(8 + STITCH_WORDS + 8 * world, dtype=torch.int64, device=device)
        self.d_result = self.d_block[0:8]
        self.d_stitch = self.d_block[8:8 + STITCH_WORDS]
        self.d_all = self.d_block[8 + STITCH_WORDS:]
        self.h_block = torch.zeros_like(self.d_block, device="cpu")
        self.event = None
        self.first_done = None   # overlap_tail: the first pass has been enqueued on the caller's stream up to here
        if device.type == "cuda":
            self.h_block = self.h_block.pin_memory()
            self.event = torch.cuda.Event()
            self.first_done = torch.cuda.Event()
        self.err = None
        self.pending = False


class ShardedStep:
    """Buffers of one rank's sharded step, allocated once: the step itself allocates nothing and never
    waits for the host until its single copy-out at the end.

        launch(first_state)            speculative stage-1 pass, record -> d_result       (caller's C-ABI call)
        all_gather_into_tensor         ONE collective, device to device (RCCL over xGMI)
        stitch_shards_device_async     one-lane kernel: entering state / tape base / totals, on the device
        reemit(d_stitch)               stage-1 launch that reads its entering state from device memory and
                                       returns at once unless it is 1                     (caller's C-ABI call)
        copy-out                       [final record | stitch | all records] -> pinned host, one synchronise

    run() is the whole step.  enqueue() / collect() are its two halves: everything up to the copy-out is enqueued
    without waiting for the host, so with depth >= 2 the next step (another file, or the next batch of this one, into
    another tape buffer) can be enqueued before the previous one's records are read — the GPU never idles between
    steps.  Every rank must enqueue and collect in the same order (the all-gather is a collective).

    overlap_tail: everything behind the first pass — all-gather, stitch kernel, re-emit launch, copy-out — is enqueued
    on a SECOND stream that waits for the first pass only.  The caller's stream is free for the next step's first pass at
    once; the tail (tens of microseconds of latencies, no bandwidth) runs beside it: in practice when that next pass
    drains, a persistent stage-1 grid leaving no wave slot free before.  A step then completes one step late, so the
    caller keeps depth >= 3 steps in flight; the re-emit launch must use ANOTHER context than the first pass (its
    scratch would be shared with the next step's first pass otherwise) — reemit() is called with that stream current.
    """

    def __init__(self, device: torch.device, group=None, gather_via_host: bool = False, depth: int = 1,
                 overlap_tail: bool = False):
        # gather_via_host: development rehearsal on a one-GPU box (several ranks share the card, the group is
        # gloo): the records make the trip through host memory; stitch kernel and re-emit launch are the real ones
        self.gather_via_host = gather_via_host
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.slots = [_StepSlot(device, self.world) for _ in range(max(1, depth))]
        # A HIGH-PRIORITY stream: HIP multiplexes a process's streams onto a few hardware queues (four by default), and
        # two streams that share one are served in submission order — the tail of step i would simply run before the
        # first pass of step i + 1 again (seen in round 4's first kernel trace: every kernel on one queue id).  Streams
        # of another priority live on other hardware queues than the caller's, and the tail's few small kernels are
        # dispatched ahead of a persistent grid's workgroups whenever a slot frees up.
        self.tail = torch.cuda.Stream(device, priority=-1) if (overlap_tail and device.type == "cuda") else None
        # slot 0 under the names a depth-1 caller uses
        s0 = self.slots[0]
        self.d_block, self.d_result, self.d_stitch, self.d_all, self.h_block = (s0.d_block, s0.d_result, s0.d_stitch,
                                                                                s0.d_all, s0.h_block)

    def stitch_via_host(self, file_in_quote_in: int = 0, slot: int = 0) -> None:
        """CPU rehearsal of the stitch kernel (gloo groups, tensors in host memory): the same arithmetic
        through csvsimd_stitch_shards, written where the kernel would write it."""
        sl = self.slots[slot]
        host = sl.d_all.tolist()
        results = [result_from_words(host[8 * i: 8 * i + 8]) for i in range(self.world)]
        st = stitch_shards(results, self.rank, file_in_quote_in)
        sl.d_stitch.copy_(torch.tensor([st.in_quote_in | (st.in_quote_final << 32), st.count, st.tape_index_base,
                                        st.total_entries, st.error | (st.reemit << 32), 0, 0, 0], dtype=torch.int64))

    def enqueue(self, launch: Callable[[int], None], reemit: Callable[[int], None], file_in_quote_in: int = 0,
                rehearsal: bool = False, slot: int = 0, first_state: int = 0) -> None:
        """Everything of a step except waiting for it.  launch(0) must enqueue the speculative pass with its record
        going to self.slots[slot].d_result; reemit(d_stitch_ptr) must enqueue csvsimd_stage1_reemit_device_async
        into the same tape / record."""
        from . import stitch_shards_device_async
        sl = self.slots[slot]
        if sl.pending:
            raise RuntimeError("ShardedStep: slot enqueued again before it was collected")
        sl.err = None
        try:
            launch(first_state)   # 0 = speculate "entered outside"; ENTER_GUESS = the kernel chooses from its first eight tiles
        except Exception as e:  # still join the collective: the peers are about to block in it
            sl.err = e
            sl.d_result.zero_()
            sl.d_result[4] = 1  # error flag set: every rank will report the failure
        if self.tail is not None:
            sl.first_done.record(torch.cuda.current_stream(self.device))
            self.tail.wait_event(sl.first_done)
        with (torch.cuda.stream(self.tail) if self.tail is not None else contextlib.nullcontext()):
            if self.gather_via_host:
                h_all = torch.empty(8 * self.world, dtype=torch.int64)
                dist.all_gather_into_tensor(h_all, sl.d_result.cpu(), group=self.group)
                sl.d_all.copy_(h_all)
            else:
                dist.all_gather_into_tensor(sl.d_all, sl.d_result, group=self.group)
            if rehearsal:
                self.stitch_via_host(file_in_quote_in, slot)
            else:
                stream = torch.cuda.current_stream(self.device).cuda_stream
                stitch_shards_device_async(sl.d_all.data_ptr(), self.world, self.rank, file_in_quote_in,
                                           sl.d_stitch.data_ptr(), stream)
            if sl.err is None:
                reemit(sl.d_stitch.data_ptr())
            sl.h_block.copy_(sl.d_block, non_blocking=True)
            if sl.event is not None:
                sl.event.record(torch.cuda.current_stream(self.device))
        sl.pending = True

    def collect(self, slot: int = 0) -> Tuple[Stitch, ShardResult, List[ShardResult]]:
        """Waits for the step enqueued in `slot` (its only synchronisation) and returns
        (stitch, this rank's final record, every rank's speculative record)."""
        sl = self.slots[slot]
        if not sl.pending:
            raise RuntimeError("ShardedStep: nothing enqueued in this slot")
        if sl.event is not None:
            sl.event.synchronize()
        sl.pending = False
        if sl.err is not None:
            raise sl.err
        h = sl.h_block.tolist()
        st = stitch_from_words(h[8:8 + STITCH_WORDS])
        final = result_from_words(h[0:8])
        if st.error or final.error:
            raise RuntimeError("stage 1 reported an internal error on some rank (no rank has a valid tape)")
        return st, final, [result_from_words(h[8 + STITCH_WORDS + 8 * i: 16 + STITCH_WORDS + 8 * i])
                           for i in range(self.world)]

    def run(self, launch: Callable[[int], None], reemit: Callable[[int], None], file_in_quote_in: int = 0,
            rehearsal: bool = False, first_state: int = 0) -> Tuple[Stitch, ShardResult, List[ShardResult]]:
        """One whole step in slot 0: enqueue, then collect."""
        self.enqueue(launch, reemit, file_in_quote_in, rehearsal, 0, first_state)
        return self.collect(0)


def index_sharded(launch: Callable[[int], None], d_result: torch.Tensor, group=None,
                  file_in_quote_in: int = 0, first_state: int = 0) -> Tuple[Stitch, ShardResult, bool]:
    """One sharded stage-1 step with the stitch on the HOST (kept for callers without a device-side
    re-emit, and as the arithmetic the gloo tests compare the device stitch with):
    launch(first_state) -> ONE all-gather of the result records -> copy to the host -> csvsimd_stitch_shards ->
    launch(true state) only if the first pass ran with another entering state than the true one
    (first_state: 0 = speculate "outside", ENTER_GUESS = let the kernel choose from the shard's first eight tiles).
    Returns (stitch, final result of this rank, re_emitted).  Every rank joins the collective even if its
    own launch raises."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    err = None
    try:
        launch(first_state)
    except Exception as e:
        err = e
        d_result.zero_()
        d_result[4] = 1
    gathered = torch.empty(8 * world, dtype=torch.int64, device=d_result.device)
    dist.all_gather_into_tensor(gathered, d_result, group=group)
    if err is not None:
        raise err
    host = gathered.cpu().tolist()
    results = [result_from_words(host[8 * i: 8 * i + 8]) for i in range(world)]
    for i, r in enumerate(results):
        if r.error:
            raise RuntimeError(f"stage 1 reported an internal error on rank {i}")
    st = stitch_shards(results, rank, file_in_quote_in)
    if st.reemit:
        launch(st.in_quote_in)
        final = result_from_words(d_result.cpu().tolist())
        if final.error:
            raise RuntimeError("stage 1 reported an internal error on the re-emit pass")
        return st, final, True
    return st, results[rank], False
