"""Chunk-sharded stage 1 across ranks (one process per GPU, torch.distributed over RCCL/xGMI).

New relative to the reference, which is single-threaded (its README.md:24 lists "splitting work
without first knowing record breaks" as a TODO).  The file is cut into contiguous byte ranges,
one per rank.  Every rank indexes its range speculatively as if it were entered outside a quoted
string and obtains the composable shard descriptor

    (quote_parity, count_enter_outside, count_enter_inside)

— the same two quantities the reference carries between 64-byte blocks (`inside_str`,
`array_idx`: src/reader.rs:217-218).  ONE all-gather of the 64-byte result record per rank stitches
them (RCCL has no exclusive scan; the payload is latency-bound, so the 7 x ~153 GB/s xGMI links
are irrelevant).  Only a rank whose true entering state turns out to be "inside a string"
re-emits its shard (never on quote-free corpora).  The tape stays sharded in rank order with
absolute offsets: concatenating the shards, after the sentinel 0, is the reference's tape.
"""
from __future__ import annotations

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
    return r


def words_from_result(r: ShardResult) -> List[int]:
    return [r.count, r.count_enter_outside, r.count_enter_inside, r.quote_parity | (r.in_quote_out << 32),
            r.error | (r.escape_out << 32), r.written, 0, 0]


def index_sharded(launch: Callable[[int], None], d_result: torch.Tensor, group=None,
                  file_in_quote_in: int = 0) -> Tuple[Stitch, ShardResult, bool]:
    """One sharded stage-1 step for this rank.

    launch(in_quote_in) ENQUEUES stage 1 over this rank's byte range into its own tape shard, with
    the 64-byte csvsimd_shard_result going to `d_result` (8 x int64, on the device the collective
    runs on).  The step is: launch(0) -> ONE all-gather of the result records, device to device
    (RCCL over xGMI) -> one copy to the host (the only synchronisation) -> csvsimd_stitch_shards ->
    launch(1) only if this rank turns out to start inside a quoted string.
    Returns (stitch, final result of this rank, re_emitted)."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    launch(0)
    gathered = torch.empty(8 * world, dtype=torch.int64, device=d_result.device)
    dist.all_gather_into_tensor(gathered, d_result, group=group)
    host = gathered.cpu().tolist()
    results = [result_from_words(host[8 * i: 8 * i + 8]) for i in range(world)]
    for i, r in enumerate(results):
        if r.error:
            raise RuntimeError(f"stage 1 reported an internal error on rank {i}")
    st = stitch_shards(results, rank, file_in_quote_in)
    if st.in_quote_in:
        launch(1)
        final = result_from_words(d_result.cpu().tolist())
        if final.error:
            raise RuntimeError("stage 1 reported an internal error on the re-emit pass")
        return st, final, True
    return st, results[rank], False
