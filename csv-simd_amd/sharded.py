"""Chunk-sharded stage 1 across ranks (one process per GPU, torch.distributed over RCCL/xGMI).

New relative to the reference, which is single-threaded (its README.md:24 lists "splitting work
without first knowing record breaks" as a TODO).  The file is cut into contiguous byte ranges,
one per rank.  Every rank indexes its range speculatively as if it were entered outside a quoted
string and obtains the composable shard descriptor

    (quote_parity, count_enter_outside, count_enter_inside)

— the same two quantities the reference carries between 64-byte blocks (`inside_str`,
`array_idx`: src/reader.rs:217-218).  ONE all-gather of 3 x int64 per rank stitches them (RCCL
has no exclusive scan; the payload is 24 B/rank, latency-bound, so the 7 x ~153 GB/s xGMI links
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


def all_gather_descriptors(local: ShardResult, device: torch.device, group=None) -> List[ShardResult]:
    """The one collective of the path: all-gather (parity, count_outside, count_inside)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([local.quote_parity, local.count_enter_outside, local.count_enter_inside],
                        dtype=torch.int64, device=device)
    out = torch.empty(3 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    host = out.cpu().tolist()
    res = []
    for i in range(world):
        r = ShardResult()
        r.quote_parity, r.count_enter_outside, r.count_enter_inside = host[3 * i: 3 * i + 3]
        res.append(r)
    return res


def index_sharded(run_pass: Callable[[int], ShardResult], device: torch.device, group=None,
                  file_in_quote_in: int = 0) -> Tuple[Stitch, ShardResult, bool]:
    """One sharded stage-1 step for this rank.

    run_pass(in_quote_in) runs stage 1 over this rank's byte range into its own tape shard and
    returns the ShardResult.  Returns (stitch, final result of this rank, re_emitted)."""
    rank = dist.get_rank(group)
    spec = run_pass(0)
    results = all_gather_descriptors(spec, device, group)
    st = stitch_shards(results, rank, file_in_quote_in)
    if st.in_quote_in:
        final = run_pass(1)
        return st, final, True
    return st, spec, False
