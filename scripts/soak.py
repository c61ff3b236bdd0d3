#!/usr/bin/env python3
"""dev tool: time-bounded soak of the stage-1 kernel variants — every launch's count and
order-sensitive tape checksum must equal the first launch's (which is checked against the analytic /
oracle value by the tests).  Looks for rare scheduling-dependent faults (look-back races)."""
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    cases = []
    for name, gib in (("16x32_q10", 1.0), ("64x31_q10", 2.0), ("1024x4_dense", 0.5), ("64x31_noquote", 4.0)):
        cols, width, seed, q = pkg.WORKLOADS[name]
        n = pkg.workload_len(name, int(gib * 2**30)) - 7      # ragged tail
        dbuf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        pkg.synth_fill_device(dbuf.data_ptr(), 0, n + 16 - (n + 16) % 4, cols, width, seed, q)
        cap = n // (width + 1) + 64
        dtape = torch.empty(cap, dtype=torch.int64, device=dev)
        for dialect in (None, pkg.Dialect(",", '"', "\\"), pkg.Dialect(";", "'")):
            for mis in (0, 5):
                cases.append((name, n, dbuf, dtape, cap, dialect, mis))
    dres = torch.zeros(8, dtype=torch.int64, device=dev)
    dsum = torch.zeros(2, dtype=torch.int64, device=dev)
    ref = {}
    hist = collections.Counter()
    t_end = time.time() + seconds
    launches = 0
    t_say = time.time() + 60
    while time.time() < t_end:
        if time.time() > t_say:   # a line a minute: a silent GPU command is taken for hung after seven
            print(f"# {launches} launches, {sum(v for k, v in hist.items() if k != 'ok')} mismatches", file=sys.stderr, flush=True)
            t_say = time.time() + 60
        for ci, (name, n, dbuf, dtape, cap, dialect, mis) in enumerate(cases):
            inq = launches % 3   # 0, 1, and 2 = CSVSIMD_ENTER_GUESS (the kernel's own choice must be the same every time too)
            # both instantiations in turn (round 4): the dense one must give the default one's record and checksum
            ctx.hint_density(1, 2) if (launches // 3) % 2 else ctx.hint_density(1, 1000)
            # ... on grids of every size (round 5: csvsimd_ctx_limit_workgroups; the signature must not depend on it)
            ctx.limit_workgroups((0, 0, 1, 2, 5, 64)[(launches // 6) % 6])
            if dialect is None:
                ctx.stage1_index_device_async(dbuf.data_ptr() + mis, n, 123, inq, dtape.data_ptr(), cap, dres.data_ptr())
            else:
                ctx.stage1_index_device_dialect_async(dialect, dbuf.data_ptr() + mis, n, 123, inq, dtape.data_ptr(), cap,
                                                      dres.data_ptr())
            r = pkg.ShardResult.from_buffer_copy(dres.cpu().numpy().tobytes())
            dsum.zero_()
            pkg.tape_checksum_device(dtape.data_ptr(), min(r.count, cap), 0, dsum.data_ptr())
            key = (ci, inq)
            sig = (r.count, r.in_quote_out, r.error, r.count_enter_outside, r.count_enter_inside, r.in_quote_in_used,
                   tuple(dsum.tolist()))
            if key not in ref:
                ref[key] = sig
            hist["ok" if sig == ref[key] else f"MISMATCH case {ci} {name} inq {inq}: {sig} != {ref[key]}"] += 1
            launches += 1
    # batched launches: slices of the corpora above as independent buffers (ragged sizes, both entering states), the
    # same batch again and again: every record and every tape checksum must repeat
    import numpy as np
    bt_end = time.time() + max(10.0, seconds / 4)
    name, n, dbuf, dtape, cap, _, _ = cases[0]
    sizes = [3 * 262144 + 17, 1, 262144, 5 * 262144 - 64, 0, 40 * 262144 + 4097, 777, 12 * 262144]
    offs = np.cumsum([0] + [s_ + 4096 for s_ in sizes])
    btapes = [torch.empty(max(s_, 1) // 8 + 64, dtype=torch.int64, device=dev) for s_ in sizes]
    items = [(dbuf.data_ptr() + int(offs[i]) + (i % 3), sizes[i], int(offs[i]), btapes[i].data_ptr(), btapes[i].numel(), i & 1)
             for i in range(len(sizes))]
    bres = torch.zeros((len(sizes), 8), dtype=torch.int64, device=dev)
    bref, batches = None, 0
    while time.time() < bt_end:
        if time.time() > t_say:
            print(f"# {batches} batched launches", file=sys.stderr, flush=True)
            t_say = time.time() + 60
        ctx.stage1_index_batch_device_async(items, bres.data_ptr())
        recs = bres.cpu().numpy().copy()
        sig = [recs.tobytes()]
        for i in range(len(sizes)):
            dsum.zero_()
            pkg.tape_checksum_device(btapes[i].data_ptr(), min(int(recs[i, 0]), btapes[i].numel()), 0, dsum.data_ptr())
            sig.append(tuple(dsum.tolist()))
        if bref is None:
            bref = sig
        hist["ok" if sig == bref else f"MISMATCH batch {batches}"] += 1
        batches += 1
    print(json.dumps({"seconds": seconds, "launches": launches, "batched_launches": batches, "cases": len(cases), "hist": dict(hist)}))
    return 0 if set(hist) == {"ok"} else 1


if __name__ == "__main__":
    sys.exit(main())
