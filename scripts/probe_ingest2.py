#!/usr/bin/env python3
"""dev tool (round 3): where does the host-buffer entry point lose its last 10 % to the H2D probe?  Times
csvsimd_stage1_index on 2 GiB of the 64x31 corpus with and without a tape (count-only: no D2H, no unload copies) and
for several chunk sizes (CSVSIMD_INGEST_CHUNK_MIB, read per call), next to a pinned H2D copy of the same bytes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
n = pkg.workload_len("64x31_noquote", 2 << 30)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
host = dbuf.cpu().numpy()
tape = np.empty(n // (width + 1) + 64, dtype=np.uint64)
none = np.empty(0, dtype=np.uint64)
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty_like(pin, device="cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h2d = 1e9
for _ in range(4):
    e0.record(); dst.copy_(pin, non_blocking=True); e1.record(); e1.synchronize()
    h2d = min(h2d, e0.elapsed_time(e1))
h2d_gib = pin.numel() / (h2d * 1e-3) / 2**30
print(f"H2D probe {h2d_gib:.2f} GiB/s")
for chunk in (None, 8, 16, 32):
    if chunk is None:
        os.environ.pop("CSVSIMD_INGEST_CHUNK_MIB", None)
    else:
        os.environ["CSVSIMD_INGEST_CHUNK_MIB"] = str(chunk)
    ctx = pkg.Context(0)
    ctx.read_into(host[: 64 << 20], tape)
    for label, t in (("with tape", tape), ("count only", none)):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            rc, tl, _ = ctx.read_into(host, t)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0 and tl == n // (width + 1) + 1
        g = n / best / 2**30
        print(f"chunk {chunk or 'default'} MiB  {label:10s}: {best * 1e3:7.2f} ms  {g:6.2f} GiB/s  = {g / h2d_gib:.3f} of the H2D probe")
    ctx.close()
