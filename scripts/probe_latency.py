#!/usr/bin/env python3
"""dev tool: end-to-end latency of the host-buffer entry point (csvsimd_stage1_index) by input size."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
oracle = graft.load_oracle()
ctx = pkg.Context(0)
cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
out = {}
for size in (1 << 10, 1 << 14, 1 << 17, 1 << 20, 1 << 23, 1 << 25, 1 << 26, 1 << 28):
    row = cols * (width + 1)
    n = max(row, size // row * row)
    host = np.frombuffer(bytes(oracle.synth(0, n, cols, width, seed, q)), dtype=np.uint8).copy()
    tape = np.empty(n // (width + 1) + 64, dtype=np.uint64)
    for _ in range(3):
        ctx.read_into(host, tape)
    reps = 200 if n < (1 << 22) else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        rc, cnt, _ = ctx.read_into(host, tape)
    dt = (time.perf_counter() - t0) / reps
    assert rc == 0
    out[str(n)] = {"us": round(dt * 1e6, 1), "GiB/s": round(n / dt / 2**30, 2)}
print(json.dumps(out))
