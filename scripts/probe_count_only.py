#!/usr/bin/env python3
"""dev tool: the emitting kernel vs the count-only instantiation (tape = NULL) on the quoted 8 GiB shard: what a
parity-only pre-pass before the all-gather would cost ("measure both")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
pkg = importlib.import_module("csv-simd_amd")
cols, width, seed = 64, 31, 0xC5F00004
n = (8 << 30) // 2048 * 2048
dev = torch.device("cuda:0")
ctx = pkg.Context(0)
dbuf = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, 10)
cap = n // 32 * 5 // 4 + 1024
dtape = torch.empty(cap, dtype=torch.int64, device=dev)
dres = torch.zeros(8, dtype=torch.int64, device=dev)
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
for name, tp, c in (("emit", dtape.data_ptr(), cap), ("count only (no tape)", 0, 0)):
    ms = ctx.stage1_time_device(dbuf.data_ptr(), n, tp, c, dres.data_ptr(), s, warmup=16, iters=20)
    print(name, round(ms, 4), "ms", round(n / ms / 1e6 / 8000 * 100, 2), "% of 8 TB/s")
