#!/usr/bin/env python3
"""dev tool: kernel time vs alignment of the input pointer (and of the tape pointer)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
n = pkg.workload_len("64x31_noquote", 4 << 30)
dbuf = torch.empty(n + 4096, dtype=torch.uint8, device="cuda:0")
cap = n // (width + 1) + 64
dtape = torch.empty(cap + 64, dtype=torch.int64, device="cuda:0")
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ctx.reserve(n + 4096)
s = torch.cuda.current_stream().cuda_stream
out = {}
for mis in (0, 16, 48, 64, 128, 256, 777, 1024):
    pkg.synth_fill_device(dbuf.data_ptr() + (mis & ~3), 0, n, cols, width, seed, q)
    ms = min(ctx.stage1_time_device(dbuf.data_ptr() + (mis & ~3), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 10) for _ in range(3))
    out[f"in+{mis & ~3}"] = round(n / ms / 1e9, 3)
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
for tmis in (8, 24, 40):
    ms = min(ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr() + tmis, cap, dres.data_ptr(), s, 2, 10) for _ in range(3))
    out[f"tape+{tmis}"] = round(n / ms / 1e9, 3)
print(json.dumps(out))
