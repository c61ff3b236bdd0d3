#!/usr/bin/env python3
"""dev tool: kernel time of the column search (equals / starts-with / contains) on 2 033 600 records of 32 bytes"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
n, stride = 2033600, 32
gen = torch.Generator(device=dev); gen.manual_seed(3)
col = torch.randint(97, 123, (n, stride), dtype=torch.uint8, device=dev, generator=gen)
lens = torch.randint(20, 33, (n,), dtype=torch.int32, device=dev, generator=gen)
col = torch.where(torch.arange(stride, device=dev)[None, :] < lens[:, None], col, torch.zeros_like(col)).contiguous()
bitmap = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
out = {}
for mode, name in ((0, "equals"), (1, "starts_with"), (2, "contains")):
    for needle in (b"abc", b"qzjxkvwy"):
        for _ in range(3): hits = pkg.columnar_search_device(ctx, col.data_ptr(), lens.data_ptr(), n, stride, needle, mode, bitmap.data_ptr())
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(20): hits = pkg.columnar_search_device(ctx, col.data_ptr(), lens.data_ptr(), n, stride, needle, mode, bitmap.data_ptr())
        dt = (time.perf_counter() - t0) / 20
        out[f"{name}:{needle.decode()}"] = {"call_us": round(dt * 1e6, 1), "hits": int(hits)}
print(json.dumps(out))
