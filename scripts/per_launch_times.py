#!/usr/bin/env python3
"""dev tool: duration of every single launch of a back-to-back sequence (one HIP event pair per launch), to see
whether the average hides slow outliers.  usage: per_launch_times.py [gib] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
pkg = importlib.import_module("csv-simd_amd")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols, width, seed = 64, 31, 0xC5F00004
row = cols * (width + 1)
n = int(gib * 2**30) // row * row
dev = torch.device("cuda:0")
ctx = pkg.Context(0)
dbuf = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, 0)
cap = n // (width + 1) + 1024
dtape = torch.empty(cap, dtype=torch.int64, device=dev)
dres = torch.zeros(8, dtype=torch.int64, device=dev)
ctx.reserve(n)
torch.cuda.synchronize()
s = torch.cuda.current_stream(dev)
def launch():
    ctx.stage1_index_device_async(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap, dres.data_ptr(), s.cuda_stream)
for mode in ("back-to-back", "sync after each", "sync + 2 ms host sleep"):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launch)]
    for a, b in ev:
        a.record(s); launch(); b.record(s)
        if mode != "back-to-back": torch.cuda.synchronize()
        if mode.endswith("sleep"): time.sleep(0.002)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    print(mode, "min %.4f med %.4f mean %.4f max %.4f" % (min(t), sorted(t)[len(t) // 2], sum(t) / len(t), max(t)))
    if len(t) <= 48:
        print("  ", " ".join("%.3f" % x for x in t))
    else:  # means of groups of ten
        print("   x10:", " ".join("%.3f" % (sum(t[i:i + 10]) / len(t[i:i + 10])) for i in range(0, len(t), 10)))
