#!/usr/bin/env python3
"""dev tool (CSVSIMD_LIB = a -DCSVSIMD_WG_END_TRACE build): when do the workgroups of ONE launch start, draw their last
tile and leave?  Three s_memrealtime stamps and a tile count per workgroup, written at exit: next to nothing is added to
the kernel.  usage: r04_wg_end_trace.py [workload] [GiB]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
lib = ctypes.CDLL(os.environ["CSVSIMD_LIB"])
name = sys.argv[1] if len(sys.argv) > 1 else "16x32_noquote"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, int(gib * 2**30))
ctx = pkg.Context(0)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
cap = int(n // (width + 1) * 1.25) + 1024
dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
ms = ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 8, 20)
torch.cuda.synchronize()
buf = np.zeros(2048 * 4, dtype=np.uint64)
assert lib.csvsimd_dev_wg_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
tr = buf.reshape(2048, 4).astype(np.int64)
tr = tr[tr[:, 2] > 0]
t0 = tr[:, 0].min()
start, last, end, tiles = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, tr[:, 3]
print(f"== {name} {gib} GiB: {ms*1e3:.1f} us per launch by events; last launch by stamps: {len(tr)} workgroups, first start 0, last exit {end.max():.1f} us")
print("start      us: p50 %.1f p99 %.1f max %.1f" % (np.percentile(start, 50), np.percentile(start, 99), start.max()))
print("last ticket us: p1 %.1f p50 %.1f p99 %.1f max %.1f" % (np.percentile(last, 1), np.percentile(last, 50), np.percentile(last, 99), last.max()))
print("exit       us: p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [1, 10, 50, 90, 99, 100])))
print("exit - last ticket us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(end - last, [10, 50, 90, 100])))
for k in sorted(set(tiles.tolist())):
    m = tiles == k
    print(f"  workgroups with {k} tiles: {m.sum():4d}  exit p50 {np.percentile(end[m], 50):.1f}  max {end[m].max():.1f}")
h, edges = np.histogram(end, bins=np.arange(np.floor(end.max()) - 40, np.floor(end.max()) + 2, 2.5))
print("exits per 2.5-us bin over the last 40 us:", h.tolist())
