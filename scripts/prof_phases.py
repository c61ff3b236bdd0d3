#!/usr/bin/env python3
"""dev tool: per-phase wall-clock breakdown of the stage-1 kernel (timing build, DBG bit 3)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "64x31_noquote"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, int(gib * 2**30))
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
dbuf = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
cap = n // (width + 1) + 64
dtape = torch.empty(cap, dtype=torch.int64, device=dev)
dres = torch.zeros(8, dtype=torch.int64, device=dev)
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
os.environ["CSVSIMD_PROBE_MODE"] = "8"
for label, tp, c in (("EMIT", dtape.data_ptr(), cap), ("COUNT-ONLY", 0, 0)):
    ms = ctx.stage1_time_device(dbuf.data_ptr(), n, tp, c, dres.data_ptr(), s, 1, 1)
    print(f"== {label} {name}: {ms:.4f} ms, {n / ms / 1e9:.3f} TB/s, tile = {pkg.tile_bytes()} B", file=sys.stderr)
