"""dev tool: three csvsimd_stage1_index calls on 2 GiB (for rocprofv3 --memory-copy-trace)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
n = 2 << 30
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
d = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(d.data_ptr(), 0, n, cols, width, seed, q)
host = d.cpu().numpy()
tape = np.zeros(n // 32 + 64, dtype=np.uint64)
ctx = pkg.Context(0)
ctx.read_into(host[: 256 << 20], tape)
for _ in range(3):
    t0 = time.perf_counter(); rc, tl, _ = ctx.read_into(host, tape); dt = time.perf_counter() - t0
    print("GiB/s", n / dt / 2**30, {k: round(v * 1e3, 2) for k, v in pkg.ingest_last_phases().items() if isinstance(v, float)})
ctx.close()
