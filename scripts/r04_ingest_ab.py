"""dev tool, run ON the GPU box: csvsimd_stage1_index on 2 GiB with and without a tape (count only: no narrow kernel, no
expander), rates of every call"""
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
ctx.read_into(host, tape)
for label, t in (("adaptive", tape), ("count only", None), ("adaptive", tape)):
    if t is not None:
        os.environ["CSVSIMD_INGEST_NARROW_WGS"] = label
        label = "narrow wgs " + label
    rates = []
    for _ in range(6):
        t0 = time.perf_counter(); rc, tl, _ = ctx.read_into(host, t); dt = time.perf_counter() - t0
        rates.append(n / dt / 2**30)
    ph = pkg.ingest_last_phases()
    print(f"{label:10s}: GiB/s", " ".join(f"{r:.1f}" for r in rates), {k: round(v * 1e3, 2) for k, v in ph.items() if isinstance(v, float)})
ctx.close()
