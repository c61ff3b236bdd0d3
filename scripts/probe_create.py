#!/usr/bin/env python3
"""dev tool (round 5): csvsimd_create(path) — file -> tape, the reference's entry point (src/lib.rs:61-74) — end to end,
next to csvsimd_stage1_index on the same bytes in a pageable buffer.  usage: probe_create.py [dir] [GiB]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
n = pkg.workload_len("64x31_noquote", int(gib * 2**30))
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
torch.cuda.synchronize()
host = dbuf.cpu().numpy()
del dbuf
path = os.path.join(d, "probe_create_%d.csv" % os.getpid())
t0 = time.perf_counter()
with open(path, "wb") as f:
    f.write(host.data)
out = {"dir": d, "bytes": n, "write_s": round(time.perf_counter() - t0, 3)}
try:
    ctx = pkg.Context(0)
    pitch = width + 1
    want_len = n // pitch + 1
    tape = np.empty(n // pitch + 64, dtype=np.uint64)
    ctx.read_into(host[: 64 << 20], tape)
    rows = []
    for i in range(4):
        t0 = time.perf_counter()
        rc, tl, _ = ctx.read_into(host, tape)
        dt = time.perf_counter() - t0
        rows.append({"ms": round(dt * 1e3, 2), "GiB/s": round(n / dt / 2**30, 2)})
        assert rc == 0 and tl == want_len
    out["buffer_ingest"] = rows
    out["buffer_ingest_phases"] = {k: (round(v * 1e3, 3) if isinstance(v, float) else v) for k, v in pkg.ingest_last_phases().items()}
    tape_fresh = []
    for i in range(2):   # a tape nobody has touched yet: what a caller who allocates per file pays
        t = np.empty(n // pitch + 64, dtype=np.uint64)
        t0 = time.perf_counter()
        rc, tl, _ = ctx.read_into(host, t)
        dt = time.perf_counter() - t0
        tape_fresh.append({"ms": round(dt * 1e3, 2), "GiB/s": round(n / dt / 2**30, 2)})
        del t
    out["buffer_ingest_fresh_tape"] = tape_fresh
    rows = []
    for i in range(4):
        t0 = time.perf_counter()
        tp = ctx.create(path)
        dt = time.perf_counter() - t0
        ph = pkg.ingest_last_phases()
        ok = tp.record_cnt == n // (cols * pitch) - 1 if hasattr(tp, "record_cnt") else None
        idx = tp.index()
        ok = bool(idx.size == want_len and idx[1] == width and idx[-1] == n - 1)
        t1 = time.perf_counter()
        tp.close()
        rows.append({"ms": round(dt * 1e3, 2), "GiB/s": round(n / dt / 2**30, 2), "ok": ok, "destroy_ms": round((time.perf_counter() - t1) * 1e3, 2),
                     "phases_ms": {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in ph.items()}})
    out["create"] = rows
finally:
    os.unlink(path)
print(json.dumps(out, indent=1))
