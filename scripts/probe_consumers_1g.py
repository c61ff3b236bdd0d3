#!/usr/bin/env python3
"""dev tool (round 5): the column consumers at a size where a roofline fraction means something — 32 Mi records x 32 bytes
(1 GiB of column): search (equals / contains), frequency count with 100 / 10 000 / all-distinct values.  Device time by events.
usage: probe_consumers_1g.py [records]; PROBE_ONLY=search|freq, PROBE_KINDS=all_distinct,100,1000 (numbers of values),
PROBE_REPS=30 (search: best of), CSVSIMD_LIB=<variant .so> for A/B runs of tuning builds."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
dev = torch.device("cuda:0")
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 32 << 20
stride = 32
ctx = pkg.Context(0)
s_ = torch.cuda.current_stream(dev).cuda_stream
out = {"records": nrec, "stride": stride}

def device_time(fn, reps=5):
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        best = dt if best is None else min(best, dt)
    return best

g = torch.Generator(device=dev); g.manual_seed(1)
col = torch.randint(97, 123, (nrec, stride), dtype=torch.uint8, device=dev, generator=g)
need = pkg.columnar_frequency_scratch_bytes(nrec)
scratch = torch.empty(need, dtype=torch.uint8, device=dev)
ent = torch.empty((nrec + 8, 2), dtype=torch.int64, device=dev)
d_status = torch.zeros(4, dtype=torch.int64, device=dev)
alg = nrec * stride
kinds = [x for x in os.environ.get("PROBE_KINDS", "all_distinct,100,10000").split(",") if x]
for label, k in (() if os.environ.get("PROBE_ONLY") == "search" else [(x, 0 if x == "all_distinct" else int(x)) for x in kinds]):
    if k:
        pick = torch.randint(0, k, (nrec,), device=dev, generator=g)
        c = col[:k][pick].contiguous()
    else:
        c = col
    st = pkg.columnar_frequency_device(ctx, c.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
    t = device_time(lambda: pkg.columnar_frequency_device_async(ctx, c.data_ptr(), 0, nrec, stride, 0, scratch.data_ptr(), need,
                                                                ent.data_ptr(), ent.shape[0], d_status.data_ptr(), s_))
    ok = int(ent[: st.n_distinct, 1].sum()) == nrec and st.overflow == 0
    if k:
        ok = ok and st.n_distinct == len(torch.unique(pick))
    a = alg + st.n_distinct * 16
    out["colfreq_" + label] = {"ms": round(t * 1e3, 4), "distinct": int(st.n_distinct), "ok": bool(ok), "GBps_algorithmic": round(a / t / 1e9, 1),
                               "frac_of_8TBps": round(a / t / 8e12, 3)}
    if k:
        del c, pick
if os.environ.get("PROBE_ONLY") == "freq":
    print(json.dumps(out)); sys.exit(0)
bm = torch.zeros((nrec + 63) // 64 + 1, dtype=torch.int64, device=dev)
row = bytes(col[1000].cpu().numpy())
for label, needle, mode in (("contains_6", row[4:10], pkg.SEARCH_CONTAINS), ("contains_1", row[4:5], pkg.SEARCH_CONTAINS),
                            ("contains_12", row[14:26], pkg.SEARCH_CONTAINS), ("equals", row, pkg.SEARCH_EQUALS),
                            ("starts_with_5", row[:5], pkg.SEARCH_STARTS_WITH)):
    hits = pkg.columnar_search_device(ctx, col.data_ptr(), 0, nrec, stride, needle, mode, bm.data_ptr())
    ts = []
    for _ in range(int(os.environ.get("PROBE_REPS", "5"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg.columnar_search_device(ctx, col.data_ptr(), 0, nrec, stride, needle, mode, bm.data_ptr())
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    out["colsearch_" + label] = {"ms_wall": round(t * 1e3, 4), "hits": int(hits), "GBps_algorithmic": round(alg / t / 1e9, 1),
                                 "frac_of_8TBps": round(alg / t / 8e12, 3)}
# check contains_6 against torch on a slice
sl = col[: 1 << 20]
needle = torch.tensor(list(row[4:10]), dtype=torch.uint8, device=dev)
m = torch.zeros(sl.shape[0], dtype=torch.bool, device=dev)
for s in range(stride - 6 + 1):
    m |= (sl[:, s: s + 6] == needle).all(dim=1)
hits = pkg.columnar_search_device(ctx, col.data_ptr(), 0, 1 << 20, stride, row[4:10], pkg.SEARCH_CONTAINS, bm.data_ptr())
out["contains_check"] = bool(int(m.sum()) == hits)
if os.environ.get("PROBE_ONLY") == "search":
    print(os.environ.get("CSVSIMD_LIB", "product"), {k[10:]: v["ms_wall"] for k, v in out.items() if k.startswith("colsearch_")}, out["contains_check"])
else:
    print(json.dumps(out, indent=1))
