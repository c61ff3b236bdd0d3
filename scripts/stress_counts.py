import os, sys, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
for name in ("16x32_noquote", "16x32_q10"):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 1024 << 20)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    S = n // (width + 1)
    cap = S + 64
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    hist = collections.Counter()
    for rep in range(150):
        r = ctx.stage1_index_device(dbuf.data_ptr(), n)
        hist[("count-only", r.count - S, r.count_enter_outside + r.count_enter_inside - (r.count_enter_outside + r.count_enter_inside if q else S))] += 1
    ref = None
    for rep in range(60):
        r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap, allow_overflow=True)
        t = dtape[:S].clone()
        if ref is None: ref = t
        hist[("emit", r.count - S, int((t != ref).sum()))] += 1
    print(name, dict(hist), flush=True)
