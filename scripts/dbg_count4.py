import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
name = "16x32_noquote"
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, 1024 << 20)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
S = n // (width + 1)
cap = S + 64
dtape = torch.full((cap,), -1, dtype=torch.int64, device="cuda:0")
want = torch.arange(1, S + 1, dtype=torch.int64, device="cuda:0") * (width + 1) - 1
nbad = 0
for rep in range(400):
    dtape.fill_(-1)
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap, allow_overflow=True)
    k = min(r.count, S)
    neq = (dtape[:k] != want[:k])
    if r.count != S or bool(neq.any()):
        nbad += 1
        i0 = int(neq.nonzero()[0]) if bool(neq.any()) else -1
        print("rep", rep, "count", r.count - S, "first bad idx", i0, "n bad", int(neq.sum()))
        if i0 >= 0:
            g = dtape[max(0, i0 - 3): i0 + 5].tolist(); w_ = want[max(0, i0 - 3): i0 + 5].tolist()
            print("  got ", g); print("  want", w_)
            p = w_[3] if i0 >= 3 else w_[i0]
            print("  want pos", p, "tile", p // 131072, "wave", (p % 131072) // 32768, "round", (p % 32768) // 4096, "stripe(lane)", (p % 4096) // 64, "bit", p % 64)
        if nbad >= 4: break
print("bad", nbad)
