import os, sys, json, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
out = {}
for cols, width in ((1024, 2), (1024, 3), (1024, 4), (1024, 5), (1024, 6), (1024, 7), (1024, 8), (1024, 9), (1024, 11), (1024, 15)):
    row = cols * (width + 1)
    n = ((1 << 30) // row) * row
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, 12345, 0)
    S = n // (width + 1)
    dtape = torch.empty(S + 64, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    ctx.reserve(n)
    s = torch.cuda.current_stream().cuda_stream
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), S + 64)
    assert r.count == S
    ms = min(ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), S + 64, dres.data_ptr(), s, 2, 10) for _ in range(3))
    out[f"{cols}x{width}"] = {"ms": round(ms, 4), "read_TBps": round(n / ms / 1e9, 3), "total_TBps": round((n + 8 * S) / ms / 1e9, 3)}
    del dbuf, dtape
print(json.dumps(out))
