import os, sys, json, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
out = {}
for name, gib in (("1024x4_dense", 1.0), ("64x31_noquote", 4.0)):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, int(gib * 2**30))
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 64
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    ctx.reserve(n)
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for label, mode, tp, c in (("emit", None, dtape.data_ptr(), cap), ("emit_nostore", "16", dtape.data_ptr(), cap), ("count_only", None, 0, 0)):
        if mode is None: os.environ.pop("CSVSIMD_PROBE_MODE", None)
        else: os.environ["CSVSIMD_PROBE_MODE"] = mode
        ms = ctx.stage1_time_device(dbuf.data_ptr(), n, tp, c, dres.data_ptr(), s, 2, 10)
        res[label] = round(ms, 4)
    out[name] = res
print(json.dumps(out))
