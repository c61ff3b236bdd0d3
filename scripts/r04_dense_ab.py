"""dev tool: default vs dense instantiation on three corpora (events around 30 launches), for the library in CSVSIMD_LIB"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
tag = os.path.basename(os.environ.get("CSVSIMD_LIB", "product"))
for name, gib in (("1024x4_dense", 1), ("16x32_noquote", 1), ("64x31_noquote", 8)):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, gib << 30)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    pkg.synth_fill_device(d.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 1024
    t = torch.empty(cap, dtype=torch.int64, device=dev)
    res = torch.zeros(8, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    want = torch.arange(1, cap - 1023, dtype=torch.int64, device=dev) * (width + 1) - 1
    for rep in range(2):
        for label, hint in (("default", (0, 0)), ("dense", (1, 2))):
            ctx = pkg.Context(0); ctx.reserve(n)
            ctx.hint_density(*hint)
            for _ in range(20): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 30
            ok = bool(torch.equal(t[: cap - 1024], want))
            print(f"{tag:10s} {name:14s} {label:8s} {ms:.4f} ms  {n / ms / 1e6 / 8000 * 100:.2f} % of 8 TB/s  ok={ok}", flush=True)
            ctx.close()
    del d, t, want
