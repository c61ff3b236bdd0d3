"""dev tool: where does the dense instantiation start to win?  1 GiB corpora of 64 columns x W-byte fields (entries per byte
1 / (W + 1)), default vs dense instantiation, events around 30 launches each"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
for width in (2, 4, 6, 8, 10, 12, 16, 24, 31):
    cols = 64
    row = cols * (width + 1)
    n = (1 << 30) // row * row
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    pkg.synth_fill_device(d.data_ptr(), 0, n, cols, width, 0xC5F00099, 0)
    cap = n // (width + 1) + 1024
    t = torch.empty(cap, dtype=torch.int64, device=dev)
    res = torch.zeros(8, dtype=torch.int64, device=dev)
    out = []
    for label, hint in (("default", (0, 0)), ("dense", (1, 2))):
        ctx = pkg.Context(0); ctx.reserve(n); ctx.hint_density(*hint)
        best = None
        for rep in range(2):
            for _ in range(15): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 30
            best = ms if best is None else min(best, ms)
        out.append(best)
        ctx.close()
    print(f"width {width:2d}  entries/byte {1 / (width + 1):.3f}  default {out[0]:.4f} ms  dense {out[1]:.4f} ms  dense/default {out[1] / out[0]:.3f}", flush=True)
    del d, t
