import os, sys, torch, collections
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
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
for mode in ("0", "4", "2", "6"):
    os.environ["CSVSIMD_PROBE_MODE"] = mode
    hist = collections.Counter()
    for rep in range(60):
        ctx.stage1_time_device(dbuf.data_ptr(), n, 0, 0, dres.data_ptr(), s, 0, 1)
        h = dres.cpu().tolist()
        hist[(h[0] - S, h[1] + h[2] - S)] += 1
    print("mode", mode, dict(hist), flush=True)
