"""dev tool: pacing knobs of the dense instantiation on the dense corpus (variants/xhook.so only)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
name = "1024x4_dense"
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, 1 << 30)
d = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(d.data_ptr(), 0, n, cols, width, seed, q)
cap = n // (width + 1) + 1024
t = torch.empty(cap, dtype=torch.int64, device=dev)
res = torch.zeros(8, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
ctx = pkg.Context(0); ctx.reserve(n); ctx.hint_density(1, 2)
for tok, prio, delay in ((1, 1, 0), (0, 1, 0), (0, 0, 0), (1, 0, 0), (2, 1, 0), (1, 1, 1), (1, 1, 2), (0, 1, 2), (1, 1, 0)):
    os.environ.update(CSVSIMD_X_TOKEN=str(tok), CSVSIMD_X_PRIO=str(prio), CSVSIMD_X_DELAY=str(delay))
    for _ in range(20): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ctx.stage1_index_device_async(d.data_ptr(), n, 0, 0, t.data_ptr(), cap, res.data_ptr(), s)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"token {tok} prio {prio} delay {delay}: {ms:.4f} ms  {n / ms / 1e6 / 8000 * 100:.2f} %", flush=True)
