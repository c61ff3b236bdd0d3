"""dev tool: colfreq on 2.03 M x 32-byte records, one cardinality per run (argv[1]: few | distinct | mid)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
case = sys.argv[1]
n, stride = 2033600, 32
gen = torch.Generator(device=dev); gen.manual_seed(1)
vocab = torch.randint(0, 256, (n, stride), dtype=torch.uint8, device=dev, generator=gen)
k = {"few": 100, "mid": 10000, "distinct": n}[case]
col = vocab if k == n else vocab[torch.randint(0, k, (n,), device=dev, generator=gen)].contiguous()
ctx = pkg.Context(0)
need = pkg.columnar_frequency_scratch_bytes(n)
scratch = torch.empty(need, dtype=torch.uint8, device=dev)
ent = torch.empty((n + 8, 2), dtype=torch.int64, device=dev)
st = torch.zeros(4, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    pkg.columnar_frequency_device_async(ctx, col.data_ptr(), 0, n, stride, 0, scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0], st.data_ptr(), s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    pkg.columnar_frequency_device_async(ctx, col.data_ptr(), 0, n, stride, 0, scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0], st.data_ptr(), s)
e1.record(); e1.synchronize()
print(case, "status", st.cpu().tolist(), "ms per call", e0.elapsed_time(e1) / 30)
