"""dev tool: phase stamps of the two colfreq kernels (a -DCSVSIMD_CF_TRACE build, CSVSIMD_LIB=.../libcftrace.so).
argv[1]: few | mid | distinct.  Prints, per pass, the mean / max over workgroups of every stamp relative to the
earliest pass-1 start of the call, in microseconds (s_memrealtime = 100 MHz)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
lib = ctypes.CDLL(os.environ["CSVSIMD_LIB"])
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
def call():
    pkg.columnar_frequency_device_async(ctx, col.data_ptr(), 0, n, stride, 0, scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0], st.data_ptr(), s)
for _ in range(20): call()
torch.cuda.synchronize()
buf = np.zeros(2 * 4096 * 8, dtype=np.uint64)
names = (("start", "hashed", "probed", "barrier", "sorted", "written", "probed b0", "hashed b1"),
         ("start", "total", "prefix", "merged", "reserved", "end p0", "end all"))
for flush in (False, True):
    if flush:
        junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev); junk.fill_(1); del junk
    torch.cuda.synchronize()
    lib.csvsimd_dev_cf_trace(buf.ctypes.data_as(ctypes.c_void_p), 1)
    call(); torch.cuda.synchronize()
    lib.csvsimd_dev_cf_trace(buf.ctypes.data_as(ctypes.c_void_p), 0)
    tr = buf.reshape(2, 4096, 8).astype(np.int64)
    t0 = tr[0][tr[0][:, 0] > 0][:, 0].min()
    print(case, "cache flushed before the call" if flush else "hot (the column was just read)", "status", st.cpu().tolist())
    for p in (0, 1):
        rows = tr[p][tr[p][:, 0] > 0]
        print("  pass", p + 1, "workgroups", len(rows))
        for i, nm in enumerate(names[p]):
            v = rows[:, i]; v = v[v > 0]
            if len(v): print("    %-9s min %6.2f mean %6.2f max %6.2f us  (n=%d)" % (nm, (v.min() - t0) / 100, (v.mean() - t0) / 100, (v.max() - t0) / 100, len(v)))
