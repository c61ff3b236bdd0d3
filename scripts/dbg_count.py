import os, sys, torch, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
L = pkg.lib()
L.csvsimd_debug_copy_scratch.restype = C.c_int
L.csvsimd_debug_copy_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
name = "16x32_noquote"
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, 1024 << 20)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
S = n // (width + 1)
T = 131072
ntiles = (n + T - 1) // T
# expected per-tile counts: delimiters at positions p = 33k+32
pos = np.arange(S, dtype=np.int64) * 33 + 32
per_tile = np.bincount(pos // T, minlength=ntiles)
expect_inc = np.cumsum(per_tile)
nbad = 0
for rep in range(200):
    rc = ctx.stage1_index_device(dbuf.data_ptr(), n)
    if rc.count != S:
        nbad += 1
        buf = np.zeros(528 // 8 + ntiles, dtype=np.uint64)
        L.csvsimd_debug_copy_scratch(ctx._h, buf.ctypes.data, buf.nbytes)
        desc = buf[66:66 + ntiles]
        lo = (desc & np.uint64(0xffffffff)).astype(np.int64); hi = (desc >> np.uint64(32)).astype(np.int64)
        status = np.where((lo >> 30) == (hi >> 30), lo >> 30, 0)
        x = (lo & 0x3fffffff) | ((hi & 0x3fffffff) << 30)
        cnt = x >> 1
        inc = status == 2
        wrong = np.nonzero(inc & (cnt != expect_inc))[0]
        print("rep", rep, "count", rc.count, "expected", S, "n_inclusive", int(inc.sum()), "n_agg", int((status == 1).sum()),
              "first wrong tile", wrong[:5], "delta", (cnt[wrong[:5]] - expect_inc[wrong[:5]]) if len(wrong) else None, flush=True)
        if len(wrong):
            w0 = wrong[0]
            for j in range(max(0, w0 - 3), min(ntiles, w0 + 3)):
                a = (int(x[j]) >> 1) & 0xffffff
                print("   tile", j, "status", status[j], "cnt/agg", cnt[j] if status[j] == 2 else a, "expect inc", expect_inc[j], "per_tile", per_tile[j])
        if nbad >= 3: break
print("bad runs", nbad)
