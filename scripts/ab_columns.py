#!/usr/bin/env python3
"""dev tool: A/B timing of csvsimd_chunk_to_columns_device over library variants (csv-simd_amd/csrc/variants/col_*.so,
e.g. `make OUT=variants/col_w16k.so EXTRA='-DCSVSIMD_COLWIN_BYTES=16384 -DCSVSIMD_COLWIN_ENTRIES=2048'`): 16x32 corpus
1 GiB, all 16 columns, stride 32; kernel time by torch events, best of 5 x 10 launches; result checked."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import __graft_entry__ as graft
pkg = graft.load_package()
dev = torch.device("cuda", 0)
name = "16x32_noquote"
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, 1 << 30)
dbytes = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(dbytes.data_ptr(), 0, n, cols, width, seed, q)
entries = n // (width + 1)
dindex = torch.zeros(entries + 2, dtype=torch.int64, device=dev)
ctx = pkg.Context(0)
r = ctx.stage1_index_device(dbytes.data_ptr(), n, 0, 0, dindex.data_ptr() + 8, entries + 1)
rows = r.count // cols
nrec = rows - 1
whole = (0, cols, rows * cols, nrec)
out = torch.empty((cols, nrec, 32), dtype=torch.uint8, device=dev)
lens = torch.empty((cols, nrec), dtype=torch.int32, device=dev)
def run():
    pkg.chunk_to_columns_device(ctx, dbytes.data_ptr(), n, dindex.data_ptr(), r.count + 1, cols, "LF", whole, None, out.data_ptr(), 32, lens.data_ptr())
for _ in range(20): run()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
table = dbytes[: rows * cols * (width + 1)].view(rows, cols, width + 1)[1:, :, :width]
ok = bool((lens == width).all()) and torch.equal(out, table.permute(1, 0, 2).contiguous())
total = (n - cols * (width + 1)) + 8 * (r.count + 1 - cols) + cols * nrec * 36
print(json.dumps({"ms": round(best, 4), "TBps_read_plus_write": round(total / best / 1e9, 3), "ok": ok}))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "csv-simd_amd", "csrc", "variants", "col_*.so")))
for rnd in range(2):
    for lib in libs:
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSVSIMD_LIB=lib), capture_output=True, text=True, timeout=300)
        print(rnd, os.path.basename(lib), p.stdout.strip() or p.stderr.strip()[-300:], flush=True)
