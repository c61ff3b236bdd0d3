#!/usr/bin/env python3
"""dev tool: time every tuning variant (csv-simd_amd/csrc/variants/*.so) in its own subprocess."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
out = {}
for name in sys.argv[1].split(","):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, int(float(os.environ.get("SWEEP_GIB", "4")) * 2**30))
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 64
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    ctx.reserve(n)
    s = torch.cuda.current_stream().cuda_stream
    r = ctx.stage1_index_device(dbuf.data_ptr(), n, 0, 0, dtape.data_ptr(), cap)
    ok = r.count == n // (width + 1)
    ms = min(ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 10) for _ in range(3))
    msc = ctx.stage1_time_device(dbuf.data_ptr(), n, 0, 0, dres.data_ptr(), s, 2, 10)
    out[name] = {"ok": ok, "emit_TBps": round(n / ms / 1e9, 3), "count_TBps": round(n / msc / 1e9, 3)}
    del dtape, dbuf
print(json.dumps(out))
''' % ROOT
names = sys.argv[1] if len(sys.argv) > 1 else "64x31_noquote"
libs = sorted(glob.glob(os.path.join(ROOT, "csv-simd_amd", "csrc", "variants", "*.so")))
for lib in libs:
    env = dict(os.environ, CSVSIMD_LIB=lib)
    p = subprocess.run([sys.executable, "-c", code, names], env=env, capture_output=True, text=True, timeout=120)
    print(os.path.basename(lib), p.stdout.strip() or p.stderr.strip()[-300:], flush=True)
