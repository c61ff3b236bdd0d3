#!/usr/bin/env python3
"""dev tool: like ab_variants.py, but the launches cycle over K tape buffers (K = 1: every launch rewrites the same
tape, whose lines may still sit dirty in the 256 MB Infinity Cache; K = 4 of a 1 GiB corpus = 1 GB of tapes: every
launch writes lines the caches no longer hold).  Separates a real gain of a store policy from a replay artefact.
usage: ab_rotating.py [workload:gib] [K,K,...] [input copies]"""
import ctypes as C, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes as C, json, sys, torch
L = C.CDLL(sys.argv[1])
W = {"64x31_noquote": (64, 31, 0xC5F00004, 0), "16x32_noquote": (16, 32, 0xC5F00002, 0), "16x32_q10": (16, 32, 0xC5F00003, 10)}
name, gib = sys.argv[2].split(":")
cols, width, seed, q = W[name]
row = cols * (width + 1)
n = int(float(gib) * 2**30) // row * row
vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
ctx = C.c_void_p()
assert L.csvsimd_ctx_create(0, C.byref(ctx)) == 0
KIN = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # input copies cycled as well
dbufs = [torch.empty(n, dtype=torch.uint8, device="cuda:0") for _ in range(KIN)]
for dbuf in dbufs:
    assert L.csvsimd_synth_fill_device(vp(dbuf.data_ptr()), u64(0), u64(n), u32(cols), u32(width), u64(seed), u32(q), None) == 0
cap = int(n // (width + 1) * 1.25) + 1024
assert L.csvsimd_ctx_reserve(ctx, u64(n)) == 0
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
out = {}
for K in [int(k) for k in sys.argv[3].split(",")]:
    tapes = [torch.empty(cap, dtype=torch.int64, device="cuda:0") for _ in range(K)]
    s = torch.cuda.current_stream()
    def launch(i):
        rc = L.csvsimd_stage1_index_device_async(ctx, vp(dbufs[i % KIN].data_ptr()), u64(n), u64(0), u32(0), vp(tapes[i % K].data_ptr()), u64(cap), vp(dres.data_ptr()), vp(s.cuda_stream))
        assert rc == 0, rc
    settle = max(8, int(0.03 / (n / 4.5e12)))
    for i in range(settle): launch(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for i, (a, b) in enumerate(ev):
        a.record(s); launch(i); b.record(s)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    out["K=%d" % K] = [round(t[len(t) // 2], 4), round(n / t[len(t) // 2] / 1e6 / 8000 * 100, 2), int(dres[0])]
    del tapes
print(json.dumps(out))
'''
spec = sys.argv[1] if len(sys.argv) > 1 else "16x32_noquote:1"
ks = sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8"
kin = sys.argv[3] if len(sys.argv) > 3 else "1"
for r in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in sorted(glob.glob(os.path.join(ROOT, "csv-simd_amd", "csrc", "variants", "*.so"))):
        p = subprocess.run([sys.executable, "-c", code, lib, spec, ks, kin], capture_output=True, text=True, timeout=300)
        print(r, os.path.basename(lib), spec, p.stdout.strip() or p.stderr.strip()[-400:], flush=True)
