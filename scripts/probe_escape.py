#!/usr/bin/env python3
"""dev tool (round 3): kernel time of the dialect variants on the 64x31 8 GiB and 16x32 1 GiB corpora through the probe
build's CSVSIMD_PROBE_DIALECT hook: reference dialect, another delimiter/quote, escape dialect with the hashed LUT
classification (<.., 3>) and with the direct compares (<.., 2>, CSVSIMD_PROBE_NO_HASHED_DIALECT=1).
usage: python scripts/probe_escape.py   (needs csv-simd_amd/csrc/libcsvsimd_probes.so)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes as C, json, os, sys, torch
L = C.CDLL(sys.argv[1])
W = {"64x31_noquote": (64, 31, 0xC5F00004, 0), "16x32_noquote": (16, 32, 0xC5F00002, 0)}
ctx = C.c_void_p()
assert L.csvsimd_ctx_create(0, C.byref(ctx)) == 0
out = {}
for spec in sys.argv[2].split(","):
    name, gib = spec.split(":")
    cols, width, seed, q = W[name]
    row = cols * (width + 1)
    n = int(float(gib) * 2**30) // row * row
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    assert L.csvsimd_synth_fill_device(vp(dbuf.data_ptr()), u64(0), u64(n), u32(cols), u32(width), u64(seed), u32(q), None) == 0
    cap = int(n // (width + 1) * 1.25) + 1024
    dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    assert L.csvsimd_ctx_reserve(ctx, u64(n)) == 0
    torch.cuda.synchronize()
    ms = C.c_float()
    L.csvsimd_stage1_time_device(ctx, vp(dbuf.data_ptr()), u64(n), vp(dtape.data_ptr()), u64(cap), vp(dres.data_ptr()), None, 0, max(8, int(0.03 / (n / 4.5e12))), C.byref(ms))
    best = 1e9
    for _ in range(3):
        assert L.csvsimd_stage1_time_device(ctx, vp(dbuf.data_ptr()), u64(n), vp(dtape.data_ptr()), u64(cap), vp(dres.data_ptr()), None, 2, 10, C.byref(ms)) == 0
        best = min(best, ms.value)
    out[spec] = [round(best, 4), round(n / best / 1e6 / 8000 * 100, 2), int(dres[0])]
    del dtape, dbuf
print(json.dumps(out))
'''
lib = os.path.join(ROOT, "csv-simd_amd", "csrc", "libcsvsimd_probes.so")
specs = "64x31_noquote:8,16x32_noquote:1"
cases = [("reference dialect", {}), ("delimiter ; quote '", {"CSVSIMD_PROBE_DIALECT": "59,39,0"}),
         ("escape \\ hashed <..,3>", {"CSVSIMD_PROBE_DIALECT": "44,34,92"}),
         ("escape \\ compares <..,2>", {"CSVSIMD_PROBE_DIALECT": "44,34,92", "CSVSIMD_PROBE_NO_HASHED_DIALECT": "1"})]
for rnd in range(2):
    for label, env in cases:
        p = subprocess.run([sys.executable, "-c", code, lib, specs], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        print(rnd, f"{label:28s}", p.stdout.strip() or p.stderr.strip()[-400:], flush=True)
