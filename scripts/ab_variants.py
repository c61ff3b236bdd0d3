#!/usr/bin/env python3
"""dev tool: A/B kernel timing of library variants (csv-simd_amd/csrc/variants/*.so), interleaved rounds, through the
few C-ABI entry points every round's library has (ctx, synth, time_device) — so older builds can be compared too."""
import ctypes as C, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes as C, json, os, sys, torch
L = C.CDLL(sys.argv[1])
W = {"64x31_noquote": (64, 31, 0xC5F00004, 0), "16x32_noquote": (16, 32, 0xC5F00002, 0), "16x32_q10": (16, 32, 0xC5F00003, 10),
     "1024x4_dense": (1024, 4, 0xC5F00005, 0), "64x31_q10": (64, 31, 0xC5F00004, 10)}
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
    best = 1e9
    # settle: an idle GPU runs its first ~10 ms of load through a power-management transient (DESIGN.md §4)
    L.csvsimd_stage1_time_device(ctx, vp(dbuf.data_ptr()), u64(n), vp(dtape.data_ptr()), u64(cap), vp(dres.data_ptr()), None, 0, max(8, int(0.03 / (n / 4.5e12))), C.byref(ms))
    for _ in range(3):
        rc = L.csvsimd_stage1_time_device(ctx, vp(dbuf.data_ptr()), u64(n), vp(dtape.data_ptr()), u64(cap), vp(dres.data_ptr()), None, 2, 10, C.byref(ms))
        assert rc == 0, rc
        best = min(best, ms.value)
    cnt = int(dres[0])
    chk = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    assert L.csvsimd_tape_checksum_device(vp(dtape.data_ptr()), u64(cnt), u64(1), vp(chk.data_ptr()), None) == 0
    out[spec] = [round(best, 4), round(n / best / 1e6 / 8000 * 100, 2), cnt, "%016x" % (int(chk[0]) & (2**64 - 1))]
    del dtape, dbuf
print(json.dumps(out))
'''
specs = sys.argv[1] if len(sys.argv) > 1 else "64x31_noquote:8,16x32_noquote:1,16x32_q10:1,1024x4_dense:1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
libs = sorted(glob.glob(os.path.join(ROOT, "csv-simd_amd", "csrc", "variants", "*.so")))
for r in range(rounds):
    for lib in libs:
        p = subprocess.run([sys.executable, "-c", code, lib, specs], capture_output=True, text=True, timeout=300)
        print(r, os.path.basename(lib), p.stdout.strip() or p.stderr.strip()[-400:], flush=True)
