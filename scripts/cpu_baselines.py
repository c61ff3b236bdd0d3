#!/usr/bin/env python3
"""CPU baseline table of BASELINE.md §2, timed on the host of the GPU box on the same bytes the GPU
scans (generated on the GPU, copied back).  All variants are the oracle's RESTATEMENTS of the
reference algorithm, never the reference binary (no Rust toolchain in this image).
liboracle_native.so is rebuilt here with -march=native for THIS host."""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
oracle = graft.load_oracle()
subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "liboracle_native.so"], check=True, capture_output=True)
nat = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_native.so"))
u64p = C.POINTER(C.c_uint64)
nat.oracle_sse_read.restype = C.c_int; nat.oracle_sse_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
nat.oracle_sse_read_mt.restype = C.c_int; nat.oracle_sse_read_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, u64p]
base = oracle.lib()

def best(fn, reps=3):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
    return t

out = {"host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for name in ("64x31_noquote", "16x32_q10"):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 2 << 30)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    host = oracle.aligned_copy(dbuf.cpu().numpy())
    S = n // (width + 1)
    tape = np.zeros(S + 200, dtype=np.uint64)
    cnt = C.c_uint64()
    res = {}
    res["ref_sse_1t (baseline flags, growing Vec)"] = n / best(lambda: oracle.sse_read_growing_timed(host)) / 2**30
    res["ref_sse_1t_native (-march=native, pre-reserved)"] = n / best(lambda: nat.oracle_sse_read(host.ctypes.data, n, tape.ctypes.data, tape.size, C.byref(cnt))) / 2**30
    assert cnt.value == S + 1
    for T in sorted({8, 16, len(os.sched_getaffinity(0))}):
        res[f"ref_sse_mt native, {T} threads"] = n / best(lambda: nat.oracle_sse_read_mt(host.ctypes.data, n, T, tape.ctypes.data, tape.size, C.byref(cnt))) / 2**30
        assert cnt.value == S + 1
    res["scalar_1t (byte loop)"] = n / best(lambda: base.oracle_scalar_read(host.ctypes.data, n, tape.ctypes.data, tape.size, C.byref(cnt)), 2) / 2**30
    out[name] = {k: round(v, 2) for k, v in res.items()}
print(json.dumps(out, indent=1))
