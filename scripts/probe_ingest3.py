#!/usr/bin/env python3
"""dev tool (round 3): time breakdown of csvsimd_stage1_index (probe build: CSVSIMD_PROBE_INGEST_TIMES=1) with the tape
returned by D2H copies vs written by the kernel into the pinned slot, + NUMA facts of the box.
usage: CSVSIMD_LIB=csv-simd_amd/csrc/libcsvsimd_probes.so python scripts/probe_ingest3.py"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as graft
pkg = graft.load_package()
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
n = pkg.workload_len("64x31_noquote", 2 << 30)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
host = dbuf.cpu().numpy()
tape = np.empty(n // (width + 1) + 64, dtype=np.uint64)
ctx = pkg.Context(0)
ctx.read_into(host[: 64 << 20], tape)
for rep in range(4):
    t0 = time.perf_counter(); rc, tl, _ = ctx.read_into(host, tape); dt = time.perf_counter() - t0
    assert rc == 0
    print(f"  call {rep}: {dt * 1e3:.2f} ms = {n / dt / 2**30:.2f} GiB/s", flush=True)
''' % ROOT
for mode in ("adaptive", "0", "1"):
    env = dict(os.environ, CSVSIMD_PROBE_INGEST_TIMES="1")
    if mode != "adaptive":
        env["CSVSIMD_PROBE_ZEROCOPY_TAPE"] = mode
    print(f"== tape mode: {mode} (0 = D2H copies, 1 = kernel writes the pinned slot)", flush=True)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=200)
    print(p.stdout, p.stderr[-3000:], flush=True)
import glob
facts = {}
for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    facts[os.path.basename(nd)] = open(nd + "/cpulist").read().strip()
for d in glob.glob("/sys/class/drm/card*/device/numa_node"):
    facts[d.split("/")[4] + "_numa"] = open(d).read().strip()
facts["affinity"] = sorted(os.sched_getaffinity(0))[:40]
print(json.dumps(facts))
