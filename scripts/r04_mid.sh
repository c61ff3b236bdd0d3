#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ingest or golden or config1 or end_to_end" 2>&1 | tail -3
python3 bench.py --only-latency 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['latency']
for r in d['sizes']: print(r)
print(d['crossover'])"
python3 - <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
ctx = pkg.Context(0)
for mib in (2, 4, 8, 16, 24, 32, 64, 100, 128, 256):
    host = oracle.aligned_copy(oracle.synth(0, mib << 20, cols, width, seed, q))
    tape = np.zeros(host.size // 8 + 64, dtype=np.uint64)
    ctx.read_into(host, tape)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); rc, tl, _ = ctx.read_into(host, tape); ts.append(time.perf_counter() - t0)
    ph = pkg.ingest_last_phases()
    print(f"{mib:4d} MiB  best {min(ts)*1e6:8.1f} us  {host.size / min(ts) / 2**30:6.2f} GiB/s  chunks {ph['chunks']} threads {ph['host_threads']}", flush=True)
PY
