#!/usr/bin/env python3
"""dev tool (round 5): where a mid-size csvsimd_stage1_index call spends its time: best-of-N wall time and the phase record of
that call, 1 ... 64 MiB, one kept context."""
import faulthandler, json, os, sys, time
import numpy as np
faulthandler.dump_traceback_later(int(os.environ.get("PROBE_WATCHDOG", "90")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
oracle = graft.load_oracle()
ctx = pkg.Context(0)
cols, width, seed, q = pkg.WORKLOADS["16x32_q10"]
sizes = [int(s) << 20 for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32,64,256".split(","))]
out = {}
for n in sizes:
    print("size", n, file=sys.stderr, flush=True)
    host = oracle.aligned_copy(oracle.synth(0, n, cols, width, seed, q))
    tape = np.zeros(host.size // 8 + 64, dtype=np.uint64)
    want = oracle.sse_read(host)
    for w_ in range(3):
        rc, tl, _ = ctx.read_into(host, tape)
        print(" warm", w_, rc, tl, file=sys.stderr, flush=True)
    assert rc == 0 and tl == want.size and np.array_equal(tape[:tl], want), n
    best, ph, ts = None, None, []
    for _ in range(40 if n <= (32 << 20) else 10):
        t0 = time.perf_counter()
        rc, tl, _ = ctx.read_into(host, tape)
        dt = time.perf_counter() - t0
        ts.append(dt)
        if best is None or dt < best:
            best, ph = dt, pkg.ingest_last_phases()
    out[f"{n >> 20} MiB"] = {"best_us": round(best * 1e6, 1), "median_us": round(sorted(ts)[len(ts) // 2] * 1e6, 1),
                            "GiB/s": round(n / best / 2**30, 2),
                            "phases_us": {k: (round(v * 1e6, 1) if isinstance(v, float) else v) for k, v in ph.items()}}
print(json.dumps(out, indent=1))
