"""dev tool, run ON the GPU box: host topology + csvsimd_stage1_index (2 GiB) by number of copying threads; every call's
rate, not the best one, and the phase record of the median call."""
import glob, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()


def rd(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return f"<{e.__class__.__name__}>"


print("affinity cpus:", len(os.sched_getaffinity(0)), "cpu.max:", rd("/sys/fs/cgroup/cpu.max"), "cpuset:", rd("/sys/fs/cgroup/cpuset.cpus.effective")[:80])
for n in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    print(os.path.basename(n), "cpus", rd(n + "/cpulist")[:60], "| MemFree", [l for l in rd(n + "/meminfo").splitlines() if "MemFree" in l][:1])
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node")):
    print(c, rd(c), rd(os.path.dirname(c) + "/vendor"))
print("mems allowed:", [l for l in rd("/proc/self/status").splitlines() if "allowed" in l])

dev = torch.device("cuda", 0)
n = 2 << 30
cols, width, seed, q = pkg.WORKLOADS["64x31_noquote"]
d = torch.empty(n, dtype=torch.uint8, device=dev)
pkg.synth_fill_device(d.data_ptr(), 0, n, cols, width, seed, q)
host = d.cpu().numpy()
tape = np.empty(n // 32 + 64, dtype=np.uint64)
tape[:] = 0
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty_like(pin, device=dev)
def h2d():
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dst.copy_(pin, non_blocking=True); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
    return pin.numel() / (best * 1e-3) / 2**30
print("h2d probe GiB/s:", round(h2d(), 2))
# chunked H2D ceiling: 64 copies of 32 MiB from pinned memory, on one stream / on two streams taking turns
pin2 = torch.empty(2 << 30, dtype=torch.uint8).pin_memory()
dst2 = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
for nstreams in (1, 2):
    ss = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(64):
            with torch.cuda.stream(ss[i % nstreams]):
                dst2[i << 25: (i + 1) << 25].copy_(pin2[i << 25: (i + 1) << 25], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"64 x 32 MiB pinned H2D on {nstreams} stream(s): {2 / best:.2f} GiB/s ({best * 1e3:.2f} ms)")
del pin2, dst2
for spec in (sys.argv[1:] or ["8:2", "8:1", "4:2", "4:1", "8:2"]):
    threads, h2ds = (int(x) for x in spec.split(":"))
    os.environ["CSVSIMD_INGEST_THREADS"] = str(threads)
    os.environ["CSVSIMD_INGEST_H2D_STREAMS"] = str(h2ds)
    ctx = pkg.Context(0)
    ctx.read_into(host[: 256 << 20], tape)
    ctx.read_into(host, tape)
    rates, phases = [], []
    for _ in range(8):
        t0 = time.perf_counter(); rc, tl, _ = ctx.read_into(host, tape); dt = time.perf_counter() - t0
        assert rc == 0
        rates.append(n / dt / 2**30); phases.append(pkg.ingest_last_phases())
    order = np.argsort(rates)
    med = phases[order[len(order) // 2]]
    print(f"threads {threads:2d} h2d streams {h2ds}: GiB/s min {min(rates):.1f} med {np.median(rates):.1f} max {max(rates):.1f} | median call ms:",
          {k: round(v * 1e3, 2) for k, v in med.items() if isinstance(v, float)})
    ctx.close()
print("h2d probe GiB/s:", round(h2d(), 2))
