#!/usr/bin/env python3
"""dev tool: PCIe-inclusive rate of the host-buffer entry point (csvsimd_stage1_index)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
out = {}
for name in ("64x31_noquote", "16x32_q10"):
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, 2 << 30)
    dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    host = dbuf.cpu().numpy()
    tape = np.empty(n // (width + 1) + 64, dtype=np.uint64)
    ctx.read_into(host[: 64 << 20], tape)  # warm: pinned buffers, pages
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); rc, cnt, _ = ctx.read_into(host, tape); dt = time.perf_counter() - t0
        assert rc == 0 and cnt == n // (width + 1) + 1
        best = min(best, dt)
    out[name] = {"bytes": n, "entries": cnt, "s": round(best, 4), "GiB/s_pcie_inclusive": round(n / best / 2**30, 2)}
def numa_facts():
    import ctypes
    f = {"cpu_now": ctypes.CDLL(None).sched_getcpu()}
    try:
        bus = torch.cuda.get_device_properties(0).pci_bus_id
        f["pci_bus"] = bus
    except Exception as e:
        f["pci_bus"] = str(e)[:40]
    try:
        import glob
        nodes = {}
        for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
            nodes[os.path.basename(nd)] = open(nd + "/cpulist").read().strip()
        f["nodes"] = nodes
        for d in glob.glob("/sys/class/drm/card*/device/numa_node"):
            f.setdefault("gpu_numa", {})[d.split("/")[4]] = open(d).read().strip()
    except Exception as e:
        f["err"] = str(e)[:60]
    return f
out["numa"] = numa_facts()
print(json.dumps(out))
