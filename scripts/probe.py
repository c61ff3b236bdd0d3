#!/usr/bin/env python3
"""GPU probe (dev tool): stage-1 kernel variants vs plain streaming kernels on the same buffer."""
import os, sys, json, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()

def ev_time(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["64x31_noquote", "16x32_q10", "1024x4_dense"]
    dev = torch.device("cuda", 0)
    ctx = pkg.Context(0)
    out = {}
    for name in names:
        cols, width, seed, q = pkg.WORKLOADS[name]
        n = pkg.workload_len(name, int(gib * 2**30))
        dbuf = torch.empty(n, dtype=torch.uint8, device=dev)
        pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
        cap = n // (width + 1) + 64
        dtape = torch.empty(cap, dtype=torch.int64, device=dev)
        dres = torch.zeros(8, dtype=torch.int64, device=dev)
        ctx.reserve(n)
        s = torch.cuda.current_stream().cuda_stream
        ms_emit = ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 10)
        ms_count = ctx.stage1_time_device(dbuf.data_ptr(), n, 0, 0, dres.data_ptr(), s, 2, 10)
        extra = {}
        for mode in (1, 4, 6, 7):  # never static tiles WITH look-back (2): non-resident tiles deadlock it
            os.environ["CSVSIMD_PROBE_MODE"] = str(mode)
            ms = ctx.stage1_time_device(dbuf.data_ptr(), n, 0, 0, dres.data_ptr(), s, 2, 10)
            extra[f"mode{mode}_TBps"] = round(n / ms / 1e9, 3)
        for bpc in (2, 4, 6, 8):
            os.environ["CSVSIMD_PROBE_MODE"] = "0"; os.environ["CSVSIMD_PROBE_BLOCKS_PER_CU"] = str(bpc)
            ms = ctx.stage1_time_device(dbuf.data_ptr(), n, 0, 0, dres.data_ptr(), s, 2, 10)
            extra[f"count_bpc{bpc}_TBps"] = round(n / ms / 1e9, 3)
            ms = ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 10)
            extra[f"emit_bpc{bpc}_TBps"] = round(n / ms / 1e9, 3)
        os.environ.pop("CSVSIMD_PROBE_MODE"); os.environ.pop("CSVSIMD_PROBE_BLOCKS_PER_CU")
        out[name] = {"bytes": n, "emit_ms": round(ms_emit, 4), "emit_TBps": round(n / ms_emit / 1e9, 3),
                     "count_ms": round(ms_count, 4), "count_TBps": round(n / ms_count / 1e9, 3), **extra}
        del dtape
    # reference streaming kernels on the last buffer
    v = dbuf[: (n // 16) * 16].view(torch.int64)
    ms_sum = ev_time(lambda: v.sum())
    dst = torch.empty_like(v)
    ms_copy = ev_time(lambda: dst.copy_(v))
    out["torch_sum_read_TBps"] = round(v.numel() * 8 / ms_sum / 1e9, 3)
    out["torch_copy_rw_TBps"] = round(2 * v.numel() * 8 / ms_copy / 1e9, 3)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
