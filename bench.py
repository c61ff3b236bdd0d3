#!/usr/bin/env python3
"""bench.py — stage-1 CSV structural indexing throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--gib-per-gpu G]

A "step" is one pass of the hot path over one batch of synthetic CSV already resident in HBM:
each rank indexes its contiguous byte shard of the file (speculative stage-1 launch through the
C ABI), the ranks exchange their shard descriptors with ONE all-gather over RCCL (N > 1), stitch
quote parity / tape bases, and a rank whose true entering state is "inside a string" re-emits.
The step ends when the tape and its length are final on every rank.  Weak scaling: every rank
always holds the same number of bytes (default 8 GiB = one GPU's shard of BASELINE config 4,
"64 GiB synthetic CSV, 64 cols, chunk-sharded 8xMI355X").

Rank 0 prints ONE JSON line.  value = whole-job GiB/s (all ranks' bytes / max-over-ranks time).
roofline: HBM-bound, algorithmic bytes = 1 byte read per CSV byte scanned;
duration = the stage-1 kernel's average launch time from HIP events recorded on its own stream.
cpu_baseline: the oracle's faithful SSE restatement of the reference loop ("ref_sse_1t": 1 thread
like the reference, growing Vec) timed on this host over a bounded sample of the same bytes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable (copy)


def build_if_needed():
    so = os.path.join(ROOT, "csv-simd_amd", "csrc", "libcsvsimd_hip.so")
    if not os.path.exists(so):
        graft.build()


class ShardBench:
    """One rank's shard: device buffer, tape shard, context."""

    def __init__(self, pkg, device, workload, shard_bytes, rank, world):
        self.pkg, self.device = pkg, device
        cols, width, seed, q = pkg.WORKLOADS[workload]
        self.cols, self.width = cols, width
        row = cols * (width + 1)
        self.n = (shard_bytes // row) * row           # whole rows per shard (boundary = row start)
        self.lo = rank * self.n
        self.total = world * self.n
        self.dbuf = torch.empty(self.n, dtype=torch.uint8, device=device)
        pkg.synth_fill_device(self.dbuf.data_ptr(), self.lo, self.n, cols, width, seed, q)
        # the bench sizes the tape from the known shape: no retry, no count pre-pass in a step
        self.cap = self.n // (width + 1) + 64
        self.dtape = torch.empty(self.cap, dtype=torch.int64, device=device)
        self.d_result = torch.zeros(8, dtype=torch.int64, device=device)
        self.h_result = torch.zeros(8, dtype=torch.int64).pin_memory()
        self.h_words = self.h_result.numpy()              # the same pinned bytes, cheap to read per step
        self.ctx = pkg.Context(device.index)
        self.ctx.reserve(self.n)
        torch.cuda.synchronize(device)

    def launch(self, in_quote_in):
        """Enqueue stage 1 over this rank's shard (asynchronous; result record -> d_result)."""
        s = torch.cuda.current_stream(self.device)
        self.ctx.stage1_index_device_async(self.dbuf.data_ptr(), self.n, self.lo, in_quote_in,
                                           self.dtape.data_ptr(), self.cap, self.d_result.data_ptr(),
                                           s.cuda_stream)

    def check(self, r):
        if r.error or r.count > self.cap:
            raise RuntimeError(f"stage1 failed: error={r.error} count={r.count} cap={self.cap}")
        return r

    def run_pass(self, in_quote_in):
        """Single-GPU step: launch, then read the result record back (the only synchronisation)."""
        from csv_simd_amd import sharded
        self.launch(in_quote_in)
        self.h_result.copy_(self.d_result, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        w = self.h_words
        if (int(w[4]) & 0xFFFFFFFF) or int(w[0]) > self.cap:   # error flag / more entries than the tape holds
            return self.check(sharded.result_from_words(self.h_result.tolist()))
        return w


def time_steps(step, steps, warmup, device, dist_on):
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def load_traffic(workload, launch_bytes):
    """HBM bytes per launch from the committed PMC profile (separate --pmc passes), if it was taken
    on this workload at this size; None otherwise."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                e = json.load(f).get(workload)
            if e and e.get("launch_bytes") == launch_bytes:
                return e.get("hbm_bytes")
        except Exception:
            return None
    return None


def cpu_baseline(oracle, sb, sample_bytes):
    """ref_sse_1t on a bounded sample of the SAME bytes (copied back from HBM)."""
    n = min(sb.n, sample_bytes)
    host = oracle.aligned_copy(sb.dbuf[:n].cpu().numpy())
    oracle.sse_read_growing_timed(host[: 1 << 24])  # warm the code path / page in
    times, entries = [], 0
    t_total = 0.0
    while len(times) < 3 or (t_total < 10.0 and len(times) < 64):   # ~10 s of CPU work
        entries, dt = oracle.sse_read_growing_timed(host)
        times.append(dt)
        t_total += dt
    best, median = min(times), sorted(times)[len(times) // 2]
    return {"value": round(n / best / 2**30, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
            "variant": "ref_sse_1t (SSE restatement of reference reader::read, growing Vec, 1 thread)",
            "sample": f"first {n / 2**30:.2f} GiB of rank 0's shard, best of {len(times)} passes "
                      f"({t_total:.1f} s of CPU work; median {n / median / 2**30:.2f} GiB/s)",
            "entries": entries, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="64x31_noquote")
    ap.add_argument("--gib-per-gpu", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE shapes at N=1")
    ap.add_argument("--native-rccl", action="store_true",
                    help="do the sharded step inside the C ABI (csvsimd_stage1_index_sharded: ncclAllGather "
                         "from C++) instead of torch.distributed.all_gather_into_tensor")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched by torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    if rank == 0:
        build_if_needed()
    # under torch.distributed.run the collective path is exercised even with a single rank
    dist_on = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("CSVSIMD_BENCH_FORCE_DIST") == "1"
    # rehearsal of the N > 1 control flow on a ONE-GPU box (dev only, never used by the driver): all
    # ranks share cuda:0 and the records travel over gloo through the host instead of RCCL
    rehearsal = os.environ.get("CSVSIMD_BENCH_REHEARSAL") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    if dist_on:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        dist.barrier()
    pkg = graft.load_package()
    from csv_simd_amd import sharded

    shard_bytes = int(args.gib_per_gpu * 2**30)
    sb = ShardBench(pkg, device, args.workload, shard_bytes, rank, world)
    state = {}

    comm = None
    if dist_on and args.native_rccl:
        import torch.distributed as dist
        uid = [pkg.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)   # the 128-byte RCCL id travels over the existing group
        comm = pkg.Comm(uid[0], rank, world, device.index)

    def step():
        if comm is not None:
            r, st = comm.index_sharded(sb.ctx, sb.dbuf.data_ptr(), sb.n, sb.lo, sb.dtape.data_ptr(), sb.cap, 0,
                                       torch.cuda.current_stream(device).cuda_stream)
            sb.check(r)
            state["count"], state["re"], state["total_entries"] = st.count, bool(st.in_quote_in), st.total_entries
        elif dist_on and rehearsal:
            def launch_via_host(inq):
                sb.launch(inq)
                sb.h_result.copy_(sb.d_result, non_blocking=True)
                torch.cuda.current_stream(device).synchronize()
            st, final, re = sharded.index_sharded(launch_via_host, sb.h_result)
            sb.check(final)
            state["count"], state["re"] = st.count, re
            state["total_entries"] = st.total_entries
        elif dist_on:
            st, final, re = sharded.index_sharded(sb.launch, sb.d_result)
            sb.check(final)
            state["count"], state["re"] = st.count, re
            state["total_entries"] = st.total_entries
        else:
            w = sb.run_pass(0)
            state["count"], state["re"], state["total_entries"] = int(w[0]), False, int(w[0]) + 1

    dt = time_steps(step, args.steps, args.warmup, device, dist_on)
    total_bytes = sb.total
    gib_s = total_bytes * args.steps / dt / 2**30
    rows = total_bytes // (sb.cols * (sb.width + 1))

    # ---- roofline leg: the stage-1 kernel alone, HIP events on its own stream -------------------
    kern_ms = sb.ctx.stage1_time_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), sb.cap,
                                        sb.d_result.data_ptr(), torch.cuda.current_stream(device).cuda_stream,
                                        warmup=2, iters=max(5, min(args.steps, 50)))
    achieved = sb.n / (kern_ms * 1e-3) / 1e9
    entries = state["count"]
    # what this GPU's HBM actually streams with the same traffic shape and no work at all (the tape
    # shard doubles as the probe's output buffer: it is rewritten by the next launch anyway)
    probed = None
    if rank == 0 and sb.cap * 8 >= sb.n // 4:
        s_ = torch.cuda.current_stream(device).cuda_stream
        ms_r = sb.ctx.hbm_probe_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), 0, s_, 1, 5)
        ms_rw = sb.ctx.hbm_probe_device(sb.dbuf.data_ptr(), sb.n, sb.dtape.data_ptr(), 4, s_, 1, 5)
        probed = {"read_only_GBps": round(sb.n / (ms_r * 1e-3) / 1e9, 1),
                  "read_with_quarter_written_GBps": round(sb.n / (ms_rw * 1e-3) / 1e9, 1),
                  "note": "bare nt stream of the same buffer, 16 waves/CU (csvsimd_hbm_probe_device); "
                          "the 64x31 corpus writes 8 B of tape per 32 B read"}
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": load_traffic(args.workload, sb.n),
        "kernel": "void csvsimd::stage1_kernel<true, 0, 0>(csvsimd::KernelArgs)", "kernel_ms": round(kern_ms, 4),
        "algorithmic_bytes_per_launch": sb.n,
        "read_plus_tape_write_GBps": round((sb.n + 8 * entries) / (kern_ms * 1e-3) / 1e9, 1),
        "probed_stream": probed,
    }

    out = {
        "metric": "csv_bytes_scanned_per_s", "value": round(gib_s, 3), "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {sb.cols} cols x {sb.width}-byte fields, LF rows, "
                               f"{sb.n / 2**30:.3f} GiB per GPU (BASELINE config 4's per-GPU shard shape)",
                   "bytes_per_gpu": sb.n, "total_bytes": total_bytes, "tape_entries": int(state["total_entries"]),
                   "parallelism": f"chunk-sharded x{world}, one all-gather of shard descriptors"
                                  + (" (native RCCL from the C ABI)" if comm is not None else "")
                                  + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearsal else "")},
        "rows_indexed_per_s": round(rows * args.steps / dt, 1),
        "gib_per_s_per_gpu": round(gib_s / world, 3),
        "roofline": roofline,
    }

    if rank == 0 and world == 1:
        if not args.no_extra:
            extra = {}
            for name in ("16x32_noquote", "16x32_q10", "1024x4_dense"):
                del_sb = ShardBench(pkg, device, name, 1 << 30, 0, 1)
                r = sharded.result_from_words(del_sb.run_pass(0).tolist())
                ms = del_sb.ctx.stage1_time_device(del_sb.dbuf.data_ptr(), del_sb.n, del_sb.dtape.data_ptr(),
                                                   del_sb.cap, del_sb.d_result.data_ptr(),
                                                   torch.cuda.current_stream(device).cuda_stream, 2, 10)
                extra[name] = {"bytes": del_sb.n, "entries": r.count, "kernel_ms": round(ms, 4),
                               "GiB/s": round(del_sb.n / (ms * 1e-3) / 2**30, 2),
                               "hbm_read_frac": round(del_sb.n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                del del_sb
            out["other_workloads"] = extra
        if not args.no_cpu_baseline:
            oracle = graft.load_oracle()
            out["cpu_baseline"] = cpu_baseline(oracle, sb, 2 << 30)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
