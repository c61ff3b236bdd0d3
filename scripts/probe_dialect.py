#!/usr/bin/env python3
"""GPU probe (dev tool): kernel time of the dialect variants on the 64x31 corpus (delimiter swapped in place)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    name = "64x31_noquote"
    cols, width, seed, q = pkg.WORKLOADS[name]
    n = pkg.workload_len(name, int(gib * 2**30))
    dev = torch.device("cuda", 0)
    ctx = pkg.Context(0)
    dbuf = torch.empty(n, dtype=torch.uint8, device=dev)
    pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
    cap = n // (width + 1) + 64
    dtape = torch.empty(cap, dtype=torch.int64, device=dev)
    dres = torch.zeros(8, dtype=torch.int64, device=dev)
    ctx.reserve(n)
    s = torch.cuda.current_stream().cuda_stream
    out = {"bytes": n}
    # the corpus has no ';' and no backslash: ',' stays a payload byte under the ';' dialects, so the
    # tape then only holds the LF of each row; "44,34,92" keeps ',' and adds the escape logic
    for label, env in (("reference", None), ("semicolon_squote", "59,39,0"), ("comma_dquote_backslash", "44,34,92"),
                       ("tab_dquote_backslash", "9,34,92")):
        if env is None:
            os.environ.pop("CSVSIMD_PROBE_DIALECT", None)
        else:
            os.environ["CSVSIMD_PROBE_DIALECT"] = env
        ms = ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 10)
        torch.cuda.synchronize()
        r = pkg.ShardResult.from_buffer_copy(dres.cpu().numpy().tobytes())
        out[label] = {"kernel_ms": round(ms, 4), "TBps": round(n / ms / 1e9, 3), "entries": r.count}
    # UTF-8 validation pass over the same (pure ASCII) corpus, then over text with no ASCII at all
    def time_utf8(buf, nbytes):
        r = torch.zeros(2, dtype=torch.int64, device=dev)
        for _ in range(2):
            ctx.utf8_validate_device_async(buf.data_ptr(), nbytes, r.data_ptr(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.utf8_validate_device_async(buf.data_ptr(), nbytes, r.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        return {"ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3), "first_invalid": int(r[0].item())}

    out["utf8_ascii"] = time_utf8(dbuf, n)
    import numpy as np
    cjk = torch.from_numpy(np.frombuffer("漢字仮名交じり文".encode(), dtype=np.uint8).copy()).to(dev)
    big = cjk.repeat((1 << 30) // cjk.numel())
    out["utf8_cjk_1GiB"] = time_utf8(big, big.numel())
    mixed = torch.from_numpy(np.frombuffer("id,name,città,東京\n".encode(), dtype=np.uint8).copy()).to(dev)
    big = mixed.repeat((1 << 30) // mixed.numel())
    out["utf8_mixed_1GiB"] = time_utf8(big, big.numel())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
