import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
name = "16x32_noquote"
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, 1024 << 20)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
S = n // (width + 1)
cap = S + 64
ntiles = (n + 131071) // 131072
dtape = torch.full((cap + ntiles * 16 + 64,), -1, dtype=torch.int64, device="cuda:0")
want = torch.arange(1, S + 1, dtype=torch.int64, device="cuda:0") * (width + 1) - 1
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
os.environ["CSVSIMD_PROBE_MODE"] = "16"
nbad = 0
for rep in range(400):
    dtape.fill_(-1)
    ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 0, 1)
    neq = (dtape[:S] != want)
    if bool(neq.any()):
        nbad += 1
        i0 = int(neq.nonzero()[0]); p = int(want[i0])
        tile, wave = p // 131072, (p % 131072) // 32768
        print("rep", rep, "first bad idx", i0, "n bad", int(neq.sum()), "tile", tile, "wave", wave)
        d = dtape[cap:].cpu().view(-1, 4)[: ntiles * 4].view(ntiles, 4, 4)
        for t in (tile - 1, tile, tile + 1):
            for w in range(4):
                run, flags, ba, sb = d[t, w].tolist()
                # expected run: index of first delimiter at/after span start
                span0 = t * 131072 + w * 32768
                exp_run = (span0 + 32) // 33 if span0 > 32 else 0
                exp_run = -(-(span0 - 32) // 33) if span0 >= 32 else 0
                print(f"   tile {t} w{w}: run {run} (expect {exp_run}) pin {flags & 1} bp {(flags >> 1) & 1} slot {(flags >> 2) & 1} iter {flags >> 8} before.a {ba & 0xffffffff} s_base {sb}")
        if nbad >= 3: break
print("bad", nbad)
