#!/usr/bin/env python3
"""dev tool (needs a -DCSVSIMD_DEV_PROBES library: CSVSIMD_LIB=.../libcsvsimd_probes.so): per-tile timeline of one
stage-1 launch.  Prints when tiles start / finish, the phase durations by position in a workgroup's sequence of
tiles, the start-up and the tail of the launch."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "16x32_noquote"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols, width, seed, q = pkg.WORKLOADS[name]
n = pkg.workload_len(name, int(gib * 2**30))
ctx = pkg.Context(0)
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
pkg.synth_fill_device(dbuf.data_ptr(), 0, n, cols, width, seed, q)
cap = int(n // (width + 1) * 1.25) + 1024
dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0")
dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ctx.reserve(n)
s = torch.cuda.current_stream().cuda_stream
path = os.environ.setdefault("CSVSIMD_PROBE_TRACE", "/tmp/csvsimd_trace.bin")
os.environ.setdefault("CSVSIMD_PROBE_MODE", "40")   # 56: the same launch without its tape stores
ms = ctx.stage1_time_device(dbuf.data_ptr(), n, dtape.data_ptr(), cap, dres.data_ptr(), s, 2, 3)
tr = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
T = pkg.tile_bytes()
nt = (n + T - 1) // T
xr = tr[nt: 2 * nt].astype(np.float64)   # second record per tile: stamps of the iteration that resolves it
tr = tr[:nt].astype(np.float64)
t0 = tr[:, 0].min()
us = (tr[:, :6] - t0) / 100.0          # s_memrealtime ticks at 100 MHz
raw6 = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)[:nt, 6]
raw7 = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)[:nt, 7]
blk = (raw6 & np.uint64(0xffff)).astype(np.int64)
xcc = ((raw6 >> np.uint64(16)) & np.uint64(0xf)).astype(np.int64)
hwid = (raw7 & np.uint64(0xffffffff)).astype(np.int64)
# the iteration that resolved each tile (10-ns ticks -> us)
it_top = ((raw6 >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64) / 100.0      # loop top -> barrier T (token + ticket)
it_cnt = ((raw6 >> np.uint64(48)) & np.uint64(0xffff)).astype(np.float64) / 100.0      # barrier T -> barrier A (other tile's count)
it_land = ((raw7 >> np.uint64(52)) & np.uint64(0xfff)).astype(np.float64) / 100.0      # barrier A -> look-back window landed
it_win = ((raw7 >> np.uint64(48)) & np.uint64(0xf)).astype(np.int64)                   # windows walked
it_spin = ((raw7 >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)               # back-off spins
print(f"== {name} {gib} GiB: {ms:.4f} ms (timing build), {nt} tiles, {len(set(blk))} workgroups drew tiles")
end = us[:, 5].max()
print(f"launch span by stamps: first ticket 0.0 .. last emit end {end:.1f} us")
order = np.argsort(us[:, 0])
# position of each tile within its workgroup's sequence
pos = np.zeros(nt, dtype=np.int64)
seen = {}
for i in order:
    pos[i] = seen.get(blk[i], 0)
    seen[blk[i]] = pos[i] + 1
names = ["count (ticket->counted)", "wait barrier A", "publish+resolve (lagged: next iteration)", "barrier B", "emit"]
print("per-position medians (us):  pos  tiles  t_start  count  waitA  [A->resolved of THIS tile]  emit_dur  t_end")
for p_ in range(int(pos.max()) + 1):
    m = pos == p_
    if m.sum() == 0:
        continue
    c = np.median(us[m, 1] - us[m, 0]); a = np.median(us[m, 2] - us[m, 1])
    lag = np.median(us[m, 3] - us[m, 2]); em = np.median(us[m, 5] - us[m, 4])
    print(f"   {p_:3d} {int(m.sum()):6d} {np.median(us[m, 0]):8.1f} {c:7.2f} {a:6.2f} {lag:8.2f} {em:8.2f} {np.median(us[m, 5]):8.1f}")
fin = np.sort(us[:, 5])
print("tape completion times (us): 1%% %.1f  50%% %.1f  90%% %.1f  99%% %.1f  max %.1f" % tuple(np.percentile(fin, [1, 50, 90, 99, 100])))
cnt_end = np.sort(us[:, 1])
print("count completion times (us): 1%% %.1f  50%% %.1f  90%% %.1f  99%% %.1f  max %.1f" % tuple(np.percentile(cnt_end, [1, 50, 90, 99, 100])))
# bytes counted per 10-us bucket -> read rate over time
b = np.histogram(us[:, 1], bins=np.arange(0, end + 10, 10))[0] * T / 10e-6 / 1e12
print("read rate by 10-us bucket of count completion (TB/s):", " ".join(f"{x:.1f}" for x in b))
b = np.histogram(us[:, 5], bins=np.arange(0, end + 10, 10))[0] * T / 10e-6 / 1e12
print("emit rate by 10-us bucket of emit completion (input TB/s):", " ".join(f"{x:.1f}" for x in b))
# which workgroups share a CU: HW_ID bits (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13]) + xcc
cu = (xcc << 8) | ((hwid >> 8) & 0xff)
per_cu = {}
for i in range(nt):
    per_cu.setdefault(int(cu[i]), set()).add(int(blk[i]))
sizes = np.bincount([len(v) for v in per_cu.values()])
print("distinct (xcc, se/sh/cu) ids:", len(per_cu), " workgroups per id histogram:", sizes.tolist())
pairs = [sorted(v) for v in per_cu.values() if len(v) == 2][:6]
print("example co-resident workgroup pairs (blockIdx):", pairs)
# the drain in detail: the workgroups that finish last — every stamp of their last two tiles
last = np.argsort(us[:, 5])[-6:]
print("the six tiles whose emit ends last:  tile  wg  pos | ticket counted  A  resolved  B  emit_end | same workgroup's previous tile: resolved emit_end")
by_wg = {}
for i in order:
    by_wg.setdefault(int(blk[i]), []).append(int(i))
for i in last:
    seq = by_wg[int(blk[i])]
    k = seq.index(int(i))
    prev = seq[k - 1] if k > 0 else None
    extra = f"{us[prev, 3]:8.1f} {us[prev, 5]:8.1f}" if prev is not None else ""
    print(f"   {int(i):6d} {int(blk[i]):4d} {int(pos[i]):3d} | " + " ".join(f"{x:8.1f}" for x in us[i, :6]) + " | " + extra
          + f" | resolving iteration: top->T {it_top[i]:.1f} T->A {it_cnt[i]:.1f} A->landed {it_land[i]:.1f} windows {it_win[i]} spins {it_spin[i]}")
cend = us[:, 1].max()
late = us[:, 5] > cend
print(f"after the last count phase ended ({cend:.1f} us): {int(late.sum())} tiles ({late.sum() * T / 2**20:.0f} MiB of input) still to emit, "
      f"done at {us[:, 5].max():.1f} us")
b = np.histogram(us[late, 5], bins=np.arange(cend, end + 5, 5))[0]
print("tiles finishing per 5-us bucket after that:", b.tolist())
print("resolving iteration, medians by position (us): pos  top->T  T->A  A->landed  windows  spins")
for p_ in range(int(pos.max()) + 1):
    m = pos == p_
    if m.sum() == 0:
        continue
    print(f"   {p_:3d} {np.median(it_top[m]):7.2f} {np.median(it_cnt[m]):6.2f} {np.median(it_land[m]):8.2f} "
          f"{np.median(it_win[m]):8.1f} {np.median(it_spin[m]):6.1f}   (max windows {it_win[m].max()}, max spins {it_spin[m].max()})")

# the resolving iteration in absolute stamps (second record): where does a tile's time go between its barrier A and its
# last store?  columns relative to the launch's first ticket
xs = (xr - t0) / 100.0
ok = xr[:, 0] > 0
print("resolving iteration, medians by position (us since launch start):")
print("  pos |  A(count)  top      A'       w0 scattered  landed   resolved | w7 scattered  B(w0)    B(w7)  | w0 flush begins  w0 stores issued  w7 stores issued")
for p_ in range(int(pos.max()) + 1):
    m = (pos == p_) & ok
    if m.sum() == 0:
        continue
    md = lambda a: np.median(a[m])
    print(f"  {p_:3d} | {md(us[:, 2]):8.1f} {md(xs[:, 0]):8.1f} {md(xs[:, 1]):8.1f} {md(xs[:, 2]):10.1f} {md(xs[:, 3]):10.1f} {md(us[:, 3]):9.1f} |"
          f" {md(xs[:, 4]):10.1f} {md(us[:, 4]):9.1f} {md(xs[:, 5]):8.1f} | {md(xs[:, 7]):12.1f} {md(us[:, 5]):16.1f} {md(xs[:, 6]):16.1f}")
print("the same for the twelve tiles whose stores are issued last:")
for i in np.argsort(np.maximum(us[:, 5], xs[:, 6]))[-12:]:
    print(f"  tile {int(i):5d} wg {int(blk[i]):3d} pos {int(pos[i])} | {us[i, 2]:8.1f} {xs[i, 0]:8.1f} {xs[i, 1]:8.1f} {xs[i, 2]:10.1f} {xs[i, 3]:10.1f} {us[i, 3]:9.1f} |"
          f" {xs[i, 4]:10.1f} {us[i, 4]:9.1f} {xs[i, 5]:8.1f} | {xs[i, 7]:12.1f} {us[i, 5]:16.1f} {xs[i, 6]:16.1f}")
# durations of the pieces, all tiles vs the tiles resolved after the last count phase ended
def piece(name, a, b, m):
    d = (b - a)[m]
    return f"{name} {np.median(d):.2f} / {np.percentile(d, 90):.2f}"
for label, m in (("all tiles", ok), ("tiles resolved after the last count ended", ok & (us[:, 3] > cend))):
    if m.sum() == 0:
        continue
    print(f"{label} ({int(m.sum())}), median / p90 us: " + "; ".join([
        piece("A' -> w0 scattered", xs[:, 1], xs[:, 2], m), piece("w0 scattered -> landed", xs[:, 2], xs[:, 3], m),
        piece("landed -> resolved", xs[:, 3], us[:, 3], m), piece("A' -> w7 scattered", xs[:, 1], xs[:, 4], m),
        piece("resolved -> B", us[:, 3], us[:, 4], m), piece("B -> w0 flush begins", us[:, 4], xs[:, 7], m),
        piece("w0 flush", xs[:, 7], us[:, 5], m), piece("B(w7) -> w7 stores issued", xs[:, 5], xs[:, 6], m)]))
