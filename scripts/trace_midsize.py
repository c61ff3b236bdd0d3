#!/usr/bin/env python3
"""dev tool: rocprofv3 --kernel-trace --memory-copy-trace csv of `probe_midsize.py <one size>` -> the GPU-side timeline of the
LAST calls: every copy and kernel with start / end relative to the call's first activity."""
import csv, glob, sys, os
d = sys.argv[1]
ev = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")), ""))
ev.sort()
# calls are separated by gaps > 100 us without GPU activity
calls, cur = [], []
for e in ev:
    if cur and e[0] - max(x[1] for x in cur) > 100_000:
        calls.append(cur); cur = []
    cur.append(e)
if cur: calls.append(cur)
print(f"# {len(ev)} events in {len(calls)} bursts; the last three bursts:")
for c in calls[-3:]:
    t0 = c[0][0]
    print(f"## burst of {len(c)} events, span {(max(x[1] for x in c) - t0) / 1e3:.1f} us")
    for s, e, name, q in c:
        print(f"  {((s - t0) / 1e3):8.1f} -> {((e - t0) / 1e3):8.1f}  ({(e - s) / 1e3:6.1f})  {name} {('q' + q) if q else ''}")
