#!/usr/bin/env python3
"""dev tool: rocprofv3 --kernel-trace csv of `bench.py --only-back-to-back` -> which stage-1 launches overlapped, by queue.
Prints, for the one-queue leg and the two-queue leg: per launch start / end relative to the leg's first start, queue id, and
the overlap with the launch before it (end of i - 1 minus start of i; positive = the two were on the GPU together)."""
import csv, sys
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if "stage1_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
print(f"# {len(rows)} stage-1 launches in the trace; times in us")
# the legs: runs of launches separated by > 2 ms gaps are split for readability; only the last 16 launches of each timed block are listed
blocks, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 2_000_000:
        blocks.append(cur); cur = []
    cur.append(r)
if cur:
    blocks.append(cur)
for b in blocks:
    queues = sorted({r[2] for r in b})
    dur = [(r[1] - r[0]) / 1e3 for r in b]
    ov = [(b[i - 1][1] - b[i][0]) / 1e3 for i in range(1, len(b))]
    span = (b[-1][1] - b[0][0]) / 1e3
    print(f"\n## block of {len(b)} launches on queue(s) {queues}: span {span:.1f} us = {span / len(b):.2f} us per launch; "
          f"kernel duration avg {sum(dur) / len(dur):.2f}; overlap with predecessor avg {sum(ov) / max(len(ov), 1):.2f} "
          f"(min {min(ov) if ov else 0:.2f}, max {max(ov) if ov else 0:.2f})")
    t0 = b[0][0]
    for i, r in enumerate(b[-12:]):
        j = len(b) - 12 + i if len(b) >= 12 else i
        o = (b[j - 1][1] - r[0]) / 1e3 if j > 0 else 0.0
        print(f"  launch {j:3d}  queue {r[2]:>3}  start {((r[0] - t0) / 1e3):10.2f}  end {((r[1] - t0) / 1e3):10.2f}  dur {((r[1] - r[0]) / 1e3):7.2f}  overlap_with_prev {o:7.2f}")
