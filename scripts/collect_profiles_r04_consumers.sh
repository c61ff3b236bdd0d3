#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): rocprofv3 passes over the tape consumers of round 4
# (bench.py --only-consumers: 16x32 corpus 1 GiB, 2.03 M records x 16 columns -> columns in one pass, frequency count and
# search on a column; the per-column path of round 2 beside it).  Kernel trace + stats in one pass, FETCH_SIZE / WRITE_SIZE
# in their own --pmc passes (never combined with a trace).  `python3` itself is the profiled program.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04_consumers
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --only-consumers"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_consumers.csv" \;
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if "csvsimd::" not in k or "stage1_kernel" in k or "synth" in k:
                continue
            acc[k.split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
res = {}
for k, c in sorted(acc.items()):
    e = {}
    for name, per in c.items():
        # the full-size call is the largest dispatch of each kernel (the 50 k-record oracle comparison launches it too)
        e[name + "_max_dispatch_KiB"] = max(per.values())
        e["dispatches"] = len(per)
    # FETCH_SIZE reports half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md §HBM): the columnar kernels read
    # with coalesced 16-byte loads, so the doubled value is the one to compare with a byte count; both are kept
    if "FETCH_SIZE_max_dispatch_KiB" in e:
        e["fetch_bytes_raw"] = e["FETCH_SIZE_max_dispatch_KiB"] * 1024
        e["fetch_bytes_x2_gfx950"] = e["FETCH_SIZE_max_dispatch_KiB"] * 2048
    if "WRITE_SIZE_max_dispatch_KiB" in e:
        e["write_bytes"] = e["WRITE_SIZE_max_dispatch_KiB"] * 1024
    res[k] = e
json.dump(res, open(os.path.join(out, "pmc_consumers.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/kernel_stats_consumers.csv"
