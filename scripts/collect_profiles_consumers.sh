#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): rocprofv3 passes over the tape consumers (bench.py --only-consumers:
# 16x32 corpus 1 GiB, 2.03 M records, one 32-byte column).  Kernel trace + stats in one pass, FETCH_SIZE / WRITE_SIZE in
# their own --pmc passes (never combined with a trace).  `python3` itself is the profiled program.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_consumers
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
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if "csvsimd::" not in k or "stage1_kernel" in k or "synth" in k:
                continue
            acc[k.split("(")[0]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
res = {}
for k, c in sorted(acc.items()):
    e = {}
    for name, vals in c.items():
        per = collections.defaultdict(float)
        for d, v in vals:
            per[d] += v
        e[name + "_avg_per_dispatch"] = sum(per.values()) / len(per)
        e["dispatches"] = len(per)
    # KiB units; FETCH_SIZE reports half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md): these kernels are
    # gathers of single sectors, not wide streams, so the raw value is kept and the doubled one shown next to it
    if "FETCH_SIZE_avg_per_dispatch" in e:
        e["fetch_MB_raw"] = round(e["FETCH_SIZE_avg_per_dispatch"] * 1024 / 1e6, 2)
        e["fetch_MB_x2"] = round(e["FETCH_SIZE_avg_per_dispatch"] * 2048 / 1e6, 2)
    if "WRITE_SIZE_avg_per_dispatch" in e:
        e["write_MB"] = round(e["WRITE_SIZE_avg_per_dispatch"] * 1024 / 1e6, 2)
    res[k] = e
json.dump(res, open(os.path.join(out, "pmc_consumers.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/kernel_stats_consumers.csv"
