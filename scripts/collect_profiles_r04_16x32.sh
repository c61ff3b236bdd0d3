#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): rocprofv3 kernel stats + per-launch durations + FETCH/WRITE passes for
# BASELINE configs 2 and 3 (16x32, 1 GiB; VERDICT r2 missing #5: their fractions existed as bench HIP-event numbers only).
# One counter group per --pmc pass, never combined with a trace; `python3` itself is the profiled program.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04_16x32
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLAGS="--gib-per-gpu 1 --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest --no-strong-check"
for W in 16x32_noquote 16x32_q10; do
    CMD="python3 $REPO/bench.py --workload $W $FLAGS"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- $CMD > "$OUT/stats_$W.log" 2>&1
    find "$OUT/stats_$W" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_${W}_1GiB.csv" \;
    echo "stats pass done: $W"
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$W/pmc$i" -- $CMD > "$OUT/pmc_${W}_$i.log" 2>&1
        echo "pmc pass $i done: $W $grp"
    done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
for w in ("16x32_noquote", "16x32_q10"):
    rows = []
    for path in glob.glob(os.path.join(out, "stats_" + w, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if "stage1_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    d = [x[1] / 1e6 for x in rows]
    with open(os.path.join(out, f"launch_durations_{w}_1GiB.txt"), "w") as f:
        f.write(f"# stage1_kernel launches in order, ms (rocprofv3 --kernel-trace); n = {len(d)}\n")
        f.write(" ".join("%.4f" % x for x in d) + "\n")
        for k in (20, 40):
            if len(d) >= k:
                f.write(f"# mean of the last {k}: {sum(d[-k:]) / k:.4f} ms; of all: {sum(d) / len(d):.4f} ms\n")
    # FETCH_SIZE / WRITE_SIZE per stage-1 launch (KiB as rocprofv3 reports them; FETCH x2 on gfx950 for wide streams)
    res = {}
    for i, name in ((1, "FETCH_SIZE"), (2, "WRITE_SIZE")):
        per = {}
        for path in glob.glob(os.path.join(out, "pmc_" + w, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    if "stage1_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals = list(per.values())
        if vals:
            res[name + "_KiB_per_launch_mean"] = sum(vals) / len(vals)
            res[name + "_launches"] = len(vals)
    if "FETCH_SIZE_KiB_per_launch_mean" in res:
        res["hbm_read_bytes_per_launch (FETCH_SIZE x 1024 x 2: gfx950 wide-stream correction)"] = res["FETCH_SIZE_KiB_per_launch_mean"] * 2048
    if "WRITE_SIZE_KiB_per_launch_mean" in res:
        res["hbm_write_bytes_per_launch"] = res["WRITE_SIZE_KiB_per_launch_mean"] * 1024
    json.dump(res, open(os.path.join(out, f"pmc_{w}_1GiB.json"), "w"), indent=1)
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
head -4 "$OUT"/kernel_stats_*.csv; cat "$OUT"/launch_durations_*.txt | grep "#"; cat "$OUT"/pmc_16x32_*_1GiB.json
