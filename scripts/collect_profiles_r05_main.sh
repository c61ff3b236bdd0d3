#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): the rocprofv3 passes behind profiles/r05_*.
# Kernel trace + stats in one pass, then ONE counter group per --pmc pass (never combined with a trace).
# Verification, the quoted leg, ingest and the CPU baseline are switched off in the profiled command: they are torch /
# host work, not the kernel being priced.  `python3` itself is the profiled program (no shell / env hop).
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r05_main
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 60 --warmup 2 --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest --no-strong-check"
BENCH="python3 $REPO/bench.py $FLAGS"
DENSE="python3 $REPO/bench.py --workload 1024x4_dense --gib-per-gpu 1 $FLAGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_64x31_8GiB.csv" \;
echo "stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dense" -- $DENSE > "$OUT/stats_dense.log" 2>&1
find "$OUT/stats_dense" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_1024x4_dense_1GiB.csv" \;
echo "dense stats pass done"
for wl in 16x32_noquote 16x32_q10; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- python3 $REPO/bench.py --workload $wl --gib-per-gpu 1 $FLAGS > "$OUT/stats_$wl.log" 2>&1
    find "$OUT/stats_$wl" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_${wl}_1GiB.csv" \;
done
# the batched launch, ONE batch size per profiled process (round 4's file mixed the 8- and the 64-buffer batches in one row)
for k in 8 64; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_batch$k" -- python3 $REPO/bench.py --only-batch --batch-k $k > "$OUT/stats_batch$k.log" 2>&1
    find "$OUT/stats_batch$k" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_batch_of_$k.csv" \;
done
echo "16x32 + batch stats passes done"
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA")
i=0
for grp in "${GROUPS_[@]}"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/main/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
    echo "pmc pass $i done: $grp"
done
python3 $REPO/scripts/summarise_pmc.py "$OUT/main" "$OUT/pmc_64x31_8GiB.json"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/dense/pmc$i" -- $DENSE > "$OUT/dense$i.log" 2>&1
    echo "dense pmc pass $i done: $grp"
done
python3 $REPO/scripts/summarise_pmc.py "$OUT/dense" "$OUT/pmc_1024x4_dense_1GiB.json" 1073740800 1717985280 "1024x4_dense 1 GiB"
# per-launch durations of the stage-1 kernel, in launch order (the stats average includes bench.py's settle launches,
# i.e. the power-management transient after idle; the last launches are the ones bench.py's HIP events time)
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for tag, sub in (("64x31_8GiB", "stats"), ("1024x4_dense_1GiB", "stats_dense"), ("16x32_noquote_1GiB", "stats_16x32_noquote"),
                 ("16x32_q10_1GiB", "stats_16x32_q10")):
    rows = []
    for path in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if "stage1_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    d = [x[1] / 1e6 for x in rows]
    with open(os.path.join(out, f"launch_durations_{tag}.txt"), "w") as f:
        f.write(f"# stage1_kernel launches in order, ms (rocprofv3 --kernel-trace); n = {len(d)}\n")
        f.write(" ".join("%.4f" % x for x in d) + "\n")
        for k in (10, 20):
            if len(d) >= k:
                f.write(f"# mean of the last {k}: {sum(d[-k:]) / k:.4f} ms; of all: {sum(d) / len(d):.4f} ms\n")
PY
# keep what travels back small: the raw traces stay on the box
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
head -6 "$OUT/kernel_stats_64x31_8GiB.csv"; head -6 "$OUT/kernel_stats_1024x4_dense_1GiB.csv"
