#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): the rocprofv3 passes behind profiles/r02_*.
# Kernel trace + stats in one pass, then ONE counter group per --pmc pass (never combined with a trace).
# Verification, the quoted leg, ingest and the CPU baseline are switched off in the profiled command: they are torch /
# host work, not the kernel being priced.  `python3` itself is the profiled program (no shell / env hop).
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r02
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest"
BENCH="python3 $REPO/bench.py $FLAGS"
DENSE="python3 $REPO/bench.py --workload 1024x4_dense --gib-per-gpu 1 $FLAGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_64x31_8GiB.csv" \;
echo "stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dense" -- $DENSE > "$OUT/stats_dense.log" 2>&1
find "$OUT/stats_dense" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_1024x4_dense_1GiB.csv" \;
echo "dense stats pass done"
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
# keep what travels back small: the raw traces stay on the box
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
head -6 "$OUT/kernel_stats_64x31_8GiB.csv"; head -6 "$OUT/kernel_stats_1024x4_dense_1GiB.csv"
