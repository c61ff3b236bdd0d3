#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): the rocprofv3 passes behind profiles/r01_*.
# Kernel trace + stats in one pass, then ONE counter group per --pmc pass (never combined with a trace).
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
    echo "pmc pass $i done: $grp"
done
python3 $REPO/scripts/summarise_pmc.py "$OUT" "$OUT/pmc_summary.json"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
head -5 "$OUT/kernel_stats.csv"
# keep what travels back small: the raw traces stay on the box
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
