#!/bin/bash
# dev tool, run ON the GPU box: (a) kernel stats of the extension kernels (dialects, UTF-8),
# (b) HBM traffic counters of the dense corpus (config 5), one counter per --pmc pass.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_extra
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ext" -- python3 $REPO/scripts/probe_dialect.py 4 > "$OUT/ext.log" 2>&1
find "$OUT/ext" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_extensions.csv" \;
BENCH="python3 $REPO/bench.py --workload 1024x4_dense --gib-per-gpu 1 --steps 10 --warmup 2 --no-extra --no-cpu-baseline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/dense$i" -- $BENCH > "$OUT/dense$i.log" 2>&1
    echo "dense pmc pass $i done: $grp"
done
python3 $REPO/scripts/summarise_pmc.py "$OUT" "$OUT/pmc_dense.json" 1073740800 1717985280 "1024x4_dense 1 GiB"
head -12 "$OUT/kernel_stats_extensions.csv"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
