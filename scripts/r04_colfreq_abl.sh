#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_colfreq_abl
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset CSVSIMD_LIB; else export CSVSIMD_LIB=$REPO/csv-simd_amd/csrc/variants/libabl$v.so; fi
  for c in few distinct; do
    echo "== variant $v case $c"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$v$c" -- python3 $REPO/scripts/r04_colfreq_cases.py $c 2>&1 | grep -E "status|rror"
    f=$(find "$OUT/$v$c" -name "*kernel_stats.csv" | head -1)
    python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "colfreq" in r["Name"]:
        print("   ", r["Name"].split("(")[0], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
    find "$OUT/$v$c" -name "*.csv" -delete
  done
done
