#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_colfreq_cases
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "" $LIBS; do
if [ -n "$lib" ]; then export CSVSIMD_LIB=$REPO/csv-simd_amd/csrc/variants/$lib; echo "== $lib"; else unset CSVSIMD_LIB; echo "== product"; fi
for c in ${CASES:-few mid distinct}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$c" -- python3 $REPO/scripts/r04_colfreq_cases.py $c 2>&1 | grep -E "status|rror"
  f=$(find "$OUT/$c" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "colfreq" in r["Name"]:
        print("   ", r["Name"].split("(")[0], "calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1))
PY
  find "$OUT/$c" -name "*.csv" -delete
done
done
unset CSVSIMD_LIB
cd $REPO && python3 bench.py --only-consumers 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['consumers']
print(d['per_column_on_row_major_file']); print(d['frequency_count']['ms'], d['frequency_count']['few_distinct_values'])"
