#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_fourth
mkdir -p "$OUT"
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_columnar.py tests/test_gpu_consumers.py tests/test_gpu_parity.py -x -q -m gpu -k "columnar or consumers or frequency or gather or spans" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
python3 bench.py --only-consumers 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['consumers']
print(d['per_column_on_row_major_file']); print(d['frequency_count']['ms'], d['frequency_count']['few_distinct_values'], d['verified'])"
