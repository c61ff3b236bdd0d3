#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 300 python -m pytest tests/test_gpu_consumers.py tests/test_gpu_columnar.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --only-consumers 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['consumers']
print(d['per_column_on_row_major_file']); print(d['frequency_count']['ms'], d['frequency_count']['few_distinct_values'], d['frequency_count']['traffic_over_algorithmic'], d['frequency_count']['traffic_over_algorithmic_bounds'], d['verified'])"
timeout -k 10 120 python3 scripts/fuzz_columnar.py 40 2>/dev/null | tail -c 300
