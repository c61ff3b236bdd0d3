#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_fourth
mkdir -p "$OUT"
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_columnar.py tests/test_gpu_consumers.py -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
bash scripts/r04_colfreq_cases.sh
