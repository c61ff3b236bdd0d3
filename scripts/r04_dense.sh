#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_dense
mkdir -p "$OUT"
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_dense_variant.py -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -6 "$OUT/pytest.log"
timeout -k 10 300 python3 scripts/r04_dense_ab.py 2>&1 | grep -E "ms|rror"
