#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_full
mkdir -p "$OUT"
cd $REPO
timeout -k 10 1000 python -m pytest tests/ -q -m gpu --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -12 "$OUT/pytest.log"
timeout -k 10 900 bash scripts/collect_profiles_r04.sh > "$OUT/prof_r04.log" 2>&1; echo "prof r04 rc=$?"; tail -14 "$OUT/prof_r04.log"
