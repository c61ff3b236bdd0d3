#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_full
mkdir -p "$OUT"
cd $REPO
timeout -k 10 500 bash scripts/collect_profiles_r04_16x32.sh > "$OUT/prof_r04_16x32.log" 2>&1; echo "16x32 rc=$?"; tail -6 "$OUT/prof_r04_16x32.log" | cut -c1-300
timeout -k 10 400 bash scripts/collect_profiles_r04_consumers.sh > "$OUT/prof_r04_consumers.log" 2>&1; echo "consumers rc=$?"
timeout -k 10 700 bash scripts/collect_profiles_r04_extra.sh > "$OUT/prof_r04_extra.log" 2>&1; echo "extra rc=$?"; tail -20 "$OUT/prof_r04_extra.log" | cut -c1-250
