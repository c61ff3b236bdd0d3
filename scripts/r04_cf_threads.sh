#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for v in "" cf512 cf256; do
  if [ -z "$v" ]; then unset CSVSIMD_LIB; else export CSVSIMD_LIB=$REPO/csv-simd_amd/csrc/variants/$v.so; fi
  echo "== pass-2 workgroup: ${v:-1024 threads (product)}"
  for c in few mid distinct; do python3 scripts/r04_colfreq_cases.py $c 2>&1 | grep status; done
done
