#!/bin/bash
# dev tool, run ON the GPU box: phase stamps of colfreq (needs csrc/variants/libcftrace.so: make EXTRA=-DCSVSIMD_CF_TRACE OUT=variants/libcftrace.so)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export CSVSIMD_LIB=$REPO/csv-simd_amd/csrc/variants/libcftrace.so
for c in few distinct; do timeout -k 10 200 python3 $REPO/scripts/r04_cf_trace.py $c 2>&1 | grep -v amdgpu.ids; done | tee $REPO/gpurun_out/r04_cf_trace.txt
