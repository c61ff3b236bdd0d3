#!/bin/bash
# dev tool, run ON the GPU box: pass-1 ablations of the frequency count on the 100-value column (variants/libablN.so built with
# -DCSVSIMD_CF_ABL=N: bit 0 = the slot's row is not read from LDS, bit 1 = no atomics for a record that met its value).
# Counts are wrong in the ablated builds by construction; only the kernel times are of interest.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
LIBS="libabl1.so libabl2.so libabl3.so" bash $REPO/scripts/r04_colfreq_cases.sh 2>&1 | grep -v "^{\|^0\." | tee $REPO/gpurun_out/r04_cf_abl2.txt
