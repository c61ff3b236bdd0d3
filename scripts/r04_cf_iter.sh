#!/bin/bash
# dev tool, run ON the GPU box: colfreq parity tests, phase stamps (trace build), then the three cardinalities under rocprofv3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd $REPO
timeout -k 10 400 python -m pytest tests/test_gpu_columnar.py tests/test_gpu_consumers.py -x -q -m gpu 2>&1 | tail -3 || exit 1
echo '== tests with 64-thread pass 2 (several groups of blocks per partition)'
CSVSIMD_LIB=$REPO/csv-simd_amd/csrc/variants/libcf64.so timeout -k 10 400 python -m pytest tests/test_gpu_columnar.py tests/test_gpu_consumers.py -x -q -m gpu 2>&1 | tail -2
bash scripts/r04_cf_trace.sh > /dev/null && grep -E "hot|flushed|pass|hashed|barrier|written|start|end all|merged" gpurun_out/r04_cf_trace.txt | grep -B40 -m1 "distinct hot" 
LIBS="${LIBS:-libprev.so}" bash scripts/r04_colfreq_cases.sh 2>&1 | tee gpurun_out/r04_cf_cases.txt
