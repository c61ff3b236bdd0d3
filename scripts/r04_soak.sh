#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r04_soak
timeout -k 10 400 python3 scripts/fuzz_gpu.py 240 20261004 > gpurun_out/r04_soak/fuzz.json 2> gpurun_out/r04_soak/fuzz.err; echo "fuzz rc=$?"; tail -c 400 gpurun_out/r04_soak/fuzz.json
timeout -k 10 300 python3 scripts/soak.py 150 > gpurun_out/r04_soak/soak.json 2> gpurun_out/r04_soak/soak.err; echo "soak rc=$?"; tail -c 600 gpurun_out/r04_soak/soak.json
timeout -k 10 300 python3 scripts/fuzz_columnar.py 90 > gpurun_out/r04_soak/fuzz_columnar.json 2> gpurun_out/r04_soak/fuzz_columnar.err; echo "fuzz_columnar rc=$?"; tail -c 400 gpurun_out/r04_soak/fuzz_columnar.json; tail -3 gpurun_out/r04_soak/fuzz_columnar.err
