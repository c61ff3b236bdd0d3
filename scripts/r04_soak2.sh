#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r04_soak
timeout -k 10 250 python3 scripts/fuzz_gpu.py 150 779 4 > gpurun_out/r04_soak/fuzz_dev2.json 2> gpurun_out/r04_soak/fuzz_dev2.err; echo "fuzz dev rc=$?"; tail -c 300 gpurun_out/r04_soak/fuzz_dev2.json
timeout -k 10 200 python3 scripts/soak.py 100 > gpurun_out/r04_soak/soak2.json 2> gpurun_out/r04_soak/soak2.err; echo "soak rc=$?"; tail -c 400 gpurun_out/r04_soak/soak2.json
