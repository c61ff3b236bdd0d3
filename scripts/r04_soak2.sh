#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r04_soak
timeout -k 10 250 python3 scripts/fuzz_gpu.py 150 777 1 > gpurun_out/r04_soak/fuzz_host.json 2> gpurun_out/r04_soak/fuzz_host.err; echo "fuzz host rc=$?"; tail -c 300 gpurun_out/r04_soak/fuzz_host.json
timeout -k 10 250 python3 scripts/fuzz_gpu.py 150 778 4 > gpurun_out/r04_soak/fuzz_dev.json 2> gpurun_out/r04_soak/fuzz_dev.err; echo "fuzz dev rc=$?"; tail -c 300 gpurun_out/r04_soak/fuzz_dev.json
