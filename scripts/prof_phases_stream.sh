#!/bin/bash
# dev tool: phase sums of both kernel structures (probe build), 8 GiB 64x31 and 1 GiB 16x32
for v in 0 1; do for wl in "64x31_noquote 8" "16x32_noquote 1"; do echo "### variant=$v $wl"; CSVSIMD_LIB=csv-simd_amd/csrc/libcsvsimd_probes.so CSVSIMD_PROBE_VARIANT=$v python scripts/prof_phases.py $wl 2>&1 | grep -v amdgpu.ids | grep -v "COUNT-ONLY" | head -16; done; done
