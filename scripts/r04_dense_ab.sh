#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
rm -f csv-simd_amd/csrc/variants/r1.so
timeout -k 10 200 python3 scripts/ab_variants.py "64x31_noquote:8,16x32_noquote:1,16x32_q10:1,1024x4_dense:1" 1 2>&1 | tail -12
