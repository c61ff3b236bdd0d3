#!/bin/bash
# dev tool: the per-CU count-phase token (probe build) against the two pacing knobs
for tok in ${TOKENS:-0 1}; do for d in ${DELAYS:-0 8}; do for p in ${PRIOS:-0 1}; do echo "token=$tok delay=$d prio=$p"; CSVSIMD_PROBE_MODE=0 CSVSIMD_PROBE_CU_TOKEN=$tok CSVSIMD_PROBE_EMIT_DELAY=$d CSVSIMD_PROBE_COUNT_PRIO=$p python scripts/ab_variants.py "${SPECS:-64x31_noquote:8,16x32_noquote:1,1024x4_dense:1}" 1 | grep p_cur; done; done; done
